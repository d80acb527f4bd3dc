"""Developer probe (GPU), round 6: is a vocoder chain launch's time a STAIRCASE in its workgroup count?
A chain launch of one resblock is W = B * ceil(L / N) workgroups on S = 256 CUs x (2 | 3 co-resident) slots.  If the launch time follows
ceil(W / S) rather than W / S, the partial last round of every launch is lost time (a stage is three dependent launches: three tails) and a
single launch over all three resblocks, longest first, would collect it.  Sweeps the tiles per utterance around the round boundaries for every
resblock of the 32- / 16- / 8-channel stages and prints ms per launch beside W / S.
    python tools/voc_tail_probe.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from diffsinger_amd.vocoder import DsvChainConv, HifiGanGenerator, padded_samples, set_chain_mode


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda', 0)
    m = HifiGanGenerator(bench.VOC_CONFIG)
    m.remove_weight_norm()
    m = m.to(dev).eval()
    B = 8
    m(torch.randn(B, 80, 64, device=dev))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            fn()
        ev1.record()
        ev1.synchronize()
        return ev0.elapsed_time(ev1) / reps

    set_chain_mode(None)
    for stage, C, per_cu in ((1, 32, 2), (2, 16, 3), (3, 8, 3)):
        e = m._chain_prep(stage)
        nres, npairs, ops = e['nres'], e['npairs'], m._ops
        slots = 256 * per_cu
        per_round = slots // B                                     # tiles per utterance that fill one round
        sub = lambda r: (DsvChainConv * (npairs * 2))(*[e['descs'][(r * npairs + q) * 2 + k] for q in range(npairs) for k in range(2)])
        for r in range(nres):
            N = ops.chain_supported(C, 1, npairs, sub(r))
            for nt in sorted({per_round // 2, per_round - 2, per_round, per_round + 2, per_round + per_round // 4, per_round + per_round // 2,
                              2 * per_round - 2, 2 * per_round, 2 * per_round + 2, 2 * per_round + per_round // 4, 2 * per_round + per_round // 2,
                              3 * per_round - 2, 3 * per_round, 3 * per_round + 2}):
                L = nt * N
                x = torch.randn(B, C, padded_samples(L), device=dev)
                x[:, :, L:] = 0
                ms = timed(lambda: ops.resblock_chain(x, L, e['wp'], e['bias'], C, 1, npairs, sub(r)))
                W = B * nt
                print(json.dumps({'stage': stage, 'C': C, 'resblock': r, 'N': N, 'tiles_per_utt': nt, 'workgroups': W, 'rounds': round(W / slots, 3),
                                  'ms': round(ms, 4), 'us_per_round_equiv': round(1e3 * ms / (W / slots), 1)}), flush=True)
                del x


if __name__ == '__main__':
    main()
