"""Developer probe (GPU), round 6: do the three parallel resblocks of a HiFi-GAN stage lose time in the TAILS of their launches?
A stage's default form is one chain launch per resblock (kernel 3 / 7 / 11), each 2-3.5 rounds of co-resident workgroups; launch r + 1 reads the
running sum launch r wrote, so the launches cannot overlap and every one pays its own partial last round.  Here the three resblocks write
SEPARATE buffers from three streams (no dependence: the GPU fills one launch's tail with the next one's workgroups) and one torch expression
forms ((y0 + y1) + y2) / 3 - the order of the running sum.  Prints, per stage: each resblock alone, the default form, the concurrent form.
    python tools/voc_parallel_probe.py [reps]"""
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
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('weight'):
                p.copy_(torch.randn(p.shape, generator=g) / (p[0].numel() if not n.startswith('ups') else p.shape[0] * 2) ** 0.5)
    m = m.to(dev).eval()
    B, T = 8, 1024
    m(torch.randn(B, 80, T, device=dev))
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

    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for stage, C in ((1, 32), (2, 16), (3, 8)):
        L = T * {1: 64, 2: 128, 3: 256}[stage]
        x = torch.randn(B, C, padded_samples(L), device=dev)
        x[:, :, L:] = 0
        set_chain_mode(None)
        e = m._chain_prep(stage)
        nres, npairs, ops = e['nres'], e['npairs'], m._ops
        sub = lambda r: (DsvChainConv * (npairs * 2))(*[e['descs'][(r * npairs + q) * 2 + k] for q in range(npairs) for k in range(2)])
        want = m._stage_resblocks(stage, x, L)
        ms_default = timed(lambda: m._stage_resblocks(stage, x, L))
        alone = [timed(lambda r=r: ops.resblock_chain(x, L, e['wp'], e['bias'], C, 1, npairs, sub(r))) for r in range(nres)]

        def concurrent():
            main_s = torch.cuda.current_stream()
            ys = []
            for r in range(nres):
                streams[r].wait_stream(main_s)
                with torch.cuda.stream(streams[r]):
                    ys.append(ops.resblock_chain(x, L, e['wp'], e['bias'], C, 1, npairs, sub(r)))
            for r in range(nres):
                main_s.wait_stream(streams[r])
            return ((ys[0] + ys[1]) + ys[2]) / float(nres)

        got = concurrent()
        torch.cuda.synchronize()
        same = bool(torch.equal(got[:, :, :L], want[:, :, :L]))
        ms_conc = timed(concurrent)
        ms_sum = timed(lambda: ((want + want) + want) / 3.0)
        print(json.dumps({'stage': stage, 'C': C, 'default_one_launch_per_resblock_ms': ms_default, 'each_resblock_alone_ms': alone, 'sum_of_alone_ms': sum(alone),
                          'three_streams_plus_torch_sum_ms': ms_conc, 'torch_sum_alone_ms': ms_sum, 'same_bits': same}), flush=True)


if __name__ == '__main__':
    main()
