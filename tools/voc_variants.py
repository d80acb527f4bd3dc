"""Developer tool (GPU): A/B of the resblock-chain kernel's instantiations (dsv_set_chain_variant, include/dsv.h) on the bench shape of row f2 -
per stage (8 x 1024 mel frames: 32 channels x 65 536 samples, 16 x 131 072, 8 x 262 144) the launch time of the whole stage for every built
(nb, in_place) pair, then the generator forward with the best of each.  JSON lines.      python tools/voc_variants.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from diffsinger_amd import _lib
from diffsinger_amd.vocoder import HifiGanGenerator, padded_samples, set_chain_mode

VARIANTS = {32: [(4, 0), (4, 1)], 16: [(2, 0), (2, 1), (4, 1)], 8: [(2, 0), (2, 1), (4, 1)]}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device('cuda', 0)
    lib = _lib.load()
    h = bench.VOC_CONFIG
    m = HifiGanGenerator(h)
    m.remove_weight_norm()
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('weight'):
                p.copy_(torch.randn(p.shape, generator=g) / (p[0].numel() if not n.startswith('ups') else p.shape[0] * 2) ** 0.5)
    m = m.to(dev).eval()
    B, T = 8, 1024
    mel = torch.randn(B, 80, T, device=dev)
    m(mel)
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

    best = {}
    for stage, C in ((1, 32), (2, 16), (3, 8)):
        L = T * {1: 64, 2: 128, 3: 256}[stage]
        x = torch.randn(B, C, padded_samples(L), device=dev)
        x[:, :, L:] = 0
        set_chain_mode('off')
        want = m._stage_resblocks(stage, x, L)
        ms_off = timed(lambda: m._stage_resblocks(stage, x, L))
        set_chain_mode(None)
        flop = sum(2 * B * L * C * C * k * 6 for k in h['resblock_kernel_sizes'])
        print(json.dumps({'stage': stage, 'C': C, 'variant': 'one launch per convolution', 'ms': ms_off, 'tflops_useful': flop / ms_off / 1e9}), flush=True)
        for nb, ip in VARIANTS[C]:
            assert lib.dsv_set_chain_variant(C, nb, ip) == 0
            for mode in ('stage', 'resblock', 'pair'):
                set_chain_mode(mode)
                got = m._stage_resblocks(stage, x, L)
                same = bool(torch.equal(got, want))
                ms = timed(lambda: m._stage_resblocks(stage, x, L))
                print(json.dumps({'stage': stage, 'C': C, 'nb': nb, 'in_place': ip, 'mode': mode, 'ms': ms, 'tflops_useful': flop / ms / 1e9,
                                  'frac_fp32_peak': flop / ms / 1e9 / bench.PEAK_FP32_MFMA_TFLOPS, 'bit_identical_to_single_convs': same}), flush=True)
                if C not in best or ms < best[C][0]:
                    best[C] = (ms, nb, ip, mode)
        # round 6: two resblocks merged into one launch (the default variants' MG instantiation), every choice of the summing resblock, and the
        # default (None: the split vocoder._merge_plan picks, or one launch per resblock)
        for nb0, ip0 in {32: [(4, 1)], 16: [(4, 1), (2, 1)], 8: [(4, 1), (2, 1)]}[C]:          # (the last one stays in force)
            assert lib.dsv_set_chain_variant(C, nb0, ip0) == 0
            m._packed.pop('merge_plans', None)
            for mode in ('resblock', 'merged0', 'merged1', 'merged2', None, 'resblock', None):
                set_chain_mode(mode)
                got = m._stage_resblocks(stage, x, L)
                same = bool(torch.equal(got, want))
                ms = timed(lambda: m._stage_resblocks(stage, x, L))
                plan = m._merge_plan_for(stage, m._chain_prep(stage), B, L) if mode is None else None
                print(json.dumps({'stage': stage, 'C': C, 'nb': nb0, 'in_place': ip0, 'mode': mode or 'default', 'plan': plan, 'ms': ms, 'tflops_useful': flop / ms / 1e9,
                                  'frac_fp32_peak': flop / ms / 1e9 / bench.PEAK_FP32_MFMA_TFLOPS, 'bit_identical_to_single_convs': same}), flush=True)
    print(json.dumps({'best': {str(C): best[C] for C in best}}), flush=True)
    for C, (_, nb, ip, mode) in best.items():
        lib.dsv_set_chain_variant(C, nb, ip)
    set_chain_mode(None)
    print(json.dumps({'generator_forward_ms': timed(lambda: m(mel)), 'chain_mode': 'default', 'variants': {str(C): best[C][1:3] for C in best}}), flush=True)
    set_chain_mode('resblock')
    print(json.dumps({'generator_forward_ms': timed(lambda: m(mel)), 'chain_mode': 'resblock (no merge)'}), flush=True)
    set_chain_mode(None)
    print(json.dumps({'generator_forward_ms': timed(lambda: m(mel)), 'chain_mode': 'default'}), flush=True)


if __name__ == '__main__':
    main()
