"""Developer tool (GPU): time of the HIP PitchExtractor forward (mel -> f0 for the NSF vocoder) on the output of the
bench shape of the diffusion loop (8 x 1024 mel frames).  One JSON line.      python tools/bench_pe.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import diffsinger_amd
from diffsinger_amd import hparams


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    hparams.clear()
    diffsinger_amd.use_preset('opencpop_ds1000')                      # the shipped config with pe_enable
    from diffsinger_amd.pe import PitchExtractor
    torch.manual_seed(1234)
    dev = torch.device('cuda', 0)
    m = PitchExtractor().to(dev).eval()
    B, T = 8, 1024
    mel = torch.randn(B, T, 80, device=dev) * 1.5 - 4
    r = m(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = m(mel)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    H, k = 256, hparams['predictor_kernel']
    flop = 2 * (80 * H * 5 + 2 * H * H * 5 + H * H) + 2 * (2 * H * H + 2 * H * H * 5) + 2 * (5 * H * H * k + H * 2)
    assert bool(torch.isfinite(r['f0_denorm_pred']).all())
    print(json.dumps({'model': 'PitchExtractor', 'B': B, 'T_mel': T, 'ms_per_forward': sec * 1e3, 'mel_frames_per_s': B * T / sec, 'flop_per_frame': flop,
                      'tflops': B * T * flop / sec / 1e12, 'note': 'eager launches from Python (13 convolution / linear, 5 + 2 + 3 normalisation launches)'}), flush=True)


if __name__ == '__main__':
    main()
