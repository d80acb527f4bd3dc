"""Developer tool (GPU): the ParallelWaveGAN generator (diffsinger_amd/pwg.py) at the vocoder row's shape - 8 x 1024 mel frames, hop 256 - and for
one utterance.  One JSON line per shape.      python tools/bench_pwg.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsinger_amd.pwg import ParallelWaveGANGenerator
from oracle.pwg_cases import synth_state


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device('cuda', 0)
    m = ParallelWaveGANGenerator()
    m.load_state_dict(synth_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 1), strict=True)
    m.remove_weight_norm()
    m = m.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(3)
    for B, T in ((8, 1024), (1, 800)):
        x = torch.randn(B, 1, T * 256, device=dev, generator=g)
        c = torch.randn(B, 80, T + 4, device=dev, generator=g)
        y = m(x, c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            y = m(x, c)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / reps
        assert bool(torch.isfinite(y).all())
        flop = B * T * 256 * 30 * 2.0 * (128 * (192 + 80) + 128 * 64)      # the 30 residual blocks (everything else is < 1 %)
        print(json.dumps({'B': B, 'T_mel': T, 'samples': B * T * 256, 'ms_per_forward': round(sec * 1e3, 3), 'mel_frames_per_s': round(B * T / sec, 1),
                          'x_realtime_24k': round(B * T * 256 / 24000 / sec, 1), 'tflops': round(flop / sec / 1e12, 2),
                          'frac_fp32_mfma_peak': round(flop / sec / 1e12 / 157.3, 3)}), flush=True)


if __name__ == '__main__':
    main()
