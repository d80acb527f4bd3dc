"""Developer tool (GPU): ONE utterance of 512 frames, K = 100, on the latency kernels with and without the branch-free K-half conv
(DSD_LAT_BF, csrc/dsd_kernels.hpp ConvB<LD, true>): results must be the same bits; prints both call times."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsinger_amd  # noqa: E402
from diffsinger_amd import hparams  # noqa: E402
from diffsinger_amd.synth import presets  # noqa: E402


def run(bf, cond, x_T, noise, K, preset='lj_ds_beta6'):
    os.environ['DSD_LAT_BF'] = '1' if bf else '0'
    pre = presets()[preset]
    hparams.clear()
    diffsinger_amd.use_preset(preset)
    torch.manual_seed(3)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=K, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).cuda().eval()
    eng = gd._engine(cond)
    with torch.no_grad():
        out = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
    return out, min(ts), eng.lat_split()


if __name__ == '__main__':
    # usage: lat_bf_probe.py [preset B T K]   (default: lj_ds_beta6 1 512 100 - the reference's one-utterance inference shape)
    preset = sys.argv[1] if len(sys.argv) > 1 else 'lj_ds_beta6'
    B, T, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1, 512, 100)
    g = torch.Generator(device='cuda').manual_seed(5)
    cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, device='cuda', generator=g)
    noise = torch.randn(K, B, 1, 80, T, device='cuda', generator=g)
    a, ta, ga = run(False, cond, x_T, noise, K, preset)
    b, tb, gb = run(True, cond, x_T, noise, K, preset)
    a2, ta2, _ = run(False, cond, x_T, noise, K, preset)
    b2, tb2, _ = run(True, cond, x_T, noise, K, preset)
    print(json.dumps({'preset': preset, 'B': B, 'T': T, 'K': K, 'lat_split': [ga, gb], 'ms_default': [ta, ta2], 'ms_branch_free': [tb, tb2],
                      'bit_identical': bool(torch.equal(a, b) and torch.equal(a2, b2)), 'finite': bool(torch.isfinite(b).all())}))
