"""Developer tool (GPU): one DiffNet training step (p_losses forward + backward, no optimiser) on the HIP operators of
diffsinger_amd/train.py, next to the same autograd graph with torch's own conv1d (MIOpen / rocBLAS) on the same GPU.
    python tools/bench_train.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import diffsinger_amd
from diffsinger_amd import hparams, train
from diffsinger_amd.synth import presets

F_FWD = 26_427_392          # GEMM FLOP / frame of one DiffNet forward (the conditioner projection is NOT hoisted in training)


def run(B, T, reps, torch_conv=False):
    pre = presets()['lj_ds_beta6']
    hparams.clear()
    diffsinger_amd.use_preset('lj_ds_beta6')
    torch.manual_seed(1234)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=100, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).cuda().train()
    orig = train.ConvCache.__call__
    if torch_conv:
        def tc(self, x, weight, bias, T_, dil=1):
            w3 = weight if weight.dim() == 3 else weight[:, :, None]
            return F.conv1d(x, w3, bias, padding=dil * (w3.shape[2] - 1) // 2, dilation=dil)
        train.ConvCache.__call__ = tc
    try:
        g = torch.Generator(device='cuda').manual_seed(3)
        x0 = torch.randn(B, 1, 80, T, device='cuda', generator=g).clamp(-1, 1)
        cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
        t = torch.randint(0, 100, (B,), device='cuda', generator=g)

        def step():
            net.zero_grad(set_to_none=True)
            loss = gd.p_losses(x0, t, cond)
            loss.backward()
            return loss
        step(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            loss = step()
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / reps
    finally:
        train.ConvCache.__call__ = orig
    frames = B * T
    print(json.dumps({'impl': 'torch conv1d (MIOpen) in the same graph' if torch_conv else 'HIP operators (dsf_conv1d_dilated / dsf_conv1d_wgrad)',
                      'B': B, 'T': T, 'ms_per_step_fwd_bwd': sec * 1e3, 'frames_per_s': frames / sec,
                      'tflops_gemm': 3 * F_FWD * frames / sec / 1e12, 'loss': float(loss)}), flush=True)
    del gd, net
    torch.cuda.empty_cache()


if __name__ == '__main__':
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    for B, T in ((8, 1024), (48, 512)):
        run(B, T, reps, torch_conv=False)
        run(B, T, reps, torch_conv=True)
