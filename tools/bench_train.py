"""Developer tool (GPU): one DiffNet training step (p_losses forward + backward, no optimiser) on the HIP operators of
diffsinger_amd/train.py, next to the same autograd graph with torch's own conv1d (MIOpen / rocBLAS) on the same GPU.
    python tools/bench_train.py [reps]
    python tools/bench_train.py [reps] --conv-ab      the fused stack with the Winograd / the direct convolution in its persistent forward"""
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


def reference_style_forward(net, spec, t, cond):
    """What the reference's DiffNet.forward does (usr/diff/net.py:107-130), written with plain torch ops on the GPU: the timing
    baseline "PyTorch-ROCm eager" (MIOpen / rocBLAS convolutions, one ATen kernel per element-wise op, torch.stack of the skips)."""
    x = F.relu(F.conv1d(spec[:, 0], net.input_projection.weight, net.input_projection.bias))
    d = train.step_embedding(t, net.residual_channels)
    h = F.linear(d, net.mlp[0].weight, net.mlp[0].bias)
    d = F.linear(h * torch.tanh(F.softplus(h)), net.mlp[2].weight, net.mlp[2].bias)
    skips = []
    for layer in net.residual_layers:
        ds = F.linear(d, layer.diffusion_projection.weight, layer.diffusion_projection.bias).unsqueeze(-1)
        c = F.conv1d(cond, layer.conditioner_projection.weight, layer.conditioner_projection.bias)
        y = F.conv1d(x + ds, layer.dilated_conv.weight, layer.dilated_conv.bias, padding=layer.dilation, dilation=layer.dilation) + c
        gate, filt = torch.chunk(y, 2, dim=1)
        y = torch.sigmoid(gate) * torch.tanh(filt)
        y = F.conv1d(y, layer.output_projection.weight, layer.output_projection.bias)
        res, sk = torch.chunk(y, 2, dim=1)
        x = (x + res) / (2.0 ** 0.5)
        skips.append(sk)
    x = torch.sum(torch.stack(skips), dim=0) / (len(net.residual_layers) ** 0.5)
    x = F.relu(F.conv1d(x, net.skip_projection.weight, net.skip_projection.bias))
    return F.conv1d(x, net.output_projection.weight, net.output_projection.bias)[:, None]


def run(B, T, reps, torch_conv=False, reference_style=False, fused=True, graph=False, conv=None):
    from diffsinger_amd import train_fused
    if conv is not None:
        train_fused.set_stack_conv(conv)
    os.environ['DSD_TRAIN_FUSED'] = '1' if (fused and not torch_conv and not reference_style) else '0'
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

        noise = torch.randn(B, 1, 80, T, device='cuda', generator=g)

        def step():
            net.zero_grad(set_to_none=True)
            if reference_style:
                shp = (B, 1, 1, 1)
                xn = gd.sqrt_alphas_cumprod.gather(-1, t).reshape(shp) * x0 + gd.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(shp) * noise
                loss = (noise - reference_style_forward(net, xn, t, cond.contiguous())).abs().mean()
            else:
                loss = gd.p_losses(x0, t, cond, noise=noise)
            loss.backward()
            return loss
        step(); step()
        torch.cuda.synchronize()
        if graph:                                   # experiment: the whole step captured once and replayed as ONE hipGraph
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                loss = step()
            run_step = cg.replay
        else:
            run_step = step
        run_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = run_step()
            loss = r if r is not None else loss
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / reps
    finally:
        train.ConvCache.__call__ = orig
    frames = B * T
    impl = ('reference-style PyTorch-ROCm eager graph (MIOpen convolutions, ATen element-wise ops)' if reference_style else
            'torch conv1d (MIOpen) inside the HIP graph (fused glue kept)' if torch_conv else
            'fused residual stack (dsf_stack_forward / dsf_stack_backward)' if fused else 'HIP operators (dsf_conv1d_dilated / dsf_conv1d_wgrad / dsf_train_*)')
    if fused and not torch_conv and not reference_style:
        impl += f', persistent forward convolution: {train_fused.stack_conv()}'
    print(json.dumps({'impl': impl + (' replayed as one hipGraph' if graph else ''),
                      'B': B, 'T': T, 'ms_per_step_fwd_bwd': sec * 1e3, 'frames_per_s': frames / sec,
                      'tflops_gemm': 3 * F_FWD * frames / sec / 1e12, 'loss': float(loss)}), flush=True)
    del gd, net
    torch.cuda.empty_cache()


if __name__ == '__main__':
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    if '--conv-ab' in sys.argv:
        for B, T in ((8, 1024), (48, 512), (16, 1024)):
            for conv in ('wino', 'direct', 'wino', 'direct'):
                run(B, T, reps, conv=conv)
        sys.exit(0)
    if '--hip-only' in sys.argv:                    # for rocprofv3 runs: only the HIP variant, one shape
        B, T = (int(v) for v in sys.argv[sys.argv.index('--hip-only') + 1].split('x'))
        run(B, T, reps, graph='--graph' in sys.argv)
        sys.exit(0)
    for B, T in ((8, 1024), (48, 512)):
        run(B, T, reps)
        run(B, T, reps, fused=False)
        run(B, T, reps, torch_conv=True)
        run(B, T, reps, reference_style=True)
