"""Developer tool (GPU): the kernel SEQUENCE of one FastSpeech2 forward / one denoiser training step / one HiFi-GAN generator forward at the bench shape (torch.profiler,
device-side names in launch order, with duration and the idle gap in front) - which launches are still torch glue between the library's
kernels.      python tools/glue_trace.py fs2|train|vocoder|path [preset]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import diffsinger_amd
from diffsinger_amd import hparams


def trace(fwd, label):
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        fwd()
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ev.sort(key=lambda e: e.time_range.start)
    t_first, t_last = ev[0].time_range.start, ev[-1].time_range.end
    print(f'# {label}: {len(ev)} device events, first start -> last end {t_last - t_first:.1f} us')
    n_torch = us_torch = 0
    prev_end = t_first
    for e in ev:
        gap = e.time_range.start - prev_end
        prev_end = e.time_range.end
        is_torch = not e.name.startswith(('dsd::', 'void dsd::'))
        n_torch += is_torch
        us_torch += (e.time_range.end - e.time_range.start) if is_torch else 0
        print(f'{"T" if is_torch else " "} {e.time_range.end - e.time_range.start:8.1f} us  gap {gap:6.1f}  {e.name[:110]}')
    print(f'# non-library launches: {n_torch}, {us_torch:.1f} us of device time')


def main_train(preset='lj_ds_beta6', B=8, T=1024):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device('cuda', 0)
    gd, pre = bench.build_model(dev)
    gd.train()
    net = gd.denoise_fn
    g = torch.Generator(device=dev).manual_seed(1234)
    x0 = torch.randn(B, 1, 80, T, device=dev, generator=g).clamp(-1, 1)
    cond = torch.randn(B, T, 256, device=dev, generator=g).transpose(1, 2)
    t = torch.randint(0, 100, (B,), device=dev, generator=g)
    noise = torch.randn(B, 1, 80, T, device=dev, generator=g)

    def step():
        net.zero_grad(set_to_none=True)
        loss = gd.p_losses(x0, t, cond, noise=noise)
        loss.backward()
        return loss
    trace(step, f'training step {preset} {B} x {T}')


def main_path(B=8, T=1024, K=100):
    import bench
    dev = torch.device('cuda', 0)
    gd, pre = bench.build_model(dev)
    gd.eval()
    g = torch.Generator(device=dev).manual_seed(1234)
    conds = [torch.randn(B, T, 256, device=dev, generator=g).transpose(1, 2) for _ in range(2)]
    x_T = torch.randn(B, 1, 80, T, device=dev, generator=g)
    noise = torch.randn(K, B, 1, 80, T, device=dev, generator=g)
    count = [0]

    def step():
        count[0] += 1
        with torch.no_grad():
            return gd.inference(conds[count[0] & 1], x_T=x_T, noise=noise, K_step=K, pndm_speedup=0)
    trace(step, f'BASELINE configs[1] step: {B} x {T}, K = {K}')


def main_vocoder(B=8, T=1024):
    import bench
    from diffsinger_amd.vocoder import HifiGanGenerator
    dev = torch.device('cuda', 0)
    m = HifiGanGenerator(bench.VOC_CONFIG)
    m.remove_weight_norm()
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('weight'):
                p.copy_(torch.randn(p.shape, generator=g) / (p[0].numel() if not n.startswith('ups') else p.shape[0] * 2) ** 0.5)
    m = m.to(dev).eval()
    mel = torch.randn(B, 80, T, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
    trace(lambda: m(mel), f'HiFi-GAN generator {B} x {T} frames')


def main(preset='lj_ds_beta6', B=8, T_txt=128, fpp=8):
    hparams.clear()
    diffsinger_amd.use_preset(preset)
    from diffsinger_amd import fs2
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    m = fs2.FastSpeech2(63, 80).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(7)
    tok = torch.randint(1, 63, (B, T_txt), device=dev, generator=g)
    T = T_txt * fpp
    mel2ph = (torch.arange(T, device=dev) // fpp + 1)[None].repeat(B, 1)
    kw = dict(mel2ph=mel2ph, f0=torch.rand(B, T, device=dev, generator=g) * 2 + 6.5, uv=torch.zeros(B, T, device=dev))
    fwd = lambda: m(tok, infer=True, **{k: (v.clone() if k == 'f0' else v) for k, v in kw.items()})
    trace(fwd, preset)


if __name__ == '__main__':
    {'train': main_train, 'vocoder': main_vocoder, 'path': main_path}.get(sys.argv[1] if len(sys.argv) > 1 else 'fs2', main)(*sys.argv[2:])
