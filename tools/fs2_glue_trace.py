"""Developer tool (GPU): the kernel SEQUENCE of one FastSpeech2 forward at the bench shape (torch.profiler, device-side names in launch order,
with the torch operator that launched each) - which launches are still torch index / mask glue.      python tools/fs2_glue_trace.py [preset]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import diffsinger_amd
from diffsinger_amd import hparams


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
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        fwd()
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ev.sort(key=lambda e: e.time_range.start)
    t_first, t_last = ev[0].time_range.start, ev[-1].time_range.end
    print(f'# {preset}: {len(ev)} device events, first start -> last end {t_last - t_first:.1f} us')
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


if __name__ == '__main__':
    main(*sys.argv[1:])
