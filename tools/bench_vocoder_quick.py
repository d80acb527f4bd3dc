"""Developer tool (GPU): the shortest A/B of the vocoder's narrow-layer kernel - plain generator, 8 x 1024 mel frames, narrow stages on
k_voc_conv (fold 0) vs k_voc_conv_fold (fold 1).  Prints a JSON line per step with the wall clock since start (flushes at once)."""
import json
import os
import sys
import time

T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsinger_amd import _lib
from diffsinger_amd.vocoder import HifiGanGenerator


def say(**kw):
    print(json.dumps(dict(kw, t=round(time.perf_counter() - T0, 2))), flush=True)


say(step='import')
CONFIG = dict(resblock='1', upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=128,
              resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], audio_sample_rate=24000,
              use_pitch_embed=False)
dev = torch.device('cuda', 0)
B, T = 8, 1024
torch.manual_seed(1)
m = HifiGanGenerator(CONFIG)
m.remove_weight_norm()
with torch.no_grad():
    for n, p in m.named_parameters():
        if n.endswith('weight'):
            p.copy_(torch.randn_like(p) / (p[0].numel() if not n.startswith('ups') else p.shape[0] * 2) ** 0.5)
m = m.to(dev).eval()
mel = torch.randn(B, 80, T, device=dev)
lib = _lib.load()
say(step='model')
out = {}
for fold in (1, 0):
    lib.dsv_set_fold(fold)
    w = m(mel)
    torch.cuda.synchronize()
    say(step='first forward', fold=fold)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        w = m(mel)
    ev[1].record()
    torch.cuda.synchronize()
    out[fold] = w
    say(step='timed', fold=fold, ms_per_forward=ev[0].elapsed_time(ev[1]) / 3, mel_frames_per_s=B * T / (ev[0].elapsed_time(ev[1]) / 3e3))
say(step='diff', max_abs_diff=float((out[0] - out[1]).abs().max()), finite=bool(torch.isfinite(out[1]).all()))
