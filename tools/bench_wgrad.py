"""Developer tool (GPU): the weight-gradient kernel of the fused training stack alone (dsf_conv1d_wgrad2), events around 50 launches.
    python tools/bench_wgrad.py [BxT ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsinger_amd import train_fused


def run(B, T, K, Co=512, Ci=256, dil=1, reps=50):
    dy = torch.randn(B, Co, T, device='cuda')
    x = torch.randn(B, Ci, T, device='cuda')
    for _ in range(5):
        train_fused.conv1d_wgrad2(dy, x, K, dil, T)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        train_fused.conv1d_wgrad2(dy, x, K, dil, T)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flop = 2.0 * Co * Ci * K * B * T
    print(json.dumps({'B': B, 'T': T, 'K': K, 'us_per_call_incl_reduce': us, 'tflops': flop / us / 1e6, 'pipe': os.environ.get('DSD_WGRAD_PIPE', '1')}), flush=True)


if __name__ == '__main__':
    shapes = [a for a in sys.argv[1:] if 'x' in a] or ['8x1024', '48x512']
    for sh in shapes:
        B, T = (int(v) for v in sh.split('x'))
        run(B, T, 3)
        run(B, T, 1)
