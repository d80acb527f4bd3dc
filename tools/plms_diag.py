"""Developer tool (GPU): where do the persistent loop and the per-layer kernels part ways on a PLMS run?  Runs the loop for 1, 2, 3, ...
PLMS iterations (K_step = interval * n) on both paths and prints the max difference of x_0 after each - the first n with a non-zero
difference names the evaluation (RAW + HEUN warm-up, AB2, AB3, AB4) to look at."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsinger_amd.synth import make_inputs
from tests.gpu_helpers import build_hip

interval = 40
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 96
for n in (1, 2, 3, 4, 6):
    K = interval * n
    gd, _, _ = build_hip('opencpop_ds1000', K)
    inp = make_inputs(5, B, T)
    cond, x_T = inp['cond'].cuda(), inp['x_T'].cuda()
    eng = gd._engine(cond)
    outs = {}
    for mode in (1, 0):
        eng.set_loop_mode(mode)
        mel, x = gd.inference(cond, x_T=x_T, K_step=K, pndm_speedup=interval, return_x=True)
        assert eng.loop_mode() == mode
        outs[mode] = x.clone()
    d = (outs[1] - outs[0]).abs()
    print(f'PLMS iterations {n} ({n + 1} evaluations): max |x_0 persistent - per-layer| = {float(d.max()):.3e} (max |x_0| {float(outs[0].abs().max()):.3e}); '
          f'timeouts {eng.loop_timeouts()}; worst element at {tuple(int(v) for v in torch.nonzero(d == d.max())[0])}', flush=True)
