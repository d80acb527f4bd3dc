"""Developer tool (GPU): first PLMS evaluation count at which the persistent loop and the per-layer graph differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers as H
from tests.gpu_helpers import build_hip

name = 'plms_opencpop_i250'
case, pre, cfg, k_step, inp, smin, smax = H.case_setup(name)
for interval, ks in ((250, (250, 500, 750, 1000)), (40, (40, 80, 120, 160, 200, 400, 1000))):
    for K in ks:
        outs = []
        for mode in (1, 0):
            gd, _, _ = build_hip(case['preset'], K)
            cond = inp['cond'].transpose(1, 2).contiguous().cuda().transpose(1, 2)
            eng = gd._engine(cond)
            eng.set_loop_mode(mode)
            with torch.no_grad():
                mel, x = gd.inference(cond, x_T=inp['x_T'].cuda(), K_step=K, pndm_speedup=interval, return_x=True)
            outs.append(x.cpu().numpy())
        d = np.abs(outs[0] - outs[1])
        nz = np.argwhere(d > 0)
        print(f'interval {interval} K_step {K}: evals {len(range(0, K, interval)) + 1}: mismatched {int((d > 0).sum())}/{d.size}, max abs {d.max():.3e}, '
              f'max|x| {np.abs(outs[1]).max():.2f}; first few idx {nz[:4].tolist()}', flush=True)
