"""Developer tool (GPU): phase timeline of the ROW-SPLIT persistent loop (csrc/dsd_loop_rs.hpp) from in-kernel s_memtime stamps: what one
residual layer costs inside the launch and how much of it is the two exchanges (x' rows in, gate rows in).

    python tools/rs_timeline.py [B x T, default 1x512] [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

shape = sys.argv[1] if len(sys.argv) > 1 else '1x512'
B, T = (int(v) for v in shape.split('x'))
K = 6
dev = torch.device('cuda', 0)
gd, pre = bench.build_model(dev)
g = torch.Generator(device=dev).manual_seed(1)
cond = torch.randn(B, T, 256, device=dev, generator=g).transpose(1, 2)
x = torch.randn(B, 80, T, device=dev, generator=g)
noise = torch.randn(K, B, 80, T, device=dev, generator=g)
eng = gd._engine(cond)
eng.set_rs_split(-1)
G = eng.rs_split()
assert G > 0, 'the shape does not take the row-split loop'
names = ['weight prefetch, cp loads, x gather: loads issued -> no sentinel left (THE x HOP as this wave sees it)', 'y = x + step -> LDS, barrier (the slowest wave\'s gather)',
         f'dilated conv, {96 * 4 // (G if G >= 8 else 4) * (2 if G == 2 else 1)} MFMAs per wave, + drain', 'K partials through LDS, gate, gate rows stored',
         'gate gather: loads issued -> no sentinel left (THE GATE HOP)', 'gate tile -> LDS, barrier', f'output projection, {32 * 4 // (G if G >= 8 else 4) * (2 if G == 2 else 1)} MFMAs per wave, + drain',
         'K partials, x\' / skip update, x\' rows stored']
summary = {'shape': {'B': B, 'T': T, 'G': G}, 'phases': {}}
clock_ghz = None
for phase in (43, 44, 65):
    ts = eng.loop_timeline(x.clone(), noise, K, phase).astype(np.int64)
    ts = ts[ts[:, 0, 0] > 0]                      # workgroups of the grid padding have no stamps
    d = np.diff(ts[:, :, :9], axis=2)
    print(f'{B} x {T}, G = {G}: phase {phase} (layer {phase % 20}), {ts.shape[0]} workgroups x 4 waves, shader-clock ticks')
    row = {}
    for i, n in enumerate(names):
        print('  %-100s: median %7.0f  mean %7.0f  min %7.0f  max %7.0f' % (n, np.median(d[:, :, i]), d[:, :, i].mean(), d[:, :, i].min(), d[:, :, i].max()))
        row[n.split(':')[0][:40]] = float(np.median(d[:, :, i]))
    tot = ts[:, :, 8] - ts[:, :, 0]
    print('  layer total: median %.0f ticks; start skew across workgroups %.0f' % (np.median(tot), ts[:, :, 0].max() - ts[:, :, 0].min()))
    row['layer_total'] = float(np.median(tot))
    summary['phases'][str(phase)] = row
print('timeouts', eng.loop_timeouts())
if len(sys.argv) > 2:
    json.dump(summary, open(sys.argv[2], 'w'), indent=1)
