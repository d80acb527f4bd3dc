"""Developer tool (GPU): the convolution's weight gradient of the fused training stack as the Winograd dual (dsf_set_wgrad_dual(1)) against the three
tap products (0), tap by tap and layer by layer, for a list of shapes - the bisect of the dual's staging (which tap, which frames).
    python tools/diag_wgrad_dual.py B,T,L,cycle [...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffsinger_amd import fs2, train_fused
from tests.test_gpu_train_fused import _make_stack

dev = torch.device('cuda', 0)
for spec in sys.argv[1:] or ['3,96,5,1', '2,50,3,4', '2,96,2,4', '1,32,1,1', '2,160,2,1']:
    B, T, L, cycle = (int(v) for v in spec.split(','))
    ws, dils = _make_stack(L, cycle, seed=L + T)
    g = torch.Generator().manual_seed(5 + T)
    x0 = torch.relu(torch.randn(B, 256, T, generator=g)); cond = torch.randn(B, 256, T, generator=g)
    step = torch.randn(B, L, 256, generator=g) * 0.5; dskip = torch.randn(B, 256, T, generator=g)
    TS = fs2.padded_frames(T)
    pad = lambda t: F.pad(t, (0, TS - T)).to(dev).contiguous()
    order = ['dc_w', 'dc_b', 'cp_w', 'cp_b', 'op_w', 'op_b']
    grads = {}
    for dual in (0, 1):
        train_fused.set_wgrad_dual(bool(dual))
        x0d, condd = pad(x0).requires_grad_(True), pad(cond).requires_grad_(True)
        stepd = step.to(dev).requires_grad_(True)
        wd = [t.to(dev).requires_grad_(True) for k in order for t in ws[k]]
        skip = train_fused._ResidualStack.apply(x0d, condd, stepd, T, dils, {}, *wd)
        skip.backward(pad(dskip))
        torch.cuda.synchronize()
        grads[dual] = [w.grad.cpu() for w in wd]
    train_fused.set_wgrad_dual(True)
    for l in range(L):
        a, b = grads[0][l], grads[1][l]           # dc_w [512][256][3]
        rel = lambda x, y: float((x - y).abs().max() / max(float(y.abs().max()), 1e-12))
        per_tap = [rel(b[:, :, k], a[:, :, k]) for k in range(3)]
        per_mt = [rel(b[128 * m:128 * m + 128], a[128 * m:128 * m + 128]) for m in range(4)]
        print(f'B={B} T={T} layer {l} d={dils[l]}: dual vs taps per tap {["%.1e" % v for v in per_tap]} per row tile {["%.1e" % v for v in per_mt]} bias {rel(grads[1][L + l], grads[0][L + l]):.1e}')
