"""Developer tool (GPU): phase timeline of the layer kernel from in-kernel s_memtime stamps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device('cuda', 0)
gd, pre = bench.build_model(dev)
cond = torch.randn(B, T, 256, device=dev).transpose(1, 2)
eng = gd._engine(cond); eng.set_layer_tile(tile); eng.prepare(cond)
for layer in (3, 19):
    ts = eng.layer_timeline(layer, 50).astype(np.int64)
    t0 = ts[:, :, 0].min()
    rel = ts[:, :, :6] - t0
    names = ['start', 'staged', 'conv', 'gate', 'outproj', 'end']
    d = np.diff(rel, axis=2)
    print(f'layer {layer} B={B} T={T} tile={tile}: blocks={ts.shape[0]}  (shader-clock ticks; s_memtime)')
    print('  start skew   : mean %.0f max %.0f' % (rel[:, :, 0].mean(), rel[:, :, 0].max()))
    for i, n in enumerate(['stage(cp+x->LDS,barrier)', 'conv K=768', 'gate+barrier', 'outproj K=256', 'epilogue']):
        print('  %-26s: mean %8.0f  min %8.0f  max %8.0f' % (n, d[:, :, i].mean(), d[:, :, i].min(), d[:, :, i].max()))
    print('  stage split   : loads issued %.0f, last LDS write %.0f, barrier released %.0f (cycles after start, mean)' % ((ts[:, :, 6] - ts[:, :, 0]).mean(), (ts[:, :, 7] - ts[:, :, 0]).mean(), (ts[:, :, 1] - ts[:, :, 0]).mean()))
    print('  wave lifetime : mean %.0f ; last end %.0f' % ((rel[:, :, 5] - rel[:, :, 0]).mean(), rel[:, :, 5].max()))
    print('  layer ms (events):', eng.time_layer_kernel(layer, 50, 50))
