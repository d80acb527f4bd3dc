"""Developer tool (GPU): time the residual-layer kernel alone over batch shapes / tile sizes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

F = bench.F_LAYER_EXEC
dev = torch.device('cuda', 0)
gd, pre = bench.build_model(dev)
for (B, T) in [(8, 1024), (16, 1024), (32, 1024), (64, 2048)]:
    cond = torch.randn(B, T, 256, device=dev).transpose(1, 2)
    eng = gd._engine(cond)
    for tile in (32, 64):
        eng.set_layer_tile(tile)
        eng.prepare(cond)
        ms = eng.time_layer_kernel(3, 50, 100)
        ms_last = eng.time_layer_kernel(19, 50, 100)
        print(json.dumps({'B': B, 'T': T, 'tile': tile, 'layer_ms': round(ms, 4), 'tflops': round(B * T * F / ms / 1e9, 2),
                          'last_layer_ms': round(ms_last, 4)}), flush=True)
