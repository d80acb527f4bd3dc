"""Developer tool (GPU): run ONLY the residual-layer kernel at the bench shape, for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 0
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device('cuda', 0)
gd, pre = bench.build_model(dev)
cond = torch.randn(B, T, 256, device=dev).transpose(1, 2)
eng = gd._engine(cond)
eng.set_layer_tile(tile)
eng.prepare(cond)
print('layer ms', eng.time_layer_kernel(3, 50, iters))
