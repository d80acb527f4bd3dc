"""Developer tool (GPU): run ONLY the persistent K-step loop at the bench shape, for rocprofv3 --pmc passes over k_loop
(`--split`: the labelled split-precision loop k_loop_split instead)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else 3
dev = torch.device('cuda', 0)
gd, pre = bench.build_model(dev)
B, T, K = bench.B_PER_GPU, bench.T_FRAMES, bench.K_STEPS
g = torch.Generator(device=dev).manual_seed(1)
cond = torch.randn(B, T, 256, device=dev, generator=g).transpose(1, 2)
x = torch.randn(B, 80, T, device=dev, generator=g)
noise = torch.randn(K, B, 80, T, device=dev, generator=g)
eng = gd._engine(cond)
if '--split' in sys.argv:
    eng.set_split_mode(True)
assert eng.loop_mode() == 1
for _ in range(reps):
    eng.sample_ddpm(x.clone(), noise, K)
torch.cuda.synchronize()
print('timeouts', eng.loop_timeouts())
