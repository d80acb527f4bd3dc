"""Developer tool (GPU): where a k_voc_chain workgroup spends its time - s_memtime stamps per convolution and wave (dsv_debug_chain_timeline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from diffsinger_amd import _lib
from diffsinger_amd.vocoder import HifiGanGenerator, set_chain_mode

set_chain_mode('resblock')        # the merged launches of the default mode (MG instantiations) carry no stamps: one launch per resblock here

dev = torch.device('cuda', 0)
m = HifiGanGenerator(bench.VOC_CONFIG)
m.remove_weight_norm()
m = m.to(dev).eval()
m(torch.zeros(1, 80, 4, device=dev))
lib = _lib.load()
B, T = 8, 1024
for stage in (1, 2, 3):
    C = 128 >> (stage + 1)
    L = T * [8, 64, 128, 256][stage]
    x = torch.randn(B, C, L, device=dev)
    m._stage_resblocks(stage, x, L)
    buf = torch.zeros(18 * 4 * 4, dtype=torch.int64, device=dev)
    lib.dsv_debug_chain_timeline(buf.data_ptr())
    m._stage_resblocks(stage, x, L)
    torch.cuda.synchronize()
    lib.dsv_debug_chain_timeline(None)
    ts = buf.cpu().numpy().reshape(18, 4, 4).astype(np.int64)
    t0 = ts[0, :, 0].min()
    print(f'--- stage {stage}: {C} channels; one workgroup, s_memtime ticks, mean over the 4 waves')
    tot = ts[-1, :, 3].max() - t0
    for n in range(18):
        gemm = ts[n, :, 1] - ts[n, :, 0]
        epi = ts[n, :, 2] - ts[n, :, 1]
        bar = ts[n, :, 3] - ts[n, :, 2]
        gap = (ts[n + 1, :, 0] - ts[n, :, 3]) if n < 17 else np.zeros(4, dtype=np.int64)
        print(f'  conv {n:2d}: contraction {gemm.mean():9.0f}  epilogue {epi.mean():8.0f}  barrier wait {bar.mean():8.0f}  to next conv {gap.mean():8.0f}')
    print(f'  total {tot} ticks; sum contraction {sum((ts[n, 0, 1] - ts[n, 0, 0]) for n in range(18))}')
