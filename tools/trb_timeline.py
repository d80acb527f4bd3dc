"""Developer tool (GPU): where a k_trb_fused_w workgroup spends its time - s_memtime stamps per wave (dsf_debug_trb_timeline), layer 1 of an
8 x 1024 training step."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsinger_amd import _lib
import tools.bench_train as BT

NAMES = ['kernel start -> first quarter of the da tile staged, barrier', 'Winograd half 0 (M1, M2: 8 periods) + output transform',
         'half 1 (M0, M3: 8 periods)', 'K-half exchange, dy -> LDS tile, epilogue rows (dx store, row sums, skip rows), barrier',
         'output-projection data gradient (32 chunks)', 'K-half exchange', 'gate derivative + 96 stores per lane']

if __name__ == '__main__':
    lib = _lib.load()
    lib.dsf_debug_trb_timeline.argtypes = [C.c_void_p]
    buf = torch.zeros(256 * 4 * 8, dtype=torch.int64, device='cuda')
    lib.dsf_debug_trb_timeline(buf.data_ptr())
    BT.run(8, 1024, 2)
    torch.cuda.synchronize()
    lib.dsf_debug_trb_timeline(None)
    st = buf.cpu().view(256, 4, 8).double()
    d = st[:, :, 1:] - st[:, :, :-1]
    print('k_trb_fused_w<false, true>, layer 1 of an 8 x 1024 step: 256 workgroups x 4 waves, s_memtime ticks (100 MHz), mean / min / max')
    for i, n in enumerate(NAMES):
        print(f'  {n:110s}: {d[:, :, i].mean():8.1f} {d[:, :, i].min():8.0f} {d[:, :, i].max():8.0f}')
    tot = st[:, :, 7] - st[:, :, 0]
    print(f'  stamp 0 -> 7: mean {tot.mean():.1f} ticks = {tot.mean() / 100:.2f} us; start skew across workgroups {(st[:, 0, 0].max() - st[:, 0, 0].min()) / 100:.2f} us; '
          f'first start -> last end {(st[:, :, 7].max() - st[:, :, 0].min()) / 100:.2f} us')
