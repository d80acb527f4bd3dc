"""CPU: lane-level numpy model of the split-precision convolution prototype (csrc/dsd_split.hpp k_split_conv + the host packing of
diffsinger_amd/experimental.py) - the staging into [plane][frame][channel], the fragment-order weight planes, the six plane products per
chunk, the accumulator map - against F.conv1d.  The only hardware fact it assumes is the one every MFMA kernel here relies on (A row /
B column = lane & 31, C/D fragment map); the k order inside an instruction cancels because packing and LDS read use the same one."""
import numpy as np
import torch
import torch.nn.functional as F

from diffsinger_amd.experimental import pack_split_weight, split3

RS, FR = 264, 48


def bf16_to_f32(a_i16):
    return (a_i16.astype(np.uint16).astype(np.uint32) << 16).view(np.float32)


def frag_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def model(x, wp, T, dil, t0):
    """One workgroup: x [256][TS] fp32, wp int16 [4][48][4][3][64][8]; returns out [512][32]."""
    TS = x.shape[1]
    ysm = np.full((3, FR, RS), np.nan, np.float32)
    for c in range(256):
        for g in range(FR // 4):
            t = t0 - 8 + 4 * g
            v = x[c, t:t + 4] if 0 <= t < TS else np.zeros(4, np.float32)
            planes = split3(torch.from_numpy(np.ascontiguousarray(v)))
            for pl in range(3):
                ysm[pl, 4 * g:4 * g + 4, c] = planes[pl].float().numpy()
    out = np.zeros((512, 32), np.float32)
    TI, TJ = [0, 1, 2, 0, 1, 0], [2, 1, 0, 1, 0, 0]
    for w in range(4):
        acc = np.zeros((4, 32, 32), np.float32)                                     # [mb][row i][col j]
        for kc in range(48):
            k16, tap = kc // 3, kc % 3
            A = np.zeros((4, 3, 32, 16), np.float32)
            Bm = np.zeros((3, 16, 32), np.float32)
            for lane in range(64):
                i, h = lane & 31, lane >> 5
                for mb in range(4):
                    for pl in range(3):
                        A[mb, pl, i, 8 * h:8 * h + 8] = bf16_to_f32(wp[w, kc, mb, pl, lane])
                frow = i + 8 + (tap - 1) * dil
                for pl in range(3):
                    Bm[pl, 8 * h:8 * h + 8, i] = ysm[pl, frow, 16 * k16 + 8 * h:16 * k16 + 8 * h + 8]
            for q in range(6):
                for mb in range(4):
                    acc[mb] = (acc[mb] + (A[mb, TI[q]] @ Bm[TJ[q]]).astype(np.float32)).astype(np.float32)
        for mb in range(4):
            out[128 * w + 32 * mb:128 * w + 32 * mb + 32] = acc[mb]
    assert not np.isnan(out).any()
    return out


def test_split_conv_model_matches_conv1d():
    g = torch.Generator().manual_seed(3)
    T, TS, dil = 70, 96, 4
    w = torch.randn(512, 256, 3, generator=g) * (256 * 3) ** -0.5
    x = torch.zeros(256, TS)
    x[:, :T] = torch.randn(256, T, generator=g) * 1.5
    wp = pack_split_weight(w).numpy()
    assert wp.shape == (4, 48, 4, 3, 64, 8)
    want = F.conv1d(x[None, :, :T].double(), w.double(), None, padding=dil, dilation=dil)[0].numpy()
    for t0 in (0, 64):
        got = model(x.numpy(), wp, T, dil, t0)
        n = min(32, T - t0)
        err = np.abs(got[:, :n] - want[:, t0:t0 + n]).max()
        fp32 = F.conv1d(x[None, :, :T], w, None, padding=dil, dilation=dil)[0].numpy()
        err32 = np.abs(fp32[:, t0:t0 + n] - want[:, t0:t0 + n]).max()
        print('t0', t0, 'split-model err vs fp64', err, 'fp32 conv err vs fp64', err32)
        assert err < 2 * err32 and err < 1e-5                             # fp32-class: no worse than torch's own fp32 convolution


def test_planes_are_exact_and_packing_is_a_permutation():
    w = torch.randn(512, 256, 3)
    a, b, c = split3(w)
    assert torch.equal(a.double() + b.double() + c.double(), w.double())
    wp = pack_split_weight(w)
    assert wp.dtype == torch.int16 and wp.numel() == 3 * w.numel()
    # spot check: wave 2, chunk (g=5, tap=1), row block 3, plane 0, lane (i=7, h=1), element 4
    v = bf16_to_f32(wp[2, 5 * 3 + 1, 3, 0, 32 + 7, 4].numpy().reshape(1))[0]
    assert v == a[128 * 2 + 32 * 3 + 7, 16 * 5 + 8 + 4, 1].float().item()
