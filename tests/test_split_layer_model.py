"""CPU: lane-level numpy model of the EXPERIMENTAL split-precision residual layer (csrc/dsd_split.hpp: k_pack_a -> k_pack_split weight
chain, frame-major bf16-plane staging of y with its halo and masks, the conv and output-projection plane GEMMs with their B addressing,
the gate written as planes, row ownership of the waves) against the oracle's residual layer (= the reference's arithmetic).  What is
NOT modelled is what k_layer_split copies verbatim from the verified k_layer: the fragment-order cp / skip images and the x' epilogue.
The GPU tests of the kernel are tests/test_gpu_split_layer.py; this is the desk check that precedes them."""
import math

import numpy as np
import torch

from oracle import diffnet_oracle as O

RS, YF = 264, 48
TI, TJ = [0, 1, 2, 0, 1, 0], [2, 1, 0, 1, 0, 0]


def frag_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def bf16_round(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).bfloat16().float().numpy()


def split3(x):
    x = np.asarray(x, np.float32)
    a = bf16_round(x)
    r1 = (x - a).astype(np.float32)
    b = bf16_round(r1)
    return a, b, bf16_round((r1 - b).astype(np.float32))


def conv_chunk(k8, tap, ntap):
    """csrc/dsd_kernels.hpp conv_chunk: the dilated conv's chunk order is centre tap first; the projections have one tap."""
    if ntap != 3:
        return ntap * k8 + tap
    return k8 if tap == 1 else 32 + 2 * k8 + (tap >> 1)


def pack_a_split_rows(W, ntap):
    """k_pack_a(nw=4, ntap, nkc=32, nmb=4, split=1, hi_base=256, centre_first): fp32 [w][chunk8 = conv_chunk(k8, tap)][mb][lane][4]."""
    nk8 = 32
    out = np.zeros((4, ntap * nk8, 4, 64, 4), np.float32)
    for w in range(4):
        for mb in range(4):
            rows = (64 * w + 32 * mb if mb < 2 else 256 + 64 * w + 32 * (mb - 2)) + np.arange(32)
            for k8 in range(nk8):
                for tap in range(ntap):
                    for h in range(2):
                        out[w, conv_chunk(k8, tap, ntap), mb, 32 * h:32 * h + 32, :] = W[rows][:, 8 * k8 + 4 * h:8 * k8 + 4 * h + 4, tap]
    return out


def pack_split(src, ng, ntap):
    """k_pack_split, index for index: planes [w][chunk16][mb][pl][lane][e]."""
    nw = 4
    dst = np.zeros((nw, ng * ntap, 4, 3, 64, 8), np.float32)
    flat = src.reshape(-1)
    for w in range(nw):
        for c16 in range(ng * ntap):
            g, tap = c16 // ntap, c16 % ntap
            for mb in range(4):
                for lane in range(64):
                    i, hp = lane & 31, lane >> 5
                    for e in range(8):
                        k8 = 2 * g + hp
                        c8 = conv_chunk(k8, tap, ntap)
                        lane_src, s = i + 32 * (e >> 2), e & 3
                        v = flat[((((w * (2 * ng * ntap) + c8) * 4 + mb) * 64 + lane_src) * 4) + s]
                        p0, p1, p2 = split3(np.array([v], np.float32))
                        dst[w, c16, mb, :, lane, e] = (p0[0], p1[0], p2[0])
    return dst


def plane_gemm(acc, A, Bm):
    """acc [mb][32][32] += six plane products; A [mb][pl][32][16], Bm [pl][16][32] - fp32 accumulate."""
    for q in range(6):
        for mb in range(A.shape[0]):
            acc[mb] = (acc[mb] + (A[mb, TI[q]] @ Bm[TJ[q]]).astype(np.float32)).astype(np.float32)


def layer_model(xt, xl, xr, have_l, have_r, ds, cp, w1s, w2s, T, t0, dil):
    """One workgroup.  xt / xl / xr: this tile and its neighbours [256][32]; ds [256] step projection; cp [512][32] hoisted conditioner
    projection (+ both biases); returns (res [256][32] without bias, skip [256][32] without bias)."""
    yp = np.full((3, YF, RS), np.nan, np.float32)
    for tid in range(256):
        c4 = tid & 7
        for sl in range(8):
            row = sl * 32 + (tid >> 3)
            for e in range(4):
                t = t0 + 4 * c4 + e
                yv = np.float32(xt[row, 4 * c4 + e] + ds[row]) if t < T else np.float32(0)
                yp[:, 8 + 4 * c4 + e, row] = [p[0] for p in split3(np.array([yv], np.float32))]
        hpart = tid & 3
        hleft = hpart < 2
        hhave = have_l if hleft else have_r
        for q in range(4):
            row = 64 * q + (tid >> 2)
            src = xl[row, 24 + 4 * hpart:24 + 4 * hpart + 4] if hleft else xr[row, 4 * (hpart - 2):4 * (hpart - 2) + 4]
            tb = t0 - 8 + 4 * hpart if hleft else t0 + 32 + 4 * (hpart - 2)
            fb = 4 * hpart if hleft else 8 + 32 + 4 * (hpart - 2)
            for e in range(4):
                yv = np.float32(src[e] + ds[row]) if (hhave and tb + e < T) else np.float32(0)
                yp[:, fb + e, row] = [p[0] for p in split3(np.array([yv], np.float32))]
    gp = np.full((3, 32, RS), np.nan, np.float32)
    accs = []
    for w in range(4):
        acc = np.zeros((4, 32, 32), np.float32)
        for kc in range(48):
            g, tap = kc // 3, kc % 3
            A = np.zeros((4, 3, 32, 16), np.float32)
            Bm = np.zeros((3, 16, 32), np.float32)
            for lane in range(64):
                i, h = lane & 31, lane >> 5
                A[:, :, i, 8 * h:8 * h + 8] = w1s[w, kc, :, :, lane, :]
                frow = i + 8 + (tap - 1) * dil
                Bm[:, 8 * h:8 * h + 8, i] = yp[:, frow, 16 * g + 8 * h:16 * g + 8 * h + 8]
            plane_gemm(acc, A, Bm)
        accs.append(acc)
        # gate: row blocks 0,1 gates (rows 64 w + 32 pr + i), 2,3 their filters (256 + ...)
        for pr in range(2):
            ga = acc[pr] + cp[64 * w + 32 * pr:64 * w + 32 * pr + 32]
            fa = acc[pr + 2] + cp[256 + 64 * w + 32 * pr:256 + 64 * w + 32 * pr + 32]
            gt = (1 / (1 + np.exp(-ga.astype(np.float64))) * np.tanh(fa.astype(np.float64))).astype(np.float32)       # [row i][frame j]
            for lane in range(64):
                j, h = lane & 31, lane >> 5
                for rg in range(4):
                    ch = 64 * w + 32 * pr + 8 * rg + 4 * h
                    vals = np.array([gt[frag_row(4 * rg + s, h), j] for s in range(4)], np.float32)
                    p0, p1, p2 = split3(vals)
                    gp[0, j, ch:ch + 4], gp[1, j, ch:ch + 4], gp[2, j, ch:ch + 4] = p0, p1, p2
    assert not np.isnan(gp[:, :, :256]).any()
    res, skip = np.zeros((256, 32), np.float32), np.zeros((256, 32), np.float32)
    for w in range(4):
        acc2 = np.zeros((4, 32, 32), np.float32)
        for kc in range(16):
            A = np.zeros((4, 3, 32, 16), np.float32)
            Bm = np.zeros((3, 16, 32), np.float32)
            for lane in range(64):
                i, h = lane & 31, lane >> 5
                A[:, :, i, 8 * h:8 * h + 8] = w2s[w, kc, :, :, lane, :]
                Bm[:, 8 * h:8 * h + 8, i] = gp[:, i, 16 * kc + 8 * h:16 * kc + 8 * h + 8]
            plane_gemm(acc2, A, Bm)
        for mb in range(2):
            res[64 * w + 32 * mb:64 * w + 32 * mb + 32] = acc2[mb]
            skip[64 * w + 32 * mb:64 * w + 32 * mb + 32] = acc2[mb + 2]
    assert not np.isnan(res).any() and not np.isnan(skip).any()
    return res, skip


def test_split_layer_model_matches_the_reference_layer():
    torch.manual_seed(0)
    cfg = O.NetConfig(80, 256, 256, 2, 4)                            # two layers, dilations 1, 2
    p = O.init_diffnet_params(cfg, 7, 0.02)
    l, dil = 1, cfg.dilation(1)
    assert dil == 2
    T, TS = 70, 96
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 256, T, generator=g)
    cond = torch.randn(1, 256, T, generator=g)
    d_emb = torch.randn(1, 256, generator=g)
    with torch.no_grad():
        want_x, want_skip = O.residual_layer(p, cfg, l, x, cond, d_emb)
        pre = f'residual_layers.{l}.'
        dsv = torch.nn.functional.linear(d_emb, p[pre + 'diffusion_projection.weight'], p[pre + 'diffusion_projection.bias'])[0].numpy()
        cpf = (torch.nn.functional.conv1d(cond, p[pre + 'conditioner_projection.weight'], p[pre + 'conditioner_projection.bias'])[0]
               + p[pre + 'dilated_conv.bias'][:, None]).numpy()
    w1 = p[pre + 'dilated_conv.weight'].numpy()                      # [512][256][3]
    w2 = p[pre + 'output_projection.weight'].numpy()                 # [512][256][1]
    w1s = pack_split(pack_a_split_rows(w1, 3), 16, 3)
    w2s = pack_split(pack_a_split_rows(w2, 1), 16, 1)
    b2 = p[pre + 'output_projection.bias'].numpy()
    xpad = np.zeros((256, TS), np.float32)
    xpad[:, :T] = x[0].numpy()
    cpp = np.zeros((512, TS), np.float32)
    cpp[:, :T] = cpf
    tiles = [xpad[:, 32 * k:32 * k + 32] for k in range(3)]
    worst = 0.0
    for tn in range(3):
        xl = tiles[tn - 1] if tn > 0 else tiles[0]
        xr = tiles[tn + 1] if tn < 2 else tiles[0]
        res, skip = layer_model(tiles[tn], xl, xr, tn > 0, tn < 2, dsv, cpp[:, 32 * tn:32 * tn + 32], w1s, w2s, T, 32 * tn, dil)
        n = min(32, T - 32 * tn)
        xo = (tiles[tn] + (res + b2[:256, None])) * np.float32(1.0 / 1.41421354)
        sk = skip + b2[256:, None]
        worst = max(worst, np.abs(xo[:, :n] - want_x[0, :, 32 * tn:32 * tn + n].numpy()).max(),
                    np.abs(sk[:, :n] - want_skip[0, :, 32 * tn:32 * tn + n].numpy()).max())
    print('split layer model: worst |x_out|, |skip| error vs the oracle layer', worst)
    assert worst < 5e-6
    assert abs(1.0 / 1.41421354 - 1 / math.sqrt(2)) < 1e-7
