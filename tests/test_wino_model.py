"""CPU model of the index algebra of the Winograd F(2,3) loop (csrc/dsd_loop_wino.hpp): the transformed-weight stream of k_pack_wino, the
pair-ordered y tile, the per-lane operand reads and input transforms, the fragment maps of v_mfma_f32_16x16x4_f32, the two halves of the
contraction with the output transform between them, the gate's (channel, frame) of every accumulator register and k_condproj's
Winograd-order output (the accumulators' initial values) - all restated lane by lane in numpy and compared with the reference's own operator (torch conv1d, usr/diff/net.py:61,71)
for every dilation the kernel supports, with halo frames and a ragged tail.  A wrong shift, row, lane group or sign anywhere shows here, before
the kernel ever reaches the GPU."""
import numpy as np
import pytest
import torch

C = 256
LDK = 260
OBASE = 24 * LDK + 16
NY = OBASE + 24 * LDK
STEPS = 128


def row_of_frame(j, e):                      # wn_row_of_frame
    d = 1 << e
    hf = (j >> e) & 1
    p = ((j >> (e + 1)) << e) | (j & (d - 1))
    return OBASE + (8 + p) * LDK if hf else p * LDK


def frame_of_pair(p, e):                     # wn_frame_of_pair
    d = 1 << e
    return ((p >> e) << (e + 1)) | (p & (d - 1))


def pack_wino(wt):
    """k_pack_wino: wt [2C][C][3] -> stream [step 128][w 4][r4 4][lane 64][s 4]"""
    out = np.zeros((STEPS, 4, 4, 64, 4), np.float32)
    w64 = wt.astype(np.float64)
    for st in range(STEPS):
        hb, pos, c, half = st & 1, (st >> 1) & 1, (st >> 2) & 15, st >> 6
        for w in range(4):
            for r4 in range(4):
                rb = 4 * hb + r4
                for lane in range(64):
                    nn, g = lane & 15, lane >> 4
                    row = 64 * w + 16 * rb + nn if rb < 4 else C + 64 * w + 16 * (rb - 4) + nn
                    ch = 64 * g + 4 * c + np.arange(4)
                    g0, g1, g2 = w64[row, ch, 0], w64[row, ch, 1], w64[row, ch, 2]
                    if half == 0:
                        u = 0.5 * (g0 - g1 + g2) if pos else 0.5 * (g0 + g1 + g2)
                    else:
                        u = g2 if pos else g0
                    out[st, w, r4, lane] = u.astype(np.float32)
    return out


def mfma16(a, b, acc):
    """v_mfma_f32_16x16x4_f32 for a wave: a, b [64] (lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]); acc [64][4]:
    lane l, register r = D[i = 4 (l >> 4) + r][j = l & 15]"""
    A = a.reshape(4, 16).T.astype(np.float64)            # [i][k]
    B = b.reshape(4, 16).astype(np.float64)              # [k][j]
    D = A @ B                                            # [i][j]
    lanes = np.arange(64)
    for r in range(4):
        acc[:, r] += D[4 * (lanes >> 4) + r, lanes & 15]


def run_model(e, wt, y_ext, cp):
    """y_ext [C][48]: frames -8 .. 39 of y (already zero where the conv pads); returns the gate pre-activation a[2C][32] the kernel's lanes hold,
    scattered back to (row, frame) through the kernel's own maps."""
    d = 1 << e
    ytile = np.full(NY, np.nan, np.float32)
    # own frames: lane (j, h) of wave w writes channels 64 w + 4 h + 32 mb + 8 q .. + 3 of frame j to the frame's row
    for j in range(32):
        ytile[row_of_frame(j, e):row_of_frame(j, e) + C] = y_ext[:, 8 + j]
    # halo rows: left frame f -> O[f - 8] (= OBASE + f rows), right frame f -> E[16 + f]
    for f in range(8):
        ytile[OBASE + f * LDK:OBASE + f * LDK + C] = y_ext[:, f]
        ytile[(16 + f) * LDK:(16 + f) * LDK + C] = y_ext[:, 40 + f]
    stream = pack_wino(wt)
    lanes = np.arange(64)
    pp, gg = lanes & 15, lanes >> 4
    pE = pp * LDK + 64 * gg
    pO = OBASE + (8 + pp) * LDK + 64 * gg
    out = np.zeros((2 * C, 32), np.float64)
    for w in range(4):
        acc = cp[w].astype(np.float64).copy()            # the accumulator sets start from k_condproj's halves of the conditioner projection
        for st in range(STEPS):
            hb, pos, c, half = st & 1, (st >> 1) & 1, (st >> 2) & 15, st >> 6
            if st == 64:                                 # output transform between the halves
                m1, m2 = acc[0].copy(), acc[1].copy()
                acc[0], acc[1] = m1 + m2, m1 - m2
            o = 4 * c
            for s in range(4):
                if half == 0:
                    r0, r1 = ytile[pE + o + s], ytile[pO + o + s]
                    v = (r0 + r1) if pos == 0 else (r1 - r0)
                else:
                    v = (ytile[pO - d * LDK + o + s] - ytile[pO + o + s]) if pos == 0 else (ytile[pE + d * LDK + o + s] - ytile[pE + o + s])
                assert not np.isnan(v).any(), 'an operand row that nobody wrote'
                for r4 in range(4):
                    mfma16(stream[st, w, r4, :, s], v.astype(np.float32), acc[pos][4 * hb + r4])
        # gate mapping: lane (p, g), acc[hf][rb][r] = row (gate rb < 4 / filter) 64 w + 16 (rb & 3) + 4 g + r, frame tE + hf d
        tE = np.array([frame_of_pair(p, e) for p in pp])
        for hf in range(2):
            for rb in range(8):
                for r in range(4):
                    rows = (0 if rb < 4 else C) + 64 * w + 16 * (rb & 3) + 4 * gg + r
                    out[rows, tE + hf * d] += acc[hf][rb][:, r]
    return out


def condproj_wino(e, cpfull):
    """k_condproj's Winograd-order output for one tile and layer: the projection cpfull [2C][32] as the INITIAL VALUES of the loop's two
    accumulator sets - column c < 16 of its (transformed) result is (cp[tE(c)] + cp[tO(c)]) / 2, column 16 + c the half difference (computed
    there by linearity from the transformed conditioner tile; restated here on the result) -> [w 4][set 2][rb 8][lane 64][4]"""
    d = 1 << e
    tE = np.array([frame_of_pair(p, e) for p in range(16)])
    R = np.concatenate([0.5 * (cpfull[:, tE] + cpfull[:, tE + d]), 0.5 * (cpfull[:, tE] - cpfull[:, tE + d])], axis=1).astype(np.float32)
    out = np.full((4, 2, 8, 64, 4), np.nan, np.float32)
    for w in range(4):
        for lane in range(64):
            j, h = lane & 31, lane >> 5
            i, pr = j >> 4, j & 15
            for mb in range(4):
                for q in range(4):
                    rows = (0 if mb < 2 else C) + 64 * w + 32 * (mb & 1) + 8 * q + 4 * h + np.arange(4)
                    rb = 2 * (mb & 1) + (q >> 1) + 4 * (mb >> 1)
                    g = 2 * (q & 1) + h
                    out[w, i, rb, pr + 16 * g] = R[rows, j]
    assert not np.isnan(out).any()
    return out


@pytest.mark.parametrize('e', [0, 1, 2, 3])
def test_winograd_loop_index_algebra_matches_conv1d(e):
    d = 1 << e
    g = torch.Generator().manual_seed(100 + e)
    wt = torch.randn(2 * C, C, 3, generator=g) * 0.05
    y = torch.randn(C, 48, generator=g)
    y[:, 8 + 29:] = 0.0                                  # a ragged tail: frames >= T are zero in y (the conv's zero padding applies to y)
    cpfull = torch.randn(2 * C, 32, generator=g)
    want = torch.nn.functional.conv1d(y[None], wt, dilation=d)[0]      # valid conv over frames -8 .. 39: output index i = frame i - 8 + d
    want = want[:, 8 - d:8 - d + 32] + cpfull
    got = run_model(e, wt.numpy(), y.numpy(), condproj_wino(e, cpfull.numpy()))
    err = np.abs(got - want.double().numpy()).max()
    assert err < 2e-5, err


def test_pair_order_covers_every_row_once_and_reads_stay_inside_the_tile():
    for e in range(4):
        d = 1 << e
        rows = sorted(row_of_frame(j, e) for j in range(32))
        want = sorted([p * LDK for p in range(16)] + [OBASE + (8 + p) * LDK for p in range(16)])
        assert rows == want
        assert sorted([frame_of_pair(p, e) for p in range(16)] + [frame_of_pair(p, e) + d for p in range(16)]) == list(range(32))
        for p in range(16):
            assert OBASE + (8 + p - d) * LDK >= OBASE and (p + d) * LDK + C <= 24 * LDK
    assert NY * 4 + 1024 + (32 * LDK + C * 32 + 2 * C) * 4 <= 160 * 1024


def test_lds_bank_groups_of_the_operand_reads_and_row_writes():
    """MI355X_MICROARCH.md section LDS: a ds_read_b128 is served in four groups of 16 lanes, conflict-free when the 16 addresses cover all 64
    banks; ds_write_b128 in eight groups of 8 contiguous lanes over 32 banks."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]
    for e in range(4):
        d = 1 << e
        for base in (0, OBASE + 8 * LDK, OBASE + (8 - d) * LDK, d * LDK):
            for c in range(16):
                for grp in groups:
                    banks = set()
                    for l in grp:
                        a = base + (l & 15) * LDK + 64 * (l >> 4) + 4 * c
                        banks |= {(a + k) % 64 for k in range(4)}
                    assert len(banks) == 64, (e, base, c)
        for h in range(2):
            for q in range(8):
                for g8 in range(4):
                    banks = set()
                    for j in range(8 * g8, 8 * g8 + 8):
                        a = row_of_frame(j, e) + 4 * h + 8 * q
                        banks |= {(a + k) % 32 for k in range(4)}
                    assert len(banks) == 32, (e, h, q, g8)


# ----------------------------------------------------------------------------------------------------------------------------------------
# the latency kernels' Winograd convolution (csrc/dsd_lat_wino.hpp): roles of the G x 4 waves of a tile, fragment offsets into the loop's
# stream, K partials, the direct-layout index of the conditioner projection, the gate tile's (channel, frame)
# ----------------------------------------------------------------------------------------------------------------------------------------
def cp_direct_layout(cpfull):
    """k_condproj's direct (32x32 fragment) order for one tile and layer: [w4][mb4][q4][lane64][4]; element (q, lane = (j, h), e) of (w, mb) is row
    (gate mb < 2 / filter) 64 w + 32 (mb & 1) + 8 q + 4 h + e at frame j"""
    out = np.zeros((4, 4, 4, 64, 4), np.float32)
    for w in range(4):
        for mb in range(4):
            for q in range(4):
                for lane in range(64):
                    j, h = lane & 31, lane >> 5
                    rows = (0 if mb < 2 else C) + 64 * w + 32 * (mb & 1) + 8 * q + 4 * h + np.arange(4)
                    out[w, mb, q, lane] = cpfull[rows, j]
    return out.reshape(-1, 4)


def run_lat_model(G, e, wt, y_ext, cpfull):
    d = 1 << e
    ytile = np.full(NY, np.nan, np.float32)
    for j in range(32):
        ytile[row_of_frame(j, e):row_of_frame(j, e) + C] = y_ext[:, 8 + j]
    for f in range(8):
        ytile[OBASE + f * LDK:OBASE + f * LDK + C] = y_ext[:, f]
        ytile[(16 + f) * LDK:(16 + f) * LDK + C] = y_ext[:, 40 + f]
    stream = pack_wino(wt).reshape(-1, 64, 4)            # 1 KiB pieces: [(step * 4 + w) * 4 + r4][lane][s]
    cpl = cp_direct_layout(cpfull)
    lanes = np.arange(64)
    pp, gg = lanes & 15, lanes >> 4
    NRB = 4 if G == 2 else 2
    NGB = NRB // 2
    NCW = 4 if G == 16 else 8 if G == 8 else 16
    tE = np.array([frame_of_pair(p, e) for p in pp])
    gate_pre = np.full((2 * C, 32), np.nan, np.float64)
    for g in range(G):
        parts = {}
        for wv in range(4):
            B0 = 8 * g + 2 * wv if G == 2 else 4 * g + wv if G == 4 else 2 * g + (wv & 1) if G == 8 else g
            c0 = 8 * (wv >> 1) if G == 8 else 4 * wv if G == 16 else 0
            pE = pp * LDK + 64 * gg + 4 * c0
            pO = OBASE + (8 + pp) * LDK + 64 * gg + 4 * c0
            acc = np.zeros((2, NRB, 64, 4), np.float64)
            for gi in range(2 * NCW):
                half = 1 if gi >= NCW else 0
                ci = gi - half * NCW
                c = c0 + ci
                if gi == NCW:
                    m1, m2 = acc[0].copy(), acc[1].copy()
                    acc[0], acc[1] = m1 + m2, m1 - m2
                o = 4 * ci
                for pos in range(2):
                    for s in range(4):
                        if half == 0:
                            r0, r1 = ytile[pE + o + s], ytile[pO + o + s]
                            v = (r0 + r1) if pos == 0 else (r1 - r0)
                        else:
                            v = (ytile[pO - d * LDK + o + s] - ytile[pO + o + s]) if pos == 0 else (ytile[pE + d * LDK + o + s] - ytile[pE + o + s])
                        assert not np.isnan(v).any()
                        for k in range(NRB):
                            B, f = B0 + (k % NGB), k // NGB
                            step = ((half * 16 + c) * 2 + pos) * 2 + f
                            piece = (step * 4 + (B >> 2)) * 4 + (B & 3)
                            mfma16(stream[piece, :, s], v.astype(np.float32), acc[pos][k])
            parts[wv] = (B0, acc)
        # K partials: G = 8: waves wv and wv ^ 2 share a block pair; G = 16: all four waves; G = 2 / 4: none
        for wv in range(4):
            B0, acc = parts[wv]
            if G == 8:
                if wv >= 2:
                    continue
                acc = parts[wv][1] + parts[wv + 2][1]
            elif G == 16:
                if wv > 0:
                    continue
                acc = ((parts[0][1] + parts[1][1]) + parts[2][1]) + parts[3][1]
            for hf in range(2):
                t = tE + hf * d
                for k in range(NRB):
                    B, f = B0 + (k % NGB), k // NGB
                    base = (((B >> 2) * 4 + ((B & 3) >> 1) + 2 * f) * 4 + 2 * (B & 1) + (gg >> 1)) * 64 + 32 * (gg & 1)
                    for r in range(4):
                        rows = (C if f else 0) + 16 * B + 4 * gg + r
                        gate_pre[rows, t] = acc[hf][k][:, r] + cpl[base + t, r]
    assert not np.isnan(gate_pre).any(), 'rows or frames nobody computed'
    return gate_pre


@pytest.mark.parametrize('G', [2, 4, 8, 16])
@pytest.mark.parametrize('e', [0, 3])
def test_latency_winograd_conv_index_algebra(G, e):
    d = 1 << e
    g = torch.Generator().manual_seed(200 + 10 * G + e)
    wt = torch.randn(2 * C, C, 3, generator=g) * 0.05
    y = torch.randn(C, 48, generator=g)
    y[:, 8 + 30:] = 0.0
    cpfull = torch.randn(2 * C, 32, generator=g)
    want = torch.nn.functional.conv1d(y[None], wt, dilation=d)[0][:, 8 - d:8 - d + 32] + cpfull
    got = run_lat_model(G, e, wt.numpy(), y.numpy(), cpfull.numpy())
    err = np.abs(got - want.double().numpy()).max()
    assert err < 2e-5, err


# ----------------------------------------------------------------------------------------------------------------------------------------
# the training kernels (csrc/train_loop_wino.hpp, train_wino_bwd.hpp): where the forward's accumulators land in the saved pre-activation (the
# 32x32 fragment order the backward reads), and the transposed convolution of the backward: its weight stream, the channel-major da tile, the
# wave roles (row half wr, K half wk), the exchange and the dy2 tile
# ----------------------------------------------------------------------------------------------------------------------------------------
def test_training_forward_saves_the_pre_activation_in_the_backward_fragment_order():
    """k_tr_stack_fwd_w::save_a_and_gate: lane (p, g), accumulator (hf, rb) -> float4 index of a_frag [w][mb4][q4][h2][j32]; k_trb_fused reads
    element (mb, q, lane = (j, h), e) as row (gate mb < 2 / filter) 32 (mb & 1) + 8 q + 4 h + e of the wave's 64, frame j."""
    for e in range(4):
        d = 1 << e
        seen = set()
        for lane in range(64):
            pp, gg = lane & 15, lane >> 4
            tE = frame_of_pair(pp, e)
            for hf in range(2):
                for rb in range(8):
                    idx = (((rb >> 2) * 2 + ((rb & 3) >> 1)) * 4 + 2 * (rb & 1) + (gg >> 1)) * 64 + (gg & 1) * 32 + tE + hf * d
                    mb, q, h, j = idx >> 8, (idx >> 6) & 3, (idx >> 5) & 1, idx & 31
                    # what the Winograd accumulator holds: rows 16 (rb & 3) + 4 g + {0..3} of the gate (rb < 4) / filter rows, frame tE + hf d
                    assert (mb >> 1) == (rb >> 2) and 32 * (mb & 1) + 8 * q + 4 * h == 16 * (rb & 3) + 4 * gg and j == tE + hf * d
                    seen.add(idx)
        assert seen == set(range(1024))                  # every float4 of the wave's 16 KiB written exactly once


def pack_wino_bwd(wt):
    """k_pack_wino_bwd_multi for one layer: wt [2C co][C ci][3] -> stream [step 128][w 4][r4 4][lane 64][s 4]"""
    out = np.zeros((STEPS, 4, 4, 64, 4), np.float32)
    w64 = wt.astype(np.float64)
    for st in range(STEPS):
        hb, pos, c, half = st & 1, (st >> 1) & 1, (st >> 2) & 15, st >> 6
        for w in range(4):
            wr, wk = w & 1, w >> 1
            for r4 in range(4):
                rb = 4 * hb + r4
                for lane in range(64):
                    nn, g = lane & 15, lane >> 4
                    ci = 128 * wr + 16 * rb + nn
                    co = 256 * wk + 16 * c + 4 * np.arange(4) + g
                    g0, g1, g2 = w64[co, ci, 2], w64[co, ci, 1], w64[co, ci, 0]
                    if half == 0:
                        u = 0.5 * (g0 - g1 + g2) if pos else 0.5 * (g0 + g1 + g2)
                    else:
                        u = g2 if pos else g0
                    out[st, w, r4, lane] = u.astype(np.float32)
    return out


@pytest.mark.parametrize('e', [0, 1, 2, 3])
def test_training_backward_index_algebra_matches_the_conv_adjoint(e):
    """k_trb_fused_w's convolution part lane by lane: dy[ci][t] = sum_co sum_tap W[co][ci][tap] da[co][t - (tap - 1) d] (the adjoint of net.py:71)
    from the da tile [512][48] (frames -8 .. 39), through WinoPipe<S, LD = 48>'s reads, the K halves' exchange and the dy2 tile."""
    d = 1 << e
    LD = 48
    g = torch.Generator().manual_seed(200 + e)
    wt = torch.randn(2 * C, C, 3, generator=g) * 0.05
    da = torch.randn(2 * C, LD, generator=g)
    y = torch.zeros(1, C, 48, dtype=torch.float64, requires_grad=True)
    # autograd of the forward conv over frames -8 .. 39 (zero padding outside does not matter for the 32 inner frames: the taps reach 8 at most)
    out = torch.nn.functional.conv1d(y, wt.double(), padding=d, dilation=d)
    out.backward(da.double()[None])
    want = y.grad[0, :, 8:40].numpy()
    stream = pack_wino_bwd(wt.numpy())
    tile = da.numpy()
    lanes = np.arange(64)
    pp, gg = lanes & 15, lanes >> 4
    tE = np.array([frame_of_pair(p, e) for p in pp])
    dy2 = np.full((C, 32), np.nan)
    fin = {}
    for w in range(4):
        wr, wk = w & 1, w >> 1
        acc = np.zeros((2, 8, 64, 4))
        rows_g = wk * C + gg                              # row g of the wave's K half
        for st in range(STEPS):
            hb, pos, c, half = st & 1, (st >> 1) & 1, (st >> 2) & 15, st >> 6
            if st == 64:
                m1, m2 = acc[0].copy(), acc[1].copy()
                acc[0], acc[1] = m1 + m2, m1 - m2
            for s in range(4):
                r = rows_g + 16 * c + 4 * s
                col = 8 + tE                              # the tile's frame 0 is column kHalo
                if half == 0:
                    r0, r1 = tile[r, col], tile[r, col + d]
                    v = (r0 + r1) if pos == 0 else (r1 - r0)
                else:
                    v = (tile[r, col - d] - tile[r, col + d]) if pos == 0 else (tile[r, col + 2 * d] - tile[r, col])
                for r4 in range(4):
                    mfma16(stream[st, w, r4, :, s], v.astype(np.float32), acc[pos][4 * hb + r4])
        fin[(wr, wk)] = acc
    for wr in range(2):
        for wk in range(2):                              # wave (wr, wk) finishes row blocks 4 wk .. 4 wk + 3: (K half 0) + (K half 1)
            for hf in range(2):
                for r4 in range(4):
                    v = fin[(wr, 0)][hf][4 * wk + r4] + fin[(wr, 1)][hf][4 * wk + r4]
                    for ee in range(4):
                        rows = 128 * wr + 64 * wk + 16 * r4 + 4 * gg + ee
                        assert np.isnan(dy2[rows, tE + hf * d]).all()
                        dy2[rows, tE + hf * d] = v[:, ee]
    assert not np.isnan(dy2).any()
    err = np.abs(dy2 - want).max()
    assert err < 2e-5, err


def test_training_backward_reads_stay_inside_the_da_tile():
    """the channel-major reads of WinoPipe<S, 48>: frames tE - d .. tE + 2 d of a 32-frame tile with 8 halo columns on both sides, rows of the wave's
    K half; the chunk the pipe requests BEHIND the last one (it runs one chunk ahead) lies inside the workgroup's LDS"""
    for e in range(4):
        d = 1 << e
        for p in range(16):
            tE = frame_of_pair(p, e)
            assert 8 + tE - d >= 0 and 8 + tE + 2 * d < 48
    assert (2 * C + 16) * 48 * 4 <= (2 * C * 48 + 4 * 32 * 64) * 4      # chunk 16 of K half 1 = rows 512 .. 527: the exchange buffer behind the tile


# ------------------------------------------------------------------------------------------------------------------------------------------
# the Winograd F(2,3) DUAL of the convolution's weight gradient (csrc/train_kernels.hpp k_tr_wgrad, loop_dual / k_tr_wgrad_reduce_dual)
# ------------------------------------------------------------------------------------------------------------------------------------------
def _dual_thread_operands(a_rows, y_rows, t0, sg, d, prod, TS, ypad):
    """What ONE staging thread (pair quad sg = tid & 7) of a dual step writes to LDS for its rows: the operand values of the pairs 4 sg .. 4 sg + 3
    of the 64-frame step at t0, formed from aligned float4 loads exactly as loop_dual forms them.  a_rows [R][TS] (da: zero in [T, TS)),
    y_rows [R'][TS + 2 ypad] (the saved y with ypad zero floats on both sides; column ypad = frame 0)."""
    ds = d if d < 4 else 4
    if ds >= 4:
        pq = 4 * sg
        blk = pq // d
        e0 = 2 * d * blk + (pq - blk * d)
        o1 = e0 + d
        o2 = e0 - d if prod == 0 else e0 + 2 * d
    else:
        e0, o1 = 8 * sg, 8 * sg + 4
        o2 = e0 - ds if prod == 0 else e0 + 8                               # a 4- / 8-byte load: the d frames in front of / behind the 8
    dead = (t0 + 32 >= TS) and sg >= 4                                    # the half of an utterance's last step that lies behind its rows
    sh = -32 if dead else 0

    def f4(rows, col0, off, n=4):                                         # an ALIGNED load of n floats at frame t0 + off
        c = col0 + t0 + off + sh
        assert c % n == 0 and c >= 0 and c + n <= rows.shape[1], (c, rows.shape)
        return rows[:, c:c + n]

    def eo(l0, l1):
        if ds == 1:
            return np.stack([l0[:, 0], l0[:, 2], l1[:, 0], l1[:, 2]], 1), np.stack([l0[:, 1], l0[:, 3], l1[:, 1], l1[:, 3]], 1)
        if ds == 2:
            return np.stack([l0[:, 0], l0[:, 1], l1[:, 0], l1[:, 1]], 1), np.stack([l0[:, 2], l0[:, 3], l1[:, 2], l1[:, 3]], 1)
        return l0, l1

    ae, ao = eo(f4(a_rows, 0, e0), f4(a_rows, 0, o1))
    A = [ae, ae + ao, ae - ao, ao][prod]
    l0, l1 = f4(y_rows, ypad, e0), f4(y_rows, ypad, o1)
    x2 = f4(y_rows, ypad, o2, 4 if ds >= 4 else ds) if prod in (0, 3) else None
    if ds >= 4:
        Bv = [x2 - l1 if prod == 0 else None, l0 + l1, l1 - l0, l0 - x2 if prod == 3 else None][prod]
    else:
        e, o = eo(l0, l1)
        if prod == 1:
            Bv = e + o
        elif prod == 2:
            Bv = o - e
        elif prod == 0:
            m = np.stack([x2[:, 0], l0[:, 1], l0[:, 3], l1[:, 1]], 1) if ds == 1 else np.stack([x2[:, 0], x2[:, 1], l0[:, 2], l0[:, 3]], 1)
            Bv = m - o
        else:
            n = np.stack([l0[:, 2], l1[:, 0], l1[:, 2], x2[:, 0]], 1) if ds == 1 else np.stack([l1[:, 0], l1[:, 1], x2[:, 0], x2[:, 1]], 1)
            Bv = e - n
    if dead:
        A, Bv = np.zeros_like(A), np.zeros_like(Bv)
    return A, Bv


@pytest.mark.parametrize('d', [1, 2, 4, 8])
@pytest.mark.parametrize('T', [64, 96, 70, 33, 200])
def test_weight_gradient_dual_index_algebra(d, T):
    """The four pair products, staged thread by thread as k_tr_wgrad's dual step stages them (aligned 16-byte loads, the E / O permutation for
    d = 1, 2, whole float4 for d = 4, 8, the +- d neighbours, the dead half of a step behind TS), summed over the steps and back-transformed
    as k_tr_wgrad_reduce_dual does - against the three tap gradients dW_k[m][n] = sum_t a[m][t] y[n][t + (k - 1) d] in float64."""
    rng = np.random.default_rng(100 * d + T)
    TS, ypad = (T + 31) // 32 * 32, 8
    R, Rb = 6, 5
    a = np.zeros((R, TS))
    a[:, :T] = rng.standard_normal((R, T))
    y = np.zeros((Rb, TS + 2 * ypad))
    y[:, ypad:ypad + T] = rng.standard_normal((Rb, T))
    want = np.zeros((3, R, Rb))
    for k in range(3):
        for t in range(T):
            tt = t + (k - 1) * d
            if 0 <= tt < T:
                want[k] += np.outer(a[:, t], y[:, ypad + tt])
    P = np.zeros((4, R, Rb))
    seen = set()
    for t0 in range(0, TS, 64):
        for prod in range(4):
            for sg in range(8):
                A, Bv = _dual_thread_operands(a, y, t0, sg, d, prod, TS, ypad)
                P[prod] += A @ Bv.T
        # every frame of the step belongs to exactly one pair of exactly one thread
        ds = d if d < 4 else 4
        for sg in range(8):
            if ds >= 4:
                pq = 4 * sg
                blk = pq // d
                e0 = 2 * d * blk + (pq - blk * d)
                fr = [t0 + e0 + i for i in range(4)] + [t0 + e0 + d + i for i in range(4)]
            else:
                fr = [t0 + 8 * sg + i for i in range(8)]
            for f in fr:
                assert f not in seen
                seen.add(f)
    assert seen == set(range(0, (TS + 63) // 64 * 64))
    hs = 0.5 * (P[1] + P[2])
    got = np.stack([P[0] + hs, 0.5 * (P[1] - P[2]), hs - P[3]])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
