// train_abi.hpp - host side of the fused training stack (C ABI in include/dsf.h, "Training, the FUSED residual stack"); included at the
// end of dsd.hip behind fs2_abi.hpp (one translation unit: shares fail(), HIP_TRY, fs_ts and the packing kernels).
#include "train_kernels.hpp"
#include "train_loop.hpp"
#include "train_loop_wino.hpp"
#include "train_wino_bwd.hpp"

namespace {

constexpr size_t kTrW3 = (size_t)4 * 96 * 256 * 4;       // floats of one packed [512][256][3] (or its transpose) weight
constexpr size_t kTrW1 = (size_t)4 * 32 * 256 * 4;       // floats of one packed [512][256] weight
constexpr size_t kTrSlack = (size_t)kWeightSlack * 4;    // floats of A-prefetch slack behind a packed region
constexpr int kTrMaxSplit = 16;

static inline size_t tr_al(size_t n) { return (n + 1023) / 1024 * 1024; }      // keeps every sub-buffer 4 KiB aligned

static int tr_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 256;
    }
    return ncu;
}
// The forward as ONE persistent launch per chunk of whole utterances (train_loop.hpp) instead of a launch per layer: on when an utterance fits
// the co-resident grid (one workgroup per CU) and the chunks fill the chip at least as well as the per-layer grid does (the rule of
// loop_applicable(), dsd.hip).  dsf_set_stack_mode: 1 automatic (default), 0 per-layer launches always (the A/B switch of tools/bench_train.py),
// 2 persistent wherever an utterance fits the grid (tests of the chunked form on any shape).  A process-wide setting, read once per call:
// the workspace layouts do not depend on it.  (The data-gradient chain of the backward pass had a persistent form too - k_trb_loop, round 2:
// measured equal to the per-layer launches for 30 % more workspace, profiles/r03z - deleted in round 3.)
static int g_tr_stack_mode = 1;
extern "C" int dsf_set_stack_mode(int32_t mode) {
    if (mode < 0 || mode > 2) return fail(DSD_ERR_INVALID, "dsf_set_stack_mode: 0 (per-layer launches), 1 (automatic) or 2 (persistent wherever it fits)");
    g_tr_stack_mode = mode;
    return DSD_OK;
}
// The convolution of the persistent forward: 1 (default) Winograd F(2,3) along the frames (train_loop_wino.hpp: 2/3 of the multiplications) where
// every dilation is 1, 2, 4 or 8; 0 the direct form (train_loop.hpp: bit-identical to the per-layer launches - the anchor of that test).
static int g_tr_stack_conv = 1;
static int g_tr_stack_touch = 16;       // steps the Winograd weight stream's L2 touch runs in front (the inference loop's default)
extern "C" int dsf_set_stack_conv(int32_t mode) {
    if (mode < 0 || mode > 1) return fail(DSD_ERR_INVALID, "dsf_set_stack_conv: 0 (direct) or 1 (Winograd F(2,3))");
    g_tr_stack_conv = mode;
    return DSD_OK;
}
extern "C" int dsf_get_stack_conv(void) { return g_tr_stack_conv; }
// The dilated convolution's WEIGHT gradient in the fused backward: 1 (default, where the stack's convolution runs as Winograd: every dilation 1, 2, 4
// or 8 and dsf_set_stack_conv(1)) the Winograd F(2,3) dual - four products over frame PAIRS instead of three over the frames
// (train_kernels.hpp k_tr_wgrad / k_tr_wgrad_reduce_dual); 0 the three tap tiles of rounds 2-5 (the A/B switch and the anchor of the tests)
static int g_tr_wgrad_dual = 1;
extern "C" int dsf_set_wgrad_dual(int32_t on) { g_tr_wgrad_dual = on ? 1 : 0; return DSD_OK; }
// developer hook (tools/trb_timeline.py): s_memtime stamps of the Winograd data-gradient kernel, [workgroup][wave 4][8] per launch (the last launch wins)
static unsigned long long* g_trb_dbg = nullptr;
extern "C" int dsf_debug_trb_timeline(uint64_t* device_stamps) { g_trb_dbg = (unsigned long long*)device_stamps; return DSD_OK; }
static bool tr_wino_applies(const dsf_stack_weights* w, int L) {
    if (g_tr_stack_conv != 1) return false;
    for (int l = 0; l < L; ++l) {
        const int d = w->dilations[l];
        if (d != 1 && d != 2 && d != 4 && d != 8) return false;
    }
    return true;
}
static bool tr_persist_applies(int B, int ntile32) {
    const int mode = g_tr_stack_mode;
    if (mode == 0) return false;
    const int ncu = tr_ncu();
    if (ncu < 8 || ntile32 > ncu) return false;
    if ((long long)B * ntile32 > 32768) return false;
    if (mode == 2) return true;
    const int ntiles = B * ntile32, upc = std::max(1, ncu / ntile32), chunks = (B + upc - 1) / upc;
    const double u_p = (double)ntiles / ((double)chunks * ncu);
    const double u_l = 0.9 * (double)ntiles / ((double)((ntiles + ncu - 1) / ncu) * ncu);
    return u_p >= u_l;
}

struct TrSave {             // offsets in floats into save_ws
    size_t w1p, wcp, w2p, b1p, cp, X, Y, A, skip, bsum, iota, flags, w1w, total;
    size_t cp_l, X_l, Y_l, A_l;      // per-layer strides
};
static TrSave tr_save_layout(int B, int TS, int L) {
    const size_t ntiles = (size_t)B * TS / 32;
    TrSave s{};
    size_t o = 0;
    // ONE region for the dilated convolution's weights: a call packs either the direct fragments (w1p) or the Winograd-transformed stream
    // (w1w, + prefetch slack) into it, never both (two always-allocated regions cost 40 MB per workspace at L = 20 and left the
    // unused one holding stale bytes behind a live pointer)
    s.w1p = o; s.w1w = o; o += std::max((size_t)L * kTrW3, (size_t)L * kWnSteps * (kWnStepBytes / 4) + kTrSlack);
    s.wcp = o; o += L * kTrW1;
    s.w2p = o; o += L * kTrW1 + kTrSlack;
    s.b1p = o; o += tr_al((size_t)L * 512);
    s.cp_l = ntiles * 16384; s.cp = o; o += L * s.cp_l;
    s.X_l = ntiles * 8192; s.X = o; o += L * s.X_l;
    s.Y_l = (size_t)B * kC * (TS + 2 * kTrYPad); s.Y = o; o += tr_al(L * s.Y_l);      // rows padded: kTrYPad zero floats on both sides
    s.A_l = ntiles * 16384; s.A = o; o += L * s.A_l;
    s.skip = o; o += ntiles * 8192;
    s.bsum = o; o += 1024;
    s.iota = o; o += tr_al((size_t)B);
    s.flags = o; o += tr_al(ntiles + 64);       // phase flags + timeout word of the persistent forward (its halo buffers alias X)
    s.total = o;
    return s;
}

struct TrBwd {              // offsets in floats into bwd_ws
    size_t wotp, wdtp, da, g, dxp0, dxp1, dds_part, part, part_b, wdw, total;
};
static TrBwd tr_bwd_layout(int B, int TS, int L) {
    const size_t ntiles = (size_t)B * TS / 32, act = (size_t)B * kC * TS;
    TrBwd s{};
    size_t o = 0;
    s.wotp = o; o += L * kTrW1;
    s.wdtp = o; s.wdw = o; o += std::max((size_t)L * kTrW3, (size_t)L * kWnSteps * (kWnStepBytes / 4)) + kTrSlack;      // direct OR Winograd-transformed transposed conv weights (one per call)
    s.da = o; o += 2 * (2 * act);           // two slots each of da / g: the fused kernel of layer l reads da(l) and writes da(l - 1), g(l - 1)
    s.g = o; o += 2 * act;
    s.dxp0 = o; o += act;
    s.dxp1 = o; o += act;
    s.dds_part = o; o += tr_al((size_t)L * ntiles * kC);
    s.part = o; o += (size_t)kTrWgMaxTiles * kTrMaxSplit * 128 * 256;
    s.part_b = o; o += tr_al((size_t)kTrWgMaxTiles * kTrMaxSplit * 128);
    s.total = o;
    return s;
}

static int tr_check(const char* who, int B, int T, int L) {
    if (B < 1 || T < 1 || L < 1 || L > kTrMaxLayers) return fail(DSD_ERR_INVALID, "%s: bad shape (B=%d T=%d L=%d; L <= %d)", who, B, T, L, kTrMaxLayers);
    return DSD_OK;
}

static TrPtrs tr_ptrs(const float* const* tbl, int L) {
    TrPtrs t{};
    for (int l = 0; l < L; ++l) t.p[l] = tbl[l];
    return t;
}

static int tr_pack_multi(hipStream_t s, const float* const* src, int L, float* dst, size_t layer_floats, int nw, int ntap, int nkc, int nmb, int split,
                         int hi_base, int rows_valid, int cols_valid, int row_stride, int col_stride, int tap_rev) {
    PackMultiParams m{};
    m.pp.dst = dst; m.pp.nw = nw; m.pp.nkc = nkc; m.pp.nmb = nmb; m.pp.ntap = ntap; m.pp.split = split; m.pp.hi_base = hi_base;
    m.pp.rows_valid = rows_valid; m.pp.cols_valid = cols_valid; m.pp.row_stride = row_stride; m.pp.col_stride = col_stride; m.pp.centre_first = 1;
    m.src = tr_ptrs(src, L);
    m.dst_layer_floats = layer_floats;
    m.tap_rev = tap_rev;
    const size_t n = (size_t)nw * ntap * nkc * nmb * 256;
    hipLaunchKernelGGL(k_pack_a_multi, dim3((unsigned)((n + 255) / 256), (unsigned)L), dim3(256), 0, s, m);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

static int tr_attrs() {
    if (!first_on_device(30)) return DSD_OK;
    HIP_TRY(hipFuncSetAttribute((const void*)k_tr_layer<false>, hipFuncAttributeMaxDynamicSharedMemorySize, layer_lds_bytes<1>()));
    HIP_TRY(hipFuncSetAttribute((const void*)k_tr_layer<true>, hipFuncAttributeMaxDynamicSharedMemorySize, layer_lds_bytes<1>()));
    HIP_TRY(hipFuncSetAttribute((const void*)k_tr_stack_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, kTrStackLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_tr_stack_fwd_w, hipFuncAttributeMaxDynamicSharedMemorySize, kTrStackWinoLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trb_gate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrbGateLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trb_conv<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrbConvLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trb_conv<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrbConvLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trb_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrbFusedLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trb_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrbFusedLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trb_fused_w<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrbFusedWinoLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trb_fused_w<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrbFusedWinoLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trb_fused_w<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrbFusedWinoLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trb_fused_w<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrbFusedWinoLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_tr_wgrad<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrWgLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)k_tr_wgrad<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrWgLdsBytes));
    return DSD_OK;
}

// split-K factor of a weight-gradient launch of `ndesc` output tiles: fill the chip once (one workgroup per CU), never more splits than frame tiles
static int tr_nsplit(int ndesc, int ntile) {
    const int ncu = tr_ncu();
    int ns = ncu / std::max(ndesc, 1);
    ns = std::max(1, std::min(ns, kTrMaxSplit));
    return std::min(ns, ntile);
}

// Measurement hook (bench.py --row train): with the probe on, every k_tr_wgrad launch of the fused stack is bracketed by two events on its own
// stream; dsf_wgrad_probe_read sums the elapsed times (the roofline figure of the dominant kernel as it runs INSIDE the step).
struct TrProbe { bool on = false; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; double flops = 0.0; };
static TrProbe& tr_probe() { static TrProbe p; return p; }

// fix: some B operand is not padded against its tap shift (k_tr_wgrad<true>)
static int tr_wgrad_launch(hipStream_t s, TrWgParams& wp, int ndesc, int B, int T, int TS, bool fix) {
    wp.B = B; wp.T = T; wp.TS = TS;
    const int nc = wp.nc, np = ndesc - nc;                  // Winograd-dual products (the first nc descriptors, four per 128-row tile) / plain tiles
    if (nc) {
        // a dual step covers 64 frames, a plain step 32: half the splits for the dual products balances the workgroups, and
        // ns_c (nc + 2 np) <= CUs fills the chip once (two layers: 32 products x 4 + 16 tiles x 8 = 256 workgroups of 32 steps at 8 x 1024)
        const int ncu = tr_ncu();
        int nsc = std::max(1, ncu / std::max(nc + 2 * np, 1));
        nsc = std::min(nsc, kTrMaxSplit / 2);
        wp.ns_c = std::max(1, std::min(nsc, B * ((TS + 63) / 64)));
        wp.nsplit = std::max(1, std::min(2 * wp.ns_c, B * TS / 32));
    } else {
        wp.ns_c = 0;
        wp.nsplit = tr_nsplit(ndesc, B * TS / 32);
    }
    const int total = nc * wp.ns_c + np * wp.nsplit;
    wp.ndesc = ndesc; wp.xcd_q = total / 8; wp.xcd_r = total % 8;
    TrProbe& pr = tr_probe();
    const bool probe = pr.on && !fix;
    if (probe) {
        if (pr.used == pr.ev.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            pr.ev.emplace_back(a, b);
        }
        HIP_TRY(hipEventRecord(pr.ev[pr.used].first, s));
    }
    if (fix) hipLaunchKernelGGL(k_tr_wgrad<true>, dim3((unsigned)total), dim3(kThreads), kTrWgLdsBytes, s, wp);
    else hipLaunchKernelGGL(k_tr_wgrad<false>, dim3((unsigned)total), dim3(kThreads), kTrWgLdsBytes, s, wp);
    HIP_TRY(hipGetLastError());
    if (probe) {
        HIP_TRY(hipEventRecord(pr.ev[pr.used].second, s));
        ++pr.used;
        pr.flops += 2.0 * 128 * 256 * ((double)np + 0.5 * (double)nc) * (double)B * (double)T;     // EXECUTED: a dual product contracts T / 2 pairs
    }
    if (nc && np) hipLaunchKernelGGL(k_tr_wgrad_reduce_all, dim3((unsigned)(nc / 4 + np), 128), dim3(256), 0, s, wp);
    else if (nc) hipLaunchKernelGGL(k_tr_wgrad_reduce_dual, dim3((unsigned)(nc / 4), 128), dim3(256), 0, s, wp);
    else if (np) hipLaunchKernelGGL(k_tr_wgrad_reduce, dim3((unsigned)np, 128), dim3(256), 0, s, wp);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

}  // namespace

extern "C" int64_t dsf_stack_workspace_floats(int32_t B, int32_t T, int32_t L, int32_t which) {
    if (B < 1 || T < 1 || L < 1 || L > kTrMaxLayers || which < 0 || which > 1) return -1;
    const int TS = fs_ts(T);
    return (int64_t)(which == 0 ? tr_save_layout(B, TS, L).total : tr_bwd_layout(B, TS, L).total);
}

extern "C" int dsf_stack_offsets(int32_t B, int32_t T, int32_t L, int32_t which, int64_t* out, int32_t n) {
    if (!out || tr_check("dsf_stack_offsets", B, T, L) != DSD_OK) return DSD_ERR_INVALID;
    const int TS = fs_ts(T);
    int64_t v[16] = {0};
    if (which == 0) {
        const TrSave s = tr_save_layout(B, TS, L);
        const int64_t t[] = {(int64_t)s.w1p, (int64_t)s.wcp, (int64_t)s.w2p, (int64_t)s.b1p, (int64_t)s.cp, (int64_t)s.X, (int64_t)s.Y, (int64_t)s.A,
                             (int64_t)s.skip, (int64_t)s.bsum, (int64_t)s.cp_l, (int64_t)s.X_l, (int64_t)s.Y_l, (int64_t)s.A_l, (int64_t)s.total, 0};
        memcpy(v, t, sizeof(v));
    } else {
        const TrBwd s = tr_bwd_layout(B, TS, L);
        const int64_t t[] = {(int64_t)s.wotp, (int64_t)s.wdtp, (int64_t)s.da, (int64_t)s.g, (int64_t)s.dxp0, (int64_t)s.dxp1, (int64_t)s.dds_part,
                             (int64_t)s.part, (int64_t)s.part_b, (int64_t)s.total, 0, 0, 0, 0, 0, 0};
        memcpy(v, t, sizeof(v));
    }
    for (int i = 0; i < n && i < 16; ++i) out[i] = v[i];
    return DSD_OK;
}

// Launches of a persistent kernel: chunks of whole utterances, at most one workgroup per CU (all workgroups of a launch wait for each other);
// persistent launches on one device are serialised across streams (two co-resident grids could starve each other), as in run_persistent()
template <typename F>
static int tr_persistent_chunks(hipStream_t s, int B, int ntile32, F launch) {
    const int upc = std::max(1, tr_ncu() / ntile32);
    int dv = 0;
    (void)hipGetDevice(&dv);
    if (dv < 0 || dv >= kMaxDevices) dv = 0;
    std::lock_guard<std::mutex> guard(g_loop_mu[dv]);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    const bool guarded = (cap == hipStreamCaptureStatusNone);
    if (guarded) {
        if (!g_loop_ev[dv]) HIP_TRY(hipEventCreateWithFlags(&g_loop_ev[dv], hipEventDisableTiming));
        if (g_loop_has[dv] && g_loop_stream[dv] != s) HIP_TRY(hipStreamWaitEvent(s, g_loop_ev[dv], 0));
    }
    for (int b0 = 0; b0 < B; b0 += upc) {
        const int nb = std::min(upc, B - b0);
        launch(b0 * ntile32, nb * ntile32);
        HIP_TRY(hipGetLastError());
    }
    if (guarded) {
        HIP_TRY(hipEventRecord(g_loop_ev[dv], s));
        g_loop_stream[dv] = s;
        g_loop_has[dv] = true;
    }
    return DSD_OK;
}

static int tr_forward_persistent(const float* x0, const float* step, const dsf_stack_weights* w, int B, int T, int L, float* ws, const TrSave& lay,
                                 float* skip_out, bool wino, hipStream_t s) {
    const int TS = fs_ts(T), ntile32 = TS / 32, ntiles = B * ntile32;
    unsigned* flags = reinterpret_cast<unsigned*>(ws + lay.flags);
    HIP_TRY(hipMemsetAsync(flags, 0, ((size_t)ntiles + 64) * sizeof(unsigned), s));
    TrLoopParams p{};
    p.w1p = (const float4*)(ws + lay.w1p); p.w2p = (const float4*)(ws + lay.w2p); p.b2 = tr_ptrs(w->out_b, L);
    p.cp = (const float4*)(ws + lay.cp); p.cp_lstride = lay.cp_l / 4;
    p.step = step; p.x0 = x0;
    p.y_cm = ws + lay.Y + kTrYPad; p.y_lstride = lay.Y_l; p.y_rs = TS + 2 * kTrYPad;
    p.a_frag = (float4*)(ws + lay.A); p.a_lstride = lay.A_l / 4;
    p.bsum = ws + lay.bsum; p.skip_out = skip_out;
    p.L = L; p.T = T; p.TS = TS; p.ntile32 = ntile32; p.ntiles_total = ntiles;
    for (int l = 0; l < L; ++l) p.dil[l] = (unsigned char)w->dilations[l];
    p.flags = flags; p.tmo = flags + ntiles;
    p.halo = ws + lay.X;                            // 2 x ntiles x 16 KiB = one layer of the (here unused) tile-major x buffers
    if (wino) {
        TrLoopWinoParams q{};
        q.w1w = (const float4*)(ws + lay.w1w);
        q.wl_bytes = (unsigned)((size_t)L * kWnSteps * kWnStepBytes);
        q.touch_ahead = g_tr_stack_touch;
        return tr_persistent_chunks(s, B, ntile32, [&](int tile_base, int n_tiles) {
            p.tile_base = tile_base; p.n_tiles = n_tiles;
            q.tp = p;
            hipLaunchKernelGGL(k_tr_stack_fwd_w, dim3((unsigned)n_tiles), dim3(kThreads), kTrStackWinoLdsBytes, s, q);
        });
    }
    return tr_persistent_chunks(s, B, ntile32, [&](int tile_base, int n_tiles) {
        p.tile_base = tile_base; p.n_tiles = n_tiles;
        hipLaunchKernelGGL(k_tr_stack_fwd, dim3((unsigned)n_tiles), dim3(kThreads), kTrStackLdsBytes, s, p);
    });
}

extern "C" int dsf_stack_forward(const float* x0, const float* cond, const float* step, const dsf_stack_weights* w, int32_t B, int32_t T, int32_t L,
                                 float* ws, float* skip_out, void* stream) {
    if (!x0 || !cond || !step || !w || !ws || !skip_out) return fail(DSD_ERR_INVALID, "dsf_stack_forward: null argument");
    DSD_TRY(tr_check("dsf_stack_forward", B, T, L));
    for (int l = 0; l < L; ++l)
        if (w->dilations[l] < 1 || w->dilations[l] > kHalo) return fail(DSD_ERR_INVALID, "dsf_stack_forward: dilation %d of layer %d (1..%d)", w->dilations[l], l, kHalo);
    DSD_TRY(tr_attrs());
    hipStream_t s = (hipStream_t)stream;
    const int TS = fs_ts(T), ntile32 = TS / 32, ntiles = B * ntile32;
    const TrSave lay = tr_save_layout(B, TS, L);
    const bool persist = tr_persist_applies(B, ntile32);
    const bool wino = persist && tr_wino_applies(w, L);
    // weights -> fragment order, all layers per launch (they change every optimiser step)
    if (wino) {
        hipLaunchKernelGGL(k_pack_wino_multi, dim3(32, (unsigned)L), dim3(256), 0, s, tr_ptrs(w->dilated_conv_w, L), ws + lay.w1w);
        HIP_TRY(hipGetLastError());           // (the slack behind the stream is only ever PREFETCHED past the last step, never multiplied: it needs no value)
    } else {
        DSD_TRY(tr_pack_multi(s, w->dilated_conv_w, L, ws + lay.w1p, kTrW3, 4, 3, 32, 4, 1, kC, 2 * kC, kC, 3 * kC, 3, 0));
    }
    DSD_TRY(tr_pack_multi(s, w->cond_w, L, ws + lay.wcp, kTrW1, 4, 1, 32, 4, 1, kC, 2 * kC, kC, kC, 1, 0));
    DSD_TRY(tr_pack_multi(s, w->out_w, L, ws + lay.w2p, kTrW1, 4, 1, 32, 4, 1, kC, 2 * kC, kC, kC, 1, 0));
    {
        PackBiasMultiParams m{};
        m.pp.dst = ws + lay.b1p; m.pp.nw = 4; m.pp.nmb = 4; m.pp.split = 1; m.pp.hi_base = kC; m.pp.rows_valid = 2 * kC;
        m.a = tr_ptrs(w->dilated_conv_b, L); m.b = tr_ptrs(w->cond_b, L); m.has_b = 1; m.dst_layer_floats = 512;
        hipLaunchKernelGGL(k_pack_bias_multi, dim3(2, (unsigned)L), dim3(256), 0, s, m);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_tr_bsum, dim3(1), dim3(256), 0, s, tr_ptrs(w->out_b, L), ws + lay.bsum, L);
    int* iota = reinterpret_cast<int*>(ws + lay.iota);
    if (!persist) {
        hipLaunchKernelGGL(k_tr_iota, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, iota, B);
        const size_t n4 = (size_t)B * kC * TS / 4;
        hipLaunchKernelGGL(k_tr_cm_to_tm, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 16384)), dim3(256), 0, s, (const float4*)x0, (float4*)(ws + lay.X), TS, n4);
        HIP_TRY(hipGetLastError());
    }
    {
        CondProjParams p{};
        p.condT = cond; p.wcp = (const float4*)(ws + lay.wcp); p.b1p = (const float4*)(ws + lay.b1p); p.cp = (float4*)(ws + lay.cp);
        p.TS = TS; p.ntile32 = ntile32; p.ntiles_total = ntiles; p.L = L;
        p.wino = wino ? 1 : 0;                      // the Winograd forward takes it as its accumulators' initial values (dsd_kernels.hpp)
        for (int l = 0; l < L; ++l) p.dil[l] = (unsigned char)w->dilations[l];
        const int G = condproj_groups(p.dil, L, ntiles);
        hipLaunchKernelGGL(k_condproj, dim3((unsigned)ntiles, (unsigned)G), dim3(kThreads), condproj_lds(L, G), s, p);
        HIP_TRY(hipGetLastError());
    }
    {
        const size_t rows = (size_t)L * B * kC;
        hipLaunchKernelGGL(k_tr_zero_pads, dim3((unsigned)std::min<size_t>((rows * 2 * kTrYPad + 255) / 256, 8192)), dim3(256), 0, s, ws + lay.Y, rows, TS + 2 * kTrYPad);
        HIP_TRY(hipGetLastError());
    }
    if (persist) return tr_forward_persistent(x0, step, w, B, T, L, ws, lay, skip_out, wino, s);
    for (int l = 0; l < L; ++l) {
        const bool last = (l == L - 1);
        LayerParams p{};
        p.x_in = ws + lay.X + (size_t)l * lay.X_l;
        p.x_out = last ? nullptr : ws + lay.X + (size_t)(l + 1) * lay.X_l;
        p.w1p = (const float4*)(ws + lay.w1p + (size_t)l * kTrW3);
        p.w2p = (const float4*)(ws + lay.w2p + (size_t)l * kTrW1);
        p.b2 = w->out_b[l];
        p.cp = (const float4*)(ws + lay.cp + (size_t)l * lay.cp_l);
        p.skip = (float4*)(ws + lay.skip);
        p.ds = step + (size_t)l * kC;
        p.t_dev = iota; p.t_uniform = 0; p.ds_tstride = L * kC;
        p.T = T; p.TS = TS; p.ntile32 = ntile32; p.tiles_per_utt = ntile32; p.dil = w->dilations[l]; p.first = (l == 0);
        p.wt_stores = 1;
        p.xcd_q = ntiles / 8; p.xcd_r = ntiles % 8;
        p.dbg = nullptr;
        const LayerSave sv{ws + lay.Y + (size_t)l * lay.Y_l + kTrYPad, (float4*)(ws + lay.A + (size_t)l * lay.A_l), TS + 2 * kTrYPad};
        if (last) hipLaunchKernelGGL(k_tr_layer<true>, dim3((unsigned)ntiles), dim3(kThreads), layer_lds_bytes<1>(), s, p, sv);
        else hipLaunchKernelGGL(k_tr_layer<false>, dim3((unsigned)ntiles), dim3(kThreads), layer_lds_bytes<1>(), s, p, sv);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_tr_skip_to_cm, dim3((unsigned)ntiles), dim3(kThreads), 0, s, (const float4*)(ws + lay.skip), ws + lay.bsum, skip_out, T, TS, ntile32);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_stack_backward(const float* dskip, const float* cond, const dsf_stack_weights* w, int32_t B, int32_t T, int32_t L, const float* sws,
                                  float* bws, const dsf_stack_grads* g, float* da_all, void* stream) {
    if (!dskip || !cond || !w || !sws || !bws || !g || !g->dx0 || !g->dstep) return fail(DSD_ERR_INVALID, "dsf_stack_backward: null argument");
    DSD_TRY(tr_check("dsf_stack_backward", B, T, L));
    DSD_TRY(tr_attrs());
    hipStream_t s = (hipStream_t)stream;
    const int TS = fs_ts(T), ntile32 = TS / 32, ntiles = B * ntile32;
    const TrSave lay = tr_save_layout(B, TS, L);
    const TrBwd bl = tr_bwd_layout(B, TS, L);
    const size_t act = (size_t)B * kC * TS;
    // transposed weights in fragment order, two 128-row groups of four row blocks: Wo^T [256 gate channels][512 output rows]; Wd^T flipped
    // [256 input channels][3 x 512]
    DSD_TRY(tr_pack_multi(s, w->out_w, L, bws + bl.wotp, kTrW1, 2, 1, 64, 4, 0, 0, kC, 2 * kC, 1, kC, 0));
    // the transposed convolution as Winograd F(2,3) (train_wino_bwd.hpp) where every dilation is 1, 2, 4 or 8 and dsf_set_stack_conv says so
    const bool wino = tr_wino_applies(w, L);
    {   // each K half (gate rows / filter rows of da) of every layer as its own [256 input channels] x [3 x 256] matrix, centre taps first
        const float* halves[2 * kTrMaxLayers];
        for (int l = 0; l < L; ++l) { halves[2 * l] = w->dilated_conv_w[l]; halves[2 * l + 1] = w->dilated_conv_w[l] + (size_t)kC * 3 * kC; }
        if (!wino) DSD_TRY(tr_pack_multi(s, halves, 2 * L, bws + bl.wdtp, kTrW3 / 2, 2, 3, 32, 4, 0, 0, kC, kC, 3, 3 * kC, 1));
    }
    if (wino) {
        hipLaunchKernelGGL(k_pack_wino_bwd_multi, dim3(32, (unsigned)L), dim3(256), 0, s, tr_ptrs(w->dilated_conv_w, L), bws + bl.wdw);
        HIP_TRY(hipGetLastError());
    }
    float* dxp[2] = {bws + bl.dxp0, bws + bl.dxp1};           // dxp[k & 1]: gradient wrt the output x of layer k
    const long long da_bs = da_all ? (long long)L * 2 * kC * TS : (long long)2 * kC * TS;
    auto da_of = [&](int l) { return da_all ? da_all + (size_t)l * 2 * kC * TS : bws + bl.da + (size_t)(l & 1) * 2 * act; };
    auto g_of = [&](int l) { return bws + bl.g + (size_t)(l & 1) * act; };
    auto dx_of = [&](int l) { return dxp[l & 1]; };           // gradient wrt the output x of layer l < L - 1
    auto gate_params = [&](int l) {
        TrbGateParams p{};
        p.dxp = (l == L - 1) ? nullptr : dxp[l & 1]; p.dsk = dskip; p.a_frag = (const float4*)(sws + lay.A + (size_t)l * lay.A_l);
        p.wotp = (const float4*)(bws + bl.wotp + (size_t)l * kTrW1);
        p.da = da_of(l); p.g = g_of(l); p.da_bstride = da_bs; p.T = T; p.TS = TS; p.ntile32 = ntile32; p.xcd_q = ntiles / 8; p.xcd_r = ntiles % 8;
        return p;
    };
    auto conv_params = [&](int l) {
        TrbConvParams p{};
        p.da = da_of(l); p.wdtp = (const float4*)(bws + bl.wdtp + (size_t)l * kTrW3); p.dxp = (l == L - 1) ? nullptr : dxp[l & 1];
        p.dx_out = (l == 0) ? g->dx0 : dxp[(l - 1) & 1];
        p.dds_part = bws + bl.dds_part + (size_t)l * ntiles * kC; p.da_bstride = da_bs;
        p.T = T; p.TS = TS; p.ntile32 = ntile32; p.dil = w->dilations[l]; p.xcd_q = ntiles / 8; p.xcd_r = ntiles % 8;
        return p;
    };
    // a layer's weight gradients: 12 (dilated conv: 4 row tiles x 3 taps) + 4 (conditioner projection) + 4 or 2 (output projection) tiles
    // pass 0: the dilated convolution's tiles - as the four products of the Winograd dual per 128-row tile (they must be the FIRST descriptors of a
    // launch: tr_wgrad_launch gives them their own split count) or, with the dual off, as the three tap tiles; pass 1: the 1 x 1 projections
    const bool dual = wino && g_tr_wgrad_dual;
    auto wgrad_tiles = [&](int l, TrWgParams& wp, int& nd, int pass) -> int {
        const bool last = (l == L - 1);
        const float* da = da_of(l);
        const float* dxp_in = last ? nullptr : dx_of(l);
        const float* y = sws + lay.Y + (size_t)l * lay.Y_l + kTrYPad;
        const int yrs = TS + 2 * kTrYPad;
        const int dil = w->dilations[l];
        if (pass == 0) {
            for (int mt = 0; mt < 4; ++mt)
                for (int k = 0; k < (dual ? 4 : 3); ++k) {
                    TrWgTile& d = wp.tile[nd++];
                    d.a = da + (size_t)mt * 128 * TS; d.a_bstride = da_bs; d.bsrc = y; d.b_bstride = (long long)kC * yrs; d.b_rs = yrs;
                    d.out_rs = 3 * kC; d.out_cs = 3; d.a_scale = 1.f;
                    if (dual) {          // product k of the tile: `out` = the tile's tap 0 (the reduction writes all three taps), the bias rides on P1 (row sums of E + O)
                        d.prod = k; d.shift = dil;
                        d.out = g->dilated_conv_w[l] + (size_t)mt * 128 * 3 * kC;
                        d.out_bias = (k == 1) ? g->dilated_conv_b[l] + mt * 128 : nullptr;
                        ++wp.nc;
                    } else {
                        d.prod = -1; d.shift = (k - 1) * dil;
                        d.out = g->dilated_conv_w[l] + (size_t)mt * 128 * 3 * kC + k;
                        d.out_bias = (k == 0) ? g->dilated_conv_b[l] + mt * 128 : nullptr;
                    }
                }
            return DSD_OK;
        }
        for (int mt = 0; mt < 4; ++mt) {
            TrWgTile& d = wp.tile[nd++];
            d.a = da + (size_t)mt * 128 * TS; d.a_bstride = da_bs; d.bsrc = cond; d.b_bstride = (long long)kC * TS; d.b_rs = TS; d.shift = 0; d.prod = -1;
            d.out = g->cond_w[l] + (size_t)mt * 128 * kC; d.out_rs = kC; d.out_cs = 1; d.out_bias = g->cond_b[l] + mt * 128; d.a_scale = 1.f;
        }
        for (int mt = last ? 2 : 0; mt < 4; ++mt) {
            TrWgTile& d = wp.tile[nd++];
            if (mt < 2) { d.a = dxp_in + (size_t)mt * 128 * TS; d.a_scale = kTrInvSqrt2; }
            else { d.a = dskip + (size_t)(mt - 2) * 128 * TS; d.a_scale = 1.f; }
            d.a_bstride = (long long)kC * TS; d.bsrc = g_of(l); d.b_bstride = (long long)kC * TS; d.b_rs = TS; d.shift = 0; d.prod = -1;
            d.out = g->out_w[l] + (size_t)mt * 128 * kC; d.out_rs = kC; d.out_cs = 1; d.out_bias = g->out_b[l] + mt * 128;
        }
        if (last) {          // the residual half of the last layer's output projection is dead (net.py:126 reads the skips only): zero gradient
            HIP_TRY(hipMemsetAsync(g->out_w[l], 0, (size_t)kC * kC * sizeof(float), s));
            HIP_TRY(hipMemsetAsync(g->out_b[l], 0, (size_t)kC * sizeof(float), s));
        }
        return DSD_OK;
    };
    // Layers whose operands are complete (da, g written by their gate kernel; the gradient wrt their output x by the conv kernel of the layer
    // above) wait here: two layers share ONE weight-gradient launch - 40 tiles x 6 frame splits instead of 2 x (20 x 12): the same 240
    // workgroups work twice as long, half the split-K partials per layer are written and reduced, half the launches.  The slots of da / g /
    // dx alternate by layer parity: a pair is launched before the next kernel overwrites the older one's slot.
    int pending[2], npend = 0;
    auto wgrad_flush = [&]() -> int {
        if (!npend) return DSD_OK;
        TrWgParams wp{};
        int nd = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < npend; ++i) DSD_TRY(wgrad_tiles(pending[i], wp, nd, pass));
        npend = 0;
        wp.part = bws + bl.part; wp.part_b = bws + bl.part_b;
        return tr_wgrad_launch(s, wp, nd, B, T, TS, false);
    };
    auto wgrad = [&](int l) -> int {
        pending[npend++] = l;
        return (npend == 2) ? wgrad_flush() : DSD_OK;
    };
    const dim3 grid((unsigned)ntiles), blk(kThreads);
    {   // gate derivative of the last layer (its x_out is dead: K = 256)
        const TrbGateParams p = gate_params(L - 1);
        hipLaunchKernelGGL(k_trb_gate<true>, grid, blk, kTrbGateLdsBytes, s, p);
        HIP_TRY(hipGetLastError());
        DSD_TRY(wgrad(L - 1));
    }
    for (int l = L - 1; l >= 0; --l) {
        const bool last = (l == L - 1);
        if (l > 0) {
            // transposed conv of layer l + gate derivative of layer l - 1 in one kernel (the dx tile stays in the workgroup)
            TrbFusedParams q{conv_params(l), gate_params(l - 1)};
            if (wino) {
                const TrbFusedWinoParams qw{q, (const float4*)(bws + bl.wdw + (size_t)l * kWnSteps * (kWnStepBytes / 4)),
                                            (unsigned)(kWnSteps * kWnStepBytes), g_tr_stack_touch, (l == 1) ? g_trb_dbg : nullptr};
                if (last) hipLaunchKernelGGL(k_trb_fused_w<true>, grid, blk, kTrbFusedWinoLdsBytes, s, qw);
                else hipLaunchKernelGGL(k_trb_fused_w<false>, grid, blk, kTrbFusedWinoLdsBytes, s, qw);
            } else if (last) hipLaunchKernelGGL(k_trb_fused<true>, grid, blk, kTrbFusedLdsBytes, s, q);
            else hipLaunchKernelGGL(k_trb_fused<false>, grid, blk, kTrbFusedLdsBytes, s, q);
        } else if (wino) {
            TrbFusedParams q{conv_params(l), TrbGateParams{}};
            const TrbFusedWinoParams qw{q, (const float4*)(bws + bl.wdw), (unsigned)(kWnSteps * kWnStepBytes), g_tr_stack_touch, nullptr};
            if (last) hipLaunchKernelGGL((k_trb_fused_w<true, false>), grid, blk, kTrbFusedWinoLdsBytes, s, qw);
            else hipLaunchKernelGGL((k_trb_fused_w<false, false>), grid, blk, kTrbFusedWinoLdsBytes, s, qw);
        } else {
            const TrbConvParams p = conv_params(l);
            if (last) hipLaunchKernelGGL(k_trb_conv<true>, grid, blk, kTrbConvLdsBytes, s, p);
            else hipLaunchKernelGGL(k_trb_conv<false>, grid, blk, kTrbConvLdsBytes, s, p);
        }
        HIP_TRY(hipGetLastError());
        if (l > 0) DSD_TRY(wgrad(l - 1));
    }
    DSD_TRY(wgrad_flush());
    hipLaunchKernelGGL(k_tr_dds_reduce, dim3((unsigned)B, (unsigned)L), dim3(kC), 0, s, bws + bl.dds_part, g->dstep, L, ntile32, ntiles);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int64_t dsf_wgrad2_workspace_floats(int32_t Co, int32_t Ci, int32_t KT) {
    if (Co < 128 || (Co % 128) || Ci < 256 || (Ci % 256) || (KT != 1 && KT != 3)) return -1;
    const int nd = (Co / 128) * (Ci / 256) * KT;
    return (int64_t)nd * kTrMaxSplit * (128 * 256 + 128);
}

extern "C" int dsf_conv1d_wgrad2(const float* dy, const float* x, float* dw, float* db, float* workspace, int32_t B, int32_t Ci, int32_t Co, int32_t KT,
                                 int32_t dil, int32_t T, void* stream) {
    if (!dy || !x || !dw || !workspace) return fail(DSD_ERR_INVALID, "dsf_conv1d_wgrad2: null argument");
    if (B < 1 || T < 1 || Co < 128 || (Co % 128) || Ci < 256 || (Ci % 256) || (KT != 1 && KT != 3) || dil < 1 || dil > kHalo)
        return fail(DSD_ERR_INVALID, "dsf_conv1d_wgrad2: bad shape (B=%d T=%d Ci=%d Co=%d K=%d dil=%d): Co %% 128 == 0, Ci %% 256 == 0, K in {1, 3}", B, T, Ci, Co, KT, dil);
    DSD_TRY(tr_attrs());
    hipStream_t s = (hipStream_t)stream;
    const int TS = fs_ts(T);
    const int per = kTrWgMaxTiles, ndtot = (Co / 128) * (Ci / 256) * KT;
    // descriptors in launches of at most kTrWgMaxTiles tiles
    int done = 0;
    while (done < ndtot) {
        TrWgParams wp{};
        int nd = 0;
        for (; nd < per && done + nd < ndtot; ++nd) {
            const int id = done + nd;
            const int tap = id % KT, nt = (id / KT) % (Ci / 256), mt = id / (KT * (Ci / 256));
            TrWgTile& d = wp.tile[nd];
            d.a = dy + (size_t)mt * 128 * TS; d.a_bstride = (long long)Co * TS;
            d.bsrc = x + (size_t)nt * 256 * TS; d.b_bstride = (long long)Ci * TS; d.b_rs = TS;
            d.shift = (tap - (KT - 1) / 2) * dil; d.prod = -1;
            d.out = dw + ((size_t)mt * 128 * Ci + (size_t)nt * 256) * KT + tap; d.out_rs = Ci * KT; d.out_cs = KT;
            d.out_bias = (db && tap == 0 && nt == 0) ? db + mt * 128 : nullptr; d.a_scale = 1.f;
        }
        wp.part = workspace + (size_t)done * kTrMaxSplit * (128 * 256);
        wp.part_b = workspace + (size_t)ndtot * kTrMaxSplit * (128 * 256) + (size_t)done * kTrMaxSplit * 128;
        DSD_TRY(tr_wgrad_launch(s, wp, nd, B, T, TS, true));
        done += nd;
    }
    return DSD_OK;
}

extern "C" int dsf_wgrad_probe(int32_t on) {
    TrProbe& pr = tr_probe();
    pr.on = on != 0;
    pr.used = 0;
    pr.flops = 0.0;
    return DSD_OK;
}

extern "C" int dsf_wgrad_probe_read(double* total_ms, int64_t* launches, double* flops) {
    if (!total_ms || !launches || !flops) return fail(DSD_ERR_INVALID, "dsf_wgrad_probe_read: null argument");
    TrProbe& pr = tr_probe();
    double ms = 0.0;
    for (size_t i = 0; i < pr.used; ++i) {
        HIP_TRY(hipEventSynchronize(pr.ev[i].second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, pr.ev[i].first, pr.ev[i].second));
        ms += t;
    }
    *total_ms = ms; *launches = (int64_t)pr.used; *flops = pr.flops;
    pr.used = 0;
    pr.flops = 0.0;
    return DSD_OK;
}
