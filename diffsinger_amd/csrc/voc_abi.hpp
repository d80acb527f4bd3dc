// voc_abi.hpp - host side of the HiFi-GAN / NSF-HiFi-GAN generator ops (C ABI in include/dsv.h); included at the end of dsd.hip so
// the library stays one translation unit (shares the error string, the HIP_TRY macros and the operand-packing kernel).
#include "voc_kernels.hpp"
#include "voc_chain.hpp"
#include "pwg_kernels.hpp"

#include "../../include/dsv.h"

static inline int voc_ls(int L) { return (L + 31) / 32 * 32; }

extern "C" int32_t dsv_padded_samples(int32_t L) { return voc_ls(L); }

extern "C" int64_t dsv_packed_floats(int32_t rows, int32_t Ci, int32_t KT) {
    if (rows < 1 || Ci < 1 || KT < 1) return -1;
    const int64_t nrb = (rows + 31) / 32, ci8 = (Ci + 7) / 8;
    return (nrb * ci8 * KT * 64 + kWeightSlack) * 4;
}

extern "C" int dsv_pack_weight(const float* w, int32_t rows, int32_t Ci, int32_t KT, float* packed, void* stream) {
    if (!w || !packed) return fail(DSD_ERR_INVALID, "dsv_pack_weight: null argument");
    if (rows < 1 || Ci < 1 || KT < 1) return fail(DSD_ERR_INVALID, "dsv_pack_weight: bad shape (rows=%d Ci=%d K=%d)", rows, Ci, KT);
    PackParams p{};
    p.src = w; p.dst = packed;
    p.nw = (rows + 31) / 32; p.nkc = (Ci + 7) / 8; p.nmb = 1; p.ntap = KT;
    p.split = 0; p.hi_base = 0;
    p.rows_valid = rows; p.cols_valid = Ci;
    p.row_stride = Ci * KT; p.col_stride = KT;
    const size_t n = (size_t)p.nw * p.ntap * p.nkc * p.nmb * 256;
    hipLaunchKernelGGL(k_pack_a, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(packed + n, 0, (size_t)kWeightSlack * 16, (hipStream_t)stream));
    return DSD_OK;
}

extern "C" int dsv_pad_rows(const float* in, float* out, int64_t R, int32_t L, void* stream) {
    if (!in || !out || R < 1 || R > 65535 || L < 1) return fail(DSD_ERR_INVALID, "dsv_pad_rows: bad argument");
    const int LS = voc_ls(L);
    hipLaunchKernelGGL(k_voc_pad_rows, dim3((unsigned)((LS + 255) / 256), (unsigned)R), dim3(256), 0, (hipStream_t)stream, in, out, L, LS);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

template <int NB, int WT, int HALO>
static void voc_conv_launch(const VocConvParams& p, int B, hipStream_t s) {
    if (first_on_device(100 + 10 * NB + WT + 1000 * (HALO != kVocHalo))) {
        (void)hipFuncSetAttribute((const void*)k_voc_conv<NB, WT, HALO>, hipFuncAttributeMaxDynamicSharedMemorySize, voc_lds_bytes<NB, WT, HALO>());
    }
    constexpr int WR = 4 / WT, SPAN = voc_span<NB, WT>();
    const size_t lds = (size_t)voc_lds_bytes<NB, WT, HALO>();
    const dim3 grid((unsigned)((p.LSi + SPAN - 1) / SPAN), (unsigned)B, (unsigned)((p.rows + 32 * WR - 1) / (32 * WR)));
    hipLaunchKernelGGL((k_voc_conv<NB, WT, HALO>), grid, dim3(kThreads), lds, s, p);
}

// the lean build for memory-shaped layers (voc_kernels.hpp): LDS by the channel count
static int g_voc_lean = 1;       // dsv_set_lean: the A/B switch of the measurement
extern "C" int dsv_set_lean(int32_t on) { g_voc_lean = on ? 1 : 0; return DSD_OK; }

template <int NB, int WT>
static bool voc_lean_try(const VocConvParams& p, int B, hipStream_t s) {
    static const int env_on = [] { const char* e = getenv("DSV_LEAN"); return e ? atoi(e) : 1; }();
    constexpr int WR = 4 / WT, SPAN = voc_span<NB, WT>(), LD = voc_ld<NB, WT, kVocHaloLean>();
    const int ci8 = (p.Ci + 7) / 8 * 8;
    if (!g_voc_lean || !env_on || p.pad > kVocHaloLean || (p.KT - 1) * p.dil - p.pad > kVocHaloLean || ci8 > voc_slab<NB, WT, kVocHaloLean>() || ci8 * LD * 4 > 40 * 1024)
        return false;
    const dim3 grid((unsigned)((p.LSi + SPAN - 1) / SPAN), (unsigned)B, (unsigned)((p.rows + 32 * WR - 1) / (32 * WR)));
    hipLaunchKernelGGL((k_voc_conv_lean<NB, WT>), grid, dim3(kThreads), (size_t)ci8 * LD * 4, s, p);
    return true;
}

static int voc_conv_fill(VocConvParams& p, const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t rows, int32_t KT,
                         int32_t pad, int32_t dil, int32_t L_in, int32_t up, float pre_slope, const float* residual, const float* sum_in, float divide,
                         int32_t act, const char* who) {
    if (!in || !wpacked || !out) return fail(DSD_ERR_INVALID, "%s: null argument", who);
    if (B < 1 || B > 65535 || Ci < 1 || rows < 1 || KT < 1 || dil < 1 || L_in < 1 || up < 1 || (rows % up) || pad < 0 || pad > kVocHaloWide ||
        (KT - 1) * dil - pad > kVocHaloWide || (KT - 1) * dil - pad < 0 || act < 0 || act > 1 || divide == 0.f || (int64_t)L_in * up > (1 << 30))
        return fail(DSD_ERR_INVALID, "%s: bad shape (B=%d Ci=%d rows=%d K=%d pad=%d dil=%d L=%d up=%d act=%d); taps must stay within +-%d samples", who,
                    B, Ci, rows, KT, pad, dil, L_in, up, act, kVocHaloWide);
    p = VocConvParams{};
    p.in = in; p.wp = reinterpret_cast<const float4*>(wpacked); p.bias = bias; p.out = out; p.res = residual; p.sum_in = sum_in;
    p.Ci = Ci; p.rows = rows; p.KT = KT; p.pad = pad; p.dil = dil;
    p.Li = L_in; p.LSi = voc_ls(L_in); p.U = up; p.Lo = L_in * up; p.LSo = voc_ls(p.Lo);
    p.pre_slope = pre_slope; p.divide = divide; p.act = act;
    return DSD_OK;
}

extern "C" int dsv_conv1d(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t rows, int32_t KT,
                          int32_t pad, int32_t dil, int32_t L_in, int32_t up, float pre_slope, const float* residual, const float* sum_in,
                          float divide, int32_t act, void* stream) {
    VocConvParams p{};
    const int rc = voc_conv_fill(p, in, wpacked, bias, out, B, Ci, rows, KT, pad, dil, L_in, up, pre_slope, residual, sum_in, divide, act, "dsv_conv1d");
    if (rc != DSD_OK) return rc;
    // narrow layers: one row block, the four waves split 512 samples; 64 rows: 2 x 2; wide (low-rate) layers: four row blocks x 32 samples
    const bool wide = pad > kVocHalo || (KT - 1) * dil - pad > kVocHalo;        // taps beyond the +-28 samples of the standard staging window
    // rows > 64 on a LONG axis (the 64 -> 32 transposed convolution of the shipped generator: 256 polyphase rows x 8 192 input samples per
    // utterance): four row blocks x NB 32-sample blocks per workgroup - a weight fragment feeds 4 NB MFMAs instead of 4 and the +-28-sample
    // staging halo is paid once per 32 NB samples (round 6: 4 096 workgroups of 2 us of matrix work each were 8 latency-bound rounds).
    // Same chunk order, same bits.
    const long tall_wgs = (long)((p.LSi + 127) / 128) * B * ((rows + 127) / 128);
    if (!wide && rows > 64 && tall_wgs >= 512) {
        voc_conv_launch<4, 1, kVocHalo>(p, B, (hipStream_t)stream);
        HIP_TRY(hipGetLastError());
        return DSD_OK;
    }
    if (wide) {
        if (rows <= 32) voc_conv_launch<4, 4, kVocHaloWide>(p, B, (hipStream_t)stream);
        else if (rows <= 64) voc_conv_launch<2, 2, kVocHaloWide>(p, B, (hipStream_t)stream);
        else voc_conv_launch<1, 1, kVocHaloWide>(p, B, (hipStream_t)stream);
    } else {
        if (rows <= 32) {
            // the stride-2 transposed convolutions: the lean build (four and more workgroups per CU) where it applies
            if (!(up == 2 && voc_lean_try<2, 4>(p, B, (hipStream_t)stream))) voc_conv_launch<4, 4, kVocHalo>(p, B, (hipStream_t)stream);
        }
        else if (rows <= 64) voc_conv_launch<2, 2, kVocHalo>(p, B, (hipStream_t)stream);
        else voc_conv_launch<1, 1, kVocHalo>(p, B, (hipStream_t)stream);
    }
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

template <int NB, int WT, int HALO>
static void voc_conv_multi_launch(const VocConvMulti& m, int ngroups, int B, hipStream_t s) {
    if (first_on_device(600 + 10 * NB + WT + 1000 * (HALO != kVocHalo))) {
        (void)hipFuncSetAttribute((const void*)k_voc_conv_multi<NB, WT, HALO>, hipFuncAttributeMaxDynamicSharedMemorySize, voc_lds_bytes<NB, WT, HALO>());
    }
    constexpr int SPAN = voc_span<NB, WT>();
    const size_t lds = (size_t)voc_lds_bytes<NB, WT, HALO>();
    const dim3 grid((unsigned)((m.g[0].LSi + SPAN - 1) / SPAN), (unsigned)B, (unsigned)(m.zc * ngroups));
    hipLaunchKernelGGL((k_voc_conv_multi<NB, WT, HALO>), grid, dim3(kThreads), lds, s, m);
}

extern "C" int dsv_conv1d_multi(int32_t ngroups, const dsv_conv_desc* d, int32_t B, int32_t Ci, int32_t rows, int32_t L_in, int32_t up,
                                float pre_slope, void* stream) {
    if (!d || ngroups < 1 || ngroups > kVocMultiMax) return fail(DSD_ERR_INVALID, "dsv_conv1d_multi: 1 .. %d convolutions", kVocMultiMax);
    VocConvMulti m{};
    bool wide = false;
    for (int g = 0; g < ngroups; ++g) {
        const int rc = voc_conv_fill(m.g[g], d[g].in, d[g].wpacked, d[g].bias, d[g].out, B, Ci, rows, d[g].K, d[g].pad, d[g].dil, L_in, up, pre_slope,
                                     d[g].residual, d[g].sum_in, d[g].divide, d[g].act, "dsv_conv1d_multi");
        if (rc != DSD_OK) return rc;
        wide = wide || d[g].pad > kVocHalo || (d[g].K - 1) * d[g].dil - d[g].pad > kVocHalo;
        for (int k = 0; k < ngroups; ++k) {
            if (k != g && d[k].out == d[g].out) return fail(DSD_ERR_INVALID, "dsv_conv1d_multi: convolutions %d and %d write the same buffer", k, g);
            if (d[k].out == d[g].in || d[k].out == d[g].residual || d[k].out == d[g].sum_in)
                return fail(DSD_ERR_INVALID, "dsv_conv1d_multi: the output of convolution %d is an operand of convolution %d (the convolutions of a call must be independent)", k, g);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    // the instantiation dsv_conv1d picks for this shape (same tiles, same chunk order: the same bits)
    const long tall_wgs = (long)((m.g[0].LSi + 127) / 128) * B * ((rows + 127) / 128);
    if (!wide && rows > 64 && tall_wgs >= 512) { m.zc = (rows + 127) / 128; voc_conv_multi_launch<4, 1, kVocHalo>(m, ngroups, B, st); }
    else if (rows <= 32) { m.zc = 1; if (wide) voc_conv_multi_launch<4, 4, kVocHaloWide>(m, ngroups, B, st); else voc_conv_multi_launch<4, 4, kVocHalo>(m, ngroups, B, st); }
    else if (rows <= 64) { m.zc = 1; if (wide) voc_conv_multi_launch<2, 2, kVocHaloWide>(m, ngroups, B, st); else voc_conv_multi_launch<2, 2, kVocHalo>(m, ngroups, B, st); }
    else { m.zc = (rows + 127) / 128; if (wide) voc_conv_multi_launch<1, 1, kVocHaloWide>(m, ngroups, B, st); else voc_conv_multi_launch<1, 1, kVocHalo>(m, ngroups, B, st); }
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

static int g_voc_fold = 1;       // dsv_set_fold: the A/B switch of the measurement (narrow layers on the unfolded kernel)

extern "C" int dsv_set_fold(int32_t on) { g_voc_fold = on ? 1 : 0; return DSD_OK; }

extern "C" int32_t dsv_fold_factor(int32_t Co, int32_t Ci, int32_t K, int32_t dil) {
    if (!g_voc_fold || Co < 1 || Ci < 1 || Ci > kFoldMaxCi || K < 1 || !(K & 1) || dil < 1) return 1;
    const int F = (Co <= 8) ? 4 : (Co <= 16) ? 2 : 1;
    if (F == 1) return 1;
    const int pad = (K - 1) * dil / 2, KT = K + F - 1;
    const int maxcol = (254 + dil) * F + dil + 30 + (KT - 1) * dil - pad;           // last LDS column a lane can read (see k_voc_conv_fold)
    const int LD = (F == 4) ? fold_ld<4>() : fold_ld<2>();
    if (pad > kVocHalo - 3 || maxcol >= LD) return 1;
    return F;
}

template <int F>
static void voc_fold_launch(const VocFoldParams& p, int B, hipStream_t s) {
    const size_t lds = (size_t)fold_lds_bytes<F>();
    if (first_on_device(200 + F)) {
        (void)hipFuncSetAttribute((const void*)k_voc_conv_fold<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int fd = F * p.dil;
    const int groups = (p.LS + fd - 1) / fd;
    const int cols = groups * p.dil;
    const dim3 grid((unsigned)((cols + kFoldCols - 1) / kFoldCols), (unsigned)B);
    hipLaunchKernelGGL((k_voc_conv_fold<F>), grid, dim3(kThreads), lds, s, p);
}

extern "C" int dsv_conv1d_folded(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t Co, int32_t K,
                                 int32_t F, int32_t dil, int32_t L, float pre_slope, const float* residual, const float* sum_in, float divide,
                                 int32_t act, void* stream) {
    if (!in || !wpacked || !out) return fail(DSD_ERR_INVALID, "dsv_conv1d_folded: null argument");
    if (B < 1 || B > 65535 || L < 1 || act < 0 || act > 1 || divide == 0.f || (F != 2 && F != 4) || Co < 1 || Co * F > 32 || Ci < 1 || Ci > kFoldMaxCi ||
        K < 1 || !(K & 1) || dil < 1)
        return fail(DSD_ERR_INVALID, "dsv_conv1d_folded: bad shape (B=%d Ci=%d Co=%d K=%d F=%d dil=%d L=%d act=%d)", B, Ci, Co, K, F, dil, L, act);
    const int pad = (K - 1) * dil / 2, KT = K + F - 1;
    const int maxcol = (254 + dil) * F + dil + 30 + (KT - 1) * dil - pad;
    if (pad > kVocHalo - 3 || maxcol >= ((F == 4) ? fold_ld<4>() : fold_ld<2>()))
        return fail(DSD_ERR_INVALID, "dsv_conv1d_folded: kernel %d at dilation %d does not fit the staged tile (ask dsv_fold_factor first)", K, dil);
    VocFoldParams p{};
    p.in = in; p.wp = reinterpret_cast<const float4*>(wpacked); p.bias = bias; p.out = out; p.res = residual; p.sum_in = sum_in;
    p.Ci = Ci; p.Co = Co; p.KT = KT; p.pad = pad; p.dil = dil; p.L = L; p.LS = voc_ls(L);
    p.pre_slope = pre_slope; p.divide = divide; p.act = act;
    if (F == 4) voc_fold_launch<4>(p, B, (hipStream_t)stream);
    else voc_fold_launch<2>(p, B, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// fused ResBlock1 chains (voc_chain.hpp)
// ------------------------------------------------------------------------------------------------------------
template <int C, int F, int NB>
static int voc_chain_geometry(const dsv_chain_conv* convs, int nres, int npairs, int* N_out, int* Hh_out) {
    // halo = the receptive field of the longest chain; every convolution must cover N + 2 Hh samples
    constexpr int NCOL = 128 * NB;
    int hh = 0, cover = NCOL * F;
    for (int r = 0; r < nres; ++r) {
        int h = 0;
        for (int i = 0; i < 2 * npairs; ++i) {
            const dsv_chain_conv& c = convs[r * 2 * npairs + i];
            const int pad = (c.K - 1) * c.dil / 2;
            if (c.K < 1 || !(c.K & 1) || c.dil < 1 || pad > kVocHalo || F * c.dil > kChainSlack || ((i & 1) && c.dil != 1)) return -1;
            h += pad;
            cover = std::min(cover, (NCOL / c.dil) * F * c.dil);
        }
        hh = std::max(hh, h);
    }
    hh = (hh + 3) / 4 * 4;
    const int n = (cover - 2 * hh) / 32 * 32;
    if (n < 64) return -1;
    *N_out = n; *Hh_out = hh;
    return 0;
}

template <int C, int F, int NB, bool IP>
static int voc_chain_launch(VocChainParams& p, const dsv_chain_conv* convs, int B, hipStream_t s) {
    if (voc_chain_geometry<C, F, NB>(convs, p.nres, p.npairs, &p.N, &p.Hh) != 0)
        return fail(DSD_ERR_INVALID, "dsv_resblock_chain: the chain does not fit the staged tile (ask dsv_chain_supported first)");
    constexpr int lds = chain_lds_bytes<C, F, NB, IP>();
    if (first_on_device(300 + C + 1000 * NB + 10000 * (int)IP))
        HIP_TRY(hipFuncSetAttribute((const void*)k_voc_chain<C, F, NB, IP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const dim3 grid((unsigned)((p.LS + p.N - 1) / p.N), (unsigned)B);
    hipLaunchKernelGGL((k_voc_chain<C, F, NB, IP>), grid, dim3(kThreads), lds, s, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// Which instantiation a channel count runs on: column blocks per wave NB (window = 128 NB F samples) and one tile in place / two tiles.
// dsv_set_chain_variant is the A/B switch of the measurement (profiles/r6_*_voc_chain_variants.jsonl); every variant is bit-identical.
struct ChainVariant { int nb, ip; };
static ChainVariant g_chain_variant[3] = {{2, 1}, {2, 1}, {4, 1}};       // C = 8, 16, 32: the fastest of each (profiles/r6_02_voc_chain_variants_modes.jsonl, r6_03_voc_chain_variants.jsonl)
static inline int chain_slot(int C) { return C == 8 ? 0 : C == 16 ? 1 : C == 32 ? 2 : -1; }
static inline bool chain_variant_built(int C, int nb, int ip) {
    if (C == 32) return nb == 4;                                         // <32,1,4,false> (rounds 3-5), <32,1,4,true>
    return (nb == 2) || (nb == 4 && ip == 1);                            // <C,F,2,false> (rounds 3-5), <C,F,2,true>, <C,F,4,true>
}

extern "C" int dsv_set_chain_variant(int32_t C, int32_t nb, int32_t in_place) {
    const int sl = chain_slot(C);
    if (sl < 0 || !chain_variant_built(C, nb, in_place ? 1 : 0))
        return fail(DSD_ERR_INVALID, "dsv_set_chain_variant: no kernel for C=%d with %d column blocks per wave, %s", C, nb, in_place ? "one tile in place" : "two tiles");
    g_chain_variant[sl] = ChainVariant{nb, in_place ? 1 : 0};
    return DSD_OK;
}

static unsigned long long* g_chain_dbg = nullptr;
extern "C" int dsv_debug_chain_timeline(uint64_t* device_stamps) { g_chain_dbg = (unsigned long long*)device_stamps; return DSD_OK; }

extern "C" int32_t dsv_chain_supported(int32_t C, int32_t nres, int32_t npairs, const dsv_chain_conv* convs) {
    if (!convs || nres < 1 || npairs < 1 || nres * npairs * 2 > kChainMaxConvs) return 0;
    int n = 0, hh = 0, rc = -1;
    const int sl = chain_slot(C);
    if (sl < 0) return 0;
    const int nb = g_chain_variant[sl].nb;
    if (C == 8) rc = (nb == 4) ? voc_chain_geometry<8, 4, 4>(convs, nres, npairs, &n, &hh) : voc_chain_geometry<8, 4, 2>(convs, nres, npairs, &n, &hh);
    else if (C == 16) rc = (nb == 4) ? voc_chain_geometry<16, 2, 4>(convs, nres, npairs, &n, &hh) : voc_chain_geometry<16, 2, 2>(convs, nres, npairs, &n, &hh);
    else rc = voc_chain_geometry<32, 1, 4>(convs, nres, npairs, &n, &hh);
    return rc == 0 ? n : 0;
}

extern "C" int32_t dsv_chain_fold(int32_t C) { return C == 8 ? 4 : C == 16 ? 2 : C == 32 ? 1 : 0; }

extern "C" int dsv_resblock_chain(const float* in, const float* wpacked, const float* bias, float* out, const float* sum_in, int32_t B, int32_t C,
                                  int32_t L, int32_t nres, int32_t npairs, const dsv_chain_conv* convs, float pre_slope, float divide, void* stream) {
    if (!in || !wpacked || !bias || !out || !convs) return fail(DSD_ERR_INVALID, "dsv_resblock_chain: null argument");
    if (in == out) return fail(DSD_ERR_INVALID, "dsv_resblock_chain: in and out must be different buffers (workgroups read their neighbours' samples)");
    if (sum_in == out) return fail(DSD_ERR_INVALID, "dsv_resblock_chain: sum_in and out must be different buffers (out holds the running sum over the resblocks of the call)");
    if (!(pre_slope >= 0.f && pre_slope <= 1.f)) return fail(DSD_ERR_INVALID, "dsv_resblock_chain: pre_slope must be in [0, 1] (leaky_relu as max(v, slope v))");
    if (B < 1 || B > 65535 || L < 1 || nres < 1 || npairs < 1 || nres * npairs * 2 > kChainMaxConvs || divide == 0.f || !dsv_chain_fold(C))
        return fail(DSD_ERR_INVALID, "dsv_resblock_chain: bad shape (B=%d C=%d L=%d nres=%d npairs=%d): 8, 16 or 32 channels, at most %d convolutions", B, C,
                    L, nres, npairs, kChainMaxConvs);
    VocChainParams p{};
    p.in = in; p.out = out; p.sum_in = sum_in; p.wp = reinterpret_cast<const float4*>(wpacked); p.bias = bias;
    p.L = L; p.LS = voc_ls(L); p.nres = nres; p.npairs = npairs; p.slope = pre_slope; p.divide = divide; p.dbg = g_chain_dbg;
    const int F = dsv_chain_fold(C);
    for (int i = 0; i < nres * npairs * 2; ++i) {
        const dsv_chain_conv& c = convs[i];
        if (c.w_offset < 0 || (c.w_offset % 256) || c.bias_offset < 0)
            return fail(DSD_ERR_INVALID, "dsv_resblock_chain: convolution %d: weight offsets are multiples of 256 floats (whole chunks)", i);
        p.conv[i].woff = (int)(c.w_offset / 4); p.conv[i].boff = c.bias_offset; p.conv[i].KT = c.K + F - 1; p.conv[i].dil = c.dil;
        p.conv[i].pad = (c.K - 1) * c.dil / 2;
    }
    const ChainVariant v = g_chain_variant[chain_slot(C)];
    hipStream_t st = (hipStream_t)stream;
    if (C == 32) return v.ip ? voc_chain_launch<32, 1, 4, true>(p, convs, B, st) : voc_chain_launch<32, 1, 4, false>(p, convs, B, st);
    if (C == 16) {
        if (v.nb == 4) return voc_chain_launch<16, 2, 4, true>(p, convs, B, st);
        return v.ip ? voc_chain_launch<16, 2, 2, true>(p, convs, B, st) : voc_chain_launch<16, 2, 2, false>(p, convs, B, st);
    }
    if (v.nb == 4) return voc_chain_launch<8, 4, 4, true>(p, convs, B, st);
    return v.ip ? voc_chain_launch<8, 4, 2, true>(p, convs, B, st) : voc_chain_launch<8, 4, 2, false>(p, convs, B, st);
}

// ------------------------------------------------------------------------------------------------------------
// several independent resblocks of a stage in ONE launch + the launch that sums them (voc_chain.hpp, MG instantiations)
// ------------------------------------------------------------------------------------------------------------
static int voc_chain_fill_convs(VocChainParams& p, const dsv_chain_conv* convs, int n, int F, const char* who) {
    for (int i = 0; i < n; ++i) {
        const dsv_chain_conv& c = convs[i];
        if (c.w_offset < 0 || (c.w_offset % 256) || c.bias_offset < 0)
            return fail(DSD_ERR_INVALID, "%s: convolution %d: weight offsets are multiples of 256 floats (whole chunks)", who, i);
        p.conv[i].woff = (int)(c.w_offset / 4); p.conv[i].boff = c.bias_offset; p.conv[i].KT = c.K + F - 1; p.conv[i].dil = c.dil;
        p.conv[i].pad = (c.K - 1) * c.dil / 2;
    }
    return DSD_OK;
}

template <int C, int F, int NB>
static int voc_chain_launch_groups(VocChainParams& p, const dsv_chain_conv* convs, int B, hipStream_t s, const char* who) {
    long blocks = 0;
    for (int g = 0; g < p.ngroups; ++g) {
        VocChainGroup& G = p.grp[g];
        if (voc_chain_geometry<C, F, NB>(convs + (size_t)g * 2 * p.npairs, 1, p.npairs, &G.N, &G.Hh) != 0)
            return fail(DSD_ERR_INVALID, "%s: resblock %d does not fit the staged tile (ask dsv_chain_supported first)", who, g);
        G.tiles = (p.LS + G.N - 1) / G.N;
        G.first = (int)blocks;
        blocks += (long)B * G.tiles;
    }
    if (blocks > 0x7fffffffL) return fail(DSD_ERR_INVALID, "%s: too many workgroups", who);
    constexpr int lds = chain_lds_bytes<C, F, NB, true>();
    if (first_on_device(500 + C + 1000 * NB))
        HIP_TRY(hipFuncSetAttribute((const void*)k_voc_chain<C, F, NB, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((k_voc_chain<C, F, NB, true, true>), dim3((unsigned)blocks), dim3(kThreads), lds, s, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

static int voc_chain_groups(VocChainParams& p, const dsv_chain_conv* convs, int B, int C, hipStream_t s, const char* who) {
    // (the window of dsv_set_chain_variant; always the one-tile-in-place form)
    if (C == 32) return voc_chain_launch_groups<32, 1, 4>(p, convs, B, s, who);
    const int nb = g_chain_variant[chain_slot(C)].nb;
    if (C == 16) return nb == 4 ? voc_chain_launch_groups<16, 2, 4>(p, convs, B, s, who) : voc_chain_launch_groups<16, 2, 2>(p, convs, B, s, who);
    return nb == 4 ? voc_chain_launch_groups<8, 4, 4>(p, convs, B, s, who) : voc_chain_launch_groups<8, 4, 2>(p, convs, B, s, who);
}

extern "C" int dsv_resblock_chain_multi(const float* in, const float* wpacked, const float* bias, float* const* outs, int32_t B, int32_t C, int32_t L,
                                        int32_t ngroups, int32_t npairs, const dsv_chain_conv* convs, float pre_slope, void* stream) {
    if (!in || !wpacked || !bias || !outs || !convs) return fail(DSD_ERR_INVALID, "dsv_resblock_chain_multi: null argument");
    if (!(pre_slope >= 0.f && pre_slope <= 1.f)) return fail(DSD_ERR_INVALID, "dsv_resblock_chain_multi: pre_slope must be in [0, 1] (leaky_relu as max(v, slope v))");
    if (B < 1 || B > 65535 || L < 1 || ngroups < 1 || ngroups > kChainMaxGroups || npairs < 1 || ngroups * npairs * 2 > kChainMaxConvs || !dsv_chain_fold(C))
        return fail(DSD_ERR_INVALID, "dsv_resblock_chain_multi: bad shape (B=%d C=%d L=%d ngroups=%d npairs=%d): 8, 16 or 32 channels, at most %d resblocks and %d convolutions",
                    B, C, L, ngroups, npairs, kChainMaxGroups, kChainMaxConvs);
    VocChainParams p{};
    p.in = in; p.wp = reinterpret_cast<const float4*>(wpacked); p.bias = bias;
    p.L = L; p.LS = voc_ls(L); p.nres = 1; p.npairs = npairs; p.slope = pre_slope; p.divide = 1.f; p.dbg = nullptr;
    p.ngroups = ngroups;
    for (int g = 0; g < ngroups; ++g) {
        if (!outs[g] || outs[g] == in) return fail(DSD_ERR_INVALID, "dsv_resblock_chain_multi: outs[%d] is null or the input", g);
        for (int k = 0; k < g; ++k)
            if (outs[k] == outs[g]) return fail(DSD_ERR_INVALID, "dsv_resblock_chain_multi: outs[%d] == outs[%d] (every resblock writes its own buffer)", k, g);
        p.grp[g].out = outs[g]; p.grp[g].final = 0;
    }
    const int rc = voc_chain_fill_convs(p, convs, ngroups * npairs * 2, dsv_chain_fold(C), "dsv_resblock_chain_multi");
    if (rc != DSD_OK) return rc;
    return voc_chain_groups(p, convs, B, C, (hipStream_t)stream, "dsv_resblock_chain_multi");
}

extern "C" int dsv_resblock_chain_sum(const float* in, const float* wpacked, const float* bias, float* out, const float* sum_in, const float* sum_in2,
                                      int32_t own_last, int32_t B, int32_t C, int32_t L, int32_t npairs, const dsv_chain_conv* convs, float pre_slope,
                                      float divide, void* stream) {
    if (!in || !wpacked || !bias || !out || !convs || !sum_in || !sum_in2) return fail(DSD_ERR_INVALID, "dsv_resblock_chain_sum: null argument");
    if (in == out || sum_in == out || sum_in2 == out) return fail(DSD_ERR_INVALID, "dsv_resblock_chain_sum: out must differ from in, sum_in and sum_in2");
    if (!(pre_slope >= 0.f && pre_slope <= 1.f)) return fail(DSD_ERR_INVALID, "dsv_resblock_chain_sum: pre_slope must be in [0, 1] (leaky_relu as max(v, slope v))");
    if (B < 1 || B > 65535 || L < 1 || npairs < 1 || npairs * 2 > kChainMaxConvs || divide == 0.f || !dsv_chain_fold(C))
        return fail(DSD_ERR_INVALID, "dsv_resblock_chain_sum: bad shape (B=%d C=%d L=%d npairs=%d): 8, 16 or 32 channels", B, C, L, npairs);
    VocChainParams p{};
    p.in = in; p.out = out; p.sum_in = sum_in; p.sum_in2 = sum_in2; p.own_last = own_last ? 1 : 0;
    p.wp = reinterpret_cast<const float4*>(wpacked); p.bias = bias;
    p.L = L; p.LS = voc_ls(L); p.nres = 1; p.npairs = npairs; p.slope = pre_slope; p.divide = divide; p.dbg = nullptr;
    p.ngroups = 1; p.grp[0].out = out; p.grp[0].final = 1;
    const int rc = voc_chain_fill_convs(p, convs, npairs * 2, dsv_chain_fold(C), "dsv_resblock_chain_sum");
    if (rc != DSD_OK) return rc;
    return voc_chain_groups(p, convs, B, C, (hipStream_t)stream, "dsv_resblock_chain_sum");
}

extern "C" int dsv_noise_conv(const float* har, const float* w, const float* bias, float* out, int32_t B, int32_t C, int32_t K, int32_t stride,
                              int32_t pad, int32_t L_har, int32_t L_out, void* stream) {
    if (!har || !w || !out) return fail(DSD_ERR_INVALID, "dsv_noise_conv: null argument");
    if (B < 1 || B > 65535 || C < 1 || C > 65535 || K < 1 || stride < 1 || pad < 0 || L_har < 1 || L_out < 1 ||
        (int64_t)(L_har + 2 * pad - K) / stride + 1 != L_out)
        return fail(DSD_ERR_INVALID, "dsv_noise_conv: bad shape (B=%d C=%d K=%d stride=%d pad=%d L_har=%d L_out=%d)", B, C, K, stride, pad, L_har, L_out);
    const int LSo = voc_ls(L_out);
    hipLaunchKernelGGL(k_voc_noise_conv, dim3((unsigned)((LSo + 255) / 256), (unsigned)C, (unsigned)B), dim3(256), 0, (hipStream_t)stream, har, w, bias,
                       out, C, K, stride, pad, L_har, voc_ls(L_har), L_out, LSo);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsv_sine_source(const float* f0, const float* rand_ini, const float* noise, const float* lin_w, const float* lin_b, float* sines_ws,
                               float* har, int32_t B, int32_t T, int32_t up, int32_t H, float sample_rate, float sine_amp, float noise_std,
                               float voiced_threshold, void* stream) {
    if (!f0 || !rand_ini || !noise || !lin_w || !lin_b || !sines_ws || !har) return fail(DSD_ERR_INVALID, "dsv_sine_source: null argument");
    if (B < 1 || B > 65535 || T < 1 || up < 1 || H < 1 || H > 64 || !(sample_rate > 0.f) || (int64_t)T * up > (1 << 30))
        return fail(DSD_ERR_INVALID, "dsv_sine_source: bad shape (B=%d T=%d up=%d H=%d sr=%g)", B, T, up, H, (double)sample_rate);
    const int L = T * up, LS = voc_ls(L);
    VocSineParams sp{};
    sp.f0 = f0; sp.rand_ini = rand_ini; sp.sw = sines_ws; sp.T = T; sp.up = up; sp.L = L; sp.H = H; sp.sr = sample_rate; sp.sine_amp = sine_amp;
    hipLaunchKernelGGL(k_voc_sine, dim3((unsigned)H, (unsigned)B), dim3(256), 0, (hipStream_t)stream, sp);
    HIP_TRY(hipGetLastError());
    VocSourceParams mp{};
    mp.f0 = f0; mp.sw = sines_ws; mp.noise = noise; mp.lin_w = lin_w; mp.lin_b = lin_b; mp.har = har;
    mp.T = T; mp.up = up; mp.L = L; mp.LS = LS; mp.H = H;
    mp.noise_std = noise_std; mp.sine_amp = sine_amp; mp.voiced_threshold = voiced_threshold;
    hipLaunchKernelGGL(k_voc_source, dim3((unsigned)((LS + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, mp);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// ParallelWaveGAN generator (csrc/pwg_kernels.hpp)
// ------------------------------------------------------------------------------------------------------------
extern "C" int dsv_pwg_layer(const float* x, const float* c, const float* w1_packed, const float* b1, const float* w2_packed, const float* b2,
                             float* x_out, float* skip, int32_t B, int32_t L, int32_t n_aux, int32_t dil, int32_t first, void* stream) {
    if (!x || !w1_packed || !w2_packed || !x_out || !skip || x == x_out) return fail(DSD_ERR_INVALID, "dsv_pwg_layer: null argument / in-place call");
    if (B < 1 || B > 65535 || L < 1 || dil < 1 || n_aux < 0 || n_aux > kPwgMaxAux || (n_aux % 8) || (n_aux && !c))
        return fail(DSD_ERR_INVALID, "dsv_pwg_layer: bad shape (B=%d L=%d dil=%d aux=%d: aux channels a multiple of 8, at most %d)", B, L, dil, n_aux, kPwgMaxAux);
    if (first_on_device(40)) HIP_TRY(hipFuncSetAttribute((const void*)k_pwg_layer, hipFuncAttributeMaxDynamicSharedMemorySize, kPwgLayerLdsBytes));
    PwgLayerParams p{};
    p.x = x; p.c = n_aux ? c : nullptr; p.w1p = reinterpret_cast<const float4*>(w1_packed); p.b1 = b1;
    p.w2p = reinterpret_cast<const float4*>(w2_packed); p.b2 = b2; p.x_out = x_out; p.skip = skip;
    p.L = L; p.LS = voc_ls(L); p.dil = dil; p.naux = n_aux; p.first = first ? 1 : 0;
    hipLaunchKernelGGL(k_pwg_layer, dim3((unsigned)(p.LS / 32), (unsigned)B), dim3(kThreads), kPwgLayerLdsBytes, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsv_pwg_upsample(const float* in, const float* filter, float* out, int64_t rows, int32_t L_in, int32_t scale, void* stream) {
    if (!in || !filter || !out) return fail(DSD_ERR_INVALID, "dsv_pwg_upsample: null argument");
    if (rows < 1 || rows > 65535 || L_in < 1 || scale < 1 || scale > 64) return fail(DSD_ERR_INVALID, "dsv_pwg_upsample: bad shape (rows=%lld L=%d scale=%d)", (long long)rows, L_in, scale);
    const int LS_in = voc_ls(L_in), LS_out = voc_ls(L_in * scale);
    hipLaunchKernelGGL(k_pwg_upsample, dim3((unsigned)((LS_out + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream, in, filter, out, L_in,
                       LS_in, scale, LS_out);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsv_pwg_first(const float* z, const float* w, const float* bias, float* out, int32_t B, int32_t C, int32_t L, void* stream) {
    if (!z || !w || !out) return fail(DSD_ERR_INVALID, "dsv_pwg_first: null argument");
    if (B < 1 || B > 65535 || C < 1 || C > 65535 || L < 1) return fail(DSD_ERR_INVALID, "dsv_pwg_first: bad shape");
    const int LS = voc_ls(L);
    hipLaunchKernelGGL(k_pwg_first, dim3((unsigned)((LS + 255) / 256), (unsigned)C, (unsigned)B), dim3(256), 0, (hipStream_t)stream, z, w, bias, out, C, L, LS);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}
