// fs2_abi.hpp - host side of the FastSpeech2 conditioner ops (C ABI in include/dsf.h); included at the end of dsd.hip so the
// library stays one translation unit (shares the error string, the HIP_TRY macros and the operand-packing kernels).
#include "fs2_kernels.hpp"
#include "fs2_train.hpp"

#include "../../include/dsf.h"

static inline int fs_ts(int T) { return (T + 31) / 32 * 32; }
// Row blocks per wave of k_fs_conv (fixes the packed weight layout): 2 = 256 rows per workgroup, two workgroups per CU.  (4 - 512 rows, one
// workgroup per CU - was measured slower on the wide layers in round 1, profiles/r01s_*, and its instantiation was dropped in round 3.)
static inline int fs_nmb(int Co) { (void)Co; return 2; }
static int g_fs_conv_split = -1;           // -1: by grid size (default), 0: never, 1: wherever the shape allows it (tests / A-B)
extern "C" int dsf_set_conv_split(int32_t mode) {
    if (mode < -1 || mode > 1) return fail(DSD_ERR_INVALID, "dsf_set_conv_split: mode must be -1 (by grid size), 0 or 1");
    g_fs_conv_split = mode;
    return DSD_OK;
}
static int fs_ncu() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

extern "C" int dsf_padded_frames(int32_t T) { return fs_ts(T); }

extern "C" int64_t dsf_packed_floats(int32_t Co, int32_t Ci, int32_t KT) {
    if (Co < 1 || Ci < 8 || (Ci % 8) || KT < 1) return -1;
    const int nmb = fs_nmb(Co);
    const int64_t mt = (Co + 128 * nmb - 1) / (128 * nmb);
    return (mt * 4 * (int64_t)(Ci / 8) * KT * nmb * 64 + kWeightSlack) * 4;
}

extern "C" int dsf_pack_weight(const float* w, int32_t Co, int32_t Ci, int32_t KT, float* packed, void* stream) {
    if (!w || !packed) return fail(DSD_ERR_INVALID, "dsf_pack_weight: null argument");
    if (Co < 1 || Ci < 8 || (Ci % 8) || KT < 1 || KT > 2 * kFsHalo + 1 || !(KT & 1))
        return fail(DSD_ERR_INVALID, "dsf_pack_weight: need Co >= 1, Ci a multiple of 8, odd kernel <= %d (got %d, %d, %d)", 2 * kFsHalo + 1, Co, Ci, KT);
    const int nmb = fs_nmb(Co);
    const int mt = (Co + 128 * nmb - 1) / (128 * nmb);
    PackParams p{};
    p.src = w; p.dst = packed;
    p.nw = mt * 4; p.nkc = Ci / 8; p.nmb = nmb; p.ntap = KT;
    p.split = 0; p.hi_base = 0;
    p.rows_valid = Co; p.cols_valid = Ci;
    p.row_stride = Ci * KT; p.col_stride = KT;
    const size_t n = (size_t)p.nw * p.ntap * p.nkc * p.nmb * 256;
    hipLaunchKernelGGL(k_pack_a, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(packed + n, 0, (size_t)kWeightSlack * 16, (hipStream_t)stream));
    return DSD_OK;
}

static int fs_conv_launch(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t Co,
                          int32_t KT, int32_t dil, int32_t T, float scale, int32_t act, const float* residual, const float* keep, void* stream);

extern "C" int dsf_conv1d(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t Co,
                          int32_t KT, int32_t T, float scale, int32_t act, const float* residual, const float* keep, void* stream) {
    return fs_conv_launch(in, wpacked, bias, out, B, Ci, Co, KT, 1, T, scale, act, residual, keep, stream);
}

extern "C" int dsf_conv1d_dilated(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t Co,
                                  int32_t KT, int32_t dil, int32_t T, void* stream) {
    return fs_conv_launch(in, wpacked, bias, out, B, Ci, Co, KT, dil, T, 1.0f, 0, nullptr, nullptr, stream);
}

static int fs_conv_launch(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t Co,
                          int32_t KT, int32_t dil, int32_t T, float scale, int32_t act, const float* residual, const float* keep, void* stream) {
    if (!in || !wpacked || !out) return fail(DSD_ERR_INVALID, "dsf_conv1d: null argument");
    if (B < 1 || T < 1 || Co < 1 || Ci < 8 || (Ci % 8) || KT < 1 || KT > 2 * kFsHalo + 1 || !(KT & 1) || act < 0 || act > 3 || dil < 1 ||
        dil * (KT - 1) / 2 > kFsHalo)
        return fail(DSD_ERR_INVALID, "dsf_conv1d: bad shape (B=%d T=%d Ci=%d Co=%d K=%d dil=%d act=%d)", B, T, Ci, Co, KT, dil, act);
    if (first_on_device(10)) {
        (void)hipFuncSetAttribute((const void*)k_fs_conv<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kFsConvLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_fs_conv_ks, hipFuncAttributeMaxDynamicSharedMemorySize, kFsConvLdsBytes);
    }
    FsConvParams p{};
    p.in = in; p.wp = reinterpret_cast<const float4*>(wpacked); p.bias = bias; p.out = out; p.res = residual; p.keep = keep;
    p.Ci = Ci; p.Co = Co; p.KT = KT; p.dil = dil; p.pad = dil * (KT - 1) / 2; p.T = T; p.TS = fs_ts(T);
    p.scale = scale; p.act = act;
    const int nmb = fs_nmb(Co);
    const dim3 grid((unsigned)(p.TS / 32), (unsigned)B, (unsigned)((Co + 128 * nmb - 1) / (128 * nmb)));
    // small grids (at most one workgroup for every second CU): 64-row workgroups whose waves split the contraction (k_fs_conv_ks)
    const long long wgs = (long long)grid.x * grid.y * grid.z;
    if (Ci % 32 == 0 && (g_fs_conv_split == 1 || (g_fs_conv_split < 0 && 2 * wgs <= fs_ncu()))) {
        const dim3 gks(grid.x, grid.y, (unsigned)((Co + 63) / 64));
        hipLaunchKernelGGL(k_fs_conv_ks, gks, dim3(kThreads), kFsConvLdsBytes, (hipStream_t)stream, p);
        HIP_TRY(hipGetLastError());
        return DSD_OK;
    }
    hipLaunchKernelGGL((k_fs_conv<2>), grid, dim3(kThreads), kFsConvLdsBytes, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_layer_norm(const float* in, const float* gamma, const float* beta, float* out, int32_t B, int32_t C, int32_t T,
                              float eps, int32_t relu_in, const float* keep, void* stream) {
    if (!in || !gamma || !beta || !out) return fail(DSD_ERR_INVALID, "dsf_layer_norm: null argument");
    if (C != kC) return fail(DSD_ERR_INVALID, "dsf_layer_norm: this build normalises over %d channels (got %d)", kC, C);
    if (B < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_layer_norm: bad shape");
    FsLnParams p{};
    p.in = in; p.out = out; p.gamma = gamma; p.beta = beta; p.keep = keep; p.T = T; p.TS = fs_ts(T); p.eps = eps; p.relu_in = relu_in;
    hipLaunchKernelGGL(k_fs_ln, dim3((unsigned)(p.TS / 32), (unsigned)B), dim3(kThreads), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_attention(const float* qkv, const uint8_t* key_pad, float* out, int32_t B, int32_t C, int32_t heads, int32_t T,
                             void* stream) {
    if (!qkv || !out) return fail(DSD_ERR_INVALID, "dsf_attention: null argument");
    if (B < 1 || T < 1 || heads < 1 || C != heads * 128)
        return fail(DSD_ERR_INVALID, "dsf_attention: this build supports head_dim 128 (C=%d, heads=%d)", C, heads);
    FsAttnParams p{};
    p.qkv = qkv; p.key_pad = key_pad; p.out = out; p.C = C; p.T = T; p.TS = fs_ts(T);
    p.scale = (float)std::sqrt(1.0 / 128.0);
    if (first_on_device(11)) {
        (void)hipFuncSetAttribute((const void*)k_fs_attn<128>, hipFuncAttributeMaxDynamicSharedMemorySize, fs_attn_lds_bytes<128>());
    }
    hipLaunchKernelGGL((k_fs_attn<128>), dim3((unsigned)(p.TS / 32), (unsigned)heads, (unsigned)B), dim3(kThreads), fs_attn_lds_bytes<128>(),
                       (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// backward of LayerNorm and of the attention core (fs2_train.hpp)
// ------------------------------------------------------------------------------------------------------------
extern "C" int64_t dsf_ln_bwd_workspace_floats(int32_t B, int32_t T) {
    if (B < 1 || T < 1) return -1;
    return (int64_t)B * (fs_ts(T) / 32) * 512;
}

extern "C" int dsf_layer_norm_bwd(const float* x, const float* gamma, const float* dy, const float* keep, float* dx, float* dgamma, float* dbeta,
                                  float* ws, int32_t B, int32_t C, int32_t T, float eps, int32_t relu_in, void* stream) {
    if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || !ws) return fail(DSD_ERR_INVALID, "dsf_layer_norm_bwd: null argument");
    if (C != kC) return fail(DSD_ERR_INVALID, "dsf_layer_norm_bwd: this build normalises over %d channels (got %d)", kC, C);
    if (B < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_layer_norm_bwd: bad shape");
    FsLnBwdParams p{};
    p.x = x; p.dy = dy; p.gamma = gamma; p.keep = keep; p.dx = dx; p.part = ws; p.T = T; p.TS = fs_ts(T); p.eps = eps; p.relu_in = relu_in;
    const int ntile = p.TS / 32;
    hipLaunchKernelGGL(k_fs_ln_bwd, dim3((unsigned)ntile, (unsigned)B), dim3(kThreads), 0, (hipStream_t)stream, p);
    // partials are [B * ntile][dgamma 256 | dbeta 256]
    hipLaunchKernelGGL(k_fs_colsum, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, dgamma, B * ntile, 256, 512);
    hipLaunchKernelGGL(k_fs_colsum, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws + 256, dbeta, B * ntile, 256, 512);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int64_t dsf_attention_bwd_workspace_floats(int32_t B, int32_t heads, int32_t T) {
    if (B < 1 || heads < 1 || T < 1) return -1;
    return (int64_t)2 * B * heads * T * T;
}

static void fs_bmm(hipStream_t s, const float* A, const float* Bm, float* C, int M, int N, int K, long long am, long long ak, long long bk, long long bn,
                   long long cm, long long cn, int outer, int inner, long long a_o, long long a_i, long long b_o, long long b_i, long long c_o, long long c_i,
                   float alpha) {
    FsBmmParams p{A, Bm, C, M, N, K, am, ak, bk, bn, cm, cn, inner, a_o, a_i, b_o, b_i, c_o, c_i, alpha};
    hipLaunchKernelGGL(k_fs_bmm, dim3((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32), (unsigned)(outer * inner)), dim3(256), 0, s, p);
}

// torch.nn.Linear on a handful of ROWS (the step-embedding MLP and the layers' step projections of the denoiser under training:
// usr/diff/net.py:94-98, :119-120, :67 - [B, in] x [in, out] with B = the batch size): y = x W^T + b, and its three gradients.  Vector-ALU
// products (k_fs_bmm): the work is a few MFLOP, the point is that no vendor BLAS sits on the path.  A product with few output tiles and a
// long contraction (dx of the 20 stacked step projections: 8 x 256 outputs over K = 5120) is split over K as the batch index of k_fs_bmm;
// the partial products are added in split order (k_fs_colsum: deterministic).
static int lin_splits(int M, int N, int K) {
    const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
    if (tiles >= 64) return 1;
    for (int sp : {32, 16, 8, 4, 2})
        if (K % (sp * 32) == 0 && K / sp >= 64) return sp;
    return 1;
}

static void lin_product(hipStream_t s, const float* A, const float* Bm, float* C, float* ws, int M, int N, int K, long long am, long long ak, long long bk,
                        long long bn) {
    const int sp = ws ? lin_splits(M, N, K) : 1;
    if (sp == 1) {
        fs_bmm(s, A, Bm, C, M, N, K, am, ak, bk, bn, N, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f);
        return;
    }
    const long long kc = K / sp;
    fs_bmm(s, A, Bm, ws, M, N, (int)kc, am, ak, bk, bn, N, 1, sp, 1, kc * ak, 0, kc * bk, 0, (long long)M * N, 0, 1.f);
    hipLaunchKernelGGL(k_fs_colsum, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, s, ws, C, sp, M * N, M * N);
}

extern "C" int64_t dsf_linear_rows_workspace_floats(int32_t rows, int32_t n_in, int32_t n_out) {
    if (rows < 1 || n_in < 1 || n_out < 1) return -1;
    return (int64_t)32 * rows * std::max(n_in, n_out);
}

extern "C" int dsf_linear_rows(const float* x, const float* w, const float* bias, float* y, float* ws, int32_t rows, int32_t n_in, int32_t n_out,
                               void* stream) {
    if (!x || !w || !y) return fail(DSD_ERR_INVALID, "dsf_linear_rows: null argument");
    if (rows < 1 || n_in < 1 || n_out < 1) return fail(DSD_ERR_INVALID, "dsf_linear_rows: empty shape");
    hipStream_t s = (hipStream_t)stream;
    lin_product(s, x, w, y, ws, rows, n_out, n_in, n_in, 1, 1, n_in);                       // y[m][o] = sum_i x[m][i] w[o][i]
    if (bias) hipLaunchKernelGGL(k_fs_add_row_bias, dim3((unsigned)((n_out + 255) / 256), (unsigned)rows), dim3(256), 0, s, y, bias, n_out);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// dy [rows][n_out] -> dx [rows][n_in] (or NULL), dw [n_out][n_in] (or NULL), db [n_out] (or NULL; fixed-order column sum)
extern "C" int dsf_linear_rows_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, float* ws, int32_t rows,
                                   int32_t n_in, int32_t n_out, void* stream) {
    if (!x || !w || !dy) return fail(DSD_ERR_INVALID, "dsf_linear_rows_bwd: null argument");
    if (rows < 1 || n_in < 1 || n_out < 1) return fail(DSD_ERR_INVALID, "dsf_linear_rows_bwd: empty shape");
    hipStream_t s = (hipStream_t)stream;
    if (dx) lin_product(s, dy, w, dx, ws, rows, n_in, n_out, n_out, 1, n_in, 1);                // dx[m][i] = sum_o dy[m][o] w[o][i]
    if (dw) lin_product(s, dy, x, dw, nullptr, n_out, n_in, rows, 1, n_out, n_in, 1);          // dw[o][i] = sum_m dy[m][o] x[m][i]
    if (db) hipLaunchKernelGGL(k_fs_colsum, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, dy, db, rows, n_out, n_out);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_attention_bwd(const float* qkv, const uint8_t* key_pad, const float* dout, float* dqkv, float* ws, int32_t B, int32_t C,
                                 int32_t heads, int32_t T, void* stream) {
    if (!qkv || !dout || !dqkv || !ws) return fail(DSD_ERR_INVALID, "dsf_attention_bwd: null argument");
    if (B < 1 || T < 1 || heads < 1 || C != heads * 128)
        return fail(DSD_ERR_INVALID, "dsf_attention_bwd: this build supports head_dim 128 (C=%d, heads=%d)", C, heads);
    if ((int64_t)B * heads > 65535)
        return fail(DSD_ERR_INVALID, "dsf_attention_bwd: B * heads = %lld exceeds the 65535 (batch, head) pairs of one launch: split the batch",
                    (long long)B * heads);
    hipStream_t s = (hipStream_t)stream;
    const int TS = fs_ts(T), HD = 128, BH = B * heads;
    const long long ts = TS, tt = T;
    const float scale = (float)std::sqrt(1.0 / 128.0);
    float* P = ws;                                       // [BH][T][T] probabilities
    float* D = ws + (size_t)BH * T * T;                  // [BH][T][T] dP, then dS
    const float *q = qkv, *k = qkv + (size_t)C * TS, *v = qkv + (size_t)2 * C * TS;
    float *dq = dqkv, *dk = dqkv + (size_t)C * TS, *dv = dqkv + (size_t)2 * C * TS;
    const long long qo = 3LL * C * TS, qi = (long long)HD * TS;          // batch strides of q / k / v and their gradients
    const long long oo = (long long)C * TS, oi = (long long)HD * TS;     // ... of the attention output / its gradient
    const long long po = (long long)heads * tt * tt, pi = tt * tt;
    HIP_TRY(hipMemsetAsync(dqkv, 0, (size_t)B * 3 * C * TS * sizeof(float), s));                        // zero tails [T, TS)
    // S[tq][tk] = scale * sum_d q[d][tq] k[d][tk]
    fs_bmm(s, q, k, P, T, T, HD, 1, ts, ts, 1, tt, 1, B, heads, qo, qi, qo, qi, po, pi, scale);
    hipLaunchKernelGGL(k_fs_softmax_rows, dim3((unsigned)T, (unsigned)BH), dim3(256), 0, s, P, (const unsigned char*)key_pad, T, heads);
    // dP[tq][tk] = sum_d dO[d][tq] v[d][tk]
    fs_bmm(s, dout, v, D, T, T, HD, 1, ts, ts, 1, tt, 1, B, heads, oo, oi, qo, qi, po, pi, 1.f);
    // dv[d][tk] = sum_tq dO[d][tq] P[tq][tk]
    fs_bmm(s, dout, P, dv, HD, T, T, ts, 1, tt, 1, ts, 1, B, heads, oo, oi, po, pi, qo, qi, 1.f);
    hipLaunchKernelGGL(k_fs_softmax_bwd_rows, dim3((unsigned)T, (unsigned)BH), dim3(256), 0, s, (const float*)P, D, T);
    // dq[d][tq] = scale * sum_tk k[d][tk] dS[tq][tk] ; dk[d][tk] = scale * sum_tq q[d][tq] dS[tq][tk]
    fs_bmm(s, k, D, dq, HD, T, T, ts, 1, 1, tt, ts, 1, B, heads, qo, qi, po, pi, qo, qi, scale);
    fs_bmm(s, q, D, dk, HD, T, T, ts, 1, tt, 1, ts, 1, B, heads, qo, qi, po, pi, qo, qi, scale);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_to_channel_major(const float* x, int64_t stride_b, int64_t stride_c, int64_t stride_t, float* out, int32_t B,
                                    int32_t C, int32_t T, void* stream) {
    if (!x || !out || B < 1 || C < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_to_channel_major: bad argument");
    const int TS = fs_ts(T);
    hipLaunchKernelGGL(k_cond_layout, dim3((unsigned)(TS / 32), (unsigned)((C + 31) / 32), (unsigned)B), dim3(32, 8), 0, (hipStream_t)stream,
                       x, out, C, T, TS, stride_b, stride_c, stride_t);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_positions(const int64_t* tokens, const float* x, int32_t* pos, int32_t B, int32_t T, int32_t C, int32_t padding_idx, void* stream) {
    if ((!tokens && !x) || !pos || B < 1 || B > 65535 || T < 1 || (x && C < 1))
        return fail(DSD_ERR_INVALID, "dsf_positions: bad argument (tokens or x, pos; B=%d T=%d C=%d)", B, T, C);
    FsPosParams p{};
    p.tok = (const long long*)tokens; p.x = x; p.pos = pos; p.T = T; p.C = C; p.pad = padding_idx;
    hipLaunchKernelGGL(k_fs_positions, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_input_cm(const int64_t* tokens, const float* emb, float emb_scale, const float* add0, const float* add1, const float* add2, const float* x,
                            const int32_t* pos, const float* pos_table, const float* alpha_dev, const uint8_t* padding_mask, float* xc, float* keep,
                            uint8_t* pad_out, int32_t B, int32_t T, int32_t C, int32_t padding_idx, int32_t mask_mode, void* stream) {
    if ((!tokens) == (!x) || (tokens && !emb) || !xc || !keep || !pad_out || (pos && !pos_table))
        return fail(DSD_ERR_INVALID, "dsf_input_cm: exactly one of tokens (+ emb) / x, and xc, keep, pad_out");
    if (B < 1 || B > 65535 || T < 1 || C < 4 || (C & 3) || C > kFsInMaxC || mask_mode < 0 || mask_mode > 1)
        return fail(DSD_ERR_INVALID, "dsf_input_cm: bad shape (B=%d T=%d C=%d mask_mode=%d; C a multiple of 4, <= %d)", B, T, C, mask_mode, kFsInMaxC);
    FsInputParams p{};
    p.tok = (const long long*)tokens; p.emb = emb; p.emb_scale = emb_scale; p.add[0] = add0; p.add[1] = add1; p.add[2] = add2; p.x = x;
    p.pos = pos; p.pos_tab = pos_table; p.alpha = alpha_dev; p.pad_in = padding_mask; p.xc = xc; p.keep = keep; p.pad_out = pad_out;
    p.T = T; p.TS = fs_ts(T); p.C = C; p.pad = padding_idx; p.mask_mode = mask_mode;
    const size_t lds = (size_t)C * 33 * sizeof(float);
    if (first_on_device(60)) HIP_TRY(hipFuncSetAttribute((const void*)k_fs_input_cm, hipFuncAttributeMaxDynamicSharedMemorySize, kFsInMaxC * 33 * (int)sizeof(float)));
    hipLaunchKernelGGL(k_fs_input_cm, dim3((unsigned)(p.TS / 32), (unsigned)B), dim3(256), lds, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_gather_frames(const float* enc, const int64_t* mel2ph, const float* spk, float* out, float* out_masked, int32_t B, int32_t T, int32_t T_src,
                                 int32_t C, void* stream) {
    if (!enc || !mel2ph || !out || B < 1 || B > 65535 || T < 1 || T_src < 1 || C < 4 || (C & 3))
        return fail(DSD_ERR_INVALID, "dsf_gather_frames: bad argument (B=%d T=%d T_src=%d C=%d; C a multiple of 4)", B, T, T_src, C);
    FsGatherParams p{};
    p.enc = enc; p.mel2ph = (const long long*)mel2ph; p.spk = spk; p.out1 = out; p.out2 = out_masked; p.T = T; p.Tp = T_src; p.C = C;
    hipLaunchKernelGGL(k_fs_gather_frames, dim3((unsigned)((T + 7) / 8), (unsigned)B), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_sum_embed(const float* dec, const int64_t* idx1, const float* tab1, const float* add1, const int64_t* idx2, const float* tab2, const float* spk,
                             const int64_t* mel2ph, float* out, int32_t B, int32_t T, int32_t C, void* stream) {
    if (!dec || !mel2ph || !out || (idx1 && !tab1) || (idx2 && !tab2) || (idx1 && add1) || B < 1 || B > 65535 || T < 1 || C < 4 || (C & 3))
        return fail(DSD_ERR_INVALID, "dsf_sum_embed: bad argument (B=%d T=%d C=%d; C a multiple of 4)", B, T, C);
    FsSumEmbedParams p{};
    p.dec = dec; p.idx1 = (const long long*)idx1; p.tab1 = tab1; p.add1 = add1; p.idx2 = (const long long*)idx2; p.tab2 = tab2; p.spk = spk;
    p.mel2ph = (const long long*)mel2ph; p.out = out; p.T = T; p.C = C;
    hipLaunchKernelGGL(k_fs_sum_embed, dim3((unsigned)((T + 7) / 8), (unsigned)B), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_q_sample_rows(const float* x_start, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac, int32_t n_steps,
                                 float* out, int32_t B, int64_t per_row, void* stream) {
    if (!x_start || !noise || !t || !sqrt_ac || !sqrt_1mac || !out || n_steps < 1 || B < 1 || B > 65535 || per_row < 4 || (per_row & 3) || per_row > ((int64_t)1 << 32))
        return fail(DSD_ERR_INVALID, "dsf_q_sample_rows: bad argument (B=%d per_row=%lld; per_row a multiple of 4)", B, (long long)per_row);
    const int per4 = (int)(per_row / 4);
    hipLaunchKernelGGL(k_fs_q_sample_rows, dim3((unsigned)std::min(64, (per4 + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, x_start, noise,
                       (const long long*)t, sqrt_ac, sqrt_1mac, out, per4, (int)n_steps);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

constexpr int kL1Blocks = 256;
extern "C" int64_t dsf_l1_workspace_floats(void) { return kL1Blocks; }

extern "C" int dsf_l1_mean(const float* a, const float* b, float* workspace, float* out, int64_t n, void* stream) {
    if (!a || !b || !workspace || !out || n < 1) return fail(DSD_ERR_INVALID, "dsf_l1_mean: bad argument");
    const int nblk = (int)std::min<int64_t>(kL1Blocks, (n + 255) / 256);
    hipLaunchKernelGGL(k_fs_l1_partial, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a, b, workspace, (size_t)n);
    hipLaunchKernelGGL(k_fs_l1_final, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk, (float)(1.0 / (double)n), out);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_l1_mean_bwd(const float* a, const float* b, const float* grad_out, float* db, int64_t n, void* stream) {
    if (!a || !b || !grad_out || !db || n < 1) return fail(DSD_ERR_INVALID, "dsf_l1_mean_bwd: bad argument");
    hipLaunchKernelGGL(k_fs_l1_bwd, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, grad_out,
                       (float)(1.0 / (double)n), db, (size_t)n);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_token_masks(const int64_t* v, float* gt0, uint8_t* eq0, float* ne0, int64_t n, void* stream) {
    if (!v || (!gt0 && !eq0 && !ne0) || n < 1 || n > ((int64_t)1 << 38)) return fail(DSD_ERR_INVALID, "dsf_token_masks: bad argument (v, one output at least, n=%lld)", (long long)n);
    FsTokMaskParams p{};
    p.v = (const long long*)v; p.gt0 = gt0; p.eq0 = eq0; p.ne0 = ne0; p.n = n;
    hipLaunchKernelGGL(k_fs_token_masks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_pitch_coarse(const float* f0, int64_t stride_b, int64_t stride_t, const float* uv_f, const uint8_t* uv_u8, const int64_t* mel2ph,
                                float* f0_denorm, float* tmp, int64_t* coarse, int32_t B, int32_t T, int32_t norm, float f0_mean, float f0_std,
                                double f0_mel_min, double f0_mel_max, int32_t f0_bin, int32_t stage, int32_t pow_in, void* stream) {
    if (!f0 || B < 1 || T < 1 || norm < 1 || norm > 2 || stage < 0 || stage > 2 || (uv_f && uv_u8) || f0_bin < 3 || !(f0_mel_max > f0_mel_min))
        return fail(DSD_ERR_INVALID, "dsf_pitch_coarse: bad argument (B=%d T=%d norm=%d stage=%d f0_bin=%d)", B, T, norm, stage, f0_bin);
    if ((stage != 2 && !f0_denorm) || (stage != 0 && !tmp) || (stage != 1 && !coarse) || (pow_in && norm != 2))
        return fail(DSD_ERR_INVALID, "dsf_pitch_coarse: stage %d needs f0_denorm (0, 1), tmp (1, 2), coarse (0, 2); pow_in only with norm 2", stage);
    FsPitchParams p{};
    p.f0 = f0; p.sb = stride_b; p.st = stride_t; p.uv_f = uv_f; p.uv_u8 = uv_u8; p.mel2ph = (const long long*)mel2ph;
    p.f0_denorm = f0_denorm; p.tmp = tmp; p.coarse = (long long*)coarse; p.T = T; p.norm = norm; p.stage = stage; p.pow_in = pow_in ? 1 : 0;
    p.n = (long long)B * T;
    p.base = 2.0f; p.mean = f0_mean; p.std = f0_std;
    // the scalars as ATen's kernels see them: a Python / numpy double narrowed to the tensor's fp32, the divisor as its fp32 reciprocal
    p.inv700 = 1.0f / 700.0f;
    p.mel_min = (float)f0_mel_min;
    p.scale = (float)(f0_bin - 2);
    p.inv_range = 1.0f / (float)(f0_mel_max - f0_mel_min);
    p.top = (float)(f0_bin - 1);
    hipLaunchKernelGGL(k_fs_pitch_coarse, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_from_channel_major(const float* in, float* out, int32_t B, int32_t C, int32_t T, void* stream) {
    if (!in || !out || B < 1 || C < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_from_channel_major: bad argument");
    const int TS = fs_ts(T);
    hipLaunchKernelGGL(k_fs_from_cm, dim3((unsigned)(TS / 32), (unsigned)((C + 31) / 32), (unsigned)B), dim3(32, 8), 0, (hipStream_t)stream,
                       in, out, C, T, TS);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_p_sample(float* x, const float* eps, const float* noise, int64_t n, float sqrt_recip_ac, float sqrt_recipm1_ac,
                            float coef1, float coef2, float sigma, void* stream) {
    if (!x || !eps || !noise || n < 1) return fail(DSD_ERR_INVALID, "dsf_p_sample: bad argument");
    hipLaunchKernelGGL(k_fs_p_sample, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, eps, noise,
                       sqrt_recip_ac, sqrt_recipm1_ac, coef1, coef2, sigma, (size_t)n);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_denorm_spec(const float* x, const float* mask, float* mel, const float* spec_min, const float* spec_max, int32_t B,
                               int32_t M, int32_t T, void* stream) {
    if (!x || !mel || !spec_min || !spec_max || B < 1 || M < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_denorm_spec: bad argument");
    hipLaunchKernelGGL(k_denorm_spec, dim3((unsigned)((T + 31) / 32), (unsigned)B), dim3(256), 32 * (M + 1) * 4, (hipStream_t)stream, x, mask, mel,
                       spec_min, spec_max, M, T);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// ---- training operators ---------------------------------------------------------------------
// Frame splits of a weight-gradient launch: 16, or - for the small 1 x 1 outputs (the denoiser's input / skip / output projections: <= 8 output
// tiles, 128 workgroups of a kernel that runs two per CU; their partials are a few MiB) - as many as put two workgroups on every CU of a 256-CU
// part, at most 64.  A function of the shape alone: the workspace query and the launch agree, and the summation order (hence every bit) does
// not depend on the device.
static int fs_wg_splits(int Co, int Ci, int KT) {
    int ns = 16;
    if (KT != 1) return ns;
    const int tiles = ((Co + 127) / 128) * ((Ci + 63) / 64);
    if (tiles > 8) return ns;                   // (wider outputs keep 16: their partials would grow to tens of MiB per layer for little)
    while (ns < 64 && tiles * ns < 512) ns *= 2;
    return ns;
}

extern "C" int64_t dsf_wgrad_workspace_floats(int32_t Co, int32_t Ci, int32_t KT) {
    if (Co < 1 || Ci < 1 || KT < 1 || !(KT & 1) || KT > 2 * kFsHalo + 1) return -1;
    const int ns = fs_wg_splits(Co, Ci, KT);
    return (int64_t)ns * Co * Ci * std::min(KT, 3) + (int64_t)ns * Co;
}

// Kernels wider than 3 taps (FastSpeech2: the k = 9 conv-FFN, the k = 5 pitch predictor) run as groups of three taps: the kernel computes the taps
// k = 0..2 at the frame shifts k * dil - pad', and pad' = pad - tap0 * dil selects the group; k_fs_wgrad_reduce_taps adds the frame splits in a
// fixed order and writes the group's taps into dW [Co][Ci][KT].
extern "C" int dsf_conv1d_wgrad(const float* dy, const float* x, float* dw, float* db, float* workspace, int32_t B, int32_t Ci, int32_t Co,
                                int32_t KT, int32_t dil, int32_t T, int32_t accumulate, void* stream) {
    if (!dy || !x || !dw || !workspace) return fail(DSD_ERR_INVALID, "dsf_conv1d_wgrad: null argument");
    if (B < 1 || T < 1 || Co < 1 || Ci < 1 || KT < 1 || !(KT & 1) || dil < 1 || dil * (KT - 1) / 2 > kFsHalo)
        return fail(DSD_ERR_INVALID, "dsf_conv1d_wgrad: bad shape (B=%d T=%d Ci=%d Co=%d K=%d dil=%d); odd kernels whose taps stay within +-%d frames", B, T, Ci, Co, KT,
                    dil, kFsHalo);
    const int pad = dil * (KT - 1) / 2;
    const int gk = std::min(KT, 3);
    const int kWgSplits = fs_wg_splits(Co, Ci, KT);
    float* part_b = workspace + (size_t)kWgSplits * Co * Ci * gk;
    const dim3 grid((unsigned)((Co + 127) / 128), (unsigned)((Ci + 63) / 64), (unsigned)kWgSplits);
    for (int tap0 = 0; tap0 < KT; tap0 += 3) {
        const int ntap = std::min(3, KT - tap0);
        FsWgradParams p{};
        p.dy = dy; p.x = x; p.part = workspace; p.part_b = (db && tap0 == 0) ? part_b : nullptr;
        p.B = B; p.Ci = Ci; p.Co = Co; p.dil = dil; p.pad = pad - tap0 * dil; p.T = T; p.TS = fs_ts(T); p.nsplit = kWgSplits;
        if (KT == 1) hipLaunchKernelGGL((k_fs_wgrad<1>), grid, dim3(kThreads), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((k_fs_wgrad<3>), grid, dim3(kThreads), 0, (hipStream_t)stream, p);
        HIP_TRY(hipGetLastError());
        const size_t nrow = (size_t)Co * Ci;
        hipLaunchKernelGGL(k_fs_wgrad_reduce_taps, dim3((unsigned)std::min<size_t>((nrow * ntap + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)workspace, dw, nrow, gk, ntap, KT, tap0, kWgSplits, accumulate);
    }
    if (db) hipLaunchKernelGGL(k_fs_wgrad_reduce, dim3((unsigned)((Co + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part_b, db, (size_t)Co,
                               kWgSplits, accumulate);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_bias_grad(const float* dy, float* db, int32_t B, int32_t C, int32_t T, int32_t accumulate, void* stream) {
    if (!dy || !db || B < 1 || C < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_bias_grad: bad argument");
    hipLaunchKernelGGL(k_fs_bias_grad, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, dy, db, B, C, T, fs_ts(T), accumulate);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

static inline dim3 ew_grid(size_t n4) { return dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 16384)); }

extern "C" int dsf_train_add_step(const float* x, const float* step, float* y, int32_t B, int32_t C, int32_t T, void* stream) {
    if (!x || !step || !y || B < 1 || C < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_train_add_step: bad argument");
    const int TS = fs_ts(T);
    const size_t n4 = (size_t)B * C * TS / 4;
    hipLaunchKernelGGL(k_tr_add_step, ew_grid(n4), dim3(256), 0, (hipStream_t)stream, (const float4*)x, step, (float4*)y, C, T, TS, n4);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_train_rowsum(const float* g, float* out, int32_t rows, int32_t T, void* stream) {
    if (!g || !out || rows < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_train_rowsum: bad argument");
    hipLaunchKernelGGL(k_tr_rowsum, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, g, out, T, fs_ts(T));
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_train_gate(const float* a, float* g, int32_t B, int32_t C, int32_t T, void* stream) {
    if (!a || !g || B < 1 || C < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_train_gate: bad argument");
    const int TS = fs_ts(T);
    const size_t n4 = (size_t)B * C * TS / 4;
    hipLaunchKernelGGL(k_tr_gate, ew_grid(n4), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (float4*)g, C, T, TS, n4);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_train_gate_bwd(const float* a, const float* dg, float* da, int32_t B, int32_t C, int32_t T, void* stream) {
    if (!a || !dg || !da || B < 1 || C < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_train_gate_bwd: bad argument");
    const int TS = fs_ts(T);
    const size_t n4 = (size_t)B * C * TS / 4;
    hipLaunchKernelGGL(k_tr_gate_bwd, ew_grid(n4), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (const float4*)dg, (float4*)da, C, T, TS, n4);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_train_res_skip(const float* x, const float* y, const float* skip, float* x_out, float* skip_out, int32_t B, int32_t C, int32_t T,
                                  void* stream) {
    if (!x || !y || !x_out || !skip_out || B < 1 || C < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_train_res_skip: bad argument");
    const int TS = fs_ts(T);
    const size_t n4 = (size_t)B * C * TS / 4;
    hipLaunchKernelGGL(k_tr_res_skip, ew_grid(n4), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const float4*)y, (const float4*)skip,
                       (float4*)x_out, (float4*)skip_out, C, T, TS, n4);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_train_res_skip_bwd(const float* dx_out, const float* dskip_out, float* dx, float* dy, int32_t B, int32_t C, int32_t T, void* stream) {
    if (!dx_out || !dskip_out || !dx || !dy || B < 1 || C < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_train_res_skip_bwd: bad argument");
    const int TS = fs_ts(T);
    const size_t n4 = (size_t)B * C * TS / 4;
    hipLaunchKernelGGL(k_tr_res_skip_bwd, ew_grid(n4), dim3(256), 0, (hipStream_t)stream, (const float4*)dx_out, (const float4*)dskip_out, (float4*)dx,
                       (float4*)dy, C, T, TS, n4);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// ---- PitchExtractor pieces ----------------------------------------------------------------------
extern "C" int dsf_channel_affine(const float* x, const float* a, const float* b, const float* keep, float* y, int32_t B, int32_t C, int32_t T,
                                  void* stream) {
    if (!x || !a || !b || !y || B < 1 || C < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsf_channel_affine: bad argument");
    const int TS = fs_ts(T);
    const size_t n4 = (size_t)B * C * TS / 4;
    hipLaunchKernelGGL(k_fs_affine, ew_grid(n4), dim3(256), 0, (hipStream_t)stream, (const float4*)x, a, b, keep, (float4*)y, C, T, TS, n4);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsf_group_norm(const float* x, const float* gamma, const float* beta, const float* residual, float* y, int32_t B, int32_t C,
                              int32_t groups, int32_t T, float eps, int32_t relu, void* stream) {
    if (!x || !gamma || !beta || !y) return fail(DSD_ERR_INVALID, "dsf_group_norm: null argument");
    if (B < 1 || B > 65535 || C < 1 || groups < 1 || (C % groups) || T < 1)
        return fail(DSD_ERR_INVALID, "dsf_group_norm: bad shape (B=%d C=%d groups=%d T=%d)", B, C, groups, T);
    hipLaunchKernelGGL(k_fs_group_norm, dim3((unsigned)groups, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, residual, y, C, groups,
                       T, fs_ts(T), eps, relu);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// ---- optimiser ---------------------------------------------------------------------------------
extern "C" int dsf_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                              double weight_decay, int64_t step, const float* grad_scale, void* stream) {
    if (!p || !g || !m || !v || n < 1 || step < 1) return fail(DSD_ERR_INVALID, "dsf_adamw_step: bad argument");
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return fail(DSD_ERR_INVALID, "dsf_adamw_step: ranges must be 16-byte aligned");
    if (!(lr >= 0) || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1) || !(eps >= 0))
        return fail(DSD_ERR_INVALID, "dsf_adamw_step: bad hyper-parameters (lr=%g betas=%g,%g eps=%g)", lr, beta1, beta2, eps);
    AdamWParams a{};
    a.p = p; a.g = g; a.m = m; a.v = v; a.gscale = grad_scale; a.n = (size_t)n;
    // the scalars exactly as torch.optim.AdamW forms them in double before handing them to its fp32 kernels
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    a.decay = (float)(1.0 - lr * weight_decay);
    a.one_minus_b1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.one_minus_b2 = (float)(1.0 - beta2);
    a.bc2_sqrt = (float)std::sqrt(bc2);
    a.eps = (float)eps;
    a.neg_step_size = (float)(-(lr / bc1));
    hipLaunchKernelGGL(k_adamw, ew_grid((size_t)(n + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}
