// train_kernels.hpp - gfx950 kernels of the FUSED training step of the DiffNet residual stack (SURVEY.md section 8 row f3:
// GaussianDiffusion.p_losses, usr/diff/shallow_diffusion_tts.py:213-231, through ResidualBlock.forward, usr/diff/net.py:66-78).
//
// Forward: the inference layer kernel itself (layer_body<1, LAST, TRAIN = true>, dsd_kernels.hpp) - it additionally writes
//   y = x + step projection (channel-major) and the gate pre-activation a (fragment order) - behind ONE k_condproj launch for all layers.
// Backward of layer l, three kernels on the tile ownership of the forward (one workgroup = one 32-frame tile, 4 waves):
//   k_trb_gate   dy2 = [dx' / sqrt(2) ; dskip]  ->  dg = Wo^T dy2 (K = 512 on fp32 MFMA)  ->  gate derivative from the saved a
//                ->  da (512 rows) and the gate output g = sigmoid * tanh (operand of the output-projection weight gradient)
//   k_trb_conv   dy = transposed dilated conv of da (ONE K = 3 x 512 contraction; taps = column offsets into the staged da tile)
//                ->  dx = dx' / sqrt(2) + dy, per-tile row sums of dy (gradient of the step projection)
//   k_tr_wgrad   every weight gradient of the layer - dilated conv (3 taps), conditioner projection, output projection - as ONE launch
//                of 128 x 256 output tiles contracted over FRAMES, split-K over frame ranges, partials reduced in a fixed order
//                (deterministic); bias gradients are the row sums of the A operands it stages anyway.
// Activations of the backward pass are channel-major [B][rows][TS] (frames contiguous, zero in [T, TS)) like the operator path of
// train.py; accumulator fragments are written to it directly (a wave store instruction covers two 128-byte row segments).
#pragma once
#include "fs2_kernels.hpp"

namespace dsd {

constexpr float kTrInvSqrt2 = 1.0f / 1.41421354f;      // the constant layer_body multiplies (x + residual) with
constexpr int kTrMaxLayers = 32;

struct TrPtrs { const float* p[kTrMaxLayers]; };        // per-layer device pointers, passed by value

// ------------------------------------------------------------------------------------------------------------
// layout converters around the fused stack
// ------------------------------------------------------------------------------------------------------------
// channel-major [B][256][TS] -> tile-major [B * TS / 32][256][32] (the x layout of layer_body)
__global__ void k_tr_cm_to_tm(const float4* __restrict__ in, float4* __restrict__ out, int TS, size_t n4) {
    const int q = TS / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / q;                           // b * 256 + c
        const int g = (int)(i - row * q);
        const size_t b = row / kC;
        const int c = (int)(row - b * kC);
        const size_t tile = b * (size_t)(TS / 32) + (g >> 3);
        out[(tile * kC + c) * 8 + (g & 7)] = in[i];
    }
}

// running skip sum (fragment order [tile][w4][mb2][q4][lane64], without biases) + sum over layers of the skip-half output biases
// -> channel-major [B][256][TS], zero tail
__global__ __launch_bounds__(kThreads) void k_tr_skip_to_cm(const float4* __restrict__ skip, const float* __restrict__ bsum, float* __restrict__ out,
                                                            int T, int TS, int ntile32) {
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, w = tid >> 6;
    const int tile = blockIdx.x, b = tile / ntile32, t = (tile - b * ntile32) * 32 + j;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = skip[(((size_t)tile * 4 + w) * 2 + ms) * 256 + q * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 64 * w + 32 * ms + 8 * q + 4 * h + e;          // frag_row(4 q + e, h)
                out[((size_t)b * kC + row) * TS + t] = (t < T) ? f4at(v, e) + bsum[row] : 0.f;
            }
        }
}

// bsum[c] = sum_l output_projection_l.bias[C + c] (net.py:77 `residual, skip = chunk(y)`: the skip halves, summed once)
__global__ void k_tr_bsum(const TrPtrs ob, float* __restrict__ bsum, int L) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= kC) return;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += ob.p[l][kC + c];
    bsum[c] = acc;
}

__global__ void k_tr_iota(int* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// dds[b][l][c] = sum over the tiles of utterance b of part[l][tile][c] (fixed order)
__global__ void k_tr_dds_reduce(const float* __restrict__ part, float* __restrict__ dds, int L, int ntile32, int ntiles) {
    const int b = blockIdx.x, l = blockIdx.y, c = threadIdx.x;
    float s = 0.f;
    for (int tn = 0; tn < ntile32; ++tn) s += part[((size_t)l * ntiles + (size_t)b * ntile32 + tn) * kC + c];
    dds[((size_t)b * L + l) * kC + c] = s;
}

// ------------------------------------------------------------------------------------------------------------
// weight packing for all layers in one launch (the weights change every optimiser step)
// ------------------------------------------------------------------------------------------------------------
struct PackMultiParams {
    PackParams pp;              // pp.src unused; pp.dst = layer 0
    TrPtrs src;
    size_t dst_layer_floats;
    int tap_rev;                // 1: read tap (ntap - 1 - tap): the flipped kernel of a transposed convolution
};

__global__ void k_pack_a_multi(const PackMultiParams m) {
    const PackParams& p = m.pp;
    const float* __restrict__ src = m.src.p[blockIdx.y];
    float* __restrict__ dst = p.dst + (size_t)blockIdx.y * m.dst_layer_floats;
    const size_t n = (size_t)p.nw * p.ntap * p.nkc * p.nmb * 256;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int s = idx & 3, lane = (idx >> 2) & 63;
        size_t r = idx >> 8;
        const int mb = r % p.nmb; r /= p.nmb;
        const int kct = r % (p.ntap * p.nkc); r /= (p.ntap * p.nkc);
        const int w = (int)r;
        int tap = kct % p.ntap, kc = kct / p.ntap;
        if (p.centre_first && p.ntap == 3) {
            if (kct < p.nkc) { tap = 1; kc = kct; }
            else { const int i2 = kct - p.nkc; kc = i2 >> 1; tap = (i2 & 1) * 2; }
        }
        const int i = lane & 31, h = lane >> 5;
        int row;
        if (p.split) {
            const int half = p.nmb / 2;
            row = (mb < half) ? (half * 32) * w + 32 * mb + i : p.hi_base + (half * 32) * w + 32 * (mb - half) + i;
        } else {
            row = (p.nmb * 32) * w + 32 * mb + i;
        }
        const int col = 8 * kc + 4 * h + s;
        const int st = m.tap_rev ? p.ntap - 1 - tap : tap;
        float v = 0.f;
        if (row < p.rows_valid && col < p.cols_valid) v = src[(size_t)row * p.row_stride + (size_t)col * p.col_stride + st];
        dst[idx] = v;
    }
}

struct PackBiasMultiParams {
    PackBiasParams pp;          // pp.a / pp.b unused
    TrPtrs a, b;
    size_t dst_layer_floats;
    int has_b;
};

__global__ void k_pack_bias_multi(const PackBiasMultiParams m) {
    const PackBiasParams& p = m.pp;
    const float* __restrict__ a = m.a.p[blockIdx.y];
    const float* __restrict__ bb = m.has_b ? m.b.p[blockIdx.y] : nullptr;
    float* __restrict__ dst = p.dst + (size_t)blockIdx.y * m.dst_layer_floats;
    const int n = p.nw * p.nmb * 2 * 16;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int r = idx & 15, h = (idx >> 4) & 1;
        const int q = idx >> 5;
        const int mb = q % p.nmb, w = q / p.nmb;
        const int i = frag_row(r, h);
        int row;
        if (p.split) {
            const int half = p.nmb / 2;
            row = (mb < half) ? (half * 32) * w + 32 * mb + i : p.hi_base + (half * 32) * w + 32 * (mb - half) + i;
        } else {
            row = (p.nmb * 32) * w + 32 * mb + i;
        }
        float v = 0.f;
        if (row < p.rows_valid) { v = a[row]; if (bb) v += bb[row]; }
        dst[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward, part 1: output projection (data gradient) + gate derivative
// ------------------------------------------------------------------------------------------------------------
struct TrbGateParams {
    const float* dxp;           // gradient wrt this layer's x_out, channel-major [B][256][TS] (not read when LAST: x_out of the last layer is dead)
    const float* dsk;           // gradient wrt the skip sum [B][256][TS] (the same tensor for every layer)
    const float4* a_frag;       // saved gate pre-activation [ntiles][w4][mb4][q4][lane64]
    const float4* wotp;         // output_projection.weight transposed, packed [w4][kc64][mb2][lane64]: row = gate channel, k = output row
    float* da;                  // gradient wrt a: da[b * da_bstride + row * TS + t], rows [0,256) gate, [256,512) filter
    float* g;                   // gate output sigmoid(a_gate) * tanh(a_filter) [B][256][TS]
    long long da_bstride;
    int T, TS, ntile32;
};
constexpr int kTrbGateLdsBytes = 2 * kC * 32 * (int)sizeof(float);

template <bool LAST>
__global__ __launch_bounds__(kThreads, 1) void k_trb_gate(const TrbGateParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // dy2 tile [512][32]: rows [0,256) dx' / sqrt(2), [256,512) dskip
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, b = tile / p.ntile32, t0 = (tile - b * p.ntile32) * 32;
    // saved pre-activation of this wave's gate rows (blocks 0,1) and their filter rows (blocks 2,3): requested first, used last
    float4 av[4][4];
    {
        const float4* al = p.a_frag + ((size_t)tile * 4 + w) * (4 * 4 * 64) + lane;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) av[mb][q] = al[(mb * 4 + q) * 64];
    }
    constexpr int NCH = LAST ? 32 : 64, CH0 = LAST ? 32 : 0;
    const TileB bof{smem + CH0 * 8 * 32 + 4 * h * 32 + j, 8 * 32, NCH};
    GemmPipe<2, 1, 32, 128, 6, TileB> pipe(p.wotp + ((size_t)w * 64 + CH0) * 128, lane, NCH, bof);
    pipe.start_a();
    {
        const int g = tid & 7, t = t0 + 4 * g;
        const bool m0 = t + 0 < p.T, m1 = t + 1 < p.T, m2 = t + 2 < p.T, m3 = t + 3 < p.T;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 32 + (tid >> 3);
            const size_t off = ((size_t)b * kC + row) * p.TS + t;
            if (!LAST) {
                const float4 v = *reinterpret_cast<const float4*>(p.dxp + off);
                *reinterpret_cast<float4*>(smem + row * 32 + 4 * g) =
                    make_float4(m0 ? v.x * kTrInvSqrt2 : 0.f, m1 ? v.y * kTrInvSqrt2 : 0.f, m2 ? v.z * kTrInvSqrt2 : 0.f, m3 ? v.w * kTrInvSqrt2 : 0.f);
            }
            const float4 s = *reinterpret_cast<const float4*>(p.dsk + off);
            *reinterpret_cast<float4*>(smem + (kC + row) * 32 + 4 * g) = make_float4(m0 ? s.x : 0.f, m1 ? s.y : 0.f, m2 ? s.z : 0.f, m3 ? s.w : 0.f);
        }
    }
    __syncthreads();
    f32x16 acc[2][1];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
    pipe.start_b();
    pipe.run(acc, 0, NCH);
    // gate derivative (net.py:73-74): g = s * th, da_gate = dg * th * s (1 - s), da_filter = dg * s * (1 - th^2)
    const int t = t0 + j;
    const bool ok = t < p.T;
    float* dab = p.da + (size_t)b * p.da_bstride + t;
    float* gb = p.g + (size_t)b * kC * p.TS + t;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ag = f4at(av[mb][r >> 2], r & 3), af = f4at(av[mb + 2][r >> 2], r & 3);
            const float sg = 1.f / (1.f + expf(-ag)), th = tanhf(af);
            const float dg = acc[mb][0][r];
            const int row = 64 * w + 32 * mb + frag_row(r, h);
            dab[(size_t)row * p.TS] = ok ? dg * th * (sg * (1.f - sg)) : 0.f;
            dab[(size_t)(kC + row) * p.TS] = ok ? dg * sg * (1.f - th * th) : 0.f;
            gb[(size_t)row * p.TS] = ok ? sg * th : 0.f;
        }
}

// ------------------------------------------------------------------------------------------------------------
// backward, part 2: data gradient of the dilated convolution + residual path + step-projection gradient
// ------------------------------------------------------------------------------------------------------------
// B functor of the transposed conv over a da tile [512][LD] (column kHalo = frame 0 of the tile): chunks [0, 64) = centre tap of channel
// group kc, then (tap 0 -> column offset -dil, tap 2 -> +dil) pairs - k_pack_a's centre-first order with 64 groups.
template <int LD>
struct ConvTB {
    const float* yc; int dil;
    __device__ __forceinline__ const float* operator()(int it, int u) const {
        int kc = 6 * it + u;
        kc = (kc < 192) ? kc : 191;
        if (kc < 64) return yc + kc * (8 * LD);
        const int idx = kc - 64;
        return yc + (idx >> 1) * (8 * LD) + ((idx & 1) ? dil : -dil);
    }
};

struct TrbConvParams {
    const float* da;            // da[b * da_bstride + row * TS + t], 512 rows, zero for t >= T
    const float4* wdtp;         // dilated_conv.weight flipped + transposed, packed [w4][kc192][mb2][lane64]: row = input channel, k = (output row, tap)
    const float* dxp;           // gradient wrt this layer's x_out [B][256][TS] (residual path; not read when LAST)
    float* dx_out;              // gradient wrt this layer's x_in [B][256][TS]
    float* dds_part;            // [ntiles][256] per-tile row sums of dy
    long long da_bstride;
    int T, TS, ntile32, dil;
};
constexpr int kTrbConvLD = 32 + 2 * kHalo;
constexpr int kTrbConvLdsBytes = 2 * kC * kTrbConvLD * (int)sizeof(float);

template <bool LAST>
__global__ __launch_bounds__(kThreads, 1) void k_trb_conv(const TrbConvParams p) {
    constexpr int LD = kTrbConvLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // da tile [512][48]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, b = tile / p.ntile32, t0 = (tile - b * p.ntile32) * 32;
    const ConvTB<LD> bof{smem + 4 * h * LD + kHalo + j, p.dil};
    GemmPipe<2, 1, LD, 128, 6, ConvTB<LD>> pipe(p.wdtp + (size_t)w * (192 * 128), lane, 192, bof);
    pipe.start_a();
    {
        const float* src = p.da + (size_t)b * p.da_bstride;
#pragma unroll 4
        for (int it = 0; it < 24; ++it) {
            const int idx = it * kThreads + tid, row = idx / 12, g = idx - row * 12;
            const int t = t0 - kHalo + 4 * g;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && t < p.TS) v = *reinterpret_cast<const float4*>(src + (size_t)row * p.TS + t);
            *reinterpret_cast<float4*>(smem + row * LD + 4 * g) = v;
        }
    }
    // residual-path gradient at this lane's fragment positions: requested before the contraction
    const int t = t0 + j;
    const bool ok = t < p.T;
    float rv[2][16];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 64 * w + 32 * mb + frag_row(r, h);
            rv[mb][r] = LAST ? 0.f : p.dxp[((size_t)b * kC + row) * p.TS + t];
        }
    __syncthreads();
    f32x16 acc[2][1];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
    pipe.start_b();
    pipe.run(acc, 0, 192);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 64 * w + 32 * mb + frag_row(r, h);
            // frames >= T are zero padding of y in the forward pass (net.py:69-71 pads the conv input): no gradient flows into them
            const float dy = ok ? acc[mb][0][r] : 0.f;
            p.dx_out[((size_t)b * kC + row) * p.TS + t] = ok ? rv[mb][r] * kTrInvSqrt2 + dy : 0.f;
            float s = dy;
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64);
            if (j == 0) p.dds_part[(size_t)tile * kC + row] = s;
        }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradients: dW[m][n] = sum_b sum_t A[m][t] * B[n][t + shift], 128 x 256 output tiles, contraction over frames
// ------------------------------------------------------------------------------------------------------------
// MFMA roles: D[i = m][j = n] += A[i][k] B[k][j] with k = frame.  Both operands are activations with the frame axis contiguous: a
// 32-frame step stages A [128][32] and B [256][32] in LDS with a row stride of 36 floats, so that the fragment of an 8-frame chunk -
// lane (row, h) needs frames 8 c + 4 h + {0,1,2,3} - is ONE aligned ds_read_b128 (36 * 4 B = 9 x 16 B: eight consecutive rows hit eight
// different 16-byte bank groups).  The tap of a dilated convolution is a frame SHIFT applied when B is fetched from global memory
// (dword-aligned 16-byte loads), so the LDS side is the same for every gradient.  Wave (wm, wn) of the 2 x 2 owns 64 x 128 of the tile:
// 2 A + 4 B fragment reads feed 32 MFMAs.  LDS is double-buffered: the global loads of step k + 1 are in flight during the MFMAs of
// step k, one barrier per step.
struct TrWgTile {
    const float* a;             // A rows of this tile: a + b * a_bstride + m * TS + t, m in [0,128)
    const float* bsrc;          // B rows: bsrc + b * b_bstride + n * TS + t, n in [0,256)
    float* out;                 // gradient: out[m * out_rs + n * out_cs] (written by k_tr_wgrad_reduce)
    float* out_bias;            // row sums of A -> out_bias[m], or nullptr
    long long a_bstride, b_bstride;
    int shift, out_rs, out_cs;
    float a_scale;
};
constexpr int kTrWgMaxTiles = 24;
struct TrWgParams {
    TrWgTile tile[kTrWgMaxTiles];
    float* part;                // [ntile_desc][nsplit][128][256]
    float* part_b;              // [ntile_desc][nsplit][128]
    int nsplit, B, T, TS;
};
constexpr int kTrWgLD = 36;
constexpr int kTrWgStage = (128 + 256) * kTrWgLD;
constexpr int kTrWgLdsBytes = 2 * kTrWgStage * (int)sizeof(float);

struct __attribute__((aligned(4))) tr_f4u { float x, y, z, w; };       // a 16-byte load that is only dword-aligned

__global__ __launch_bounds__(kThreads, 1) void k_tr_wgrad(const TrWgParams p) {
    constexpr int LD = kTrWgLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TrWgTile& d = p.tile[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 1, wn = w >> 1;
    const int split = blockIdx.y;
    const int tiles_per_utt = p.TS / 32, ntile = p.B * tiles_per_utt;
    const int per = (ntile + p.nsplit - 1) / p.nsplit;
    const int tile_lo = split * per, tile_hi = min(ntile, tile_lo + per);
    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
    // staging map: thread tid moves float4 column g = tid & 7 of rows (tid >> 3) + 32 q
    const int srow = tid >> 3, sg = tid & 7;
    float4 av[4], bv[8];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    int a_valid = 0;
    const float a_scale = d.a_scale;
    const int shift = d.shift;
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_utt, t0 = (tile - b * tiles_per_utt) * 32;
        const int t = t0 + 4 * sg;
        const float* ap = d.a + (size_t)b * d.a_bstride + (size_t)srow * p.TS + t;
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const float4*>(ap + (size_t)(32 * q) * p.TS);
        a_valid = p.T - t;                                  // frames >= T carry no gradient: masked when the tile is written to LDS
        const int tb = t + shift;
        const float* bp = d.bsrc + (size_t)b * d.b_bstride + (size_t)srow * p.TS + tb;
        if (tb >= 0 && tb + 3 < p.TS) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const tr_f4u v = *reinterpret_cast<const tr_f4u*>(bp + (size_t)(32 * q) * p.TS);
                bv[q] = make_float4(v.x, v.y, v.z, v.w);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float* r = bp + (size_t)(32 * q) * p.TS;
                float4 v;
                v.x = (tb + 0 >= 0 && tb + 0 < p.TS) ? r[0] : 0.f; v.y = (tb + 1 >= 0 && tb + 1 < p.TS) ? r[1] : 0.f;
                v.z = (tb + 2 >= 0 && tb + 2 < p.TS) ? r[2] : 0.f; v.w = (tb + 3 >= 0 && tb + 3 < p.TS) ? r[3] : 0.f;
                bv[q] = v;
            }
        }
        DSD_SB();
    };
    auto stash = [&](int buf) {
        float* As = smem + buf * kTrWgStage;
        float* Bs = As + 128 * LD;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = av[q];
            v.x = (a_valid > 0) ? v.x * a_scale : 0.f; v.y = (a_valid > 1) ? v.y * a_scale : 0.f;
            v.z = (a_valid > 2) ? v.z * a_scale : 0.f; v.w = (a_valid > 3) ? v.w * a_scale : 0.f;
            *reinterpret_cast<float4*>(As + (srow + 32 * q) * LD + 4 * sg) = v;
            bsum[q] += (v.x + v.y) + (v.z + v.w);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(Bs + (srow + 32 * q) * LD + 4 * sg) = bv[q];
    };
    if (tile_lo < tile_hi) {
        fetch(tile_lo);
        stash(0);
    }
    __syncthreads();
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int cur = (tile - tile_lo) & 1;
        const bool more = tile + 1 < tile_hi;
        if (more) fetch(tile + 1);
        const float* ap = smem + cur * kTrWgStage + (64 * wm + i) * LD + 4 * h;
        const float* bp = smem + cur * kTrWgStage + 128 * LD + (128 * wn + i) * LD + 4 * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 af[2], bf[4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) af[mb] = *reinterpret_cast<const float4*>(ap + 32 * mb * LD + 8 * c);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bf[nb] = *reinterpret_cast<const float4*>(bp + 32 * nb * LD + 8 * c);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = mfma32(f4at(af[mb], s), f4at(bf[nb], s), acc[mb][nb]);
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
    }
    float* out = p.part + ((size_t)blockIdx.x * p.nsplit + split) * (128 * 256);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(64 * wm + 32 * mb + frag_row(r, h)) * 256 + 128 * wn + 32 * nb + i] = acc[mb][nb][r];
    if (d.out_bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sb = bsum[q];
            sb += __shfl_xor(sb, 1, 64); sb += __shfl_xor(sb, 2, 64); sb += __shfl_xor(sb, 4, 64);
            if (sg == 0) p.part_b[((size_t)blockIdx.x * p.nsplit + split) * 128 + srow + 32 * q] = sb;
        }
    }
}

// dW = sum over the splits, in split order; one thread per element of a 128 x 256 tile, grid (tile descriptor, 128)
__global__ __launch_bounds__(256) void k_tr_wgrad_reduce(const TrWgParams p) {
    const TrWgTile& d = p.tile[blockIdx.x];
    const int m = blockIdx.y, n = threadIdx.x;
    const float* src = p.part + (size_t)blockIdx.x * p.nsplit * (128 * 256) + m * 256 + n;
    float s = 0.f;
    for (int k = 0; k < p.nsplit; ++k) s += src[(size_t)k * (128 * 256)];
    d.out[(size_t)m * d.out_rs + (size_t)n * d.out_cs] = s;
    if (d.out_bias && n == 0) {
        const float* sb = p.part_b + (size_t)blockIdx.x * p.nsplit * 128 + m;
        float t = 0.f;
        for (int k = 0; k < p.nsplit; ++k) t += sb[(size_t)k * 128];
        d.out_bias[m] = t;
    }
}

}  // namespace dsd
