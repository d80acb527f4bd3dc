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
#include <type_traits>
#include "fs2_kernels.hpp"

namespace dsd {

constexpr float kTrInvSqrt2 = 1.0f / 1.41421354f;      // the constant layer_body multiplies (x + residual) with
constexpr int kTrMaxLayers = 32;

// Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2: XCD x takes the CONTIGUOUS range
// [x q + min(x, r), ...) of a 1-D index space of 8 q + r items, so that neighbours in that space share an L2 (the map of layer_body).
__device__ __forceinline__ int xcd_item(int lin, int q, int r) {
    const int xcd = lin & 7;
    return xcd * q + min(xcd, r) + (lin >> 3);
}

struct TrPtrs { const float* p[2 * kTrMaxLayers]; };           // per-layer device pointers, passed by value

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
#pragma unroll 4
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
#pragma unroll 8
    for (int tn = 0; tn < ntile32; ++tn) s += part[((size_t)l * ntiles + (size_t)b * ntile32 + tn) * kC + c];      // (eight loads in flight, the additions in tile order)
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
// Wave roles of both backward contractions (M = 256 output rows): wave (wr = w & 1, wk = w >> 1) multiplies the 128 rows [128 wr, +128) -
// four row blocks, 16 MFMAs per 4 weight loads + 4 LDS reads, the operand ratio of the layer kernel - over HALF of the K range; the two
// K halves are then summed through LDS (a fixed order: deterministic) and wave (wr, wk) finishes the row blocks 2 wk, 2 wk + 1 of its
// row half, i.e. rows [128 wr + 64 wk, +64) = [64 w', +64) with w' = 2 wr + wk - the rows wave w' of the forward kernel owned.
// xl: this wave's exchange slice in LDS (2 blocks x 16 registers x 64 lanes), xp: the partner's.
__device__ __forceinline__ void trb_exchange(f32x16 (&acc)[4][1], f32x16 (&fin)[2], float* xbuf, int wr, int wk, int lane) {
    float* mine = xbuf + ((wr * 2 + wk) * 32) * 64 + lane;              // what the partner finishes: blocks 2 (1 - wk) + {0, 1}
    const float* theirs = xbuf + ((wr * 2 + (1 - wk)) * 32) * 64 + lane;
#pragma unroll
    for (int mbb = 0; mbb < 2; ++mbb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = wk ? acc[mbb][0][r] : acc[2 + mbb][0][r];
            mine[(mbb * 16 + r) * 64] = v;
        }
    __syncthreads();
#pragma unroll
    for (int mbb = 0; mbb < 2; ++mbb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float own = wk ? acc[2 + mbb][0][r] : acc[mbb][0][r];
            const float oth = theirs[(mbb * 16 + r) * 64];
            fin[mbb][r] = wk ? oth + own : own + oth;                   // always (K half 0) + (K half 1)
        }
}

struct TrbGateParams {
    const float* dxp;           // gradient wrt this layer's x_out, channel-major [B][256][TS] (not read when LAST: x_out of the last layer is dead)
    const float* dsk;           // gradient wrt the skip sum [B][256][TS] (the same tensor for every layer)
    const float4* a_frag;       // saved gate pre-activation [ntiles][w4][mb4][q4][lane64]
    const float4* wotp;         // output_projection.weight transposed, packed [wr2][kc64][mb4][lane64]: row = gate channel, k = output row
    float* da;                  // gradient wrt a: da[b * da_bstride + row * TS + t], rows [0,256) gate, [256,512) filter
    float* g;                   // gate output sigmoid(a_gate) * tanh(a_filter) [B][256][TS]
    long long da_bstride;
    int T, TS, ntile32;
    int xcd_q, xcd_r;           // XCD-aware workgroup -> tile map (ntiles = 8 xcd_q + xcd_r)
};
constexpr int kTrbGateLdsBytes = (2 * kC * 32 + 4 * 32 * 64) * (int)sizeof(float);

template <bool LAST>
__global__ __launch_bounds__(kThreads, 1) void k_trb_gate(const TrbGateParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // dy2 tile [512][32]: rows [0,256) dx' / sqrt(2), [256,512) dskip; exchange
    float* xbuf = smem + 2 * kC * 32;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w & 1, wk = w >> 1;
    const int tile = xcd_item(blockIdx.x, p.xcd_q, p.xcd_r), b = tile / p.ntile32, t0 = (tile - b * p.ntile32) * 32;
    // K = 64 chunks (output rows 8 kc ..): K half 0 = the residual rows, K half 1 = the skip rows; the last layer has no residual half
    // (its x_out is dead) and splits the skip rows
    constexpr int NCH = LAST ? 16 : 32;
    const int ch0 = (LAST ? 32 : 0) + NCH * wk;
    const TileB bof{smem + ch0 * 8 * 32 + 4 * h * 32 + j, 8 * 32, NCH};
    GemmPipe<4, 1, 32, 256, 6, TileB> pipe(p.wotp + ((size_t)wr * 64 + ch0) * 256, lane, NCH, bof);
    const int sg = tid & 7, st = t0 + 4 * sg;
    const bool m0 = st + 0 < p.T, m1 = st + 1 < p.T, m2 = st + 2 < p.T, m3 = st + 3 < p.T;
    // the tile is requested in four quarters - 64 rows of each K half, the rows chunks [8 q, 8 q + 8) of either half read - all up front;
    // quarter q + 1 is written (one barrier) at the 6-chunk boundary in front of its first chunk, so only the first quarter's latency sits
    // in front of the first MFMA (the progressive staging of the forward layer kernel)
    float4 vx[4][2], vs[4][2];
    auto request = [&](int q) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const size_t off = ((size_t)b * kC + 64 * q + 32 * it + (tid >> 3)) * p.TS + st;
            if (!LAST) vx[q][it] = *reinterpret_cast<const float4*>(p.dxp + off);
            vs[q][it] = *reinterpret_cast<const float4*>(p.dsk + off);
        }
        DSD_SB();
    };
    auto write = [&](int q) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = 64 * q + 32 * it + (tid >> 3);
            if (!LAST) {
                const float4 v = vx[q][it];
                *reinterpret_cast<float4*>(smem + row * 32 + 4 * sg) =
                    make_float4(m0 ? v.x * kTrInvSqrt2 : 0.f, m1 ? v.y * kTrInvSqrt2 : 0.f, m2 ? v.z * kTrInvSqrt2 : 0.f, m3 ? v.w * kTrInvSqrt2 : 0.f);
            }
            const float4 s = vs[q][it];
            *reinterpret_cast<float4*>(smem + (kC + row) * 32 + 4 * sg) = make_float4(m0 ? s.x : 0.f, m1 ? s.y : 0.f, m2 ? s.z : 0.f, m3 ? s.w : 0.f);
        }
    };
    request(0);
    pipe.template start_a<0, 2>();
    request(1); request(2); request(3);
    pipe.template start_a<2, 5>();
    // saved pre-activation of the rows this wave finishes (forward wave w' = 2 wr + wk: gate blocks 0,1, filter blocks 2,3): behind the
    // tile in the memory queue, in flight during the contraction
    float4 av[4][4];
    {
        const float4* al = p.a_frag + ((size_t)tile * 4 + 2 * wr + wk) * (4 * 4 * 64) + lane;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) av[mb][q] = al[(mb * 4 + q) * 64];
    }
    DSD_SB();
    f32x16 acc[4][1];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
    if (LAST) {
        write(0); write(1); write(2); write(3);
        __syncthreads();
        pipe.start_b();
        pipe.run(acc, 0, NCH);
    } else {
        write(0);
        __syncthreads();
        pipe.start_b();
        pipe.run(acc, 0, 6);
        write(1);
        __syncthreads();
        pipe.run(acc, 6, 12);
        write(2);
        __syncthreads();
        pipe.run(acc, 12, 18);
        write(3);
        __syncthreads();
        pipe.run(acc, 18, NCH);
    }
    f32x16 fin[2];
    trb_exchange(acc, fin, xbuf, wr, wk, lane);
    // gate derivative (net.py:73-74): g = s * th, da_gate = dg * th * s (1 - s), da_filter = dg * s * (1 - th^2)
    const int t = t0 + j;
    const bool ok = t < p.T;
    float* dab = p.da + (size_t)b * p.da_bstride;            // wave-uniform bases of the write-through stores
    float* gb = p.g + (size_t)b * kC * p.TS;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ag = f4at(av[mb][r >> 2], r & 3), af = f4at(av[mb + 2][r >> 2], r & 3);
            const float sg_ = sigmoid_f(ag), th = tanh_f(af);      // the forward's own functions (dsd_kernels.hpp): the gate the forward multiplied by, at a tenth of libm's instructions
            const float dg = fin[mb][r];
            const int row = 128 * wr + 64 * wk + 32 * mb + frag_row(r, h);
            store4_wt(dab, row * p.TS + t, ok ? dg * th * (sg_ * (1.f - sg_)) : 0.f);
            store4_wt(dab, (kC + row) * p.TS + t, ok ? dg * sg_ * (1.f - th * th) : 0.f);
            store4_wt(gb, row * p.TS + t, ok ? sg_ * th : 0.f);
        }
}

// ------------------------------------------------------------------------------------------------------------
// backward, part 2: data gradient of the dilated convolution + residual path + step-projection gradient
// ------------------------------------------------------------------------------------------------------------
// K order of the transposed conv: K half hk = the 256 da rows [256 hk, +256) (hk = 0 gate rows, 1 filter rows) as 96 chunks in the order of
// the forward conv (ConvB: 32 centre-tap chunks of 8 rows, then the (offset -dil, +dil) pairs) - each half is packed as its own matrix
// [hk][wr2][kc96][mb4][lane64], so a wave walks ITS rows in ascending order and the tile can be staged progressively.
struct TrbConvParams {
    const float* da;            // da[b * da_bstride + row * TS + t], 512 rows, zero for t >= T
    const float4* wdtp;         // dilated_conv.weight flipped + transposed, packed [hk2][wr2][kc96][mb4][lane64]: row = input channel, k = (da row, tap)
    const float* dxp;           // gradient wrt this layer's x_out [B][256][TS] (residual path; not read when LAST)
    float* dx_out;              // gradient wrt this layer's x_in [B][256][TS]
    float* dds_part;            // [ntiles][256] per-tile row sums of dy
    long long da_bstride;
    int T, TS, ntile32, dil;
    int xcd_q, xcd_r;           // XCD-aware workgroup -> tile map: a tile's halo columns come from tiles behind the same L2
};
constexpr int kTrbConvLD = 32 + 2 * kHalo;
constexpr int kTrbConvLdsBytes = (2 * kC * kTrbConvLD + 4 * 32 * 64) * (int)sizeof(float);

template <bool LAST>
__global__ __launch_bounds__(kThreads, 1) void k_trb_conv(const TrbConvParams p) {
    constexpr int LD = kTrbConvLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // da tile [512][48]; exchange
    float* xbuf = smem + 2 * kC * LD;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w & 1, wk = w >> 1;
    const int tile = xcd_item(blockIdx.x, p.xcd_q, p.xcd_r), b = tile / p.ntile32, t0 = (tile - b * p.ntile32) * 32;
    const ConvB<LD> bof{smem + (wk * kC + 4 * h) * LD + kHalo + j, p.dil, 0};
    GemmPipe<4, 1, LD, 256, 6, ConvB<LD>> pipe(p.wdtp + ((size_t)(wk * 2 + wr) * 96) * 256, lane, 96, bof);
    // progressive staging (see k_trb_gate): quarter q = rows [64 q, +64) of either K half with their halo columns, 6 float4 per thread
    const float* src = p.da + (size_t)b * p.da_bstride;
    float4 sv[4][6];
    auto request = [&](int q) {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int idx = it * kThreads + tid, rq = idx / 12, g = idx - rq * 12;
            const int row = (rq < 64) ? 64 * q + rq : kC + 64 * q + (rq - 64);
            const int t = t0 - kHalo + 4 * g;
            const bool in = (t >= 0) && (t < p.TS);
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)row * p.TS + (in ? t : t0));
            sv[q][it] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        DSD_SB();
    };
    auto write = [&](int q) {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int idx = it * kThreads + tid, rq = idx / 12, g = idx - rq * 12;
            const int row = (rq < 64) ? 64 * q + rq : kC + 64 * q + (rq - 64);
            *reinterpret_cast<float4*>(smem + row * LD + 4 * g) = sv[q][it];
        }
    };
    request(0);
    pipe.template start_a<0, 2>();
    request(1); request(2); request(3);
    pipe.template start_a<2, 5>();
    // residual-path gradient at the fragment positions this wave finishes: behind the tile in the memory queue
    const int t = t0 + j;
    const bool ok = t < p.T;
    float rv[2][16];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 128 * wr + 64 * wk + 32 * mb + frag_row(r, h);
            rv[mb][r] = LAST ? 0.f : p.dxp[((size_t)b * kC + row) * p.TS + t];
        }
    DSD_SB();
    f32x16 acc[4][1];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
    write(0);
    __syncthreads();
    pipe.start_b();
    pipe.run(acc, 0, 6);
    write(1);
    __syncthreads();
    pipe.run(acc, 6, 12);
    write(2);
    __syncthreads();
    pipe.run(acc, 12, 18);
    write(3);
    __syncthreads();
    pipe.run(acc, 18, 96);
    f32x16 fin[2];
    trb_exchange(acc, fin, xbuf, wr, wk, lane);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 128 * wr + 64 * wk + 32 * mb + frag_row(r, h);
            // frames >= T are zero padding of y in the forward pass (net.py:69-71 pads the conv input): no gradient flows into them
            const float dy = ok ? fin[mb][r] : 0.f;
            store4_wt(uniform_ptr(p.dx_out + (size_t)b * kC * p.TS), row * p.TS + t, ok ? rv[mb][r] * kTrInvSqrt2 + dy : 0.f);      // (uniform_ptr: hipcc had this base in VGPRs - 32 waterfall loops)
            float s = dy;
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64);
            if (j == 0) p.dds_part[(size_t)tile * kC + row] = s;
        }
}

// ------------------------------------------------------------------------------------------------------------
// backward, parts 2 + 1 fused: k_trb_conv of layer l, then k_trb_gate of layer l - 1 on the same tile
// ------------------------------------------------------------------------------------------------------------
// The gate kernel of layer l - 1 reads exactly the dx tile the conv kernel of layer l just produced (the output projection is a 1 x 1
// convolution: no halo) - so it runs behind it in the SAME workgroup: dx goes from the accumulators into the dy2 tile in LDS (and to global
// memory for the weight gradient and the next residual path), one kernel node and one round trip through memory less per layer.  The
// kernel boundary stays where a tile needs its neighbours: in front of the transposed conv (halo columns of da).  Arithmetic and summation
// orders are those of the two kernels: results are bit-identical to them.
struct TrbFusedParams {
    TrbConvParams c;            // layer l
    TrbGateParams g;            // layer l - 1 (g.dxp is not read: the gradient comes from this workgroup's accumulators)
};
constexpr int kTrbFusedLdsBytes = kTrbConvLdsBytes;

template <bool LAST>            // LAST: layer l is the last layer (no residual path into its conv gradient)
__global__ __launch_bounds__(kThreads, 1) void k_trb_fused(const TrbFusedParams q) {
    constexpr int LD = kTrbConvLD;
    const TrbConvParams& p = q.c;
    const TrbGateParams& pg = q.g;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // da tile [512][48], then dy2 tile [512][32]; exchange
    float* xbuf = smem + 2 * kC * LD;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w & 1, wk = w >> 1;
    const int tile = xcd_item(blockIdx.x, p.xcd_q, p.xcd_r), b = tile / p.ntile32, t0 = (tile - b * p.ntile32) * 32;
    const ConvB<LD> bof{smem + (wk * kC + 4 * h) * LD + kHalo + j, p.dil, 0};
    GemmPipe<4, 1, LD, 256, 6, ConvB<LD>> pipe(p.wdtp + ((size_t)(wk * 2 + wr) * 96) * 256, lane, 96, bof);
    const float* src = p.da + (size_t)b * p.da_bstride;
    float4 sv[4][6];
    auto request = [&](int qq) {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int idx = it * kThreads + tid, rq = idx / 12, g = idx - rq * 12;
            const int row = (rq < 64) ? 64 * qq + rq : kC + 64 * qq + (rq - 64);
            const int t = t0 - kHalo + 4 * g;
            const bool in = (t >= 0) && (t < p.TS);
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)row * p.TS + (in ? t : t0));
            sv[qq][it] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        DSD_SB();
    };
    auto write = [&](int qq) {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int idx = it * kThreads + tid, rq = idx / 12, g = idx - rq * 12;
            const int row = (rq < 64) ? 64 * qq + rq : kC + 64 * qq + (rq - 64);
            *reinterpret_cast<float4*>(smem + row * LD + 4 * g) = sv[qq][it];
        }
    };
    request(0);
    pipe.template start_a<0, 2>();
    request(1); request(2); request(3);
    pipe.template start_a<2, 5>();
    const int t = t0 + j;
    const bool ok = t < p.T;
    float rv[2][16];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 128 * wr + 64 * wk + 32 * mb + frag_row(r, h);
            rv[mb][r] = LAST ? 0.f : p.dxp[((size_t)b * kC + row) * p.TS + t];
        }
    DSD_SB();
    f32x16 acc[4][1];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
    write(0);
    __syncthreads();
    pipe.start_b();
    pipe.run(acc, 0, 6);
    write(1);
    __syncthreads();
    pipe.run(acc, 6, 12);
    write(2);
    __syncthreads();
    pipe.run(acc, 12, 18);
    write(3);
    __syncthreads();
    // the skip rows of the next contraction's B tile and its first weight chunks are requested half way through this one
    pipe.run(acc, 18, 48);
    constexpr int NCH = 32;
    const int ch0 = NCH * wk;
    const TileB bofg{smem + ch0 * 8 * 32 + 4 * h * 32 + j, 8 * 32, NCH};
    GemmPipe<4, 1, 32, 256, 6, TileB> pipeg(pg.wotp + ((size_t)wr * 64 + ch0) * 256, lane, NCH, bofg);
    const int sg = tid & 7, st = t0 + 4 * sg;
    float4 vs[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) vs[it] = *reinterpret_cast<const float4*>(pg.dsk + ((size_t)b * kC + it * 32 + (tid >> 3)) * p.TS + st);
    DSD_SB();
    pipe.run(acc, 48, 96);
    f32x16 fin[2];
    trb_exchange(acc, fin, xbuf, wr, wk, lane);          // its barrier: every wave is done reading the da tile
    pipeg.start_a();
    // saved pre-activation of layer l - 1 at the rows this wave finishes there
    float4 av[4][4];
    {
        const float4* al = pg.a_frag + ((size_t)tile * 4 + 2 * wr + wk) * (4 * 4 * 64) + lane;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) av[mb][qq] = al[(mb * 4 + qq) * 64];
    }
    DSD_SB();
    // conv gradient epilogue (k_trb_conv) + the residual rows of the dy2 tile
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 128 * wr + 64 * wk + 32 * mb + frag_row(r, h);
            const float dy = ok ? fin[mb][r] : 0.f;
            const float dx = ok ? rv[mb][r] * kTrInvSqrt2 + dy : 0.f;
            store4_wt(p.dx_out + (size_t)b * kC * p.TS, row * p.TS + t, dx);
            smem[row * 32 + j] = dx * kTrInvSqrt2;
            float s = dy;
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64);
            if (j == 0) p.dds_part[(size_t)tile * kC + row] = s;
        }
    {
        const bool m0 = st + 0 < p.T, m1 = st + 1 < p.T, m2 = st + 2 < p.T, m3 = st + 3 < p.T;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 32 + (tid >> 3);
            const float4 s4 = vs[it];
            *reinterpret_cast<float4*>(smem + (kC + row) * 32 + 4 * sg) = make_float4(m0 ? s4.x : 0.f, m1 ? s4.y : 0.f, m2 ? s4.z : 0.f, m3 ? s4.w : 0.f);
        }
    }
    __syncthreads();
    // output-projection data gradient + gate derivative of layer l - 1 (k_trb_gate<false>)
    f32x16 accg[4][1];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accg[mb][0][r] = 0.f;
    pipeg.start_b();
    pipeg.run(accg, 0, NCH);
    f32x16 fing[2];
    trb_exchange(accg, fing, xbuf, wr, wk, lane);
    float* dab = pg.da + (size_t)b * pg.da_bstride;
    float* gb = pg.g + (size_t)b * kC * p.TS;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ag = f4at(av[mb][r >> 2], r & 3), af = f4at(av[mb + 2][r >> 2], r & 3);
            const float sg_ = sigmoid_f(ag), th = tanh_f(af);      // the forward's own functions (dsd_kernels.hpp): the gate the forward multiplied by, at a tenth of libm's instructions
            const float dg = fing[mb][r];
            const int row = 128 * wr + 64 * wk + 32 * mb + frag_row(r, h);
            store4_wt(dab, row * p.TS + t, ok ? dg * th * (sg_ * (1.f - sg_)) : 0.f);
            store4_wt(dab, (kC + row) * p.TS + t, ok ? dg * sg_ * (1.f - th * th) : 0.f);
            store4_wt(gb, row * p.TS + t, ok ? sg_ * th : 0.f);
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
    const float* bsrc;          // B rows: bsrc + b * b_bstride + n * b_rs + t, n in [0,256)
    float* out;                 // gradient: out[m * out_rs + n * out_cs] (written by k_tr_wgrad_reduce)
    float* out_bias;            // row sums of A -> out_bias[m], or nullptr
    long long a_bstride, b_bstride;
    int shift, out_rs, out_cs, b_rs;
    float a_scale;
    int prod;                   // -1: a plain tile (frames contracted as they are); 0..3: product P_prod of the Winograd dual of a 3-tap dilated
                                // convolution's weight gradient (below), contracted over frame PAIRS; shift = the dilation d in {1, 2, 4, 8}
};
// The Winograd F(2,3) DUAL for the convolution's weight gradient (round 6).  dW_k[m][n] = sum_t a[m][t] y[n][t + (k - 1) d], k = 0, 1, 2: three
// products over the frames.  Over a frame PAIR (tE, tO = tE + d) with e = a[tE], f = a[tO], d0..d3 = y[tE - d], y[tE], y[tO], y[tO + d]:
//     e d0 + f d1 = Q0 + Q1 + Q2      e d1 + f d2 = Q1 - Q2      e d2 + f d3 = Q1 + Q2 + Q3
//     Q0 = e (d0 - d2)    Q1 = (e + f) (d1 + d2) / 2    Q2 = (e - f) (d2 - d1) / 2    Q3 = f (d3 - d1)
// FOUR products over the pairs (half as many as frames) instead of three over the frames: 2/3 of the multiplications of 12 of a layer's 20
// weight-gradient tiles.  A workgroup computes ONE product of one 128-row tile: it stages 64 frames per step, forms its operand pair
//     P0: E . (y[tE - d] - y[tO])    P1: (E + O) . (y[tE] + y[tO])    P2: (E - O) . (y[tO] - y[tE])    P3: O . (y[tE] - y[tO + d])
// in registers on the way into LDS (32 pairs per step: the LDS tiles, the fragment reads and the MFMA loop are those of a plain tile), and
// k_tr_wgrad_reduce_dual back-transforms  dW_0 = P0 + (P1 + P2) / 2,  dW_1 = (P1 - P2) / 2,  dW_2 = (P1 + P2) / 2 - P3.  Pairs: a 64-frame step
// splits into blocks of 2 d frames, pair p = blk d + i <-> tE = 2 d blk + i (the forward's pair order, csrc/dsd_loop_wino.hpp); a thread forms
// the pairs 4 q .. 4 q + 3 (q = tid & 7) of its rows from two or three ALIGNED 16-byte loads (d = 1, 2: the frames 8 q - 4 .. 8 q + 11, the
// E / O split is a register permutation; d = 4, 8: E, O and their +- d neighbours are whole float4).  The bias gradient (row sums of a) is
// the row sum of P1's operand E + O.
constexpr int kTrWgMaxTiles = 48;        // two layers per launch: 2 x (16 dual products + 8 plain tiles)
struct TrWgParams {
    TrWgTile tile[kTrWgMaxTiles];
    float* part;                // split-K partials [partial tile][128][256]: descriptor n < nc: n * ns_c + split; else nc * ns_c + (n - nc) * nsplit + split
    float* part_b;              // the same indices, [128] each
    int nsplit, B, T, TS;       // nsplit: frame splits of the PLAIN tiles
    int ndesc, xcd_q, xcd_r;    // 1-D grid of nc * ns_c + (ndesc - nc) * nsplit workgroups, XCD x takes a contiguous range of (split, tile) pairs: the workgroups
                                // behind one L2 contract (nearly) the same frames, so every operand line is fetched from the fabric by ~1.3 L2s, not 8
    int nc, ns_c;               // the first nc descriptors are dual products, ns_c splits each (a pair step covers 64 frames: half the steps of a
                                // plain tile, so ns_c = nsplit / 2 balances the workgroups)
};
__host__ __device__ __forceinline__ int tr_wg_part_index(const TrWgParams& p, int desc, int split) {
    return desc < p.nc ? desc * p.ns_c + split : p.nc * p.ns_c + (desc - p.nc) * p.nsplit + split;
}
constexpr int kTrWgLD = 36;
constexpr int kTrWgStage = (128 + 256) * kTrWgLD;
constexpr int kTrWgLdsBytes = 2 * kTrWgStage * (int)sizeof(float);
constexpr int kTrYPad = 8;      // zero floats on both sides of every row of the saved y (>= the largest tap shift): a shifted 16-byte load stays in its row

struct __attribute__((aligned(4))) tr_f4u { float x, y, z, w; };       // a 16-byte load that is only dword-aligned

// {1 MFMA, then up to NV VALU, 1 LDS write, 1 global load, 1 LDS read} x n: the non-MFMA work of a step is issued in the shadow of the MFMAs
// (one wave per SIMD: nothing else would fill the pipe while this wave computes addresses or writes the next tile)
template <int NV, bool DSW, bool VMEM, bool DSR>
__device__ __forceinline__ void tr_interleave(int) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (DSR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        if (DSW) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (VMEM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
}

// FIX: the B rows are NOT padded (the stand-alone operator on a caller's tensor): the shifted load is clamped into the row and the first /
// last tile of an utterance patches the elements when it writes the tile (a branch in the step).  The fused stack pads its y rows (kTrYPad)
// and runs the branch-free form.
template <bool FIX>
__global__ __launch_bounds__(kThreads, 1) void k_tr_wgrad(const TrWgParams p) {
    constexpr int LD = kTrWgLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int item = xcd_item(blockIdx.x, p.xcd_q, p.xcd_r);
    const int n_dual = p.nc * p.ns_c;
    const bool dual = !FIX && item < n_dual;                // (wave-uniform) one product of a convolution's Winograd-dual weight gradient
    const int jt = dual ? item : item - n_dual, nd_cls = dual ? p.nc : p.ndesc - p.nc, ns_cls = dual ? p.ns_c : p.nsplit;
    const int split = jt / nd_cls, desc = (dual ? 0 : p.nc) + jt - split * nd_cls;
    const TrWgTile& d = p.tile[desc];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 1, wn = w >> 1;
    const int tiles_per_utt = dual ? (p.TS + 63) / 64 : p.TS / 32, ntile = p.B * tiles_per_utt;     // steps: 64 frames = 32 pairs / 32 frames
    const int per = (ntile + ns_cls - 1) / ns_cls;
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
    int a_valid = 0, b_shift = 0;
    const float a_scale = d.a_scale;
    const int shift = d.shift, b_rs = d.b_rs;
    // The fused stack's form (!FIX) keeps the step free of vector-ALU work: beside an fp32 MFMA every vector instruction costs ~8 cycles of matrix
    // time (tools/mfma_filler_probe.hip; the round-4 step carried 94 of them per 128 MFMAs).  Addresses: one buffer descriptor per operand whose
    // base walks with the tile (scalar), the thread's offset is loop-invariant, the row groups are scalar offsets.  The scale of A multiplies the
    // accumulators once at the end; frames >= T are masked only in an utterance's partial tile (a scalar branch).
    const unsigned a_vo = (unsigned)(srow * p.TS + 4 * sg) * 4u, b_vo = (unsigned)(srow * b_rs + 4 * sg) * 4u;
    int mask_t0 = 0;
    int f_tile = tile_lo, f_b = tile_lo / tiles_per_utt, f_tn = tile_lo - f_b * tiles_per_utt;      // the next tile to fetch (scalar counters: no division in the step)
    auto fetch_s = [&]() {
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const int b = __builtin_amdgcn_readfirstlane(f_b), t0 = __builtin_amdgcn_readfirstlane(f_tn * 32);
        {                                                   // behind the last tile the staging registers repeat it (selects, no branch: the step stays ONE block)
            const int adv = (f_tile + 1 < tile_hi) ? 1 : 0;
            f_tile += adv;
            const int tn1 = f_tn + adv, wrap = (tn1 == tiles_per_utt) ? 1 : 0;
            f_tn = wrap ? 0 : tn1;
            f_b += wrap;
        }
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.a + (size_t)b * d.a_bstride + t0), 0, 0x7ffffff0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bsrc + (size_t)b * d.b_bstride + (t0 + shift)), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4_ f = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)a_vo, q * (32 * 4) * p.TS, 0));
            av[q] = make_float4(f.x, f.y, f.z, f.w);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4_ f = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)b_vo, q * (32 * 4) * b_rs, 0));
            bv[q] = make_float4(f.x, f.y, f.z, f.w);
        }
        mask_t0 = t0;
    };
    const bool has_bias = d.out_bias != nullptr;
    // (a scalar branch at the TOP of a step: the utterance's partial tile, or a tile of padding - frames >= T carry no gradient)
    auto mask_s = [&](int t0) {
        if (t0 + 32 > p.T) {
            const int valid = p.T - (t0 + 4 * sg);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = av[q];
                v.x = (valid > 0) ? v.x : 0.f; v.y = (valid > 1) ? v.y : 0.f; v.z = (valid > 2) ? v.z : 0.f; v.w = (valid > 3) ? v.w : 0.f;
                av[q] = v;
            }
        }
    };
    auto stash_s = [&](int buf, float live, bool bias) {
        float* As = smem + buf * kTrWgStage;
        float* Bs = As + 128 * LD;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(As + (srow + 32 * q) * LD + 4 * sg) = av[q];
        if (bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bsum[q] += live * ((av[q].x + av[q].y) + (av[q].z + av[q].w));
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(Bs + (srow + 32 * q) * LD + 4 * sg) = bv[q];
    };
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_utt, t0 = (tile - b * tiles_per_utt) * 32;
        const int t = t0 + 4 * sg;
        const float* ap = d.a + (size_t)b * d.a_bstride + (size_t)srow * p.TS + t;
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const float4*>(ap + (size_t)(32 * q) * p.TS);
        a_valid = p.T - t;                                  // frames >= T carry no gradient: masked when the tile is written to LDS
        // B: frames t + shift .. + 3 of 256 rows, one dword-aligned 16-byte load per row
        const int tb = t + shift;
        const int tbc = FIX ? min(max(tb, 0), p.TS - 4) : tb;
        b_shift = tb - tbc;
        const float* bp = d.bsrc + (size_t)b * d.b_bstride + (size_t)srow * b_rs + tbc;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const tr_f4u v = *reinterpret_cast<const tr_f4u*>(bp + (size_t)(32 * q) * b_rs);
            bv[q] = make_float4(v.x, v.y, v.z, v.w);
        }
    };
    auto stash = [&](int buf, float live) {
        float* As = smem + buf * kTrWgStage;
        float* Bs = As + 128 * LD;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = av[q];
            v.x = (a_valid > 0) ? v.x * a_scale : 0.f; v.y = (a_valid > 1) ? v.y * a_scale : 0.f;
            v.z = (a_valid > 2) ? v.z * a_scale : 0.f; v.w = (a_valid > 3) ? v.w * a_scale : 0.f;
            *reinterpret_cast<float4*>(As + (srow + 32 * q) * LD + 4 * sg) = v;
            bsum[q] += live * ((v.x + v.y) + (v.z + v.w));
        }
        if (FIX && b_shift != 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = bv[q];
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = e + b_shift;            // element of the loaded window that holds frame t + shift + e, if any
                    o[e] = (idx == 0) ? v.x : (idx == 1) ? v.y : (idx == 2) ? v.z : (idx == 3) ? v.w : 0.f;
                }
                bv[q] = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(Bs + (srow + 32 * q) * LD + 4 * sg) = bv[q];
    };
    // Software pipeline over the 32-frame steps (one wave per SIMD: nothing else hides a bubble).  At the top of step k LDS buffer k & 1
    // holds tile k, the staging registers hold tile k + 1 (requested a whole step ago) and the fragments of chunk 0 are already in
    // registers.  Step k: chunk 1's fragments are requested, tile k + 1 goes to the other buffer (free since the barrier of step k - 1) and
    // tile k + 2 is requested - all of it issued BETWEEN the MFMAs of chunk 0 (tr_interleave); chunks 1, 2 run with the next chunk's
    // fragments in flight; behind chunk 3's reads comes the ONE barrier of the step (every wave is done reading this buffer, every wave's
    // writes of the other one have landed), the fragments of the next step's chunk 0 are requested, and only then chunk 3's 32 MFMAs are
    // issued - they cover barrier skew and LDS latency.  The step is branch-free: behind the last tile the staging registers repeat it
    // (`live` = 0 keeps it out of the bias sums; the buffer it lands in is never multiplied).
    const int nstep = tile_hi - tile_lo;
    float4 fa[2][2], fb[2][4];
    auto frags = [&](int set, int buf, int c) {
        const float* ap = smem + buf * kTrWgStage + (64 * wm + i) * LD + 4 * h + 8 * c;
        const float* bp = smem + buf * kTrWgStage + 128 * LD + (128 * wn + i) * LD + 4 * h + 8 * c;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) fa[set][mb] = *reinterpret_cast<const float4*>(ap + 32 * mb * LD);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) fb[set][nb] = *reinterpret_cast<const float4*>(bp + 32 * nb * LD);
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = mfma32(f4at(fa[set][mb], s), f4at(fb[set][nb], s), acc[mb][nb]);
    };
    // the step loop; BIAS (a compile-time constant of the instance that runs: the row sums are only kept for tiles with a bias gradient)
    auto loop = [&](auto bias_c) {
        constexpr bool BIAS = decltype(bias_c)::value;
        if constexpr (FIX) {
            fetch(tile_lo);
            stash(0, 1.f);
            fetch(min(tile_lo + 1, tile_hi - 1));
        } else {
            fetch_s();
            mask_s(mask_t0);
            stash_s(0, 1.f, BIAS);
            fetch_s();
        }
        __syncthreads();
        frags(0, 0, 0);
        for (int k = 0; k < nstep; ++k) {
            const int cur = k & 1;
            const float live = (k + 1 < nstep) ? 1.f : 0.f;
            if constexpr (!FIX) mask_s(mask_t0);
            frags(1, cur, 1);
            if constexpr (FIX) {
                stash(cur ^ 1, live);
                fetch(min(tile_lo + k + 2, tile_hi - 1));
            } else {
                stash_s(cur ^ 1, live, BIAS);
                fetch_s();
            }
            mma(0);
            if (!FIX) tr_interleave<BIAS ? 1 : 0, true, true, true>(0);
            DSD_SB();
            frags(0, cur, 2);
            mma(1);
            tr_interleave<0, false, false, true>(0);
            DSD_SB();
            frags(1, cur, 3);
            mma(0);
            tr_interleave<0, false, false, true>(0);
            DSD_SB();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            frags(0, cur ^ 1, 0);
            mma(1);
            tr_interleave<0, false, false, true>(0);
            DSD_SB();
        }
    };
    // ---- the Winograd-dual step: 64 frames -> 32 pairs per step, operands formed in registers on the way into LDS ------------------------------
    // DS: 1, 2 (E / O interleaved inside 8 frames: a register permutation) or 4 (d = 4, 8: E, O and their neighbours are whole float4)
    float4 pa[4][2], pb[8][3];
    auto loop_dual = [&](auto ds_c, auto pr_c, auto bias_c) {
        constexpr int DS = decltype(ds_c)::value, PR = decltype(pr_c)::value;
        constexpr bool BIAS = decltype(bias_c)::value;
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const int dd = d.shift;                             // the dilation
        int e0;                                             // frame (inside the step) of the E float4 of this thread's pairs 4 sg .. 4 sg + 3
        if constexpr (DS >= 4) { const int pq = 4 * sg, blk = pq / dd; e0 = 2 * dd * blk + (pq - blk * dd); } else e0 = 8 * sg;
        const int o1 = (DS >= 4) ? e0 + dd : e0 + 4;        // second float4: O (d >= 4) / frames 8 q + 4 .. + 7
        // third float4 of the B operand: P0 reaches back (y[tE - d]), P3 reaches forward (y[tO + d])
        // (d = 1, 2: only the last / first one or two frames of the neighbouring float4 are needed - a 4- or 8-byte load, kept in .x / .x, .y)
        const int o2 = (PR == 0) ? ((DS >= 4) ? e0 - dd : e0 - DS) : ((DS >= 4) ? e0 + 2 * dd : e0 + 8);
        constexpr bool B3 = (PR == 0 || PR == 3);
        const unsigned avo0 = (unsigned)(srow * p.TS + e0) * 4u, avo1 = (unsigned)(srow * p.TS + o1) * 4u;
        // (the B descriptor's base sits 16 floats in front of the step: o2 reaches back to frame -8, and a NEGATIVE lane offset is not an address
        // 8 frames earlier but an out-of-range offset of a raw buffer - the load returns 0.  Found by tools/diag_wgrad_dual.py: tap 0 wrong in every
        // step but an utterance's first, where the zeros were the padding anyway)
        constexpr int kBack = 16;
        const unsigned bvo0 = (unsigned)(srow * b_rs + e0 + kBack) * 4u, bvo1 = (unsigned)(srow * b_rs + o1 + kBack) * 4u, bvo2 = (unsigned)(srow * b_rs + o2 + kBack) * 4u;
        auto ld = [&](const __amdgpu_buffer_rsrc_t& r, unsigned vo, int so) -> float4 {
            const f32x4_ f = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, so, 0));
            return make_float4(f.x, f.y, f.z, f.w);
        };
        // The step's staging is spread over its first three MFMA groups (32 MFMAs each) so that every group carries a share the interleave pattern
        // can place between its MFMAs: group 0 - A (4 row slots: form, write, request the next step's), group 1 - B row slots 0-3, group 2 - B row
        // slots 4-7; the barrier sits in front of group 3 as in the plain step.  (The first version formed and wrote everything in front of group 0:
        // ~120 vector instructions and 12 LDS writes in a row with the matrix pipe idle - 185 us per launch, no faster than the three tap tiles.)
        __amdgpu_buffer_rsrc_t ra, rb;
        unsigned a0 = 0, a1 = 0, b0 = 0, b1 = 0, b2 = 0;
        auto next_step = [&]() {
            const int b = __builtin_amdgcn_readfirstlane(f_b), t0 = __builtin_amdgcn_readfirstlane(f_tn * 64);
            {
                const int adv = (f_tile + 1 < tile_hi) ? 1 : 0;
                f_tile += adv;
                const int tn1 = f_tn + adv, wrap = (tn1 == tiles_per_utt) ? 1 : 0;
                f_tn = wrap ? 0 : tn1;
                f_b += wrap;
            }
            ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.a + (size_t)b * d.a_bstride + t0), 0, 0x7ffffff0, 0x00020000);
            rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bsrc + (size_t)b * d.b_bstride + t0 - kBack), 0, 0x7ffffff0, 0x00020000);
            // the second half of an utterance's last step may lie behind its rows (TS = 32 mod 64): those threads ask for an offset beyond the
            // descriptor's range - a raw buffer load answers 0 there: zero operands, no select in the step
            const bool dead = (t0 + 32 >= p.TS) && (sg >= 4);
            const unsigned oob = 0xfffffff0u;
            a0 = dead ? oob : avo0; a1 = dead ? oob : avo1;
            b0 = dead ? oob : bvo0; b1 = dead ? oob : bvo1; b2 = dead ? oob : bvo2;
        };
        auto fetch_a = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (PR != 3 || DS < 4) pa[q][0] = ld(ra, a0, q * (32 * 4) * p.TS);
                if (PR != 0 || DS < 4) pa[q][1] = ld(ra, a1, q * (32 * 4) * p.TS);
            }
        };
        auto fetch_b = [&](int q0) {
#pragma unroll
            for (int q = q0; q < q0 + 4; ++q) {
                if (!(PR == 0 && DS >= 4)) pb[q][0] = ld(rb, b0, q * (32 * 4) * b_rs);
                if (!(PR == 3 && DS >= 4)) pb[q][1] = ld(rb, b1, q * (32 * 4) * b_rs);
                if (B3) {
                    if constexpr (DS >= 4) pb[q][2] = ld(rb, b2, q * (32 * 4) * b_rs);
                    else if constexpr (DS == 2) {
                        typedef float f32x2_ __attribute__((ext_vector_type(2)));
                        const f32x2_ f = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(rb, (int)b2, q * (32 * 4) * b_rs, 0));
                        pb[q][2].x = f.x; pb[q][2].y = f.y;
                    } else pb[q][2].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (int)b2, q * (32 * 4) * b_rs, 0));
                }
            }
        };
        // E / O of four pairs from the two float4 (frames 8 q .. 8 q + 7; d >= 4: they ARE E and O)
        auto eo = [&](const float4& l0, const float4& l1, float4& e, float4& o) {
            if constexpr (DS == 1) { e = make_float4(l0.x, l0.z, l1.x, l1.z); o = make_float4(l0.y, l0.w, l1.y, l1.w); }
            else if constexpr (DS == 2) { e = make_float4(l0.x, l0.y, l1.x, l1.y); o = make_float4(l0.z, l0.w, l1.z, l1.w); }
            else { e = l0; o = l1; }
        };
        auto f4sub = [](const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
        auto f4add = [](const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
        auto stash_a = [&](int buf, float live) {
            float* As = smem + buf * kTrWgStage;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 e, o, v;
                if constexpr (DS >= 4) { e = pa[q][0]; o = pa[q][1]; } else eo(pa[q][0], pa[q][1], e, o);
                v = (PR == 0) ? e : (PR == 1) ? f4add(e, o) : (PR == 2) ? f4sub(e, o) : o;
                *reinterpret_cast<float4*>(As + (srow + 32 * q) * LD + 4 * sg) = v;
                if (BIAS) bsum[q] += live * ((v.x + v.y) + (v.z + v.w));
            }
        };
        auto stash_b = [&](int buf, int q0) {
            float* Bs = smem + buf * kTrWgStage + 128 * LD;
#pragma unroll
            for (int q = q0; q < q0 + 4; ++q) {
                float4 v;
                if constexpr (DS >= 4) {
                    // pb[.][0] = y[tE], [1] = y[tO], [2] = y[tE - d] (P0) / y[tO + d] (P3)
                    v = (PR == 0) ? f4sub(pb[q][2], pb[q][1]) : (PR == 1) ? f4add(pb[q][0], pb[q][1]) : (PR == 2) ? f4sub(pb[q][1], pb[q][0]) : f4sub(pb[q][0], pb[q][2]);
                } else {
                    float4 e, o;
                    eo(pb[q][0], pb[q][1], e, o);
                    const float4 &l0 = pb[q][0], &l1 = pb[q][1], &x2 = pb[q][2];
                    if constexpr (PR == 1) v = f4add(e, o);
                    else if constexpr (PR == 2) v = f4sub(o, e);
                    else if constexpr (PR == 0) {           // y[tE - d] - y[tO]; x2 = frames 8 q - 4 .. 8 q - 1
                        const float4 m = (DS == 1) ? make_float4(x2.x, l0.y, l0.w, l1.y) : make_float4(x2.x, x2.y, l0.z, l0.w);      // x2 = y[8 q - d ..]
                        v = f4sub(m, o);
                    } else {                                // y[tE] - y[tO + d]; x2 = frames 8 q + 8 .. 8 q + 11
                        const float4 n = (DS == 1) ? make_float4(l0.z, l1.x, l1.z, x2.x) : make_float4(l1.x, l1.y, x2.x, x2.y);
                        v = f4sub(e, n);
                    }
                }
                *reinterpret_cast<float4*>(Bs + (srow + 32 * q) * LD + 4 * sg) = v;
            }
        };
        next_step(); fetch_a(); fetch_b(0); fetch_b(4);
        stash_a(0, 1.f); stash_b(0, 0); stash_b(0, 4);
        next_step(); fetch_a(); fetch_b(0); fetch_b(4);
        __syncthreads();
        frags(0, 0, 0);
        for (int k = 0; k < nstep; ++k) {
            const int cur = k & 1;
            const float live = (k + 1 < nstep) ? 1.f : 0.f;
            frags(1, cur, 1);
            stash_a(cur ^ 1, live);
            next_step();                                     // (scalar: the descriptors and offsets of the step behind the one in the registers)
            fetch_a();
            mma(0);
            tr_interleave<2, true, true, true>(0);
            DSD_SB();
            frags(0, cur, 2);
            stash_b(cur ^ 1, 0);
            fetch_b(0);
            mma(1);
            tr_interleave<3, true, true, true>(0);
            DSD_SB();
            frags(1, cur, 3);
            stash_b(cur ^ 1, 4);
            fetch_b(4);
            mma(0);
            tr_interleave<3, true, true, true>(0);
            DSD_SB();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            frags(0, cur ^ 1, 0);
            mma(1);
            tr_interleave<0, false, false, true>(0);
            DSD_SB();
        }
    };
    auto dual_dispatch = [&](auto ds_c) {
        using T_ = std::true_type; using F_ = std::false_type;
        if (d.prod == 0) loop_dual(ds_c, std::integral_constant<int, 0>{}, F_{});
        else if (d.prod == 1) { if (has_bias) loop_dual(ds_c, std::integral_constant<int, 1>{}, T_{}); else loop_dual(ds_c, std::integral_constant<int, 1>{}, F_{}); }
        else if (d.prod == 2) loop_dual(ds_c, std::integral_constant<int, 2>{}, F_{});
        else loop_dual(ds_c, std::integral_constant<int, 3>{}, F_{});
    };
    if (nstep > 0) {
        if constexpr (!FIX) {
            if (dual) {
                if (d.shift == 1) dual_dispatch(std::integral_constant<int, 1>{});
                else if (d.shift == 2) dual_dispatch(std::integral_constant<int, 2>{});
                else dual_dispatch(std::integral_constant<int, 4>{});
            } else if (has_bias) loop(std::true_type{});
            else loop(std::false_type{});
        } else {
            loop(std::true_type{});
        }
    }
    float* out = p.part + (size_t)tr_wg_part_index(p, desc, split) * (128 * 256);       // wave-uniform; write-through: the reduction kernel reads it next
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) store4_wt(out, (64 * wm + 32 * mb + frag_row(r, h)) * 256 + 128 * wn + 32 * nb + i, FIX ? acc[mb][nb][r] : acc[mb][nb][r] * a_scale);
    if (d.out_bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sb = FIX ? bsum[q] : bsum[q] * a_scale;
            sb += __shfl_xor(sb, 1, 64); sb += __shfl_xor(sb, 2, 64); sb += __shfl_xor(sb, 4, 64);
            if (sg == 0) p.part_b[(size_t)tr_wg_part_index(p, desc, split) * 128 + srow + 32 * q] = sb;
        }
    }
}

// rows of the saved y carry kTrYPad zero floats on both sides: written once per forward
__global__ void k_tr_zero_pads(float* __restrict__ y, size_t rows, int rs) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < rows * (2 * kTrYPad); i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / (2 * kTrYPad);
        const int e = (int)(i - row * (2 * kTrYPad));
        y[row * rs + (e < kTrYPad ? e : rs - 2 * kTrYPad + e)] = 0.f;
    }
}

// dW = sum over the splits, in split order; one thread per element of a 128 x 256 tile, grid (tile descriptor, 128)
// grid (plain descriptors, 128): the descriptors [nc, ndesc)
__device__ __forceinline__ void tr_wgrad_reduce_body(const TrWgParams& p, int bx) {
    const int desc = p.nc + bx;
    const TrWgTile& d = p.tile[desc];
    const int m = blockIdx.y, n = threadIdx.x;
    const float* src = p.part + (size_t)tr_wg_part_index(p, desc, 0) * (128 * 256) + m * 256 + n;
    float s = 0.f;
#pragma unroll 6
    for (int k = 0; k < p.nsplit; ++k) s += src[(size_t)k * (128 * 256)];       // (the splits' loads in flight together, the additions in split order)
    d.out[(size_t)m * d.out_rs + (size_t)n * d.out_cs] = s;
    if (d.out_bias && n == 0) {
        const float* sb = p.part_b + (size_t)tr_wg_part_index(p, desc, 0) * 128 + m;
        float t = 0.f;
        for (int k = 0; k < p.nsplit; ++k) t += sb[(size_t)k * 128];
        d.out_bias[m] = t;
    }
}

// grid (nc / 4, 128): the four products of one 128-row tile (descriptors 4 g .. 4 g + 3 = P0 .. P3, each summed over its splits in split
// order) back-transformed into the three tap gradients; descriptor 4 g carries `out` (tap 0: taps are out + 1, out + 2) and 4 g + 1 the bias
__device__ __forceinline__ void tr_wgrad_reduce_dual_body(const TrWgParams& p, int bx) {
    const int g = bx, m = blockIdx.y, n = threadIdx.x;
    float P[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* src = p.part + (size_t)tr_wg_part_index(p, 4 * g + i, 0) * (128 * 256) + m * 256 + n;
        float s = 0.f;
        for (int k = 0; k < p.ns_c; ++k) s += src[(size_t)k * (128 * 256)];
        P[i] = s;
    }
    const TrWgTile& d = p.tile[4 * g];
    const float hs = 0.5f * (P[1] + P[2]);
    float* o = d.out + (size_t)m * d.out_rs + (size_t)n * d.out_cs;
    o[0] = P[0] + hs;
    o[1] = 0.5f * (P[1] - P[2]);
    o[2] = hs - P[3];
    const TrWgTile& d1 = p.tile[4 * g + 1];
    if (d1.out_bias && n == 0) {
        const float* sb = p.part_b + (size_t)tr_wg_part_index(p, 4 * g + 1, 0) * 128 + m;
        float t = 0.f;
        for (int k = 0; k < p.ns_c; ++k) t += sb[(size_t)k * 128];
        d1.out_bias[m] = t;
    }
}

__global__ __launch_bounds__(256) void k_tr_wgrad_reduce(const TrWgParams p) { tr_wgrad_reduce_body(p, (int)blockIdx.x); }
__global__ __launch_bounds__(256) void k_tr_wgrad_reduce_dual(const TrWgParams p) { tr_wgrad_reduce_dual_body(p, (int)blockIdx.x); }
// both reductions of a weight-gradient launch in ONE grid (round 6: ten pairs of 6-7 us launches per training step): blocks [0, nc / 4) are the
// dual groups, the rest the plain descriptors - the same sums in the same order
__global__ __launch_bounds__(256) void k_tr_wgrad_reduce_all(const TrWgParams p) {
    const int ng = p.nc / 4;
    if ((int)blockIdx.x < ng) tr_wgrad_reduce_dual_body(p, (int)blockIdx.x);
    else tr_wgrad_reduce_body(p, (int)blockIdx.x - ng);
}

}  // namespace dsd
