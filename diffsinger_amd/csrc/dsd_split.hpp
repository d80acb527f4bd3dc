// dsd_split.hpp - EXPERIMENTAL residual-layer kernel on the bf16 matrix pipe with fp32-class accuracy ("beyond the
// fp32-MFMA ceiling").  Opt-in (dsd_set_split_mode / DSD_SPLIT=1), per-layer kernel path only; the default path does not touch it.
//
// Same computation and the same global data structures as k_layer<1, LAST> (usr/diff/net.py:66-78: y = x + step, dilated conv + hoisted
// conditioner projection, sigmoid * tanh gate, output projection, x' = (x + res) / sqrt(2), skip sum) - tile-major x, fragment-order cp /
// skip, the step table - but the two GEMMs run as SIX bf16 plane products per fp32 product on v_mfma_f32_32x32x16_bf16:
//     every fp32 operand = p0 + p1 + p2 exactly (three bf16 planes, 8 + 8 + 8 mantissa bits);  a * b ~= sum_{i + j <= 2} a_i * b_j,
//     each plane product exact in fp32, accumulated in fp32: 6 x 32 = 192 matrix-pipe cycles per K = 16 instead of 8 x 64 = 512.
// Operands:
//   A  weights as planes in 32x32x16 fragment order [wave 4][chunk][row block 4][plane 3][lane 64] x 8 bf16 (16 B per lane and plane),
//      built on the device from the fp32 fragment-order weights the handle already holds (k_pack_split); chunk = 3 g + tap for the
//      dilated conv (g = 16-channel group), = g for the output projection.  Lane (i, h) element e stands for channel 16 g + 8 h + e.
//   B  activations as bf16 planes in LDS, FRAME-major [plane][frame][channel] with a padded 528-byte frame stride: a lane's fragment is
//      8 consecutive channels of one frame = one ds_read_b128 per plane, conflict-free.  The k order inside an MFMA is whatever the
//      hardware defines - A and B use the same (h, e) -> channel map, so it cancels.
// First version: the whole y tile is staged before the first MFMA (no progressive quarters), 2-byte LDS writes in the staging pass.
#pragma once
#include "dsd_kernels.hpp"

namespace dsd {

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short su16;

constexpr int kSpRS = 264;                              // bf16 per frame row: 256 channels + 8 pad
constexpr int kSpYFrames = 32 + 2 * kHalo;              // 48
constexpr int kSpYPlane = kSpYFrames * kSpRS;           // elements per y plane
constexpr int kSpGPlane = 32 * kSpRS;                   // elements per gate plane
constexpr int kSplitStages1 = 6;                        // weight-stream register stages of the conv (5 chunks x 768 MFMA cycles ahead)
constexpr int kSplitLayerLdsBytes = (3 * kSpYPlane + 3 * kSpGPlane) * 2;        // 126 720 B

__device__ __forceinline__ void sp_split3(float x, su16& a, su16& b, su16& c) {
    const __bf16 p0 = (__bf16)x;
    const float r1 = x - (float)p0;
    const __bf16 p1 = (__bf16)r1;
    const __bf16 p2 = (__bf16)(r1 - (float)p1);
    a = __builtin_bit_cast(su16, p0); b = __builtin_bit_cast(su16, p1); c = __builtin_bit_cast(su16, p2);
}

// The pair format (dsd_loop_split.hpp, SplitPipeF): x = h0 + 2^-11 * h1 with h0 = fp16(x) and h1 = fp16((x - h0) * 2^11) - 11 + 11 mantissa
// bits, the second plane scaled up so that it stays out of the fp16 denormals for every x whose first plane is normal.
constexpr float kPairScale = 2048.f, kPairInv = 1.f / 2048.f;
__device__ __forceinline__ void sp_split2h(float x, su16& a, su16& b) {
    const _Float16 h0 = (_Float16)x;
    const _Float16 h1 = (_Float16)((x - (float)h0) * kPairScale);
    a = __builtin_bit_cast(su16, h0); b = __builtin_bit_cast(su16, h1);
}

// fp32 fragment-order weights (k_pack_a, 32x32x2: [w][chunk8][mb 4][lane][4], chunk8 = conv_chunk(k8, tap) for the dilated conv, = k8 for
// the projections; lane (i, h), s -> channel 8 k8 + 4 h + s)
// -> bf16 planes in 32x32x16 fragment order [w][chunk16 = ntap * g + tap][mb 4][plane 3][lane (i, h')][e 8], channel 16 g + 8 h' + e.
// centre_first (ntap == 3, the persistent split loop dsd_loop_split.hpp): destination chunk order = the ng centre-tap chunks, then the (-dil, +dil)
// pairs of every group - like conv_chunk() for the fp32 stream - instead of "taps of one group consecutive".
// wave_stride / chunk_stride (in bf16 elements; 0 = the dense [w][chunk16] order above): the persistent loop keeps a layer's planes in
// CONSUMPTION order [chunk16 of W1 then of W2][w][mb][plane][lane] - a chunk of all four waves is 48 KiB of consecutive lines, which is
// what lets the workgroups of an XCD touch the stream ahead of themselves line by line (dsd_loop_split.hpp, L2 touch).
// PAIR: the two fp16 planes of the pair format instead of the three bf16 planes ([mb][plane 2][lane] x 8: 4096 elements per (chunk, wave)).
template <bool PAIR = false>
__global__ void k_pack_split(const float* __restrict__ src, su16* __restrict__ dst, int nw, int ng, int ntap, int centre_first,
                             long long wave_stride, long long chunk_stride) {
    constexpr int NPL = PAIR ? 2 : 3;
    const size_t ws = wave_stride ? (size_t)wave_stride : (size_t)ng * ntap * (4 * NPL * 512), cs = chunk_stride ? (size_t)chunk_stride : (4 * NPL * 512);
    const size_t n = (size_t)nw * ng * ntap * 4 * 64 * 8;              // (w, chunk16, mb, lane, e)
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63;
        size_t r = idx >> 9;
        const int mb = r & 3; r >>= 2;
        const int c16 = r % (ng * ntap); r /= (ng * ntap);
        const int w = (int)r;
        int g = c16 / ntap, tap = c16 - g * ntap;
        if (centre_first && ntap == 3) {
            if (c16 < ng) { g = c16; tap = 1; }
            else { const int ix = c16 - ng; g = ix >> 1; tap = (ix & 1) * 2; }
        }
        const int i = lane & 31, hp = lane >> 5;
        const int k8 = 2 * g + hp, c8 = (ntap == 3) ? conv_chunk(k8, tap) : ntap * k8 + tap;    // source chunk (the fp32 stream's order)
        const int lane_src = i + 32 * (e >> 2), s = e & 3;
        const float v = src[((((size_t)w * (2 * ng * ntap) + c8) * 4 + mb) * 64 + lane_src) * 4 + s];
        const size_t o = (size_t)w * ws + (size_t)c16 * cs + (size_t)(mb * NPL) * 512 + (size_t)lane * 8 + e;
        if constexpr (PAIR) {
            su16 h0, h1;
            sp_split2h(v, h0, h1);
            dst[o] = h0; dst[o + 512] = h1;
        } else {
            su16 p0, p1, p2;
            sp_split3(v, p0, p1, p2);
            dst[o] = p0; dst[o + 512] = p1; dst[o + 1024] = p2;
        }
    }
}

// Operand pipeline: STAGES (3 or 6) register stages of the weight stream (chunk kc + STAGES - 1 requested while chunk kc is multiplied; every
// chunk is a first touch of the XCD's L2, the fp32 kernels needed ~5 k cycles of distance), B one chunk ahead,
// loads interleaved one-by-one behind the first MFMAs of a step.  NMB row blocks starting at MB0 (the last layer computes the skip half only).
// TAPS = 3: chunk = 3 g + tap, B row of tap from btap[tap]; TAPS = 1: chunk = g.
template <int NMB, int MB0, int TAPS, int STAGES>
struct SplitPipeL {
    static_assert(STAGES == 3 || STAGES == 6, "register rotation period is 6");
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned aoff;
    int n;
    const su16* btap[TAPS];
    int bplane;                  // elements between the planes of the B tile
    uint4 a[STAGES][NMB][3];
    sbf16x8 b[2][3];

    __device__ __forceinline__ SplitPipeL(const uint4* wave_base, int lane, int n_, int bplane_)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wave_base), 0, 0x7ffffff0, 0x00020000)), aoff((unsigned)lane * 16u), n(n_),
          bplane(bplane_) {}
    __device__ __forceinline__ void lda(uint4 (&dst)[NMB][3], int kc) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int kcc = (kc < n) ? kc : n - 1;                             // prefetches past the end re-read the last chunk
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)aoff, kcc * 12288 + ((MB0 + mb) * 3 + pl) * 1024, 0);
                dst[mb][pl] = make_uint4(v.x, v.y, v.z, v.w);
            }
    }
    template <int I>
    __device__ __forceinline__ void ldb(sbf16x8 (&dst)[3], int it) {          // chunk 6 it + I
        int kc = 6 * it + I;
        kc = (kc < n) ? kc : n - 1;
        const su16* bp = (TAPS == 3) ? btap[I % 3] + (kc / 3) * 16 : btap[0] + kc * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl] = __builtin_bit_cast(sbf16x8, *reinterpret_cast<const uint4*>(bp + pl * bplane));
    }
    __device__ __forceinline__ void pattern() {
#pragma unroll
        for (int i = 0; i < 3 * NMB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NMB - 3 * NMB - 3, 0);
    }
    __device__ __forceinline__ void start_a() {
#pragma unroll
        for (int i = 0; i < STAGES - 1; ++i) lda(a[i], i);
        DSD_SB();
    }
    __device__ __forceinline__ void start_b() {
        ldb<0>(b[0], 0);
        DSD_SB();
    }
    template <int I>
    __device__ __forceinline__ void step(f32x16 (&acc)[NMB], int it) {
        lda(a[(I + STAGES - 1) % STAGES], 6 * it + I + STAGES - 1);
        if (I == 5) ldb<0>(b[0], it + 1); else ldb<(I + 1) % 6>(b[(I + 1) & 1], it);
        constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};      // smallest plane products first
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sbf16x8, a[I % STAGES][mb][TI[q]]), b[I & 1][TJ[q]], acc[mb], 0, 0, 0);
        pattern();
        DSD_SB();
    }
    __device__ __forceinline__ void run(f32x16 (&acc)[NMB]) {
        for (int it = 0; 6 * it < n; ++it) {
            const int kc = 6 * it;
            step<0>(acc, it);
            if (kc + 1 >= n) break;
            step<1>(acc, it);
            if (kc + 2 >= n) break;
            step<2>(acc, it);
            if (kc + 3 >= n) break;
            step<3>(acc, it);
            if (kc + 4 >= n) break;
            step<4>(acc, it);
            if (kc + 5 >= n) break;
            step<5>(acc, it);
        }
    }
};

template <bool LAST>
__global__ __launch_bounds__(kThreads, 1) void k_layer_split(const LayerParams p) {
    constexpr int TILE = kC * 32;
    extern __shared__ __attribute__((aligned(16))) su16 spl[];
    su16* yp = spl;                                   // [3][48][264]
    su16* gp = spl + 3 * kSpYPlane;                   // [3][32][264]

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, tn;
    if (p.xcd_q >= 0) {
        const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
        const int tl = xcd * p.xcd_q + min(xcd, p.xcd_r) + k;
        b = tl / p.tiles_per_utt;
        tn = tl - b * p.tiles_per_utt;
    } else {
        b = blockIdx.y; tn = blockIdx.x;
    }
    const int tile0 = b * p.ntile32 + tn;
    const float* __restrict__ xt = p.x_in + (size_t)tile0 * TILE;
    const int t0 = tn * 32;
    const int tstep = p.t_dev ? p.t_dev[b] : p.t_uniform;
    const float* __restrict__ dsl = p.ds + (size_t)tstep * p.ds_tstride;

    // weight streams do not depend on the tile: request the first chunks of the conv before anything else
    SplitPipeL<4, 0, 3, kSplitStages1> pipe1(reinterpret_cast<const uint4*>(p.w1p) + (size_t)w * (48 * 12 * 64), lane, 48, kSpYPlane);
    pipe1.start_a();

    // 1. stage y = x + step_proj (zero outside [0, T): the conv pads y) as three bf16 planes, frame-major.
    //    Tile: thread tid owns bytes [16 tid, +16) of every 4 KiB slab = row sl * 32 + tid / 8, frames 4 (tid & 7) .. +3.
    {
        const int c4 = tid & 7;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const int row = sl * 32 + (tid >> 3);
            const float4 v = reinterpret_cast<const float4*>(xt)[sl * 256 + tid];
            const float d = dsl[row];
            const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = t0 + 4 * c4 + e;
                const float yv = (t < p.T) ? ve[e] + d : 0.f;
                su16 a0, a1, a2;
                sp_split3(yv, a0, a1, a2);
                const int o = (kHalo + 4 * c4 + e) * kSpRS + row;
                yp[o] = a0; yp[kSpYPlane + o] = a1; yp[2 * kSpYPlane + o] = a2;
            }
        }
        // halo: 8 frames on each side, rows 64 q + tid / 4; hpart 0,1 = left neighbour's last 8 frames, 2,3 = right neighbour's first 8
        const int hpart = tid & 3;
        const bool hleft = hpart < 2;
        const bool hhave = hleft ? (tn > 0) : (tn + 1 < p.ntile32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 64 * q + (tid >> 2);
            const float* src = hleft ? xt - TILE + row * 32 + 24 + 4 * hpart : xt + TILE + row * 32 + 4 * (hpart - 2);
            const float4 v = *reinterpret_cast<const float4*>(hhave ? src : xt + row * 32);
            const float d = dsl[row];
            const float ve[4] = {v.x, v.y, v.z, v.w};
            const int tb = hleft ? t0 - kHalo + 4 * hpart : t0 + 32 + 4 * (hpart - 2);
            const int fb = hleft ? 4 * hpart : kHalo + 32 + 4 * (hpart - 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yv = (hhave && tb + e < p.T) ? ve[e] + d : 0.f;
                su16 a0, a1, a2;
                sp_split3(yv, a0, a1, a2);
                const int o = (fb + e) * kSpRS + row;
                yp[o] = a0; yp[kSpYPlane + o] = a1; yp[2 * kSpYPlane + o] = a2;
            }
        }
    }
    __syncthreads();

    // 2. dilated conv, K = 3 taps x 256 channels = 48 chunks of 16
    f32x16 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) pipe1.btap[tp] = yp + (j + kHalo + (tp - 1) * p.dil) * kSpRS + 8 * h;
    pipe1.start_b();
    float4 cpv[4][4];
    {
        const float4* cpl = p.cp + ((size_t)tile0 * 4 + w) * (4 * 4 * 64) + lane;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) cpv[mb][q] = cpl[(mb * 4 + q) * 64];
    }
    DSD_SB();
    pipe1.run(acc);

    // out-proj weight stream: first chunks requested before the gate arithmetic
    constexpr int NMB2 = LAST ? 2 : 4;
    constexpr int MB0 = LAST ? 2 : 0;
    SplitPipeL<NMB2, MB0, 1, 3> pipe2(reinterpret_cast<const uint4*>(p.w2p) + (size_t)w * (16 * 12 * 64), lane, 16, kSpGPlane);
    pipe2.start_a();

    // 3. gate in registers (rows [64 w, 64 w + 64) are gates, row blocks 2,3 their filters, net.py:73-74), written as bf16 planes
    //    [plane][frame j][channel]: a register quad is four consecutive channels = one 8-byte write per plane
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            su16 q0[4], q1[4], q2[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int r = 4 * rg + s;
                const float vg = f4at(cpv[pr][rg], s), vf = f4at(cpv[pr + 2][rg], s);
                const float g = sigmoid_f(acc[pr][r] + vg) * tanh_f(acc[pr + 2][r] + vf);
                sp_split3(g, q0[s], q1[s], q2[s]);
            }
            const int o = j * kSpRS + 64 * w + 32 * pr + 8 * rg + 4 * h;
            typedef su16 su16x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<su16x4*>(gp + o) = su16x4{q0[0], q0[1], q0[2], q0[3]};
            *reinterpret_cast<su16x4*>(gp + kSpGPlane + o) = su16x4{q1[0], q1[1], q1[2], q1[3]};
            *reinterpret_cast<su16x4*>(gp + 2 * kSpGPlane + o) = su16x4{q2[0], q2[1], q2[2], q2[3]};
        }
    __syncthreads();

    // 4. output projection (K = 256 = 16 chunks): row blocks 0,1 = residual rows, 2,3 = skip rows (the last layer: skips only)
    f32x16 acc2[NMB2];
#pragma unroll
    for (int m = 0; m < NMB2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;
    pipe2.btap[0] = gp + j * kSpRS + 8 * h;
    pipe2.start_b();
    float4 xrow[8], skp[2][4];
    float brow[8];
    if (!LAST) {
#pragma unroll
        for (int it = 0; it < 8; ++it) brow[it] = p.b2[64 * w + it * 8 + (lane >> 3)];
#pragma unroll
        for (int it = 0; it < 8; ++it) xrow[it] = reinterpret_cast<const float4*>(xt + 64 * w * 32)[it * 64 + lane];
    }
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const float4* sl = p.skip + (((size_t)tile0 * 4 + w) * 2 + ms) * (4 * 64) + lane;
        const bool keep = !p.first;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = sl[q * 64];
            skp[ms][q] = make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
        }
    }
    DSD_SB();
    pipe2.run(acc2);

    // 5. epilogue - as in k_layer: the residual goes through this wave's slice of the (free) y region into row layout,
    //    x' = (x + (res + bias)) / sqrt(2), tile-major x_out in 1 KiB-per-instruction stores; the skip sum in fragment order
    float* tw = reinterpret_cast<float*>(yp) + w * (64 * 32);
    if (!LAST) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tw[(32 * mb + frag_row(r, h)) * 32 + j] = acc2[mb][r];
        __builtin_amdgcn_wave_barrier();
        float* __restrict__ xo = p.x_out + (size_t)tile0 * TILE + 64 * w * 32;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const float4 v = reinterpret_cast<const float4*>(tw)[it * 64 + lane];
            const float4 x = xrow[it];
            const float bv = brow[it];
            constexpr float kInvSqrt2 = 1.0f / 1.41421354f;
            float4 o;
            o.x = (x.x + (v.x + bv)) * kInvSqrt2;
            o.y = (x.y + (v.y + bv)) * kInvSqrt2;
            o.z = (x.z + (v.z + bv)) * kInvSqrt2;
            o.w = (x.w + (v.w + bv)) * kInvSqrt2;
            if (p.wt_stores) store16<true>(reinterpret_cast<float4*>(xo), it * 64 + lane, o);
            else store16<false>(reinterpret_cast<float4*>(xo), it * 64 + lane, o);
        }
    }
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        float4* sl = p.skip + (((size_t)tile0 * 4 + w) * 2 + ms) * (4 * 64);
        const int m = (LAST ? 0 : 2) + ms;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = get4(acc2[m], q), s = skp[ms][q];
            const float4 sv = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
            if (p.wt_stores) store16<true>(sl, q * 64 + lane, sv);
            else store16<false>(sl, q * 64 + lane, sv);
        }
    }
}

// ---- debug plumbing (dsd_debug_layer): logical [B][C][TS] <-> the layer kernels' images ------------------------------------------------
// x: tile-major [tile = b * ntile32 + tn][C][32]
__global__ void k_dbg_to_tiles(float* __restrict__ x, float* __restrict__ xt, int TS, int ntile32, int to_tiles) {
    const int tile = blockIdx.x, b = tile / ntile32, tn = tile - b * ntile32;
    for (int idx = threadIdx.x; idx < kC * 32; idx += blockDim.x) {
        const int c = idx >> 5, f = idx & 31;
        const size_t lo = ((size_t)b * kC + c) * TS + tn * 32 + f, ti = (size_t)tile * kC * 32 + idx;
        if (to_tiles) xt[ti] = x[lo]; else x[lo] = xt[ti];
    }
}
// skip sum: fragment order [tile][wave 4][ms 2][q 4][lane 64] float4 -> logical; element (q, lane = (j, h), s) is channel
// 64 w + 32 ms + frag_row(4 q + s, h) at frame j
__global__ void k_dbg_skip_to_logical(const float4* __restrict__ skip, float* __restrict__ out, int TS, int ntile32) {
    const int tile = blockIdx.x, b = tile / ntile32, tn = tile - b * ntile32;
    for (int idx = threadIdx.x; idx < 4 * 2 * 4 * 64; idx += blockDim.x) {
        const int lane = idx & 63, q = (idx >> 6) & 3, ms = (idx >> 8) & 1, w = idx >> 9;
        const float4 v = skip[(size_t)tile * 2048 + idx];
        const int j = lane & 31, h = lane >> 5;
        const float ve[4] = {v.x, v.y, v.z, v.w};
        for (int e = 0; e < 4; ++e) out[((size_t)b * kC + 64 * w + 32 * ms + frag_row(4 * q + e, h)) * TS + tn * 32 + j] = ve[e];
    }
}

}  // namespace dsd
