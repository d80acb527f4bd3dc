// train_wino_bwd.hpp - the data-gradient kernel of the fused training stack (k_trb_fused, train_kernels.hpp) with the TRANSPOSED dilated
// convolution as WINOGRAD F(2,3) along the frame axis (gfx950).
//
// dy[ci][t] = sum_co  W[co][ci][2] da[co][t - d] + W[co][ci][1] da[co][t] + W[co][ci][0] da[co][t + d]   (the adjoint of net.py:71's Conv1d)
// is a 3-tap dilated convolution with the taps flipped and the channel roles exchanged: for an output pair (t, t + d), with g'k = W[..][2 - k],
//     M0 = g'0 (d0 - d2)      M1 = (g'0+g'1+g'2)/2 (d1 + d2)      M2 = (g'0-g'1+g'2)/2 (d2 - d1)      M3 = g'2 (d3 - d1)
//     dy[t] = M0 + M1 + M2        dy[t+d] = M1 - M2 + M3          (d0..d3 = da at t-d, t, t+d, t+2d)
// - four [256 x 512] . [512 x 16 pairs] products per 32-frame tile instead of three [256 x 512] . [512 x 32]: 16.8 M instead of 25.2 M FLOP.
// The operand pipeline IS the forward's (WinoPipe, dsd_loop_wino.hpp: v_mfma_f32_16x16x4_f32 over the 16 pairs, four row blocks per step,
// transformed weights in consumption order + L2 touch, the halo-free products M1 / M2 first, then t = M1 + M2, u = M1 - M2 in place and
// M0 / M3 on top) with a CHANNEL-MAJOR B tile: the da tile stays [512][48] as k_trb_fused stages it (its producer and the weight-gradient
// kernel want frames contiguous), a lane reads its k rows 16 c + 4 s + g at the pair's four frames as ds_read_b32.
// Wave roles are k_trb_fused's: wave (wr, wk) multiplies the 128 input-channel rows [128 wr, +128) (8 row blocks of 16) over the K half wk
// (the 256 gate / filter rows of da), the halves are summed through LDS in a fixed order, wave (wr, wk) finishes rows [128 wr + 64 wk, +64).
// dy then goes through the dy2 tile in LDS (it is the next contraction's operand anyway) and leaves as float4 rows - the direct kernel's
// 32 scalar stores per lane become 8 x 16 bytes per thread.  The second part (output-projection data gradient + gate derivative of layer
// l - 1) is k_trb_fused's code unchanged.  Results differ from k_trb_fused by reduction order and the transforms' roundings (every gradient
// against float64 autograd inside the same tolerance, tests/test_gpu_train_fused.py); dsf_set_stack_conv(0) selects the direct kernels.
#pragma once
#include "dsd_loop_wino.hpp"
#include "train_kernels.hpp"

namespace dsd {

// Transformed, transposed, flipped conv weights in consumption order: dst[l][step 128][w4][r4][lane64][s4], step = ((half * 16 + c) * 2 + pos) * 2 + hb;
// wave w = (wr = w & 1, wk = w >> 1); row block rb = 4 hb + r4: input channel ci = 128 wr + 16 rb + n, n = lane & 15; da row co = 256 wk + 16 c + 4 s + g,
// g = lane >> 4.  half 0: U1, U2 (the halo-free products); half 1: U0 = g'0 = W[co][ci][2], U3 = g'2 = W[co][ci][0].  src = dilated_conv.weight [2C][C][3].
// One workgroup per (layer, wave w, row block rb): its operands are the 16 input channels ci of the row block against the 256 da rows of the
// wave's K half - 48 contiguous floats of each of 256 source rows, staged in LDS as [co 256][52] and written as the 64 fragment rows (1 KiB
// each) of the steps the row block takes part in.  fp64 sums, one rounding.
__global__ __launch_bounds__(256) void k_pack_wino_bwd_multi(const TrPtrs src, float* __restrict__ dst) {
    constexpr int LDP = 52;
    __shared__ __attribute__((aligned(16))) float wt[256 * LDP];
    const int tid = threadIdx.x, w = blockIdx.x >> 3, rb = blockIdx.x & 7, hb = rb >> 2, r4 = rb & 3, wr = w & 1, wk = w >> 1;
    const float* sp = src.p[blockIdx.y] + ((size_t)(256 * wk) * kC + 128 * wr + 16 * rb) * 3;
    for (int i = tid; i < 256 * 12; i += 256) {
        const int co = i / 12, c4 = i - co * 12;
        *reinterpret_cast<float4*>(wt + co * LDP + 4 * c4) = *reinterpret_cast<const float4*>(sp + (size_t)co * kC * 3 + 4 * c4);
    }
    __syncthreads();
    float4* d = reinterpret_cast<float4*>(dst + (size_t)blockIdx.y * ((size_t)kWnSteps * 4 * 4 * 64 * 4));
    const int lane = tid & 63, nn = lane & 15, g = lane >> 4;
    for (int combo = tid >> 6; combo < 64; combo += 4) {            // combo = (half * 16 + c) * 2 + pos
        const int pos = combo & 1, c = (combo >> 1) & 15, half = combo >> 5;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* wp = wt + (16 * c + 4 * e + g) * LDP + 3 * nn;
            const double g0 = wp[2], g1 = wp[1], g2 = wp[0];         // the taps flipped
            double u;
            if (half == 0) u = pos ? 0.5 * (g0 - g1 + g2) : 0.5 * (g0 + g1 + g2);
            else u = pos ? g2 : g0;
            o[e] = (float)u;
        }
        const int st = combo * 2 + hb;
        d[(((size_t)st * 4 + w) * 4 + r4) * 64 + lane] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

struct TrbFusedWinoParams {
    TrbFusedParams f;           // c.wdtp unused
    const float4* wdw;          // this layer's transformed weights [128 steps][w4][r4][lane64]
    unsigned wl_bytes;          // bytes of one layer's stream (+ slack behind it): the L2 touch's buffer bound
    int touch_ahead;            // steps the L2 touch runs in front (0 = off)
    unsigned long long* dbg;    // optional s_memtime stamps (dsf_debug_trb_timeline): [workgroup][wave 4][8]
};
constexpr int kTrbFusedWinoLdsBytes = kLoopTouchLds + kTrbConvLdsBytes;

// the K halves of two waves -> the finishing wave: acc[hf][rb] (8 row blocks x 2 output halves); wave (wr, wk) keeps row blocks 4 wk .. 4 wk + 3
__device__ __forceinline__ void trb_exchange_w(f32x4w (&acc)[2][8], f32x4w (&fin)[2][4], float* xbuf, int wr, int wk, int lane) {
    float4* mine = reinterpret_cast<float4*>(xbuf) + ((wr * 2 + wk) * 8) * 64 + lane;             // what the partner finishes
    const float4* theirs = reinterpret_cast<const float4*>(xbuf) + ((wr * 2 + (1 - wk)) * 8) * 64 + lane;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4w v = wk ? acc[hf][r4] : acc[hf][4 + r4];
            mine[(hf * 4 + r4) * 64] = make_float4(v[0], v[1], v[2], v[3]);
        }
    __syncthreads();
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4w own = wk ? acc[hf][4 + r4] : acc[hf][r4];
            const float4 o4 = theirs[(hf * 4 + r4) * 64];
            const f32x4w oth = {o4.x, o4.y, o4.z, o4.w};
            fin[hf][r4] = wk ? oth + own : own + oth;               // always (K half 0) + (K half 1)
        }
}

// LAST: layer l is the last layer (no residual path into its conv gradient); GATE = false: the convolution part alone (layer 0: k_trb_conv's role)
template <bool LAST, bool GATE = true>
__global__ __launch_bounds__(kThreads, 1) void k_trb_fused_w(const TrbFusedWinoParams qw) {
    constexpr int LD = kTrbConvLD, S = 4;
    const TrbConvParams& p = qw.f.c;
    const TrbGateParams& pg = qw.f.g;
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    float* smem = smem_all + kLoopTouchLds / 4;                       // da tile [512][48], then dy2 tile [512][32]; exchange
    float* xbuf = smem + 2 * kC * LD;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int pp = lane & 15, gg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w & 1, wk = w >> 1;
    const int tile = xcd_item(blockIdx.x, p.xcd_q, p.xcd_r), b = tile / p.ntile32, t0 = (tile - b * p.ntile32) * 32;
    const int dil = p.dil, de = __builtin_ctz((unsigned)dil);
    L2TouchP tc;
    {
        const unsigned long long wb = (unsigned long long)qw.wdw;
        const int nt = (int)gridDim.x, xcd = (int)(blockIdx.x & 7), nwx = 4 * ((nt - xcd + 7) >> 3), q = 4 * (int)(blockIdx.x >> 3) + w;
        const bool en = qw.touch_ahead > 0 && nwx >= 8;
        tc.rs = L2Touch::i32x4_{(int)(unsigned)wb, (int)(unsigned)((wb >> 32) & 0xffffu), (int)qw.wl_bytes, 0x00020000};
        tc.ahead = (unsigned)(qw.touch_ahead + 7) / 8u;
        tc.nwx = nwx;
        tc.dec = en ? 16 % nwx : 0;
        tc.r = en ? q : 1 << 20;
        tc.gtot = (unsigned)(kWnSteps / 8);
        tc.lds = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float*)smem_all + (unsigned)w * 256u;
        tc.lane128 = (unsigned)lane * 128u;
    }
    const bool stamp = qw.dbg != nullptr && lane == 0;
    auto mark = [&](int k) { if (stamp) qw.dbg[((size_t)blockIdx.x * 4 + w) * 8 + k] = __builtin_amdgcn_s_memtime(); };
    mark(0);
    // row g of this wave's K half at the pair's even frame (the tile's frame 0 is column kHalo)
    const float* pE = smem + (wk * kC + gg) * LD + kHalo + wn_frame_of_pair(pp, de);
    WinoPipe<S, LD> pipe(qw.wdw + (size_t)w * 256, lane, 0, pE, pE + dil, dil, tc);

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
    pipe.template start_a<S - 1>();
    request(1); request(2); request(3);
    // residual-path gradient and the skip gradient of the rows / frames this THREAD carries through the dy2 tile: row 32 it + tid / 8, frames 4 (tid % 8) ..
    const int sg = tid & 7, st = t0 + 4 * sg;
    float4 vx[8];
#pragma unroll
    for (int it = 0; it < 8; ++it)
        vx[it] = LAST ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(p.dxp + ((size_t)b * kC + it * 32 + (tid >> 3)) * p.TS + st);
    DSD_SB();
    f32x4w acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) acc[i][rb] = f32x4w{0.f, 0.f, 0.f, 0.f};
    // progressive staging: quarter q = rows [64 q, +64) of either K half = chunks 4 q .. 4 q + 3; the pipe reads one chunk ahead (period n
    // takes chunks 2 n + 1, 2 n + 2), so quarter q + 1 is written in front of period 2 q + 1
    write(0);
    __syncthreads();
    mark(1);
    pipe.start_b();
    pipe.template run<1, 0, 0>(acc);
    write(1);
    __syncthreads();
    pipe.template run<2, 0, 0>(acc);
    write(2);
    __syncthreads();
    pipe.template run<2, 0, 0>(acc);
    write(3);
    __syncthreads();
    pipe.template run<2, 0, 0>(acc);
    pipe.template run<1, 0, 1>(acc);
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        const f32x4w m1 = acc[0][rb], m2 = acc[1][rb];
        acc[0][rb] = m1 + m2;
        acc[1][rb] = m1 - m2;
    }
    DSD_SB();
    mark(2);
    // the skip rows of the next contraction's B tile are requested half way through the second half
    pipe.template run<4, 1, 1>(acc);
    constexpr int NCH = 32;
    const int ch0 = NCH * wk;
    const TileB bofg{smem + ch0 * 8 * 32 + 4 * h * 32 + j, 8 * 32, NCH};
    GemmPipe<4, 1, 32, 256, 6, TileB> pipeg(pg.wotp + ((size_t)wr * 64 + ch0) * 256, lane, NCH, bofg);
    float4 vs[8];
    if constexpr (GATE) {
#pragma unroll
        for (int it = 0; it < 8; ++it) vs[it] = *reinterpret_cast<const float4*>(pg.dsk + ((size_t)b * kC + it * 32 + (tid >> 3)) * p.TS + st);
    }
    DSD_SB();
    pipe.template run<4, 1, 1>(acc);
    mark(3);
    f32x4w fin[2][4];
    trb_exchange_w(acc, fin, xbuf, wr, wk, lane);          // its barrier: every wave is done reading the da tile
    if constexpr (GATE) pipeg.start_a();
    // saved pre-activation of layer l - 1 at the rows this wave finishes there
    float4 av[4][4];
    if constexpr (GATE) {
        const float4* al = pg.a_frag + ((size_t)tile * 4 + 2 * wr + wk) * (4 * 4 * 64) + lane;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) av[mb][qq] = al[(mb * 4 + qq) * 64];
    }
    DSD_SB();
    // dy: lane (p, g) holds rows 128 wr + 64 wk + 16 r4 + 4 g + {0..3} of frames tE(p) (hf 0) and tE(p) + d (hf 1) -> the dy2 tile [256][32]
    {
        const int tE = wn_frame_of_pair(pp, de);
        float* dst = smem + (128 * wr + 64 * wk + 4 * gg) * 32 + tE;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(16 * r4 + e) * 32 + (hf ? dil : 0)] = fin[hf][r4][e];
    }
    __syncthreads();
    // conv gradient epilogue (k_trb_conv) by rows: dx = dx'/sqrt(2) + dy leaves as float4, dx/sqrt(2) replaces dy in the tile (the residual rows of
    // the next contraction), the row sums of dy are the step-projection gradient; the skip rows behind them
    {
        const bool m0 = st + 0 < p.T, m1 = st + 1 < p.T, m2 = st + 2 < p.T, m3 = st + 3 < p.T;
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(p.dx_out + (size_t)b * kC * p.TS, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 32 + (tid >> 3);
            float4* cell = reinterpret_cast<float4*>(smem + row * 32 + 4 * sg);
            const float4 dv = *cell, r4v = vx[it];
            // frames >= T are zero padding of y in the forward pass (net.py:69-71 pads the conv input): no gradient flows into them
            const float d0 = m0 ? dv.x : 0.f, d1 = m1 ? dv.y : 0.f, d2 = m2 ? dv.z : 0.f, d3 = m3 ? dv.w : 0.f;
            const f32x4_ dx = {m0 ? r4v.x * kTrInvSqrt2 + d0 : 0.f, m1 ? r4v.y * kTrInvSqrt2 + d1 : 0.f,
                               m2 ? r4v.z * kTrInvSqrt2 + d2 : 0.f, m3 ? r4v.w * kTrInvSqrt2 + d3 : 0.f};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, dx), rdx, (row * p.TS + st) * 4, 0, 16);     // aux 16 = sc1 (store4_wt)
            *cell = make_float4(dx[0] * kTrInvSqrt2, dx[1] * kTrInvSqrt2, dx[2] * kTrInvSqrt2, dx[3] * kTrInvSqrt2);
            float s = (d0 + d1) + (d2 + d3);
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            if (sg == 0) p.dds_part[(size_t)tile * kC + row] = s;
            if constexpr (GATE) {
                const float4 s4 = vs[it];
                *reinterpret_cast<float4*>(smem + (kC + row) * 32 + 4 * sg) = make_float4(m0 ? s4.x : 0.f, m1 ? s4.y : 0.f, m2 ? s4.z : 0.f, m3 ? s4.w : 0.f);
            }
        }
    }
    if constexpr (!GATE) return;
    __syncthreads();
    mark(4);
    // output-projection data gradient + gate derivative of layer l - 1 (k_trb_gate<false>)
    const int t = t0 + j;
    const bool ok = t < p.T;
    f32x16 accg[4][1];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accg[mb][0][r] = 0.f;
    pipeg.start_b();
    pipeg.run(accg, 0, NCH);
    mark(5);
    f32x16 fing[2];
    trb_exchange(accg, fing, xbuf, wr, wk, lane);
    mark(6);
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
    mark(7);
}

}  // namespace dsd
