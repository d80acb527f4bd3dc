// dsd_loop_wino.hpp - the persistent K-step loop of dsd_loop.hpp with the 3-tap dilated convolution (usr/diff/net.py:61,71: nn.Conv1d(C, 2C, 3,
// padding=dilation, dilation=dilation)) evaluated as WINOGRAD F(2,3) along the frame axis: fp32 in, fp32 out, exact-fp32 MFMA - the same dtype as
// the reference, 2/3 of the convolution's multiplications.
//
// For an output pair (t, t + d) of a layer with dilation d, inputs d0 = y[t-d], d1 = y[t], d2 = y[t+d], d3 = y[t+2d] and taps g0, g1, g2
// (out[t] = g0 y[t-d] + g1 y[t] + g2 y[t+d]):
//     M0 = g0 (d0 - d2)      M1 = (g0+g1+g2)/2 (d1 + d2)      M2 = (g0-g1+g2)/2 (d2 - d1)      M3 = g2 (d3 - d1)
//     out[t] = M0 + M1 + M2        out[t+d] = M1 - M2 + M3
// Four [512 x 256] . [256 x 16 pairs] products per 32-frame tile instead of three [512 x 256] . [256 x 32]: 16.8 M instead of 25.2 M FLOP per
// tile and layer.  The transformed weights U0..U3 are made once per model (fp64 sums, ONE rounding: k_pack_wino); the input transform is one
// add per B operand, made in registers from the frame-major y tile when the fragment is read; the output transform is register arithmetic
// between the two halves of the contraction.
//
// How it maps onto the loop (everything not named here IS k_loop: tile ownership, x / skip sum in registers, halo exchange, out-projection,
// head, sampler update, failure protocol):
//   * a 32-frame tile splits into 32 / (2 d) blocks of d pairs for every d in {1, 2, 4, 8}: always 16 pairs p = blk * d + i <-> frames
//     tE = 2 d blk + i and tO = tE + d.  The y tile is kept in PAIR order - E[p] = y[tE(p)], O[p] = y[tO(p)] - so that the four operands of
//     pair p are rows E[p], O[p], O[p - d] (= y[tE - d]) and E[p + d] (= y[tO + d]): consecutive lanes read consecutive rows, and the
//     neighbours' frames are rows O[-8 .. -1] (left halo) and E[16 .. 23] (right halo) - 24 + 24 rows of 260 floats, what k_loop's 48 rows take.
//   * N = 16 pairs: v_mfma_f32_16x16x4_f32 (32 cycles, the same 64 FLOP / clk / SIMD as the 32x32x2 form).  Lane (p = lane & 15, g = lane >> 4)
//     supplies B[k = g][pair p]; g <-> channels [64 g, 64 g + 64): one ds_read_b128 per operand row = the four MFMAs of a 16-channel chunk
//     (channels 64 g + 4 c + s), conflict-free for the b128 lane groups (MI355X_MICROARCH.md section LDS: row stride 65 slots, +16 g slots).
//     A wave owns the gate rows [64 w, 64 w + 64) and their filter rows as 8 row blocks of 16; D: lane (p, g), register r = row 4 g + r of the
//     block - gate and filter of a channel meet in one lane, four consecutive channels are one float4 of the gate tile.
//   * halo-free half first: M1 and M2 read only the tile's own frames, so the neighbours' frames travel under them (k_loop's "centre taps
//     first").  Then t = M1 + M2, u = M1 - M2 in place, and the second half accumulates M0 onto t and M3 onto u: 64 accumulator registers.
//   * the weight stream is 2 MiB per layer (k_loop: 1.5) at half the MFMA time per byte: 32 B / clk and CU.  It is kept in CONSUMPTION order
//     ([layer][step][wave]: a step = 16 MFMAs per wave = 4 KiB per wave) and the waves of an XCD fetch it into their L2 ahead of themselves
//     (L2Touch, dsd_loop_split.hpp); eight register stages of 4 KiB per wave.
//   * the hoisted conditioner projection is written by k_condproj as the INITIAL VALUES of the two accumulator sets, in this kernel's
//     accumulator order (CondProjParams::wino), and fetched into them while they are dead: under the previous layer's out-projection.
// Results differ from the direct form by reduction order and the transforms' roundings (tests/test_gpu_wino.py: within 2e-5 of k_loop on a
// K = 100 loop, the oracle parity budget of 1e-4 holds with a 10 x margin); k_loop stays the bit-identity anchor of the per-layer kernels.
#pragma once
#include "dsd_loop.hpp"
#include "dsd_loop_split.hpp"

namespace dsd {

typedef float f32x4w __attribute__((ext_vector_type(4)));

constexpr int kWnSteps = 128;                                   // steps of a layer's convolution: 2 halves x 16 chunks x 2 products x 2 row-block halves
constexpr int kWnStepBytes = 4 * 4096;                          // one step of all four waves
constexpr int kWnOBase = 24 * kFmLDK + 16;                      // O rows behind the 24 E rows, shifted by 4 slots (ds_write_b128 groups of 8 lanes stay conflict-free)
constexpr int kWnY = kWnOBase + 24 * kFmLDK;                    // floats of the pair-ordered y tile
constexpr int kLoopWinoLdsBytes = kLoopTouchLds + (kWnY + kFmG + kC * 32 + 2 * kC) * (int)sizeof(float);
static_assert(kWnY >= kC * 32, "the head reuses the y tile as [256][32]");
static_assert(kLoopWinoLdsBytes <= 160 * 1024, "LDS");

// frame j of a tile -> float offset of its row in the pair-ordered y tile, for dilation d = 1 << e
__host__ __device__ __forceinline__ int wn_row_of_frame(int j, int e) {
    const int d = 1 << e, hf = (j >> e) & 1, p = ((j >> (e + 1)) << e) | (j & (d - 1));
    return hf ? kWnOBase + (8 + p) * kFmLDK : p * kFmLDK;
}
// pair p -> its even frame tE (the odd frame is tE + d)
__host__ __device__ __forceinline__ int wn_frame_of_pair(int p, int e) {
    const int d = 1 << e;
    return ((p >> e) << (e + 1)) | (p & (d - 1));
}

// Transformed weights in consumption order: dst[l][step][w][r4][lane][s] (float), step = ((half * 16 + c) * 2 + pos) * 2 + hb; row block
// rb = 4 hb + r4: rows 64 w + 16 rb + n (gate, rb < 4) / C + 64 w + 16 (rb - 4) + n (filter), n = lane & 15; channel 64 g + 4 c + s, g = lane >> 4.
// half 0: U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2 (the halo-free products); half 1: U0 = g0, U3 = g2.  src = dilated_conv.weight [2C][C][3].
__global__ void k_pack_wino(const float* __restrict__ src, float* __restrict__ dst) {
    const size_t n = (size_t)kWnSteps * 4 * 4 * 64 * 4;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int s = idx & 3, lane = (idx >> 2) & 63, r4 = (idx >> 8) & 3, w = (idx >> 10) & 3, st = (int)(idx >> 12);
        const int hb = st & 1, pos = (st >> 1) & 1, c = (st >> 2) & 15, half = st >> 6;
        const int nn = lane & 15, g = lane >> 4, rb = 4 * hb + r4;
        const int row = (rb < 4) ? 64 * w + 16 * rb + nn : kC + 64 * w + 16 * (rb - 4) + nn;
        const int ch = 64 * g + 4 * c + s;
        const float* wp = src + ((size_t)row * kC + ch) * 3;
        const double g0 = wp[0], g1 = wp[1], g2 = wp[2];
        double u;
        if (half == 0) u = pos ? 0.5 * (g0 - g1 + g2) : 0.5 * (g0 + g1 + g2);
        else u = pos ? g2 : g0;
        dst[idx] = (float)u;
    }
}

// L2 touch of the transformed-weight stream by PERIODS (the mechanism of L2Touch, dsd_loop_split.hpp: one dword per 128-byte line through
// `buffer_load_dword ... lds` into a 256-byte scratch, no destination register).  A period = 8 steps of all four waves = 128 KiB = 16 pieces of
// 8 KiB; piece t of period n belongs to wave (16 n + t) mod nwx of the XCD's nwx waves.  A wave keeps r = (q - 16 n) mod nwx as a running
// counter - ONE scalar test per period (the per-step form cost six scalar instructions per 16 MFMAs, and every instruction beside a 32-cycle
// fp32 MFMA is paid for in matrix time: tools/mfma_filler_probe.hip) - and fetches the pieces t = r, r + nwx, ... < 16.
struct L2TouchP {
    L2Touch::i32x4_ rs;
    int r, nwx, dec;             // running turn, the XCD's waves, 16 mod nwx (off: r out of reach, dec 0)
    unsigned ahead, gtot;        // periods the touch runs in front; periods in the whole stream (16 per layer)
    unsigned lds, lane128;
    __device__ __forceinline__ void issue(unsigned soff) const {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" : : "s"(lds), "v"(lane128), "s"(rs), "s"(soff) : "m0");
#pragma clang diagnostic pop
    }
    // gper: index of the period being multiplied in the stream of all layers' periods
    __device__ __forceinline__ void period(unsigned gper) {
        for (int t = r; t < 16; t += nwx) {
            unsigned g = gper + ahead;
            if (g >= gtot) g -= gtot;
            issue(g * (unsigned)(8 * kWnStepBytes) + (unsigned)t * 8192u);
        }
        r -= dec;
        if (r < 0) r += nwx;
    }
};

// Operand pipeline of the Winograd contraction.  A: S register stages of one step (4 float4 = the four row blocks of a half) straight from
// global / L2, step k + S - 1 requested while step k is multiplied.  B: the raw operand rows of the NEXT 16-channel chunk are read at the first
// step of a group of four, transformed (one add each) at its last.  Everything that walks - the stream offset, the chunk pointers into the y
// tile, the touch's turn - is RUNNING state advanced once per period of eight steps, so that a step carries 16 MFMAs, 4 loads, their waits and
// one scalar add.  What the fillers cost beside a 32-cycle fp32 MFMA (tools/mfma_filler_probe.hip, profiles/r5_02_mfma_filler_probe.jsonl):
// a satisfied s_waitcnt nothing, a ds_read_b128 under one cycle, a VECTOR-ALU instruction 8 cycles of matrix time (+ 5 for the first one
// in a gap: the fp32 matrix pipe and the vector ALU do not overlap) - so the input transform's eight adds go into ONE gap of a group's last step.
// CM = 0: the B tile is FRAME-MAJOR in pair order (this file's loop).  CM = LD > 0: the B tile is CHANNEL-MAJOR [k row][LD frames] (the
// transposed convolution of the training backward, train_wino_bwd.hpp): lane (p, g) reads k rows 16 c + 4 s + g at frames tE(p) + {-d, 0, d, 2 d}
// as ds_read_b32 - the pointers are frame pointers of row g, a chunk is 16 rows further.
template <int S, int CM = 0>
struct WinoPipe {
    static_assert(S == 4 || S == 8, "the register rotation has period 8");
    static constexpr int kCS = CM ? 16 * CM : 4;        // floats between two chunks of the B tile
    __amdgpu_buffer_rsrc_t rsrc;    // over the whole stream behind this wave's 4 KiB of step 0 / layer 0
    unsigned vo[4], so;             // lane * 16 + 1024 r4 (kept in registers: rematerialised, they are four vector-ALU instructions per period beside
                                    // the MFMAs); byte offset of step 0 of the CURRENT period
    const float *qE, *qO;           // halo-free half: rows E[p], O[p] at the chunk the next read takes
    const float *rOm, *rO, *rEp, *rE;   // second half: rows O[p - d], O[p], E[p + d], E[p]
    L2TouchP& tc;
    unsigned gper;                  // index of the current period in the stream of all layers' periods
    float4 a[S][4];
    float4 raw[4];
    float v[2][2][4];               // [group parity][product][k step]

    __device__ __forceinline__ WinoPipe(const float4* wave_base, int lane, int l, const float* pE, const float* pO, int dilrow, L2TouchP& tc_)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(CM ? uniform_ptr(const_cast<float4*>(wave_base)) : const_cast<float4*>(wave_base), 0, 0x7ffffff0, 0x00020000)),      // (the persistent loops' descriptors are scalar as they are: their code stays the measured one)
          so((unsigned)l * (unsigned)(kWnSteps * kWnStepBytes)), qE(pE), qO(pO), rOm(pO - dilrow - 4), rO(pO - 4), rEp(pE + dilrow - 4), rE(pE - 4),
          tc(tc_), gper((unsigned)l * (unsigned)(kWnSteps / 8)) {
        if constexpr (CM != 0) {                            // pE: row g at frame tE, pO = pE + d, dilrow = d (frames)
            rOm = pE - dilrow - kCS; rO = pO - kCS; rEp = pO + dilrow - kCS; rE = pE - kCS;
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            vo[r4] = (unsigned)lane * 16u + (unsigned)r4 * 1024u;
            asm volatile("" : "+v"(vo[r4]));
        }
    }

    template <int KOFF>
    __device__ __forceinline__ void lda(float4 (&dst)[4]) {
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const int soff = (int)so + KOFF * kWnStepBytes;                  // past the layer's last step: the next layer's first ones (or the slack)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4_ f = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)vo[r4], soff, 0));
            dst[r4] = make_float4(f.x, f.y, f.z, f.w);
        }
    }
    // raw operand rows of the next chunk (+ O floats): half 0 (halo-free) E[p], O[p]; half 1 O[p - d], O[p], E[p + d], E[p]
    template <int HALF, int O>
    __device__ __forceinline__ void ldb_raw() {
        if constexpr (CM != 0) {
            constexpr int OC = (O / 4) * kCS;
            auto rd = [](const float* q) -> float4 { return make_float4(q[OC], q[OC + 4 * CM], q[OC + 8 * CM], q[OC + 12 * CM]); };
            if constexpr (HALF == 0) {
                raw[0] = rd(qE); raw[1] = rd(qO);
            } else {
                raw[0] = rd(rOm); raw[1] = rd(rO); raw[2] = rd(rEp); raw[3] = rd(rE);
            }
        } else if constexpr (HALF == 0) {
            raw[0] = *reinterpret_cast<const float4*>(qE + O);
            raw[1] = *reinterpret_cast<const float4*>(qO + O);
        } else {
            raw[0] = *reinterpret_cast<const float4*>(rOm + O);
            raw[1] = *reinterpret_cast<const float4*>(rO + O);
            raw[2] = *reinterpret_cast<const float4*>(rEp + O);
            raw[3] = *reinterpret_cast<const float4*>(rE + O);
        }
    }
    // one add per operand, as v_pk_add_f32 on pairs: a vector-ALU instruction beside the fp32 MFMAs costs 8 cycles of matrix time whether it is
    // packed or not (tools/mfma_filler_probe.hip) - four instructions per chunk and product pair instead of eight
    template <int HALF>
    __device__ __forceinline__ void transform(float (&dst)[2][4]) {
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        // (the differences as explicit v_pk_add_f32 with the second source negated: hipcc packs the sums and leaves half of the differences scalar)
        auto pk_sub = [](f32x2_ a, f32x2_ b) -> f32x2_ {
            f32x2_ d;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            return d;
        };
#pragma unroll
        for (int s = 0; s < 4; s += 2) {
            const f32x2_ r0 = {f4at(raw[0], s), f4at(raw[0], s + 1)}, r1 = {f4at(raw[1], s), f4at(raw[1], s + 1)};
            f32x2_ p0, p1;
            if constexpr (HALF == 0) {
                p0 = r0 + r1;                                            // d1 + d2
                p1 = pk_sub(r1, r0);                                     // d2 - d1
            } else {
                const f32x2_ r2 = {f4at(raw[2], s), f4at(raw[2], s + 1)}, r3 = {f4at(raw[3], s), f4at(raw[3], s + 1)};
                p0 = pk_sub(r0, r1);                                     // d0 - d2
                p1 = pk_sub(r2, r3);                                     // d3 - d1
            }
            dst[0][s] = p0[0]; dst[0][s + 1] = p0[1];
            dst[1][s] = p1[0]; dst[1][s + 1] = p1[1];
        }
    }
    template <int I, int NH>
    __device__ __forceinline__ void pattern() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if constexpr ((I & 3) == 0) {
#pragma unroll
            for (int i = 0; i < (NH ? 4 : 2); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, CM ? 4 : 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 12 - (NH ? 4 : 2), 0);
        } else if constexpr ((I & 3) == 3) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);          // the four packed adds of the next chunk's input transform: one gap
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
    }
    template <int N, int I = 0>
    __device__ __forceinline__ void start_a() {
        lda<I>(a[I]);
        if constexpr (I + 1 < N) start_a<N, I + 1>();
        else DSD_SB();
    }
    __device__ __forceinline__ void start_b() {
        ldb_raw<0, 0>();
        transform<0>(v[0]);
        qE += kCS; qO += kCS;                                           // the first group's read takes chunk 1
        DSD_SB();
    }
    // step I of a period: product pos = (I >> 1) & 1 of the group's chunk, row blocks 4 (I & 1) .. + 3; NH = half of the NEXT group
    template <int I, int NH>
    __device__ __forceinline__ void step(f32x4w (&acc)[2][8]) {
        constexpr int grp = I >> 2, pos = (I >> 1) & 1, hb = I & 1;
        lda<I + S - 1>(a[(I + S - 1) % S]);
        if constexpr ((I & 3) == 0) ldb_raw<NH, 4 * (I >> 2)>();
        if constexpr ((I & 3) == 3) transform<NH>(v[grp ^ 1]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                acc[pos][4 * hb + r4] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4at(a[I % S][r4], s), v[grp][pos][s], acc[pos][4 * hb + r4], 0, 0, 0);
        pattern<I, NH>();
        DSD_SB();
    }
    template <int NH0, int NH1>
    __device__ __forceinline__ void period(f32x4w (&acc)[2][8]) {
        tc.period(gper);
        step<0, NH0>(acc); step<1, NH0>(acc); step<2, NH0>(acc); step<3, NH0>(acc);
        step<4, NH1>(acc); step<5, NH1>(acc); step<6, NH1>(acc); step<7, NH1>(acc);
        ++gper;
        so += 8u * (unsigned)kWnStepBytes;
        if constexpr (NH0 == 0 || NH1 == 0) { qE += 2 * kCS; qO += 2 * kCS; }
        if constexpr (NH0 == 1 || NH1 == 1) { rOm += 2 * kCS; rO += 2 * kCS; rEp += 2 * kCS; rE += 2 * kCS; }
    }
    // N periods; NH0 / NH1: the half the group behind the first / second group of a period belongs to
    template <int N, int NH0, int NH1>
    __device__ __forceinline__ void run(f32x4w (&acc)[2][8]) {
#pragma nounroll
        for (int n = 0; n < N; ++n) period<NH0, NH1>(acc);
    }
};

struct LoopWinoParams {
    LoopParams lp;              // everything k_loop takes (w1p unused here; cp in THIS kernel's accumulator order)
    const float4* w1w;          // transformed conv weights of all layers, consumption order [L][128 steps][w4][r4][lane64]
    unsigned wl_bytes;          // bytes of that buffer (the L2 touch's buffer bound)
    int touch_ahead;            // steps the L2 touch runs in front (0 = off)
};

template <int MODE, int S>
__global__ __launch_bounds__(kThreads, 1) void k_loop_wino(const LoopWinoParams pw) {
    constexpr int LDK = kFmLDK;
    const LoopParams& p = pw.lp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem + kLoopTouchLds / 4;   // pair-ordered y tile: E rows [0, 24), O rows [-8, 16) at kWnOBase; head: scaled skip sum [256][32]
    float* gtile = ytile + kWnY;               // [32][260] gate tile, frame-major, natural frame order; head: relu(skip_projection) [256][32]
    float* xt = gtile + kFmG;                  // [256][32] scratch: spec tile of the in-projection
    float* dsbuf = xt + kC * 32;               // [2][256]  step projection of phase ph in dsbuf[ph & 1], fetched one phase ahead

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int pp = lane & 15, gg = lane >> 4;   // the 16x16x4 fragment's pair column and k group
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tl;
    {
        const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
        const int q = p.n_tiles >> 3, r = p.n_tiles & 7;
        tl = xcd * q + min(xcd, r) + k;
    }
    L2TouchP tc;
    {
        const unsigned long long wb = (unsigned long long)pw.w1w;
        const int xcd = (int)(blockIdx.x & 7), nwx = 4 * ((p.n_tiles - xcd + 7) >> 3), q = 4 * (int)(blockIdx.x >> 3) + w;
        const bool en = pw.touch_ahead > 0 && nwx >= 8;
        tc.rs = L2Touch::i32x4_{(int)(unsigned)wb, (int)(unsigned)((wb >> 32) & 0xffffu), (int)pw.wl_bytes, 0x00020000};
        tc.ahead = (unsigned)(pw.touch_ahead + 7) / 8u;         // the lead in periods
        tc.nwx = nwx;
        tc.dec = en ? 16 % nwx : 0;
        tc.r = en ? q : 1 << 20;                                 // period 0: piece t belongs to wave t mod nwx
        tc.gtot = (unsigned)p.L * (unsigned)(kWnSteps / 8);
        tc.lds = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float*)smem + (unsigned)w * 256u;
        tc.lane128 = (unsigned)lane * 128u;
    }
    const int tile = p.tile_base + tl;
    const int b = tile / p.ntile32, tn = tile - b * p.ntile32, t0 = tn * 32;
    const bool has_left = tn > 0, has_right = tn + 1 < p.ntile32;
    const int M = p.head.M, T = p.T;
    const bool in_t = t0 + j < T;           // this lane's frame is a frame of the utterance

    float4 xq[2][4];        // x tile in fragment order: xq[mb][q] = channels 64 w + 32 mb + 8 q + 4 h + {0,1,2,3} of frame j
    float4 skp[2][4];       // running skip sum of this wave's skip rows, the same order
    const int ch0 = 64 * w + 4 * h;         // channel of xq[0][0].x

    auto timed_out = [&]() -> bool { return __hip_atomic_load((gu32*)p.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; };

    // in-projection of the tile in xt (as [kMPad][32]) -> xq, through the (free) y tile region as [256][32]
    auto inproj_to_xq = [&]() {
        inproj_tile(xt, p.head.winp, p.head.binp, p.head.nk_in, ytile, w, lane);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* src = ytile + (ch0 + 32 * mb + 8 * q) * 32 + j;
                xq[mb][q] = make_float4(src[0], src[32], src[64], src[96]);
            }
        __syncthreads();    // every wave has its rows before the region becomes the y tile again
    };

    for (int idx = tid; idx < kMPad * 32; idx += kThreads) {
        const int m = idx >> 5, t = t0 + (idx & 31);
        xt[idx] = (m < M && t < T) ? p.spec0[((size_t)b * M + m) * T + t] : 0.f;
    }
    dsbuf[tid] = p.ds_table[(size_t)p.eval_t[0] * p.L * kC + tid];       // phase 0 = (evaluation 0, layer 0)
    __syncthreads();
    inproj_to_xq();

    // the halo protocol of k_loop: first / last 8 frames of x as write-through stores, every storing wave drained, barrier, ONE flag store
    auto publish_issue = [&](unsigned phase) {
        float* hb = p.halo + ((size_t)(phase & 1) * p.ntiles_total + tile) * (2 * kC * 8);
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(hb, 0, 0x7ffffff0, 0x00020000);
        if (j < 8 || j >= 24) {
            int oz;
            asm volatile("v_mov_b32 %0, 0" : "=v"(oz));        // keeps the offset arithmetic in this block (hoisted, it would live across every contraction)
            const int side = (j >= 24) ? 1 : 0, f = j & 7;
            const int vo = ((side * 8 + f) * kC + ch0) * 4 + oz;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_ v = {xq[mb][q].x, xq[mb][q].y, xq[mb][q].z, xq[mb][q].w};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r, vo + (32 * mb + 8 * q) * 4, 0, 16);
                }
        }
    };
    // The second half of a publication - every storing wave drained, barrier, ONE flag store - is NOT done where the stores are issued: it is
    // merged into the top of the next layer, whose barrier behind the y tile it shares (k_loop pays a drain of ~1 us and a barrier of its own
    // at the end of every layer: 2.6 k of its 147 k cycles; here the stores drain under the skip-sum update, the weight prefetch and the
    // staging of y).  The protocol is unchanged: the flag of phase ph is raised after the stores of all four waves are visible.
    const bool stamp = p.dbg != nullptr;
#define LOOP_STAMP(i) do { if (stamp && ph == (unsigned)p.dbg_phase && lane == 0) p.dbg[((size_t)tl * 4 + w) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define HEAD_STAMP(i) do { if (stamp && e == p.dbg_phase / p.L && lane == 0) p.dbg[((size_t)tl * 4 + w) * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

    // The two accumulator sets of the convolution live across the layers: between the gate of one layer and the contraction of the next they
    // are dead, and that window (the out-projection: 35 k cycles) is where the NEXT layer's conditioner projection is fetched straight into
    // them - k_condproj leaves it as the sets' initial values ((cp[tE] + cp[tO]) / 2 and (cp[tE] - cp[tO]) / 2, which the output transform
    // between the two halves turns into cp[tE] and cp[tO]): no registers for cp beside the accumulators, no adds in the gate.
    f32x4w acc[2][8];
    auto load_cp = [&](int l) {
        const float4* cpl = p.cp + (size_t)l * p.cp_lstride + ((size_t)tile * 4 + w) * (2 * 8 * 64);       // wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rb = 0; rb < 8; ++rb) {
                const float4 c = ld16_u(cpl, ((i * 8 + rb) * 64 + lane) * 16);
                acc[i][rb] = f32x4w{c.x, c.y, c.z, c.w};
            }
    };
    load_cp(0);

    unsigned ph = 0;
    publish_issue(0);
    for (int e = 0; e < p.n_evals; ++e) {
        const int t_e = p.eval_t[e];
        for (int l = 0; l < p.L; ++l, ++ph) {
            const bool last = (l == p.L - 1);
            const float* dsl = dsbuf + (ph & 1) * kC;
            LOOP_STAMP(0);
            const int dil = (int)p.dil[l], de = __builtin_ctz((unsigned)dil);

            // (c) the weight stream does not depend on anything computed here: request its first steps now
            WinoPipe<S> pipe1(pw.w1w + (size_t)w * 256, lane, l, ytile + pp * LDK + 64 * gg, ytile + kWnOBase + (8 + pp) * LDK + 64 * gg, dil * LDK, tc);
            pipe1.template start_a<S - 1>();

            // (b) own frames of y = x + step_proj (zero at frames >= T: the conv's zero padding applies to y, net.py:69-71): the lane's 32
            //     channels of frame j as 8 ds_write_b128 into the frame's row of the pair-ordered tile
            {
                float* yrow = ytile + wn_row_of_frame(j, de);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = ch0 + 32 * mb + 8 * q;
                        const float4 d = *reinterpret_cast<const float4*>(dsl + c);
                        *reinterpret_cast<float4*>(yrow + c) = fm_add_masked(xq[mb][q], d, in_t);
                    }
            }
            // (a) this tile's halo frames of phase ph (stored at the end of the previous phase / behind the head's input projection) are
            //     visible once every wave has drained; the barrier is the one the y tile needs anyway
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store((gu32*)(p.flags + tile), ph + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            LOOP_STAMP(1);
            // (d1) every wave reads the two neighbour flags now (lanes 0 / 1), tested behind the first period
            unsigned fv = 0xffffffffu;
            if (lane < 2) {
                const bool have = lane ? has_right : has_left;
                if (have) fv = __hip_atomic_load((const gu32*)(p.flags + tile + (lane ? 1 : -1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            DSD_SB();

            // (g) first half: M1 (acc[0]) and M2 (acc[1]) - on top of the conditioner projection's halves they were loaded with - read the tile's
            //     own frames only: the exchange with the neighbours runs under them
            pipe1.start_b();
            pipe1.template run<1, 0, 0>(acc);
            // (d2) both neighbours have published phase ph?  Lanes whose early read was too early poll (bounded, sticky timeout)
            if (fv < ph + 1u) {
                const gu32* f = (const gu32*)(p.flags + tile + (lane ? 1 : -1));
                for (int spins = 0;; ++spins) {
                    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ph + 1u) break;
                    if ((spins & 255) == 255 && timed_out()) break;
                    if (spins >= kLoopSpinLimit) { __hip_atomic_store((gu32*)p.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            // (e1) request the neighbours' frames: 8 frames x 256 channels per side = 512 float4, two per thread (sc1 loads)
            float4 hv[2][2];
            {
                const float* hbase = p.halo + (size_t)(ph & 1) * p.ntiles_total * (2 * kC * 8);
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const bool have = side ? has_right : has_left;
                    // my left halo = left neighbour's LAST 8 frames (its side 1); my right halo = right neighbour's first 8 (side 0)
                    const int off = (((tile + (side ? 1 : -1)) * 2 + (side ? 0 : 1)) * (8 * kC) + 4 * tid) * 4;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        hv[side][g] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (have) hv[side][g] = ld16_sc1(hbase, off + g * (4 * kC * 4));       // float4 index tid + 256 g: frame 4 g + tid / 64
                    }
                }
            }
            DSD_SB();
            pipe1.template run<4, 0, 0>(acc);
            // (e2) halo rows: left frame f (t = t0 - 8 + f) is O[f - 8], right frame f (t = t0 + 32 + f) is E[16 + f]; float4 index
            //      tid + 256 g = (frame f = 4 g + tid / 64, channels 4 (tid % 64) ..)
            {
                const int c = 4 * (tid & 63);
                const float4 d = *reinterpret_cast<const float4*>(dsl + c);
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const bool have = side ? has_right : has_left;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int f = 4 * g + (tid >> 6);
                        const int t = side ? t0 + 32 + f : t0 - kHalo + f;
                        float* dst = side ? ytile + (16 + f) * LDK + c : ytile + kWnOBase + f * LDK + c;
                        *reinterpret_cast<float4*>(dst) = fm_add_masked(hv[side][g], d, have && t < T);
                    }
                }
            }
            __syncthreads();
            LOOP_STAMP(2);
            pipe1.template run<2, 0, 0>(acc);
            pipe1.template run<1, 0, 1>(acc);
            // output transform, first part: t = M1 + M2 (frame tE), u = M1 - M2 (frame tO); the second half accumulates M0 onto t, M3 onto u
#pragma unroll
            for (int rb = 0; rb < 8; ++rb) {
                const f32x4w m1 = acc[0][rb], m2 = acc[1][rb];
                acc[0][rb] = m1 + m2;
                acc[1][rb] = m1 - m2;
            }
            DSD_SB();
            pipe1.template run<8, 1, 1>(acc);
            // step projection of the NEXT phase (next layer, or layer 0 of the next evaluation)
            float ds_next = 0.f;
            {
                const bool more = !last || (e + 1 < p.n_evals);
                const int tn_ = last ? p.eval_t[min(e + 1, p.n_evals - 1)] : t_e, ln_ = last ? 0 : l + 1;
                if (more) ds_next = p.ds_table[((size_t)tn_ * p.L + ln_) * kC + tid];
            }

            const TileBT bof2{gtile + j * LDK + 4 * h, 32};
            // gate (net.py:73-74) in registers -> frame-major gate tile: lane (p, g) holds channels 64 w + 16 rb + 4 g + {0..3} of frames tE, tE + d
            auto do_gate = [&]() {
                const int tE = wn_frame_of_pair(pp, de);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float* grow = gtile + (tE + (hf ? dil : 0)) * LDK + 64 * w + 4 * gg;
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        float g4[4];
#pragma unroll
                        for (int ee = 0; ee < 4; ++ee) g4[ee] = sigmoid_f(acc[hf][rb][ee]) * tanh_f(acc[hf][rb + 4][ee]);
                        *reinterpret_cast<float4*>(grow + 16 * rb) = make_float4(g4[0], g4[1], g4[2], g4[3]);
                    }
                }
            };
            LOOP_STAMP(3);
            if (!last) {
                // output projection, all four row blocks (0,1 residual, 2,3 skip) in one pass
                GemmPipe<4, 1, LDK, 256, 6, TileBT, 1, true> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256), lane, 32, bof2);
                pipe2.start_a();
                do_gate();
                // the next layer's conditioner projection into the (dead) accumulators, under the out-projection.  (Issued HERE, in front of
                // the barrier: beside the out-projection's MFMAs the 16 cold loads cost 2.5 k cycles of in-order waits, profiles/r5_07)
                load_cp(l + 1);
                dsbuf[((ph + 1u) & 1u) * kC + tid] = ds_next;       // visible behind the barrier (that half was last read in phase ph - 1)
                __syncthreads();
                LOOP_STAMP(4);
                f32x16 acc2[4][1];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[m][0][r] = 0.f;
                float4 bq[2][4];            // residual-half bias of this lane's channels
                pipe2.start_b();
                pipe2.run(acc2, 0, 6);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[mb][q] = *reinterpret_cast<const float4*>(p.b2raw + (size_t)l * 2 * kC + ch0 + 32 * mb + 8 * q);
                DSD_SB();
                pipe2.run(acc2, 6, 32);
                LOOP_STAMP(5);
                // residual in place: x' = (x + res + b) / sqrt(2) - the accumulators hold exactly the elements of xq
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = get4(acc2[mb][0], q), x = xq[mb][q], bv = bq[mb][q];
                        constexpr float kInvSqrt2 = 1.0f / 1.41421354f;
                        xq[mb][q] = make_float4((x.x + (v.x + bv.x)) * kInvSqrt2, (x.y + (v.y + bv.y)) * kInvSqrt2,
                                                (x.z + (v.z + bv.z)) * kInvSqrt2, (x.w + (v.w + bv.w)) * kInvSqrt2);
                    }
                LOOP_STAMP(6);
                publish_issue(ph + 1u);                             // the halo stores drain under the skip sum, the next prefetch and y tile
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = get4(acc2[2 + ms][0], q), s = skp[ms][q];
                        skp[ms][q] = (l == 0) ? a : make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                    }
                LOOP_STAMP(7);
            } else {
                // last layer: only the skip half (net.py:126 reads the skips; the residual is dead)
                GemmPipe<2, 1, LDK, 256, 6, TileBT, 1, true> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256) + 2 * 64, lane, 32, bof2);
                pipe2.start_a();
                do_gate();
                dsbuf[((ph + 1u) & 1u) * kC + tid] = ds_next;       // visible behind the barrier
                __syncthreads();
                f32x16 acc2[2][1];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[m][0][r] = 0.f;
                pipe2.start_b();
                pipe2.run(acc2, 0, 32);
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = get4(acc2[ms][0], q), s = skp[ms][q];
                        skp[ms][q] = (l == 0) ? a : make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                    }
            }
        }

        // ---- head (net.py:126-129) + sampler epilogue for this tile, then the next evaluation's input projection: k_loop's code -----------
        HeadParams hp = p.evals[e];
        const bool fuse = (e + 1 < p.n_evals);
        float* stile = ytile;               // [256][32]
        float* htile = gtile;               // [256][32]
        float* ptile = xt;                  // [96][32]
        HEAD_STAMP(0);
        __syncthreads();                    // all waves are out of the last layer's out-proj (gate tile reads)
        const float* sl = stile + 4 * h * 32 + j;
        GemmPipe<2, 1, 32, 128, 6, TileB> pipe_s(p.head.wsp + (size_t)w * (32 * 128), lane, 32, TileB{sl, 8 * 32, 32});
        pipe_s.start_a();
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 s = skp[ms][q], bs = p.head.bskp[((w * 2 + ms) * 2 + h) * 4 + q];
                const float v[4] = {s.x + bs.x, s.y + bs.y, s.z + bs.z, s.w + bs.w};
#pragma unroll
                for (int ee = 0; ee < 4; ++ee)
                    stile[(64 * w + 32 * ms + frag_row(4 * q + ee, h)) * 32 + j] = __fdiv_rn(v[ee], p.head.sqrt_L);
            }
        __syncthreads();
        HEAD_STAMP(1);
        {
            f32x16 acc[2][1];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) set4(acc[mb][0], q, p.head.bsp[((w * 2 + mb) * 2 + h) * 4 + q]);
            pipe_s.start_b();
            pipe_s.run(acc, 0, 32);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    htile[(64 * w + 32 * mb + frag_row(r, h)) * 32 + j] = fmaxf(acc[mb][0][r], 0.f);
        }
        HEAD_STAMP(2);
        const float* hl = htile + 4 * h * 32 + j;
        GemmPipe<1, 1, 32, 192, 6, TileB> pipe_o(p.head.woutp + (size_t)min(w, 2) * 64, lane, 32, TileB{hl, 8 * 32, 32});
        if (w < 3) pipe_o.start_a();
        __syncthreads();
        HEAD_STAMP(3);
        if (w < 3) {
            f32x16 acc[1][1];
#pragma unroll
            for (int q = 0; q < 4; ++q) set4(acc[0][0], q, p.head.boutp[(w * 2 + h) * 4 + q]);
            pipe_o.start_b();
            pipe_o.run(acc, 0, 32);
            HEAD_STAMP(4);
            // (an opaque zero defined HERE keeps the 16 element indices - and the Philox products that hang on them - in this block: as loop
            // invariants of the evaluation loop they would live, spilled to scratch, across every contraction of the kernel)
            int oz;
            asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
            const int t = t0 + j + oz;
            // sampler arithmetic (p_sample :134-166 / p_sample_plms :168-204): all global reads of the 16 elements first, then the math, then the stores
            size_t idxs[16];
            bool oks[16];
            float xv[16], av[16], bv[16], cv[16];
            const float* nz = nullptr;
            unsigned long long seed = 0;
            if (MODE == HEAD_DDPM) {
                nz = *hp.noise_cell;
                if (nz) nz += hp.noise_off; else seed = *hp.seed_cell;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * w + frag_row(r, h);
                oks[r] = (m < M) && (t < T);
                idxs[r] = oks[r] ? ((size_t)b * M + m) * T + t : 0;
                xv[r] = hp.x_base[idxs[r]];
                av[r] = bv[r] = cv[r] = 0.f;
                if (MODE == HEAD_DDPM) {
                    av[r] = nz ? nz[idxs[r]] : philox_normal(seed, hp.step_id, idxs[r]);
                } else {
                    if (hp.order >= PLMS_HEUN) av[r] = hp.e1[idxs[r]];
                    if (hp.order >= PLMS_AB3) bv[r] = hp.e2[idxs[r]];
                    if (hp.order >= PLMS_AB4) cv[r] = hp.e3[idxs[r]];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * w + frag_row(r, h);
                const bool ok = oks[r];
                const size_t idx = idxs[r];
                const float eps = acc[0][0][r];
                const float x = xv[r];
                float xn;
                if (MODE == HEAD_DDPM) {
                    float x0 = __fsub_rn(__fmul_rn(hp.sa, x), __fmul_rn(hp.sb, eps));
                    x0 = fminf(fmaxf(x0, -1.f), 1.f);
                    const float mean = __fadd_rn(__fmul_rn(hp.c1, x0), __fmul_rn(hp.c2, x));
                    xn = __fadd_rn(mean, __fmul_rn(hp.sigma, av[r]));
                } else {
                    float ep;
                    if (hp.order == PLMS_RAW) {
                        ep = eps;
                    } else if (hp.order == PLMS_HEUN) {
                        ep = __fmul_rn(__fadd_rn(av[r], eps), 0.5f);
                    } else if (hp.order == PLMS_AB2) {
                        ep = __fmul_rn(__fsub_rn(__fmul_rn(3.f, eps), av[r]), 0.5f);
                    } else if (hp.order == PLMS_AB3) {
                        ep = __fdiv_rn(__fadd_rn(__fsub_rn(__fmul_rn(23.f, eps), __fmul_rn(16.f, av[r])), __fmul_rn(5.f, bv[r])), 12.f);
                    } else {
                        ep = __fdiv_rn(__fsub_rn(__fadd_rn(__fsub_rn(__fmul_rn(55.f, eps), __fmul_rn(59.f, av[r])),
                                                           __fmul_rn(37.f, bv[r])), __fmul_rn(9.f, cv[r])), 24.f);
                    }
                    if (ok && hp.eps_out) hp.eps_out[idx] = eps;
                    const float delta = __fmul_rn(hp.dA, __fsub_rn(__fmul_rn(hp.cx, x), __fmul_rn(hp.ce, ep)));
                    xn = __fadd_rn(x, delta);
                }
                if (ok) hp.x_out[idx] = xn;
                ptile[m * 32 + j] = ok ? xn : 0.f;
            }
        }
        HEAD_STAMP(5);
        __syncthreads();
        HEAD_STAMP(6);
        if (fuse) { inproj_to_xq(); publish_issue(ph); load_cp(0); }     // (layer 0's conditioner projection LAST: live across the in-projection, the 64 registers cost 35 spills in the sampler update)     // (layer 0's conditioner projection: under the input projection)
        HEAD_STAMP(7);
    }
#undef LOOP_STAMP
#undef HEAD_STAMP
    // a wait that hit its spin bound leaves garbage: make it LOUD - poison this tile of the result with NaN
    if (timed_out()) {
        float* xo = const_cast<float*>(p.spec0);
        for (int idx = tid; idx < M * 32; idx += kThreads) {
            const int m = idx >> 5, t = t0 + (idx & 31);
            if (t < T) xo[((size_t)b * M + m) * T + t] = __builtin_nanf("");
        }
    }
}

}  // namespace dsd
