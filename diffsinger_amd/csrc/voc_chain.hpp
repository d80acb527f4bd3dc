// voc_chain.hpp - gfx950: whole ResBlock1 chains of the HiFi-GAN generator in ONE kernel, activations resident in LDS.
//
// What is computed, and where the reference computes it (paths relative to the reference root): for nres parallel resblocks r and npairs conv
// pairs q per resblock (modules/hifigan/hifigan.py:54-61 inside :161-166),
//     y = x
//     for q:  xt = convs1[q](leaky_relu(y)) ; xt = convs2[q](leaky_relu(xt)) ; y = xt + y              ResBlock1.forward
//     xs = y_0 ; xs = xs + y_1 ; ... ; out = (sum_in + xs) / divide                                   `xs += resblock(x)`, `x = xs / num_kernels`
// k_voc_conv / k_voc_conv_fold run every convolution as its own launch: a stage of the shipped generator reads and writes its activation
// ~40 times (round 2: 0.36 of the HBM roof on the narrow stages).  Here a workgroup owns N output samples of all C channels of one utterance
// plus a halo of Hh samples per side (the receptive field of the chain), stages leaky_relu(x) once, and runs the convolutions back to back:
//   * two LDS tiles [C][LD]: A = leaky_relu(y) (input of convs1), T = leaky_relu(xt) (input of convs2); the raw y lives in REGISTERS in the
//     accumulator-fragment order of convs2 (always dilation 1), so the residual add is register arithmetic; the running sum over the parallel
//     resblocks likewise.  Per stage the activation is read nres times and written once.
//   * C * F == 32: the F-fold of k_voc_conv_fold for the narrow stages (row = co * F + e computes output sample pos(c) + e * dil of channel co,
//     pos(c) = (c / dil) * F * dil + c % dil for column c; the A operand holds the F shifted copies of the filter, K + F - 1 taps) - F = 4 for
//     8 channels, 2 for 16, 1 (no fold) for 32.  All 32 MFMA rows carry work.
//   * values outside [0, L) are written to the tiles as ZERO (the convolutions' zero padding applies to every intermediate activation);
//     columns whose samples fall outside the range a later convolution needs compute garbage that is never read for a needed sample: the
//     host picks N so that N + 2 Hh <= the samples EVERY convolution of the chain covers (min over dilations of floor(NCOL / dil) * F * dil).
// Arithmetic and summation order are those of the one-convolution kernels (chunk order ci8 * KT + tap, bias, + residual, sum_in + v, / divide):
// results are BIT-IDENTICAL to the unfused path (tests/test_gpu_vocoder.py).
#pragma once
#include <type_traits>

#include "voc_kernels.hpp"

namespace dsd {

constexpr int kChainMaxConvs = 18;         // 3 resblocks x 3 pairs x 2
constexpr int kChainSlack = 32;            // zero / scratch columns on both sides of a tile row (>= the largest pad, kVocHalo, rounded to 32)

struct VocChainConv {
    int woff;               // float4 index of this convolution's packed weight [chunk = ci8 * KT + s][lane64] inside wp
    int boff;               // float index of its bias [C] inside bias
    int KT;                 // folded taps K + F - 1
    int dil, pad;           // pad = (K - 1) * dil / 2
};

constexpr int kChainMaxGroups = 3;
struct VocChainGroup {
    float* out;             // [B][C][LS]: this group's result
    int first, tiles;       // first block of the group, tiles per utterance (ceil(LS / N))
    int N, Hh;              // its chain's output samples per workgroup and halo
    int final;              // 0: out = the raw y of the chain; 1: out = the summed, divided, masked result (sum_in / sum_in2 / own_last / divide)
};

struct VocChainParams {
    const float* in;        // [B][C][LS]
    float* out;             // [B][C][LS]
    const float* sum_in;    // [B][C][LS] or nullptr
    const float4* wp;
    const float* bias;
    int L, LS;
    int N, Hh;              // output samples per workgroup (multiple of 32), halo per side (multiple of 4)
    int nres, npairs;
    float slope, divide;
    unsigned long long* dbg;                // optional s_memtime stamps of ONE workgroup (dsv_debug_chain_timeline): [conv][wave 4][4]
    VocChainConv conv[kChainMaxConvs];      // [res][pair][2]
    // MG instantiations only (round 6: several INDEPENDENT single-resblock chains in one launch, and the launch that sums them)
    const float* sum_in2;   // a second summand of the final result: ((sum_in + y) + sum_in2) / divide, or ((sum_in + sum_in2) + y) / divide (own_last)
    int own_last;
    int ngroups;            // 1 .. kChainMaxGroups; the grid is ONE-dimensional: group g owns the blocks [grp[g].first, grp[g].first + B * grp[g].tiles)
                            // and the convolutions conv[2 npairs g ...]
    VocChainGroup grp[3];
};

template <int F, int NB> constexpr int chain_wpos() { return 128 * NB * F; }                     // samples a dilation-1 convolution covers
template <int F, int NB> constexpr int chain_ld() { return chain_wpos<F, NB>() + 2 * kChainSlack; }
// IP = ONE tile, rewritten IN PLACE: a convolution's epilogue writes leaky_relu(result) over the tile the contraction just read (one more
// barrier per convolution - behind the last MFMA, where the waves arrive together anyway).  Half the LDS: twice the workgroups per CU, or
// twice the window (the receptive-field overlap of a 512-sample window is 1.33 x the useful FLOPs, of a 1024-sample window 1.13 x).
template <int C, int F, int NB, bool IP> constexpr int chain_lds_bytes() { return ((IP ? 1 : 2) * C * chain_ld<F, NB>() + 256) * (int)sizeof(float); }
template <int C, int F, int NB, bool IP> constexpr int chain_wg_per_cu() {
    return (chain_lds_bytes<C, F, NB, IP>() <= 40 * 1024 && NB <= 2) ? 3 : (chain_lds_bytes<C, F, NB, IP>() <= 80 * 1024) ? 2 : 1;
}

// F consecutive floats (F = 1, 2, 4) of the row at byte offset `soff` (wave-uniform: an SGPR) behind a buffer descriptor, lane offset `voff` bytes
template <int F>
__device__ __forceinline__ void chain_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff, float (&d)[F]) {
    if constexpr (F == 1) {
        d[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    } else if constexpr (F == 2) {
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        const f32x2_ f = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
        d[0] = f.x; d[1] = f.y;
    } else {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const f32x4_ f = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
        d[0] = f.x; d[1] = f.y; d[2] = f.z; d[3] = f.w;
    }
}
template <int F>
__device__ __forceinline__ void chain_st(__amdgpu_buffer_rsrc_t r, int voff, int soff, const float (&d)[F]) {
    if constexpr (F == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d[0]), r, voff, soff, 0);
    } else if constexpr (F == 2) {
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        const f32x2_ f = {d[0], d[1]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, f), r, voff, soff, 0);
    } else {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const f32x4_ f = {d[0], d[1], d[2], d[3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, f), r, voff, soff, 0);
    }
}

// K loop of one convolution: A fragments stream from global / L2 (6 register stages), B from the LDS tile with a lane-specific offset per
// column block (pos(c) is not linear in the column once dil > 1) and a RUNNING chunk pointer (tap + 1, wrap to the next 8-channel group).
// CONSTB (F == 1: pos(c) = c for every dilation): the column blocks of a wave sit 32 floats apart - compile-time ds_read offsets instead of
// NB lane-offset additions per chunk (0.4 vector-ALU instructions per MFMA in rounds 3-5, profiles/r5_32_isa_scan.txt).
template <int NB, int LD, bool CONSTB>
struct ChainPipe {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned aoff;
    const float* cur;
    int KT, dil, left, tap;
    int boff[NB];
    float4 a[6][1];
    float b[2][4][NB];

    __device__ __forceinline__ ChainPipe(const float4* abase_uniform, int lane)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(abase_uniform), 0, 0x7ffffff0, 0x00020000)),
          aoff((unsigned)lane * 16u), cur(nullptr), KT(1), dil(1), left(0), tap(0) {}
    // The weight stream of a convolution depends on nothing the kernel computes: it is re-targeted and its first five chunks are requested
    // BEHIND the previous convolution's last MFMA - in front of that convolution's epilogue and the barrier - so that the L2 round trip of a
    // stream's first touch is not paid at the head of every (short: 6 - 56 chunks) contraction.
    __device__ __forceinline__ void set_a(const float4* abase_uniform) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(abase_uniform), 0, 0x7ffffff0, 0x00020000);
    }
    __device__ __forceinline__ void set_b(int n, const float* bbase, int KT_, int dil_, const int (&boff_)[NB]) {
        cur = bbase; KT = KT_; dil = dil_; left = n - 1; tap = 0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) boff[nb] = boff_[nb];
    }
    __device__ __forceinline__ void lda(float4 (&dst)[1], int kc) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)aoff, kc * (64 * 16), 0);
        const f32x4 f = __builtin_bit_cast(f32x4, v);
        dst[0] = make_float4(f.x, f.y, f.z, f.w);
    }
    __device__ __forceinline__ void ldb(float (&dst)[4][NB]) {
        const float* bp = cur;
        const bool adv = left > 0, wrap = (tap + 1 == KT);
        const int step = wrap ? 8 * LD - (KT - 1) * dil : dil;
        cur += adv ? step : 0;
        tap = adv ? (wrap ? 0 : tap + 1) : tap;
        left -= adv ? 1 : 0;
        // VOLATILE LDS loads (round 6): left alone, hipcc pairs the sixteen ds_read_b32 of a chunk into ds_read2_b32, whose two 8-bit offsets
        // do not reach across a tile row - one v_add_u32 per row for a new base, 4-5 vector-ALU instructions per chunk, each paid in matrix
        // time (tools/mfma_filler_probe.hip: 8 cycles beside an fp32 MFMA).  A volatile access is not merged: ds_read_b32 with the row and
        // block offsets in its own 16-bit offset field, ONE vector instruction per chunk (the advance of the running pointer).
        typedef const volatile __attribute__((address_space(3))) float lds_cvf;
        lds_cvf* vp = (lds_cvf*)bp;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) dst[s][nb] = CONSTB ? vp[s * LD + 32 * nb] : vp[s * LD + boff[nb]];
    }
    __device__ __forceinline__ void pattern() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) {                      // (4 NB single ds_read_b32 per chunk: two behind each of 2 NB MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB - 1 - 2 * NB, 0);
    }
    __device__ __forceinline__ void start_a() {
#pragma unroll
        for (int i = 0; i < 5; ++i) lda(a[i], i);
        DSD_SB();
    }
    __device__ __forceinline__ void start_b() {
        ldb(b[0]);
        DSD_SB();
    }
    template <int I>
    __device__ __forceinline__ void step(f32x16 (&acc)[1][NB], int it) {
        lda(a[(I + 5) % 6], 6 * it + I + 5);
        ldb(b[(I + 1) & 1]);
        mma_chunk<1, NB>(acc, a[I % 6], b[I & 1]);
        pattern();
        DSD_SB();
    }
    // whole groups of six chunks as ONE basic block, the tail behind it (GemmPipe::run_blocks)
    __device__ __forceinline__ void run_blocks(f32x16 (&acc)[1][NB], int end) {
        int it = 0;
        for (; 6 * it + 6 <= end; ++it) {
            step<0>(acc, it); step<1>(acc, it); step<2>(acc, it); step<3>(acc, it); step<4>(acc, it); step<5>(acc, it);
        }
        const int kc = 6 * it;
        if (kc >= end) return;
        step<0>(acc, it);
        if (kc + 1 >= end) return;
        step<1>(acc, it);
        if (kc + 2 >= end) return;
        step<2>(acc, it);
        if (kc + 3 >= end) return;
        step<3>(acc, it);
        if (kc + 4 >= end) return;
        step<4>(acc, it);
    }
};

// grid (ceil(LS / N), B); 4 waves, wave w owns the columns [32 NB w, 32 NB (w + 1)) of every convolution
// MG (round 6): the launch holds several INDEPENDENT single-resblock chains (p.grp[]: each with its own tile size, halo and output buffer) - the
// workgroups of the first group are dispatched first.  A launch of W workgroups on S co-resident slots costs ceil(W / S) rounds
// (profiles/r6_27_voc_tail_probe.jsonl: a staircase), so three dependent launches pay three partial rounds; with the longest chain first and a
// short one behind it in the same grid the short workgroups fill the long ones' last round.  No workgroup waits for another: the groups write
// separate buffers and the sum is formed by the NEXT launch (sum_in / sum_in2) behind an ordinary kernel boundary.
template <int C, int F, int NB, bool IP, bool MG = false>
__global__ __launch_bounds__(kThreads, (chain_wg_per_cu<C, F, NB, IP>())) void k_voc_chain(const VocChainParams p) {
    static_assert(C * F == 32 && (C % 8) == 0, "one 32-row MFMA block: 8 channels x 4, 16 x 2 or 32 x 1");
    constexpr int LD = chain_ld<F, NB>(), SLK = kChainSlack, NCOL4 = LD / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufA = smem;                     // [C][LD] leaky_relu(y): input of convs1
    float* bufT = IP ? smem : smem + C * LD;   // [C][LD] leaky_relu(xt): input of convs2 (IP: the same tile, rewritten in place)
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t0, b, ws, gidx = 0;                // ws: sample of tile column SLK
    if constexpr (MG) {
        const int bid = blockIdx.x;
        if (p.ngroups > 1 && bid >= p.grp[1].first) gidx = 1;
        if (p.ngroups > 2 && bid >= p.grp[2].first) gidx = 2;
        const int local = bid - p.grp[gidx].first, tiles = p.grp[gidx].tiles;
        b = __builtin_amdgcn_readfirstlane(local / tiles);         // (the division runs on the vector ALU: without this every buffer descriptor
                                                                    // derived from b is a waterfall loop - tests/test_verified_isa.py)
        t0 = (local - b * tiles) * p.grp[gidx].N;
        ws = t0 - p.grp[gidx].Hh;
    } else {
        t0 = blockIdx.x * p.N; b = blockIdx.y;
        ws = t0 - p.Hh;
    }
    const int cw = w * (32 * NB);
    const int L = p.L, LS = p.LS;
    const float slope = p.slope;
    const float* inb = p.in + (size_t)b * C * LS;

    // the scratch columns of T are read (for samples nobody needs) before anything wrote them: make them finite once
    // (IP: every column of the one tile is staged from x; only the 256 floats behind it are cleared)
    for (int idx = tid + (IP ? C * LD / 4 : 0); idx < (C * LD + 256) / 4; idx += kThreads) *reinterpret_cast<float4*>(bufT + 4 * idx) = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x16 y[NB];
    // The running sum over the parallel resblocks lives in `out` itself (round 6; rounds 3-5 kept it in 16 NB registers): the workgroup owns the
    // samples [t0, t0 + N) of every channel, writes y_0 there, then y_0 + y_1, then (sum_in + (y_0 + y_1 + y_2)) / divide - the additions of the
    // one-convolution path in their order; the same lanes store and reload the same addresses (coherent inside a CU).  What it buys: the
    // 64 registers that stood between the 32-channel kernel and two workgroups per CU.
    // Register quad q of a column = rows 8 q + 4 h + (0..3) of the 32-row block = 4 / F channels x F consecutive samples from t = ws + c F:
    // group g = 4 q / F + i of the lane holds channel (8 / F) q + i + (4 / F) h - the half-wave term goes into the lane's byte offset, the rest
    // is a wave-uniform row offset; t0, N, Hh are multiples of 4 and LS of 32, so the F samples of a group are inside a range together.
    const int tend = min(t0 + p.N, LS);       // (MG: flush_mg has its own)
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((MG ? const_cast<float*>(p.in) : p.out) + (size_t)b * C * LS, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_sin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.sum_in ? p.sum_in + (size_t)b * C * LS : p.in), 0, 0x7ffffff0, 0x00020000);
    const bool have_sin = p.sum_in != nullptr;
    constexpr int NG = 16 / F;               // groups of F registers per column
    auto flush = [&](bool first, bool last) {
        // (nothing below depends on the convolution index: without the opaque zero hipcc computes the addresses once, in front of the loop,
        // and keeps them in registers across every contraction)
        int oz = 0;
        asm volatile("" : "+v"(oz));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int t = ws + (cw + 32 * nb + j + oz) * F;
            const bool ok = t >= t0 && t < tend;
            const int voff = ((4 / F) * h * LS + (ok ? t : t0)) * 4;
            float pv[NG][F], sv[NG][F];
#pragma unroll
            for (int g = 0; g < NG; ++g) {                       // all the reads first
                const int soff = ((8 / F) * (g * F / 4) + (g * F % 4) / F) * LS * 4;
                if (!first) chain_ld<F>(r_out, voff, soff, pv[g]);
                if (last && have_sin) chain_ld<F>(r_sin, voff, soff, sv[g]);
            }
            if (!ok) continue;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int soff = ((8 / F) * (g * F / 4) + (g * F % 4) / F) * LS * 4;
                float v[F];
#pragma unroll
                for (int e = 0; e < F; ++e) {
                    v[e] = y[nb][g * F + e];
                    if (!first) v[e] = pv[g][e] + v[e];
                    if (last) {
                        if (have_sin) v[e] = sv[g][e] + v[e];
                        if (p.divide != 1.f) v[e] = v[e] / p.divide;
                        if (t + e >= L) v[e] = 0.f;
                    }
                }
                chain_st<F>(r_out, voff, soff, v);
            }
        }
    };

    // MG: one resblock per workgroup.  The group's record (tile size, output buffer, role) is read from the kernel arguments again HERE - scalar
    // loads behind an opaque copy of the group index - instead of living in registers across the contractions (the 32-channel kernel has
    // none to spare).  A final group forms the stage's result from two summands in the order of `xs += resblock(x)`.
    auto flush_mg = [&]() {
        int oz = 0, gz = gidx;
        asm volatile("" : "+v"(oz));
        asm volatile("" : "+s"(gz));
        float* const gout = p.grp[gz].out;
        const int tend_g = min(t0 + p.grp[gz].N, LS);
        const bool fin = p.grp[gz].final != 0;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(gout + (size_t)b * C * LS, 0, 0x7ffffff0, 0x00020000);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((fin ? p.sum_in : p.in) + (size_t)b * C * LS), 0, 0x7ffffff0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((fin ? p.sum_in2 : p.in) + (size_t)b * C * LS), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int t = ws + (cw + 32 * nb + j + oz) * F;
            const bool ok = t >= t0 && t < tend_g;
            const int voff = ((4 / F) * h * LS + (ok ? t : t0)) * 4;
            float av[NG][F], bv2[NG][F];
            if (fin) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {                   // all the reads first
                    const int soff = ((8 / F) * (g * F / 4) + (g * F % 4) / F) * LS * 4;
                    chain_ld<F>(ra, voff, soff, av[g]);
                    chain_ld<F>(rb, voff, soff, bv2[g]);
                }
            }
            if (!ok) continue;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int soff = ((8 / F) * (g * F / 4) + (g * F % 4) / F) * LS * 4;
                float v[F];
#pragma unroll
                for (int e = 0; e < F; ++e) {
                    v[e] = y[nb][g * F + e];
                    if (fin) {
                        // (s + y) + s2, or with this resblock as the last of the three (s + s2) + y
                        if (p.own_last) { const float t2 = av[g][e] + bv2[g][e]; v[e] = t2 + v[e]; }
                        else { v[e] = av[g][e] + v[e]; v[e] = bv2[g][e] + v[e]; }
                        if (p.divide != 1.f) v[e] = v[e] / p.divide;
                        if (t + e >= L) v[e] = 0.f;
                    }
                }
                chain_st<F>(ro, voff, soff, v);
            }
        }
    };

    const int total = (MG ? 1 : p.nres) * p.npairs * 2;
    const int n0 = MG ? gidx * 2 * p.npairs : 0;          // the group's first convolution in conv[]
    // every sample this workgroup can write to a tile (columns SLK .. SLK + WPOS + F * dil) lies inside [0, L): no range masks in the epilogues
    const bool interior = (ws >= 0) && (ws + chain_wpos<F, NB>() + SLK <= L);
    const bool stamp = !MG && p.dbg != nullptr && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && lane == 0;
    auto mark = [&](int n, int k) { if (stamp) p.dbg[(n * 4 + w) * 4 + k] = __builtin_amdgcn_s_memtime(); };
    ChainPipe<NB, LD, F == 1> pipe(p.wp + p.conv[n0].woff, lane);
    pipe.start_a();
    int q = 0;                              // pair of the convolution n inside its resblock
#pragma unroll 1
    for (int n = 0; n < total; ++n) {
        const int ci = n & 1;
        if (ci == 0 && q == 0) {
            __syncthreads();                // the previous resblock's last readers of A are done
            // stage A = leaky_relu(x) for the samples [ws - SLK, ws - SLK + LD), zero outside [0, LS) (x is zero in [L, LS) already).  ALL the
            // loads of the tile are requested before the first one is used: a load -> write -> load chain pays the memory latency once per
            // float4 (18 times per thread at 32 channels - a third of the workgroup's time in the first version, profiles/r08_*)
            // (the addresses below do not depend on the convolution index: hipcc hoists all of them - 18 + 64 pointers at 32 channels - out of the
            // loop and keeps them in registers across every contraction; an opaque zero defined HERE keeps the index arithmetic in this block)
            int oz = 0;
            asm volatile("" : "+v"(oz));
            constexpr int NST = (C * NCOL4 + kThreads - 1) / kThreads, SB = (NST > 9) ? 6 : NST;                 // batches of at most six loads per thread beyond nine
#pragma unroll
            for (int it0 = 0; it0 < NST; it0 += SB) {
                float4 sv[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int idx = (it0 + i) * kThreads + tid + oz;
                    const int row = idx / NCOL4, g = idx - row * NCOL4;
                    const int t = ws - SLK + 4 * g;
                    const bool ok = (idx < C * NCOL4) && t >= 0 && t < LS;
                    const float4 v = *reinterpret_cast<const float4*>(inb + (size_t)(ok ? row : 0) * LS + (ok ? t : 0));
                    sv[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                DSD_SB();
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int idx = (it0 + i) * kThreads + tid + oz;
                    const int row = idx / NCOL4, g = idx - row * NCOL4;
                    float4 v = sv[i];
                    v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope); v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
                    if (idx < C * NCOL4) *reinterpret_cast<float4*>(bufA + row * LD + 4 * g) = v;
                }
                DSD_SB();
            }
            // y = x in the fragment order of a dilation-1 convolution: register r of column c holds channel row / F, sample ws + c F + row % F
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int t = ws + (cw + 32 * nb + j + oz) * F;
                const bool ok = t >= 0 && t < LS;
                const int voff = ((4 / F) * h * LS + (ok ? t : 0)) * 4;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float v[F];
                    chain_ld<F>(r_in, voff, ((8 / F) * (g * F / 4) + (g * F % 4) / F) * LS * 4, v);
#pragma unroll
                    for (int e = 0; e < F; ++e) y[nb][g * F + e] = ok ? v[e] : 0.f;
                }
            }
            __syncthreads();
        }
        mark(n, 0);
        const VocChainConv cv = p.conv[n0 + n];
        const float* src = ci ? bufT : bufA;
        float* dst = ci ? bufA : bufT;
        const int dil = cv.dil;
        // column -> first sample (relative to ws) of its F-step group
        int rel[NB], boff[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int c = cw + 32 * nb + j;
            const int g = c / dil;
            rel[nb] = g * (F * dil) + (c - g * dil);
            boff[nb] = rel[nb] - rel[0];
        }
        f32x16 acc[1][NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][nb][r] = 0.f;
        const int nch = (C / 8) * cv.KT;
        pipe.set_b(nch, src + 4 * h * LD + SLK + rel[0] - cv.pad, cv.KT, dil, boff);
        pipe.start_b();
        pipe.run_blocks(acc, nch);
        mark(n, 1);
        if (n + 1 < total) {                // the next convolution's weights: requested now, used behind the epilogue and the barrier
            pipe.set_a(p.wp + p.conv[n0 + n + 1].woff);
            pipe.start_a();
        }
        const bool last_of_res = (ci == 1 && q == p.npairs - 1);        // the last convolution of a resblock feeds no further one
        if constexpr (IP) {
            // in place: nobody may write the tile before every wave has read its last operand of THIS convolution
            if (!last_of_res) __syncthreads();
        }
        // the bias of this lane's channels through SCALAR loads (constant address space: s_load, lgkmcnt): a vector load issued here and used in the
        // epilogue made hipcc wait for the next convolution's weight prefetch (vmcnt counts in order) - an L2 round trip in front of every epilogue.
        // Requested BEHIND the contraction (round 6): 16 / F registers less while it runs
        typedef const __attribute__((address_space(4))) float cfloat;
        cfloat* bias_c = (cfloat*)(p.bias + cv.boff);
        float bv[16 / F];
#pragma unroll
        for (int i = 0; i < 16 / F; ++i) {
            const int c0 = (((i * F) & 3) + 8 * ((i * F) >> 2)) / F;            // frag_row(i F, 0) / F; the other half-wave: + 4 / F
            const float b0 = bias_c[c0], b1 = bias_c[c0 + 4 / F];
            bv[i] = h ? b1 : b0;
        }
        // Epilogue: v = acc + bias (+ y -> the new y), leaky_relu(v) -> the other tile.  The first version spent 6 650 cycles per convolution here
        // (64 values x ~100 cycles of index / mask arithmetic, profiles/r10_voc_chain_timeline.txt): row and step of a register are compile-time
        // constants up to the half-wave term, which moves into the lane's base pointer; tiles whose every writable sample lies inside [0, L) -
        // all but the first and last of an utterance - skip the range masks, and leaky_relu is max(v, slope v) (the same value for 0 <= slope <= 1)
        auto epilogue = [&](auto ci_tag, auto last_tag, auto interior_tag) {
            constexpr bool CI = decltype(ci_tag)::value, LAST = decltype(last_tag)::value, INTERIOR = decltype(interior_tag)::value;
            int ed[F];
#pragma unroll
            for (int e = 0; e < F; ++e) ed[e] = e * dil;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float* dbase = dst + (h * (4 / F)) * LD + SLK + rel[nb];
                const int tb = ws + rel[nb];
                if constexpr (CI && F == 4) {
                    // dilation 1: rows 8 rg + 4 h + (0..3) of a column are four consecutive samples of channel 2 rg + h: 16-byte tile writes
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        float4 v = get4(acc[0][nb], rg), yv = get4(y[nb], rg);
                        const float b0 = bv[rg];
                        v.x = (v.x + b0) + yv.x; v.y = (v.y + b0) + yv.y; v.z = (v.z + b0) + yv.z; v.w = (v.w + b0) + yv.w;
                        set4(y[nb], rg, v);
                        if constexpr (!LAST) {
                            float4 o = make_float4(fmaxf(v.x, v.x * slope), fmaxf(v.y, v.y * slope), fmaxf(v.z, v.z * slope), fmaxf(v.w, v.w * slope));
                            if (!INTERIOR) {
                                if (!(tb + 0 >= 0 && tb + 0 < L)) o.x = 0.f;
                                if (!(tb + 1 >= 0 && tb + 1 < L)) o.y = 0.f;
                                if (!(tb + 2 >= 0 && tb + 2 < L)) o.z = 0.f;
                                if (!(tb + 3 >= 0 && tb + 3 < L)) o.w = 0.f;
                            }
                            *reinterpret_cast<float4*>(dbase + (2 * rg) * LD) = o;
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row0 = (r & 3) + 8 * (r >> 2);            // frag_row without the half-wave term (4 h: a multiple of F)
                        const int co0 = row0 / F, e = row0 % F;
                        float v = acc[0][nb][r] + bv[r / F];
                        if constexpr (CI) { v += y[nb][r]; y[nb][r] = v; }
                        if constexpr (!LAST) {
                            float o = fmaxf(v, v * slope);
                            if (!INTERIOR) { const int t = tb + ed[e]; if (!(t >= 0 && t < L)) o = 0.f; }
                            dbase[co0 * LD + ed[e]] = o;
                        }
                    }
                }
            }
        };
        {
            using T_ = std::true_type; using F_ = std::false_type;
            if (ci == 0) { if (interior) epilogue(F_{}, F_{}, T_{}); else epilogue(F_{}, F_{}, F_{}); }
            else if (!last_of_res) { if (interior) epilogue(T_{}, F_{}, T_{}); else epilogue(T_{}, F_{}, F_{}); }
            else epilogue(T_{}, T_{}, T_{});                  // the last convolution of a resblock writes no tile: nothing to mask
        }
        mark(n, 2);
        __syncthreads();
        mark(n, 3);
        if (ci == 1) {
            if (last_of_res) {
                if constexpr (MG) { flush_mg(); return; }     // (one resblock per workgroup: no path from these stores back into the K loop)
                else flush(n == 2 * p.npairs - 1, n == total - 1);
                q = 0;
            } else {
                ++q;
            }
        }
    }

}

}  // namespace dsd
