// voc_pipe.hpp - gfx950: the one-convolution kernel of the HiFi-GAN generator as a PERSISTENT, double-buffered pipeline (SURVEY.md section 8 row
// f2; round 6).
//
// k_voc_conv (voc_kernels.hpp) runs a tile as three phases - stage the input slab, contract, store - and a launch of the long layers is one or
// two rounds of co-resident workgroups that all sit in the same phase at the same time: the per-CU memory time of a tile (at 64 channels
// 94 KiB of staging + 64 KiB of residual + 64 KiB of output at the ~10 B / clk a CU sustains = ~10 us) stands BESIDE its 27 us of matrix
// time, not under it (a third resident workgroup and a mixed order of long and short workgroups were measured and lost).
// Here ONE workgroup per CU walks its tiles - item = (convolution of the call, row-block group, utterance, tile), the items of a launch dealt
// round-robin - with two LDS slabs:
//     A prefetch(k) | residual / running-sum operands(k) | contraction(k), the staging loads of item k + 1 issued in three gaps between its
//     first groups of six chunks | leaky_relu + LDS write of item k + 1 into the other slab | barrier | epilogue(k): stores in flight under k + 1
// Vector memory returns in order: the weight prefetch is issued BEFORE the loads it must not wait for, and the staging loads are spread over
// three gaps so that the weight loads queued behind them stay inside the five-chunk prefetch distance.  A staging load is ONE instruction
// without address arithmetic - buffer_load_dwordx4 with the thread's fixed offset inside the utterance in a register and (utterance, tile) in
// the scalar offset; the range checks of the first and last tile of an utterance run in a slow path.  The contraction walks the chunks in
// k_voc_conv's order and the fused tail is voc_conv_epilogue: BIT-IDENTICAL results (tests/test_gpu_vocoder.py).
#pragma once
#include "voc_kernels.hpp"

namespace dsd {

constexpr int kVocPipeLds = 78 * 1024;      // per slab; two slabs are a CU's LDS
constexpr int kPipeLoads = 18;              // staging float4 per thread at most (32 channels x 568 columns)

template <int NB, int WT, int HALO> constexpr int pipe_slab() {          // channels per slab: the LDS, and kPipeLoads float4 per thread
    const int s = kVocPipeLds / (voc_ld<NB, WT, HALO>() * 4) / 8 * 8;
    const int r = kPipeLoads * kThreads / (voc_ld<NB, WT, HALO>() / 4) / 8 * 8;
    return (s < r ? s : r) > 256 ? 256 : (s < r ? s : r);
}

struct VocPipeParams {
    VocConvParams g[kVocMultiMax];          // the convolutions of the call: one shape (B, Ci, rows, L, up), their own kernel / operands
    int ngroups, tiles, zc, B;              // x tiles per utterance, row-block groups per convolution
    int per_group;                          // tiles * B * zc
    int nitems;                             // per_group * ngroups
    int slab8;                              // channels staged per tile: Ci rounded up to 8 (<= pipe_slab)
};

template <int FROM, int TO, int HALO>
__device__ __forceinline__ void pipe_issue(float4 (&sv)[kPipeLoads], const int (&goff)[kPipeLoads], const int (&cg)[kPipeLoads], unsigned rowok,
                                           const float* in_utt_uniform, int nbytes, int t0, int LSi, bool interior) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_utt_uniform), 0, nbytes, 0x00020000);
    if (interior) {
        const int soff = (t0 - HALO) * 4;
#pragma unroll
        for (int i = FROM; i < TO; ++i) {
            const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, goff[i], soff, 0));
            sv[i] = make_float4(f.x, f.y, f.z, f.w);
        }
    } else {
#pragma unroll
        for (int i = FROM; i < TO; ++i) {
            const int t = t0 + cg[i];
            const bool ok = ((rowok >> i) & 1u) && t >= 0 && t < LSi;
            const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ok ? goff[i] + (t0 - HALO) * 4 : 0, 0, 0));
            sv[i] = make_float4(f.x, f.y, f.z, f.w);
        }
    }
}

__device__ __forceinline__ void pipe_write(const float4 (&sv)[kPipeLoads], const int (&loff)[kPipeLoads], const int (&cg)[kPipeLoads], unsigned live,
                                           unsigned rowok, float* slab, int t0, int LSi, float slope, bool interior) {
#pragma unroll
    for (int i = 0; i < kPipeLoads; ++i) {
        bool ok = (rowok >> i) & 1u;
        if (!interior) { const int t = t0 + cg[i]; ok = ok && t >= 0 && t < LSi; }
        float4 v = ok ? sv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = voc_lrelu(v.x, slope); v.y = voc_lrelu(v.y, slope); v.z = voc_lrelu(v.z, slope); v.w = voc_lrelu(v.w, slope);
        if ((live >> i) & 1u) *reinterpret_cast<float4*>(slab + loff[i]) = v;
    }
}

// grid = min(items, CUs) workgroups, one per CU; dynamic LDS = 2 slabs of slab8 x LD floats
template <int NB, int WT, int HALO>
__global__ __launch_bounds__(kThreads, 1) void k_voc_conv_pipe(const VocPipeParams m) {
    constexpr int LD = voc_ld<NB, WT, HALO>(), SPAN = voc_span<NB, WT>(), WR = 4 / WT, NCOL4 = LD / 4;
    static_assert(pipe_slab<NB, WT, HALO>() * NCOL4 <= kPipeLoads * kThreads, "a slab is at most kPipeLoads float4 per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w % WR, wt = w / WR;
    const int Ci = m.g[0].Ci, LSi = m.g[0].LSi, rows = m.g[0].rows;
    const int nrb = (rows + 31) / 32;
    const int nstage = m.slab8 * NCOL4, slabsz = m.slab8 * LD, nbytes = Ci * LSi * 4;

    // this thread's staging float4 i: channel row idx / NCOL4, column group idx % NCOL4 of the slab - fixed for the whole launch
    int goff[kPipeLoads], loff[kPipeLoads], cg[kPipeLoads];
    unsigned live = 0, rowok = 0;
#pragma unroll
    for (int i = 0; i < kPipeLoads; ++i) {
        const int idx = i * kThreads + tid;
        const int row = idx / NCOL4, g4 = idx - row * NCOL4;
        const bool ex = idx < nstage, rk = ex && row < Ci;
        goff[i] = rk ? (row * LSi + 4 * g4) * 4 : 0;       // bytes inside the utterance, relative to sample t0 - HALO of row 0
        loff[i] = row * LD + 4 * g4;
        cg[i] = 4 * g4 - HALO;                              // first sample of the float4 relative to t0
        live |= ex ? (1u << i) : 0u;
        rowok |= rk ? (1u << i) : 0u;
    }

    // (the divisions run on the vector ALU: readfirstlane, or every descriptor derived from them is a waterfall loop)
    auto decode = [&](int it, int& grp, int& bz, int& b, int& t0) {
        const int g_ = it / m.per_group, r = it - g_ * m.per_group;
        const int tb = m.tiles * m.B;
        const int z = r / tb, r2 = r - z * tb;
        const int bb = r2 / m.tiles, x = r2 - bb * m.tiles;
        grp = __builtin_amdgcn_readfirstlane(g_); bz = __builtin_amdgcn_readfirstlane(z); b = __builtin_amdgcn_readfirstlane(bb);
        t0 = __builtin_amdgcn_readfirstlane(x) * SPAN;
    };
    auto is_interior = [&](int t0) { return (t0 - HALO >= 0) && (t0 + SPAN + HALO <= LSi); };

    int it = blockIdx.x;
    if (it >= m.nitems) return;
    int grp, bz, b, t0;
    decode(it, grp, bz, b, t0);
    float4 sv[kPipeLoads];
    {
        const VocConvParams& p = m.g[grp];
        const bool in0 = is_interior(t0);
        pipe_issue<0, kPipeLoads, HALO>(sv, goff, cg, rowok, p.in + (size_t)b * Ci * LSi, nbytes, t0, LSi, in0);
        pipe_write(sv, loff, cg, live, rowok, smem, t0, LSi, p.pre_slope, in0);
        __syncthreads();
    }
    int cur = 0;
#pragma unroll 1
    while (true) {
        const VocConvParams& p = m.g[grp];
        const int nxt = it + (int)gridDim.x;
        const bool more = nxt < m.nitems;
        int ngrp = grp, nbz = bz, nbb = b, nt0 = t0;
        if (more) decode(nxt, ngrp, nbz, nbb, nt0);
        const VocConvParams& pn = m.g[ngrp];
        const bool nin = is_interior(nt0);
        const float* nbase = pn.in + (size_t)nbb * Ci * LSi;

        const int rb = bz * WR + wr;
        const int rbc = (rb < nrb) ? rb : nrb - 1;
        const int nch = (m.slab8 / 8) * p.KT;
        f32x16 acc[1][NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][nb][r] = 0.f;
        const float* slab = smem + cur * slabsz;
        VocTapB<LD> bof(slab + 4 * h * LD + HALO + wt * (32 * NB) + j - p.pad, p.KT, p.dil, nch);
        GemmPipe<1, NB, LD, 64, 6, VocTapB<LD>, 1, false, true> pipe(p.wp + (size_t)rbc * nch * 64, lane, nch, bof);
        pipe.start_a();
        // a plain convolution's residual / running-sum operands: behind the weight prefetch, in front of everything else of this item
        constexpr bool PRE = (NB <= 2);
        float rpre[PRE ? NB : 1][16], spre[PRE ? NB : 1][16];
        const bool pre = PRE && p.U == 1 && rb < nrb && (p.res || p.sum_in);
        if constexpr (PRE) {
            if (pre) {
                const int qp = t0 + wt * (32 * NB) + j;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rb * 32 + frag_row(r, h), n = qp + 32 * nb;
                        const bool ok = row < p.rows && n < p.LSo;
                        const size_t o = ((size_t)b * p.rows + (ok ? row : 0)) * p.LSo + (ok ? n : 0);
                        rpre[nb][r] = p.res ? p.res[o] : 0.f;
                        spre[nb][r] = p.sum_in ? p.sum_in[o] : 0.f;
                    }
            }
        }
        pipe.start_b();
        if (nch >= 18) {
            pipe.run_group(acc, 0);
            if (more) pipe_issue<0, 6, HALO>(sv, goff, cg, rowok, nbase, nbytes, nt0, LSi, nin);
            pipe.run_group(acc, 1);
            if (more) pipe_issue<6, 12, HALO>(sv, goff, cg, rowok, nbase, nbytes, nt0, LSi, nin);
            pipe.run_group(acc, 2);
            if (more) pipe_issue<12, kPipeLoads, HALO>(sv, goff, cg, rowok, nbase, nbytes, nt0, LSi, nin);
            pipe.run_from(acc, 3, nch);
        } else if (nch >= 6) {
            pipe.run_group(acc, 0);
            if (more) pipe_issue<0, kPipeLoads, HALO>(sv, goff, cg, rowok, nbase, nbytes, nt0, LSi, nin);
            pipe.run_from(acc, 1, nch);
        } else {
            if (more) pipe_issue<0, kPipeLoads, HALO>(sv, goff, cg, rowok, nbase, nbytes, nt0, LSi, nin);
            pipe.run_from(acc, 0, nch);
        }
        // item k + 1 into the other slab: its last readers (contraction k - 1) passed the barrier of the previous iteration
        if (more) pipe_write(sv, loff, cg, live, rowok, smem + (cur ^ 1) * slabsz, nt0, LSi, pn.pre_slope, nin);
        __syncthreads();
        if (rb < nrb) voc_conv_epilogue<NB, PRE>(p, acc, rb, b, t0 + wt * (32 * NB) + j, h, pre, rpre, spre);
        if (!more) break;
        it = nxt; grp = ngrp; bz = nbz; b = nbb; t0 = nt0; cur ^= 1;
    }
}

}  // namespace dsd
