// voc_chain16.hpp - gfx950: the ResBlock1 chains of the HiFi-GAN generator's NARROW stages (16 and 8 channels) on a 16-ROW matrix shape
// (round 6).
//
// voc_chain.hpp runs these stages on one 32-row MFMA block by folding F = 2 / 4 output samples into the rows (row = co F + e): the A operand then
// holds the F shifted copies of a filter, K + F - 1 taps for K - 14 % (16 channels) / 43 % (8 channels) of the matrix work multiplies zeros.
// v_mfma_f32_16x16x1_4B_f32 is four independent 16 x 16 x 1 blocks per instruction - 64 COLUMNS x 16 rows x ONE k at the FLOP rate of the 32x32x2
// form (32.04 cycles, profiles/r6_25_mfma_shape_probe.jsonl) - so 16 channels need no fold and 8 channels a 2-fold (K + 1 taps): 12.5 % / 20 %
// less matrix work.  The same probe shows that its accumulation rounds like the 32x32x2 form's (c + a0 b0 + a1 b1 in k order: 0 mismatches in
// 512 000 elements), so with the contributions of an output sample visited in the one-convolution kernels' order - 8-channel group, tap,
// channels 0 4 1 5 2 6 3 7 of the group (the k = 0 / 1 halves of their four MFMAs per chunk) - the results stay BIT-IDENTICAL to theirs (the
// fold's extra taps only ever add a zero product).
//
// Everything else is voc_chain.hpp's in-place form: a workgroup owns N output samples of all C channels of one utterance plus the chain's
// receptive field, ONE LDS tile [C][LD] rewritten in place by every convolution, the raw y in registers in the accumulator order of the
// dilation-1 convolutions, one launch per resblock with the running sum over the parallel resblocks in the output buffer.
//   * lanes: a wave owns 64 NBLK columns; as the B operand lane l supplies column 64 nb + l, as the accumulator lane (q = l >> 4, j = l & 15)
//     holds, in register r = 4 blk + rr, row 4 q + rr of column 64 nb + 16 blk + j.  Row = co F + e computes output sample pos(c) + e dil of
//     channel co, pos(c) = (c / dil) F dil + c % dil (voc_kernels.hpp's fold; F = 1: pos(c) = c).
//   * the A operand of chunk (8-channel group g, folded tap s) is EIGHT k steps = two 1 KiB fragment rows [chunk][half][lane] float4, lane l
//     carrying row l & 15 (the four blocks multiply the same filter): packed on the host (diffsinger_amd/vocoder.py pack_chain16).
//   * the result of a resblock leaves through the (free) tile: raw y written in the dilation-1 layout, then every thread moves whole float4 of
//     a channel row - coalesced, and the running-sum arithmetic (out + y, (sum_in + .) / divide, zero beyond L) is the one of voc_chain.hpp.
#pragma once
#include "voc_chain.hpp"

namespace dsd {

template <int F, int NBLK> constexpr int chain16_wpos() { return 256 * NBLK * F; }                // samples a dilation-1 convolution covers
template <int F, int NBLK> constexpr int chain16_ld() { return chain16_wpos<F, NBLK>() + 2 * kChainSlack; }
template <int C, int F, int NBLK> constexpr int chain16_lds_bytes() { return (C * chain16_ld<F, NBLK>() + 256) * (int)sizeof(float); }

typedef float f32x16c __attribute__((ext_vector_type(16)));

// K loop of one convolution on the 16-row shape: A = two float4 per chunk and lane (4 register stages), B = eight volatile ds_read_b32 per
// chunk and column block (row offsets in the instruction's 16-bit field), a RUNNING chunk pointer per column block (tap + 1, wrap to the next
// 8-channel group).  CONSTB (F == 1: pos(c) = c): the column blocks of a wave sit 64 floats apart - one pointer.
template <int NBLK, int LD, bool CONSTB>
struct Chain16Pipe {
    static constexpr int ST = 4, NP = CONSTB ? 1 : NBLK;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned aoff;
    const float* cur[NP];
    int KT, dil, left, tap;
    float4 a[ST][2];
    float b[2][8][NBLK];

    __device__ __forceinline__ Chain16Pipe(const float4* abase_uniform, int lane)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(abase_uniform), 0, 0x7ffffff0, 0x00020000)), aoff((unsigned)lane * 16u), KT(1), dil(1),
          left(0), tap(0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) cur[i] = nullptr;
    }
    __device__ __forceinline__ void set_a(const float4* abase_uniform) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(abase_uniform), 0, 0x7ffffff0, 0x00020000);
    }
    __device__ __forceinline__ void set_b(int n, const float* (&bbase)[NBLK], int KT_, int dil_) {
        KT = KT_; dil = dil_; left = n - 1; tap = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) cur[i] = bbase[i];
    }
    __device__ __forceinline__ void lda(float4 (&dst)[2], int kc) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)aoff + hf * 1024, kc * 2048, 0));
            dst[hf] = make_float4(f.x, f.y, f.z, f.w);
        }
    }
    __device__ __forceinline__ void ldb(float (&dst)[8][NBLK]) {
        typedef const volatile __attribute__((address_space(3))) float lds_cvf;
        lds_cvf* vp[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) vp[i] = (lds_cvf*)cur[i];
        const bool adv = left > 0, wrap = (tap + 1 == KT);
        const int step = wrap ? 8 * LD - (KT - 1) * dil : dil;
#pragma unroll
        for (int i = 0; i < NP; ++i) cur[i] += adv ? step : 0;
        tap = adv ? (wrap ? 0 : tap + 1) : tap;
        left -= adv ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = (i >> 1) + 4 * (i & 1);              // the one-convolution kernels' order inside a chunk: channels 0 4 1 5 2 6 3 7
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) dst[i][nb] = CONSTB ? vp[0][ch * LD + 64 * nb] : vp[nb][ch * LD];
        }
    }
    __device__ __forceinline__ void pattern() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 8 * NBLK - 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    __device__ __forceinline__ void start_a() {
#pragma unroll
        for (int i = 0; i < ST - 1; ++i) lda(a[i], i);
        DSD_SB();
    }
    __device__ __forceinline__ void start_b() {
        ldb(b[0]);
        DSD_SB();
    }
    template <int I>
    __device__ __forceinline__ void step(f32x16c (&acc)[NBLK], int it) {
        lda(a[(I + ST - 1) % ST], ST * it + I + ST - 1);
        ldb(b[(I + 1) & 1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 av4 = a[I % ST][i >> 2];
            const float av = ((i & 3) == 0) ? av4.x : ((i & 3) == 1) ? av4.y : ((i & 3) == 2) ? av4.z : av4.w;
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x1f32(av, b[I & 1][i][nb], acc[nb], 0, 0, 0);
        }
        pattern();
        DSD_SB();
    }
    // whole groups of four chunks as ONE basic block, the tail behind it
    __device__ __forceinline__ void run_blocks(f32x16c (&acc)[NBLK], int end) {
        int it = 0;
        for (; ST * it + ST <= end; ++it) {
            step<0>(acc, it); step<1>(acc, it); step<2>(acc, it); step<3>(acc, it);
        }
        const int kc = ST * it;
        if (kc >= end) return;
        step<0>(acc, it);
        if (kc + 1 >= end) return;
        step<1>(acc, it);
        if (kc + 2 >= end) return;
        step<2>(acc, it);
    }
};

// grid (ceil(LS / N), B); 4 waves, wave w owns the columns [64 NBLK w, 64 NBLK (w + 1)) of every convolution
template <int C, int F, int NBLK>
__global__ __launch_bounds__(kThreads, 3) void k_voc_chain16(const VocChainParams p) {
    static_assert(C * F == 16 && (C % 8) == 0 && (F == 1 || F == 2), "one 16-row MFMA block: 16 channels, or 8 channels x 2");
    constexpr int LD = chain16_ld<F, NBLK>(), SLK = kChainSlack, NCOL4 = LD / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                      // [C][LD]: leaky_relu of the current activation, rewritten in place
    const int tid = threadIdx.x, lane = tid & 63, j16 = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * p.N, b = blockIdx.y;
    const int ws = t0 - p.Hh;                // sample of tile column SLK
    const int L = p.L, LS = p.LS;
    const float slope = p.slope;
    const float* inb = p.in + (size_t)b * C * LS;
    float* outb = p.out + (size_t)b * C * LS;
    const float* sinb = p.sum_in ? p.sum_in + (size_t)b * C * LS : nullptr;

    for (int idx = tid + C * LD / 4; idx < (C * LD + 256) / 4; idx += kThreads) *reinterpret_cast<float4*>(tile + 4 * idx) = make_float4(0.f, 0.f, 0.f, 0.f);

    // accumulator register r = 4 blk + rr of column block nb: row 4 q + rr -> channel co = row / F, step e = row % F; column 64 (NBLK w + nb) + 16 blk + j16
    f32x16c y[NBLK];
    const int tend = min(t0 + p.N, LS);
    const int total = p.nres * p.npairs * 2;
    const bool interior = (ws >= 0) && (ws + chain16_wpos<F, NBLK>() + SLK <= L);
    Chain16Pipe<NBLK, LD, F == 1> pipe(p.wp + p.conv[0].woff, lane);
    pipe.start_a();
    int pq = 0;                              // pair of the convolution n inside its resblock
#pragma unroll 1
    for (int n = 0; n < total; ++n) {
        const int ci = n & 1;
        if (ci == 0 && pq == 0) {
            __syncthreads();                // the previous resblock's copy-out has left the tile
            int oz = 0;
            asm volatile("" : "+v"(oz));     // (keeps the staging's index arithmetic in this block: voc_chain.hpp)
            constexpr int NST = (C * NCOL4 + kThreads - 1) / kThreads, SB = (NST > 9) ? 6 : NST;
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
                    if (idx < C * NCOL4) *reinterpret_cast<float4*>(tile + row * LD + 4 * g) = v;
                }
                DSD_SB();
            }
            // y = x in the fragment order of a dilation-1 convolution: register (blk, rr) of column c holds channel co(rr), sample ws + c F + e(rr)
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 4 * q + (r & 3), co = row / F, e = row % F;
                    const int c = 64 * (NBLK * w + nb) + 16 * (r >> 2) + j16 + oz;
                    const int t = ws + c * F + e;
                    const bool ok = t >= 0 && t < LS;
                    const float v = inb[(size_t)co * LS + (ok ? t : 0)];
                    y[nb][r] = ok ? v : 0.f;
                }
            __syncthreads();
        }
        const VocChainConv cv = p.conv[n];
        const int dil = cv.dil;
        // B operand: lane l supplies column 64 (NBLK w + nb) + l; accumulator: column 64 (NBLK w + nb) + 16 blk + j16
        const float* bbase[NBLK];
        int reld[NBLK][4];                  // first sample (relative to ws) of the accumulator columns
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const int cb = 64 * (NBLK * w + nb) + lane, gb = cb / dil;
            bbase[nb] = tile + SLK + gb * (F * dil) + (cb - gb * dil) - cv.pad;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const int c = 64 * (NBLK * w + nb) + 16 * blk + j16, g = c / dil;
                reld[nb][blk] = g * (F * dil) + (c - g * dil);
            }
        }
        f32x16c acc[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const int nch = (C / 8) * cv.KT;
        pipe.set_b(nch, bbase, cv.KT, dil);
        pipe.start_b();
        pipe.run_blocks(acc, nch);
        // this lane's bias values, requested in FRONT of the next convolution's weight prefetch (vector-memory returns are in order: an older
        // load can be waited for with the prefetch still in flight)
        float bv[4 / F];
#pragma unroll
        for (int i = 0; i < 4 / F; ++i) bv[i] = p.bias[cv.boff + (4 / F) * q + i];
        if (n + 1 < total) {
            pipe.set_a(p.wp + p.conv[n + 1].woff);
            pipe.start_a();
        }
        const bool last_of_res = (ci == 1 && pq == p.npairs - 1);
        __syncthreads();                    // in place: nobody writes the tile before every wave has read its last operand of THIS convolution
        // Epilogue: v = acc + bias (+ y -> the new y); leaky_relu(v) -> the tile, or - behind the last convolution of a resblock - the RAW v
        // (the tile is the staging buffer of the coalesced copy-out)
        auto epilogue = [&](auto ci_tag, auto last_tag, auto interior_tag) {
            constexpr bool CI = decltype(ci_tag)::value, LAST = decltype(last_tag)::value, INTERIOR = decltype(interior_tag)::value;
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = r & 3, blk = r >> 2;
                    const int co_i = rr / F, e = rr % F;                    // channel (4 / F) q + co_i
                    float v = acc[nb][r] + bv[co_i];
                    if constexpr (CI) { v += y[nb][r]; y[nb][r] = v; }
                    float o = LAST ? v : fmaxf(v, v * slope);
                    const int off = reld[nb][blk] + e * dil;
                    if (!INTERIOR && !LAST) { const int t = ws + off; if (!(t >= 0 && t < L)) o = 0.f; }
                    tile[((4 / F) * q + co_i) * LD + SLK + off] = o;
                }
        };
        {
            using T_ = std::true_type; using F_ = std::false_type;
            if (ci == 0) { if (interior) epilogue(F_{}, F_{}, T_{}); else epilogue(F_{}, F_{}, F_{}); }
            else if (!last_of_res) { if (interior) epilogue(T_{}, F_{}, T_{}); else epilogue(T_{}, F_{}, F_{}); }
            else epilogue(T_{}, T_{}, T_{});
        }
        __syncthreads();
        if (ci == 1) {
            if (last_of_res) {
                // copy-out: samples [t0, tend) of every channel, whole float4 of a row per thread; the running sum over the parallel resblocks
                // lives in `out` (voc_chain.hpp): y_0, then out + y_r, and for the last one (sum_in + .) / divide, zero beyond L
                const bool first = (n == 2 * p.npairs - 1), last = (n == total - 1);
                const int n4 = (tend - t0) / 4;                             // t0, N, LS are multiples of 4
                for (int idx = tid; idx < C * n4; idx += kThreads) {
                    const int row = idx / n4, g = idx - row * n4;
                    const int t = t0 + 4 * g;
                    float4 v = *reinterpret_cast<const float4*>(tile + row * LD + SLK + p.Hh + 4 * g);
                    const size_t o = (size_t)row * LS + t;
                    if (!first) { const float4 pv = *reinterpret_cast<const float4*>(outb + o); v.x = pv.x + v.x; v.y = pv.y + v.y; v.z = pv.z + v.z; v.w = pv.w + v.w; }
                    if (last) {
                        if (sinb) { const float4 sv = *reinterpret_cast<const float4*>(sinb + o); v.x = sv.x + v.x; v.y = sv.y + v.y; v.z = sv.z + v.z; v.w = sv.w + v.w; }
                        if (p.divide != 1.f) { v.x = v.x / p.divide; v.y = v.y / p.divide; v.z = v.z / p.divide; v.w = v.w / p.divide; }
                        if (t + 0 >= L) v.x = 0.f;
                        if (t + 1 >= L) v.y = 0.f;
                        if (t + 2 >= L) v.z = 0.f;
                        if (t + 3 >= L) v.w = 0.f;
                    }
                    *reinterpret_cast<float4*>(outb + o) = v;
                }
                pq = 0;
            } else {
                ++pq;
            }
        }
    }
}

}  // namespace dsd
