// voc_kernels.hpp - gfx950 kernels of the HiFi-GAN / NSF-HiFi-GAN generator (the step AFTER the
// diffusion hot path, mel [B,80,T] (+ f0 [B,T]) -> waveform [B,1,T*hop]).
//
// What is computed, and where the reference computes it (paths relative to the reference root):
//   k_voc_conv        every Conv1d AND every ConvTranspose1d of HifiGanGenerator.forward (modules/hifigan/hifigan.py:144-169) and of
//                     ResBlock1 / ResBlock2 (:30-92) as one fp32-MFMA contraction over (input channel, tap), with the element-wise
//                     neighbours fused: the leaky_relu in FRONT of the convolution (applied while the input tile is staged), the
//                     bias, the residual `xt + x`, the running sum over the parallel resblocks `xs += ...`, the `/ num_kernels`,
//                     the harmonic-source add `x + x_source` and the final tanh.
//                     A transposed convolution (stride u, kernel k, padding (k-u)/2) is the SAME kernel: its polyphase form is a
//                     stride-1 convolution at the INPUT rate with u * Co output rows (row = co * u + phase), whose rows are written
//                     interleaved - out[co][q * u + phase] (a "pixel shuffle" in the store addresses, no scatter, no zero stuffing).
//   k_voc_sine        SineGen._f02sine + forward (modules/parallel_wavegan/models/source.py:45-77, :101-137): per (utterance,
//                     harmonic) the phase is a cumulative sum over the SAMPLE axis; two block scans in fp64 (the reference's CPU
//                     cumsum accumulates in double too)
//   k_voc_source      uv / noise mix of SineGen.forward (:124-135) + SourceModuleHnNSF's Linear(9 -> 1) + tanh (:518-531)
//   k_voc_noise_conv  the strided Conv1d(1 -> C, kernel 2s, stride s) of noise_convs (hifigan.py:124-130): har [B,1,L] -> [B,C,L/s]
//   k_voc_pad_rows    [R][L] contiguous -> [R][LS] (row stride padded to 32, zero tail)
//
// Activations live channel-major [B][C][LS] like everywhere in this library (sample axis contiguous, LS = L rounded up to 32,
// ZERO in [L, LS)).  The generator is narrow (128 -> 64 -> 32 -> 16 -> 8 channels on the shipped config) and LONG (up to 256
// samples per mel frame), the opposite of the denoiser: a wave therefore owns ONE 32-row block and NB consecutive 32-sample
// blocks (the weight fragment it streams from L2 is reused NB times), and the four waves of a workgroup split the time axis
// when the layer has fewer than 128 rows.  Rows are padded to 32 with zero weights (the 8- and 16-channel layers waste MFMA
// issue slots; they are bound by their HBM traffic, not by the matrix pipe).
#pragma once
#include "dsd_kernels.hpp"

namespace dsd {

constexpr int kVocHalo = 28;               // taps reach +-25 samples (kernel 11, dilation 5) on the shipped generators; multiple of 4 (float4 staging)
constexpr int kVocHaloWide = 48;           // second instantiation of k_voc_conv: taps up to +-48 samples (the official v3 generator: kernel 7 at
                                           // dilation 12 = 36), hifigan.py:104-179 accepts any config
constexpr int kVocLdsBudget = 72 * 1024;   // two workgroups per CU
constexpr int kVocStageBatch = 12;         // staging loads a thread keeps in flight (round 6: 8 -> 12 - the 64-channel slab of k_voc_conv<2, 2> is 11.5
                                           // float4 per thread: ONE round trip to memory instead of two)

template <int NB, int WT> constexpr int voc_span() { return 32 * NB * WT; }                    // samples per workgroup
template <int NB, int WT, int HALO = kVocHalo> constexpr int voc_ld() { return voc_span<NB, WT>() + 2 * HALO; }  // LDS row stride
template <int NB, int WT, int HALO = kVocHalo> constexpr int voc_slab() {                     // input channels staged per pass
    const int s = kVocLdsBudget / (voc_ld<NB, WT, HALO>() * 4) / 8 * 8;
    return s > 256 ? 256 : s;
}
template <int NB, int WT, int HALO = kVocHalo> constexpr int voc_lds_bytes() { return voc_slab<NB, WT, HALO>() * voc_ld<NB, WT, HALO>() * 4; }

enum VocAct { VOC_ACT_NONE = 0, VOC_ACT_TANH = 1 };

struct VocConvParams {
    const float* in;        // [B][Ci][LSi]
    const float4* wp;       // packed [row block 32][chunk = ci8 * KT + tap][lane64] float4 (k_pack_a, nmb = 1)
    const float* bias;      // [Co] or nullptr
    float* out;             // [B][Co][LSo],  Co = rows / U
    const float* res;       // [B][Co][LSo] or nullptr: added after the bias
    const float* sum_in;    // [B][Co][LSo] or nullptr: running sum the result is added TO
    int Ci, rows, KT, pad, dil;
    int Li, LSi, U, Lo, LSo;
    float pre_slope;        // leaky_relu slope applied to the input (1 = identity)
    float divide;           // the result is divided by this (1 = no division)
    int act;
};

// B-operand functor of GemmPipe: chunk kc = ci8 * KT + tap of the staged slab, as a RUNNING pointer.  GemmPipe asks for the chunks strictly in
// order, one call per chunk (start_b, then every step): tap + 1, wrap to the next 8-channel group, freeze on the last chunk (the prefetch
// behind it must stay inside the slab) - a handful of selects.  (Rounds 1-2 recomputed the map per chunk: a chain of uniform branches, one per
// supported kernel size, and a constant-divisor division in front of EVERY chunk - ~85 scalar / vector instructions between two groups of
// 4 NB MFMAs with the matrix pipe idle; -4.5 % on the whole generator forward, profiles/r05_fm_conv_inc_ab.jsonl.)
template <int LD>
struct VocTapB {
    const float* cur;       // the chunk handed out next (base = slab + 4 h LD + halo + this wave's first sample + j - pad)
    int KT, dil, left, tap; // left: chunks that may still be advanced past
    __device__ __forceinline__ VocTapB(const float* base, int KT_, int dil_, int n) : cur(base), KT(KT_), dil(dil_), left(n - 1), tap(0) {}
    __device__ __forceinline__ const float* operator()(int, int) {
        const float* r = cur;
        const bool adv = left > 0, wrap = (tap + 1 == KT);
        const int step = wrap ? 8 * LD - (KT - 1) * dil : dil;
        cur += adv ? step : 0;
        tap = adv ? (wrap ? 0 : tap + 1) : tap;
        left -= adv ? 1 : 0;
        return r;
    }
};

__device__ __forceinline__ float voc_lrelu(float v, float slope) { return (v > 0.f) ? v : v * slope; }

// The fused tail of a convolution: bias, residual, running sum, divisor, tanh, zero beyond L, in the store form of its phase count
// (voc_conv_body; the persistent pipelined form measured in r6_36 / r6_37 shared it - git 4b97f88).  acc: this wave's 32 rows x NB 32-sample blocks; q0: input-rate sample of
// column j of block 0; rpre / spre: a plain convolution's residual / running-sum operands if they were requested in front of the contraction.
template <int NB, bool PRE>
__device__ __forceinline__ void voc_conv_epilogue(const VocConvParams& p, f32x16 (&acc)[1][NB], int rb, int b, int q0, int h, bool pre,
                                                  const float (&rpre)[PRE ? NB : 1][16], const float (&spre)[PRE ? NB : 1][16]) {
    const int U = p.U, Co = p.rows / U;
    if ((U & 3) == 0) {
        // rows 8 rg + 4 h + (0..3) of this lane are 4 consecutive phases of ONE output channel: 16-byte accesses
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int q = q0 + 32 * nb;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = rb * 32 + 8 * rg + 4 * h;
                const int co = row / U, ph = row - co * U;
                const int n = q * U + ph;
                if (row >= p.rows || n >= p.LSo) continue;
                const size_t o = ((size_t)b * Co + co) * p.LSo + n;
                const float bv = p.bias ? p.bias[co] : 0.f;
                float4 v = get4(acc[0][nb], rg);
                v.x += bv; v.y += bv; v.z += bv; v.w += bv;
                if (p.res) { const float4 r4 = *reinterpret_cast<const float4*>(p.res + o); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
                if (p.sum_in) { const float4 s4 = *reinterpret_cast<const float4*>(p.sum_in + o); v.x = s4.x + v.x; v.y = s4.y + v.y; v.z = s4.z + v.z; v.w = s4.w + v.w; }
                if (p.divide != 1.f) { v.x = v.x / p.divide; v.y = v.y / p.divide; v.z = v.z / p.divide; v.w = v.w / p.divide; }
                if (p.act == VOC_ACT_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
                if (n + 0 >= p.Lo) v.x = 0.f;
                if (n + 1 >= p.Lo) v.y = 0.f;
                if (n + 2 >= p.Lo) v.z = 0.f;
                if (n + 3 >= p.Lo) v.w = 0.f;
                *reinterpret_cast<float4*>(p.out + o) = v;
            }
        }
        return;
    }
    if (U == 2) {
        // registers (r, r + 1), r even, of this lane are the two phases of ONE output channel (row = 2 co + phase): 8-byte accesses, the lanes
        // of a half wave write 256 contiguous bytes (round 6; the scalar form below wrote every second float per instruction and spent ~2 900
        // vector instructions per wave on its index arithmetic - the stride-2 transposed convolutions ran at a third of their memory roof)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = (q0 + 32 * nb) * 2;
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                const int row = rb * 32 + frag_row(2 * rp, h);
                const int co = row >> 1;
                if (row >= p.rows || n >= p.LSo) continue;
                const size_t o = ((size_t)b * Co + co) * p.LSo + n;
                const float bv = p.bias ? p.bias[co] : 0.f;
                float2 v = make_float2(acc[0][nb][2 * rp] + bv, acc[0][nb][2 * rp + 1] + bv);
                if (p.res) { const float2 r2 = *reinterpret_cast<const float2*>(p.res + o); v.x += r2.x; v.y += r2.y; }
                if (p.sum_in) { const float2 s2 = *reinterpret_cast<const float2*>(p.sum_in + o); v.x = s2.x + v.x; v.y = s2.y + v.y; }
                if (p.divide != 1.f) { v.x = v.x / p.divide; v.y = v.y / p.divide; }
                if (p.act == VOC_ACT_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); }
                if (n + 0 >= p.Lo) v.x = 0.f;
                if (n + 1 >= p.Lo) v.y = 0.f;
                *reinterpret_cast<float2*>(p.out + o) = v;
            }
        }
        return;
    }
    // general phase count (1: plain convolution, lanes j write consecutive samples)
    float bv[16];
    int cov[16], phv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + frag_row(r, h);
        const int rc = (row < p.rows) ? row : 0;
        cov[r] = rc / U;
        phv[r] = rc - cov[r] * U;
        bv[r] = p.bias ? p.bias[cov[r]] : 0.f;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int q = q0 + 32 * nb;
        float rv[16], sv[16];
        bool have = false;
        if constexpr (PRE) {
            if (pre) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { rv[r] = rpre[nb][r]; sv[r] = spre[nb][r]; }
                have = true;
            }
        }
        if (!have) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {                           // all the reads first
                const int n = q * U + phv[r];
                const bool ok = (rb * 32 + frag_row(r, h) < p.rows) && n < p.LSo;
                const size_t o = ((size_t)b * Co + cov[r]) * p.LSo + (ok ? n : 0);
                rv[r] = (p.res && ok) ? p.res[o] : 0.f;
                sv[r] = (p.sum_in && ok) ? p.sum_in[o] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = q * U + phv[r];
            const bool ok = (rb * 32 + frag_row(r, h) < p.rows) && n < p.LSo;
            float v = acc[0][nb][r] + bv[r];
            if (p.res) v += rv[r];
            if (p.sum_in) v = sv[r] + v;
            if (p.divide != 1.f) v = v / p.divide;
            if (p.act == VOC_ACT_TANH) v = tanhf(v);
            if (ok) p.out[((size_t)b * Co + cov[r]) * p.LSo + n] = (n < p.Lo) ? v : 0.f;
        }
    }
}

// Workgroup = WR = 4 / WT row blocks of 32 x (WT * NB * 32) samples of one utterance.  Wave w: row block (w % WR), time part (w / WR).
template <int NB, int WT, int HALO, bool OPERANDS_AHEAD = true>
__device__ __forceinline__ void voc_conv_body(const VocConvParams& p, int bz) {
    constexpr int LD = voc_ld<NB, WT, HALO>(), SPAN = voc_span<NB, WT>(), SLAB = voc_slab<NB, WT, HALO>(), WR = 4 / WT;
    constexpr int NCOL4 = LD / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [SLAB][LD]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w % WR, wt = w / WR;
    const int t0 = blockIdx.x * SPAN, b = blockIdx.y;
    const int rb = bz * WR + wr;                                     // this wave's 32-row block
    const int nrb = (p.rows + 31) / 32;
    const int rbc = (rb < nrb) ? rb : nrb - 1;                       // waves past the last block walk valid memory and store nothing
    const int ci8 = (p.Ci + 7) / 8;
    const int nchunk_total = ci8 * p.KT;
    f32x16 acc[1][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] = 0.f;
    const float* inb = p.in + (size_t)b * p.Ci * p.LSi;
    const float slope = p.pre_slope;
    // A plain convolution's residual / running-sum operands (the second convolution of every ResBlock1 pair) are requested HERE, in front of
    // the weight prefetch and the staging loads (round 6): their round trip to memory ran behind the contraction, with nothing to hide under.
    // Only for the narrow tiles (NB <= 2: 16 NB registers per operand).
    constexpr bool PRE = OPERANDS_AHEAD && (NB <= 2);              // (the lean build has no registers for them: its layers carry no residual)
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
    for (int c0 = 0; c0 < p.Ci; c0 += SLAB) {
        const int nc = min(SLAB, p.Ci - c0);
        const int nc8 = (nc + 7) / 8 * 8;                            // rows [nc, nc8) are staged as zeros (their weights are zero too)
        // stage channels [c0, c0 + nc) x samples [t0 - halo, t0 + SPAN + halo), zero outside [0, LSi), leaky_relu applied here
        // (eight loads per thread in flight before the first LDS write: a load -> write -> load chain pays the memory latency once per float4 -
        // 12 round trips for a 64-channel slab, ~16 us of every launch of the 64-channel stage in rounds 2-5, profiles/r57_vocoder_kernel_stats.txt)
        const int nstage = nc8 * NCOL4;
        // the weight stream depends on nothing the kernel computes: its first chunks are requested in FRONT of the staging loads (round 6; vector
        // memory returns in order - the staging wait covers them - and their L2 / Infinity-Cache latency, ~1.5 us of every launch, leaves the
        // critical path)
        const int nch = (nc8 / 8) * p.KT;
        const float4* ap = p.wp + ((size_t)rbc * nchunk_total + (size_t)(c0 / 8) * p.KT) * 64;
        VocTapB<LD> bof(smem + 4 * h * LD + HALO + wt * (32 * NB) + j - p.pad, p.KT, p.dil, nch);
        GemmPipe<1, NB, LD, 64, 6, VocTapB<LD>, 1, false, true> pipe(ap, lane, nch, bof);
        pipe.start_a();
        for (int i0 = 0; i0 < nstage; i0 += kVocStageBatch * kThreads) {
            float4 sv[kVocStageBatch];
#pragma unroll
            for (int i = 0; i < kVocStageBatch; ++i) {
                const int idx = i0 + i * kThreads + tid;
                const int row = idx / NCOL4, g = idx - row * NCOL4;
                const int t = t0 - HALO + 4 * g;
                const bool ok = idx < nstage && row < nc && t >= 0 && t < p.LSi;
                const float4 v = *reinterpret_cast<const float4*>(inb + (size_t)(c0 + (ok ? row : 0)) * p.LSi + (ok ? t : 0));
                sv[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            DSD_SB();
#pragma unroll
            for (int i = 0; i < kVocStageBatch; ++i) {
                const int idx = i0 + i * kThreads + tid;
                const int row = idx / NCOL4, g = idx - row * NCOL4;
                float4 v = sv[i];
                v.x = voc_lrelu(v.x, slope); v.y = voc_lrelu(v.y, slope); v.z = voc_lrelu(v.z, slope); v.w = voc_lrelu(v.w, slope);
                if (idx < nstage) *reinterpret_cast<float4*>(smem + row * LD + 4 * g) = v;
            }
            DSD_SB();
        }
        __syncthreads();
        pipe.start_b();
        pipe.run_blocks(acc, nch);
        __syncthreads();
    }
    if (rb >= nrb) return;
    const int q0 = t0 + wt * (32 * NB) + j;                          // input-rate sample index of frame block 0
    voc_conv_epilogue<NB, PRE>(p, acc, rb, b, q0, h, pre, rpre, spre);
}

template <int NB, int WT, int HALO = kVocHalo>
__global__ __launch_bounds__(kThreads, 2) void k_voc_conv(const VocConvParams p) {
    voc_conv_body<NB, WT, HALO>(p, blockIdx.z);
}

// The same body built LEAN for the memory-shaped layers (round 6, r6_52): the stride-2 transposed convolutions of the shipped generator
// (32 -> 16 and 16 -> 8 channels: 64 / 32 reduction rows, 3.5 us of matrix work per tile beside 72 + 64 KiB of staging and stores) moved
// 2.2 TB/s with two 240-register workgroups per CU all staging, contracting and storing in the same phase.  Half the tile (NB = 2: 256 samples),
// a 4-sample halo (two polyphase taps reach 1 sample), the LDS slab sized by the channel count and at most 128 registers: four and more
// workgroups per CU, whose phases drift apart over twice the rounds.  Same chunk order: the same bits.
constexpr int kVocHaloLean = 4;
template <int NB, int WT>
__global__ __launch_bounds__(kThreads, 3) void k_voc_conv_lean(const VocConvParams p) {
    voc_conv_body<NB, WT, kVocHaloLean, false>(p, blockIdx.z);
}

// Several INDEPENDENT convolutions of the same shape (B, Ci, rows, L, up) in ONE launch (round 6): the three parallel resblocks of a stage
// are three chains of six convolutions; level by level their convolutions depend on nothing of each other.  On the 64-channel stage of the
// shipped generator - no chain kernel: a 64-channel tile with its receptive field does not fit the LDS - a convolution is 512 workgroups =
// exactly one round of co-resident workgroups, all of them staging, contracting and storing at the same time.  Three convolutions in one
// grid (group = blockIdx.z / zc, the longest kernel first) pay the kernel boundary once, and the workgroups of the second and third round
// stage under the contractions of the round before: 125 us against 3 x 45 (profiles/r6_31_vocoder_kernel_stats.txt).
// (A variant built for three workgroups per CU - 168 registers: no operand prefetch, the phase count a compile-time 1, six staging loads in
// flight, the LDS slab sized by the channel count, 28 registers spilled - measured 3.670 against 3.651 ms for the row, r6_33: not kept.)
constexpr int kVocMultiMax = 3;
struct VocConvMulti {
    VocConvParams g[kVocMultiMax];
    int zc;                 // row-block groups of one convolution (blockIdx.z = group * zc + bz)
};

template <int NB, int WT, int HALO = kVocHalo>
__global__ __launch_bounds__(kThreads, 2) void k_voc_conv_multi(const VocConvMulti m) {
    const int grp = blockIdx.z / m.zc;
    voc_conv_body<NB, WT, HALO>(m.g[grp], (int)blockIdx.z - grp * m.zc);
}

// ------------------------------------------------------------------------------------------------------------
// Narrow layers (Co <= 16): F output samples folded into the 32 MFMA rows
// ------------------------------------------------------------------------------------------------------------
// With 8 (16) output channels a 32-row MFMA tile multiplies 75 % (50 %) zeros.  Here row = co * F + e computes output sample
// pos(c) + e * dil of channel co (F = 4 for Co <= 8, 2 for Co <= 16), column c stands for the sample group
//     pos(c) = (c / dil) * F * dil + c % dil
// (the F * dil samples [g F dil, (g+1) F dil) are the dil phases x F steps of group g: every sample is produced exactly once),
// and the reduction index is (ci, s) with the F shifted copies of the filter in the A operand (a banded Toeplitz block,
// built on the host: W'[co F + e][ci][s] = w[co][ci][s - e], K' = K + F - 1 taps):
//     out[co][pos(c) + e dil] = sum_ci sum_s W'[co F + e][ci][s] * in[ci][pos(c) + s dil - pad]
// The MFMA count per output sample drops by F * K / (K + F - 1) (3.1x for K = 11, F = 4), all 32 rows carry work.
// GemmPipe with a lane-specific B offset per frame block (pos is not linear in the column index once dil > 1).
template <int NB, int LD, typename BOff>
struct GemmPipeOff {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned aoff;
    int n;
    BOff bof;
    int boff[NB];
    float4 a[6][1];
    float b[2][4][NB];

    __device__ __forceinline__ GemmPipeOff(const float4* abase_uniform, int lane, int n_, BOff bof_, const int (&boff_)[NB])
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(abase_uniform), 0, 0x7ffffff0, 0x00020000)),
          aoff((unsigned)lane * 16u), n(n_), bof(bof_) {
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
    __device__ __forceinline__ void ldb(float (&dst)[4][NB], int it, int u) {
        const float* bp = bof(it, u);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) dst[s][nb] = bp[s * LD + boff[nb]];
    }
    __device__ __forceinline__ void pattern() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB - 1 - 2 * NB, 0);
    }
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int i = 0; i < 5; ++i) lda(a[i], i);
        DSD_SB();
        ldb(b[0], 0, 0);
        DSD_SB();
    }
    template <int I>
    __device__ __forceinline__ void step(f32x16 (&acc)[1][NB], int it) {
        lda(a[(I + 5) % 6], 6 * it + I + 5);
        ldb(b[(I + 1) & 1], it, I + 1);
        mma_chunk<1, NB>(acc, a[I % 6], b[I & 1]);
        pattern();
        DSD_SB();
    }
    __device__ __forceinline__ void run(f32x16 (&acc)[1][NB], int end) {
        for (int it = 0; 6 * it < end; ++it) {
            const int kc = 6 * it;
            step<0>(acc, it);
            if (kc + 1 >= end) break;
            step<1>(acc, it);
            if (kc + 2 >= end) break;
            step<2>(acc, it);
            if (kc + 3 >= end) break;
            step<3>(acc, it);
            if (kc + 4 >= end) break;
            step<4>(acc, it);
            if (kc + 5 >= end) break;
            step<5>(acc, it);
        }
    }
};

constexpr int kFoldCols = 256;             // columns (sample groups) per workgroup: 4 waves x 2 blocks of 32
template <int F> constexpr int fold_ld() { return 260 * F + 80; }          // LDS row stride (bound in dsv_conv1d_folded)
constexpr int kFoldMaxCi = 16;
template <int F> constexpr int fold_lds_bytes() { return kFoldMaxCi * fold_ld<F>() * 4; }

struct VocFoldParams {
    const float* in;        // [B][Ci][LS]
    const float4* wp;       // packed W' [32 rows][chunk = ci8 * KT + s][lane64] float4
    const float* bias;      // [Co] or nullptr
    float* out;             // [B][Co][LS]
    const float* res;
    const float* sum_in;
    int Ci, Co, KT, pad, dil, L, LS;         // KT = K + F - 1 (folded taps), pad = the convolution's own padding
    float pre_slope, divide;
    int act;
};

struct VocFoldB {
    const float* base;      // slab + 4 h LD + (pos(column block 0) - t_org - pad)
    int KT, dil, n, ld8;
    __device__ __forceinline__ const float* operator()(int it, int u) const {
        int kc = 6 * it + u;
        kc = (kc < n) ? kc : n - 1;
        int g;                                  // folded tap counts of the shipped kernels 3 / 7 / 11 with F = 4 and F = 2
        if (KT == 14) g = kc / 14;
        else if (KT == 10) g = kc / 10;
        else if (KT == 6) g = kc / 6;
        else if (KT == 12) g = kc / 12;
        else if (KT == 8) g = kc / 8;
        else if (KT == 4) g = kc / 4;
        else g = kc / KT;
        const int s = kc - g * KT;
        return base + g * ld8 + s * dil;
    }
};

template <int F>
__global__ __launch_bounds__(kThreads, 2) void k_voc_conv_fold(const VocFoldParams p) {
    constexpr int NB = 2, LD = fold_ld<F>(), NCOL4 = LD / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [nc8][LD]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = blockIdx.x * kFoldCols, b = blockIdx.y;
    const int dil = p.dil, fd = F * dil;
    const int t_org = ((cb / dil) * fd - kVocHalo) & ~3;             // first staged sample (may be negative), 16-byte aligned
    const int nc8 = (p.Ci + 7) / 8 * 8;
    const float* inb = p.in + (size_t)b * p.Ci * p.LS;
    const float slope = p.pre_slope;
    const int nstage = nc8 * NCOL4;
    for (int i0 = 0; i0 < nstage; i0 += kVocStageBatch * kThreads) {          // loads in flight in batches, as in k_voc_conv
        float4 sv[kVocStageBatch];
#pragma unroll
        for (int i = 0; i < kVocStageBatch; ++i) {
            const int idx = i0 + i * kThreads + tid;
            const int row = idx / NCOL4, g = idx - row * NCOL4;
            const int t = t_org + 4 * g;
            const bool ok = idx < nstage && row < p.Ci && t >= 0 && t < p.LS;
            const float4 v = *reinterpret_cast<const float4*>(inb + (size_t)(ok ? row : 0) * p.LS + (ok ? t : 0));
            sv[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        DSD_SB();
#pragma unroll
        for (int i = 0; i < kVocStageBatch; ++i) {
            const int idx = i0 + i * kThreads + tid;
            const int row = idx / NCOL4, g = idx - row * NCOL4;
            float4 v = sv[i];
            v.x = voc_lrelu(v.x, slope); v.y = voc_lrelu(v.y, slope); v.z = voc_lrelu(v.z, slope); v.w = voc_lrelu(v.w, slope);
            if (idx < nstage) *reinterpret_cast<float4*>(smem + row * LD + 4 * g) = v;
        }
        DSD_SB();
    }
    __syncthreads();
    int pos[NB], boff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int c = cb + w * (32 * NB) + nb * 32 + j;
        const int grp = c / dil;
        pos[nb] = grp * fd + (c - grp * dil);
        boff[nb] = pos[nb] - pos[0];
    }
    f32x16 acc[1][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] = 0.f;
    const int nch = (nc8 / 8) * p.KT;
    const VocFoldB bof{smem + 4 * h * LD + (pos[0] - t_org - p.pad), p.KT, dil, nch, 8 * LD};
    GemmPipeOff<NB, LD, VocFoldB> pipe(p.wp, lane, nch, bof, boff);
    pipe.start();
    pipe.run(acc, nch);
    const int Co = p.Co;
    if (dil == 1 && F == 4) {
        // rows 8 rg + 4 h + (0..3) = channel 2 rg + h, steps e = 0..3 = four consecutive samples from pos = 4 c: 16-byte accesses
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int co = 2 * rg + h, n = pos[nb];
                if (co >= Co || n >= p.LS) continue;
                const size_t o = ((size_t)b * Co + co) * p.LS + n;
                const float bv = p.bias ? p.bias[co] : 0.f;
                float4 v = get4(acc[0][nb], rg);
                v.x += bv; v.y += bv; v.z += bv; v.w += bv;
                if (p.res) { const float4 r4 = *reinterpret_cast<const float4*>(p.res + o); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
                if (p.sum_in) { const float4 s4 = *reinterpret_cast<const float4*>(p.sum_in + o); v.x = s4.x + v.x; v.y = s4.y + v.y; v.z = s4.z + v.z; v.w = s4.w + v.w; }
                if (p.divide != 1.f) { v.x = v.x / p.divide; v.y = v.y / p.divide; v.z = v.z / p.divide; v.w = v.w / p.divide; }
                if (p.act == VOC_ACT_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
                if (n + 0 >= p.L) v.x = 0.f;
                if (n + 1 >= p.L) v.y = 0.f;
                if (n + 2 >= p.L) v.z = 0.f;
                if (n + 3 >= p.L) v.w = 0.f;
                *reinterpret_cast<float4*>(p.out + o) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float rv[16], sv[16], bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {                               // all the reads first
            const int row = frag_row(r, h);
            const int co = row / F, e = row - co * F;
            const int n = pos[nb] + e * dil;
            const bool ok = co < Co && n < p.LS;
            const size_t o = ((size_t)b * Co + (ok ? co : 0)) * p.LS + (ok ? n : 0);
            bv[r] = (p.bias && ok) ? p.bias[co] : 0.f;
            rv[r] = (p.res && ok) ? p.res[o] : 0.f;
            sv[r] = (p.sum_in && ok) ? p.sum_in[o] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = frag_row(r, h);
            const int co = row / F, e = row - co * F;
            const int n = pos[nb] + e * dil;
            const bool ok = co < Co && n < p.LS;
            float v = acc[0][nb][r] + bv[r];
            if (p.res) v += rv[r];
            if (p.sum_in) v = sv[r] + v;
            if (p.divide != 1.f) v = v / p.divide;
            if (p.act == VOC_ACT_TANH) v = tanhf(v);
            if (ok) p.out[((size_t)b * Co + co) * p.LS + n] = (n < p.L) ? v : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// NSF harmonic source
// ------------------------------------------------------------------------------------------------------------
struct VocSineParams {
    const float* f0;        // [B][T] frame-rate f0 in Hz (0 = unvoiced)
    const float* rand_ini;  // [B][H] initial phases (column 0 is ignored: the fundamental starts at phase 0, source.py:59-60)
    float* sw;              // [B][H][L] sin(2 pi phase) * sine_amp
    int T, up, L, H;
    float sr, sine_amp;
};

// torch's `x % 1` on fp32 (aten remainder: fmod, then + divisor when the signs differ)
__device__ __forceinline__ float voc_mod1(float a) {
    float m = fmodf(a, 1.f);
    if (m != 0.f && m < 0.f) m += 1.f;
    return m;
}

// exclusive scan of one double per thread over the 256 threads of the block (in place in `part`)
__device__ __forceinline__ void voc_block_scan(double* part, int tid) {
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; ++i) { const double v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
}

// One workgroup per (harmonic, utterance); thread k owns the k-th contiguous piece of the sample axis.
//   rad[i]   = (f0_up[i] * (h+1) / sr) % 1, rad[0] += rand_ini                                   source.py:52-60
//   over[i]  = cumsum(rad)[i] % 1 ; shift[i] = -1 where over[i] < over[i-1]                        :64-70
//   phase[i] = cumsum(rad + shift)[i] ; sw = sin(phase * 2 * pi) * sine_amp                        :72-73, :121
// Both cumulative sums accumulate in fp64 and are rounded to fp32 per element, like aten's CPU cumsum on a float tensor.
__global__ __launch_bounds__(256) void k_voc_sine(const VocSineParams p) {
    __shared__ double part1[256];
    __shared__ double part2[256];
    const int tid = threadIdx.x, hidx = blockIdx.x, b = blockIdx.y;
    const int per = (p.L + 255) / 256;
    const int i0 = min(p.L, tid * per), i1 = min(p.L, i0 + per);
    const float mult = (float)(hidx + 1);
    const float ini = (hidx == 0) ? 0.f : p.rand_ini[(size_t)b * p.H + hidx];
    const float* f0b = p.f0 + (size_t)b * p.T;
    auto rad_at = [&](int i) -> float {
        float f = f0b[i / p.up];
        if (hidx > 0) f = f * mult;
        float r = voc_mod1(f / p.sr);
        if (i == 0) r = r + ini;
        return r;
    };
    double s1 = 0.0;
    for (int i = i0; i < i1; ++i) s1 += (double)rad_at(i);
    part1[tid] = s1;
    voc_block_scan(part1, tid);
    const double base1 = part1[tid];
    double c = base1, s2 = 0.0;
    float over_prev = voc_mod1((float)c);
    for (int i = i0; i < i1; ++i) {
        const float r = rad_at(i);
        c += (double)r;
        const float over = voc_mod1((float)c);
        const float shift = (i > 0 && (over - over_prev) < 0.f) ? -1.f : 0.f;
        s2 += (double)(r + shift);
        over_prev = over;
    }
    part2[tid] = s2;
    voc_block_scan(part2, tid);
    c = base1;
    double c2 = part2[tid];
    over_prev = voc_mod1((float)c);
    float* dst = p.sw + ((size_t)b * p.H + hidx) * p.L;
    for (int i = i0; i < i1; ++i) {
        const float r = rad_at(i);
        c += (double)r;
        const float over = voc_mod1((float)c);
        const float shift = (i > 0 && (over - over_prev) < 0.f) ? -1.f : 0.f;
        c2 += (double)(r + shift);
        over_prev = over;
        const float ph = (float)c2;
        dst[i] = sinf(ph * 2.f * 3.14159265358979323846f) * p.sine_amp;
    }
}

struct VocSourceParams {
    const float* f0;        // [B][T]
    const float* sw;        // [B][H][L]
    const float* noise;     // [B][L][H] standard normal draws (torch.randn_like(sine_waves), source.py:131)
    const float* lin_w;     // [H]   m_source.l_linear.weight[0]
    const float* lin_b;     // [1]
    float* har;             // [B][LS]
    int T, up, L, LS, H;
    float noise_std, sine_amp, voiced_threshold;
};

__global__ __launch_bounds__(256) void k_voc_source(const VocSourceParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= p.LS) return;
    float out = 0.f;
    if (i < p.L) {
        const float f = p.f0[(size_t)b * p.T + i / p.up];
        const float uv = (f > p.voiced_threshold) ? 1.f : 0.f;
        const float namp = uv * p.noise_std + ((1.f - uv) * p.sine_amp) / 3.f;      // source.py:129, evaluated left to right in fp32
        float acc = 0.f;
        for (int hh = 0; hh < p.H; ++hh) {
            const float s = p.sw[((size_t)b * p.H + hh) * p.L + i] * uv + namp * p.noise[((size_t)b * p.L + i) * p.H + hh];
            acc += s * p.lin_w[hh];
        }
        out = tanhf(acc + p.lin_b[0]);
    }
    p.har[(size_t)b * p.LS + i] = out;
}

// xs[b][c][n] = bias[c] + sum_j w[c][j] * har[b][n * s - pad + j]     (zero outside [0, Lh))
__global__ __launch_bounds__(256) void k_voc_noise_conv(const float* __restrict__ har, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, int C, int K, int s, int pad, int Lh, int LSh, int Lo, int LSo) {
    const int n = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (n >= LSo) return;
    float v = 0.f;
    if (n < Lo) {
        const float* hb = har + (size_t)b * LSh;
        const float* wc = w + (size_t)c * K;
        const int i0 = n * s - pad;
        for (int jj = 0; jj < K; ++jj) {
            const int i = i0 + jj;
            if (i >= 0 && i < Lh) v += wc[jj] * hb[i];
        }
        v += bias ? bias[c] : 0.f;
    }
    out[((size_t)b * C + c) * LSo + n] = v;
}

__global__ __launch_bounds__(256) void k_voc_pad_rows(const float* __restrict__ in, float* __restrict__ out, int L, int LS) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const size_t r = blockIdx.y;
    if (t < LS) out[r * LS + t] = (t < L) ? in[r * L + t] : 0.f;
}

}  // namespace dsd
