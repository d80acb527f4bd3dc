// dsd_loop_split.hpp - EXPERIMENT, labelled as such everywhere it surfaces (bench.py `secondary`, dsd_set_split_mode): the persistent K-step loop
// of dsd_loop.hpp with the two contractions of a residual layer on the 16-bit matrix pipe at fp32-class accuracy - the only route past the fp32
// MFMA ceiling k_loop sits at (0.98 of the bare v_mfma_f32_32x32x2_f32 stream).
//
// Two arithmetic formats (template parameter WF of k_loop_split, DSD_SPLIT_W):
//   * the PAIR format (WF = 2, the default; SplitPipeF): every fp32 operand as two fp16 planes, x = h0 + 2^-11 h1 (11 + 11 mantissa bits), a
//     product as h0 g0 + 2^-11 (h0 g1 + h1 g0): three v_mfma_f32_32x32x16_f16 per 16-deep chunk, 4 bytes per weight;
//   * three exact bf16 planes p0 + p1 + p2 (8 + 8 + 8 bits; dsd_split.hpp), a product as the six plane products with i + j <= 2, smallest first,
//     on v_mfma_f32_32x32x16_bf16, the planes on the wire (WF = 0, SplitPipeR: 6 bytes per weight) - the cross-check stream of the tests.  (A
//     third stream - fp32 on the wire, split into the same planes in registers, bit-identical and slower - is in git at 138668f.)
// Either way the products are exact in fp32 and accumulated in fp32; what is dropped is <= 2^-22 relative per product - of the order of the
// fp32 MFMA chain's own rounding (tests/test_gpu_split_loop.py measures all of them against an fp64 evaluation of the oracle).
//
// Everything else IS k_loop: a workgroup owns a 32-frame tile for the whole loop, x and the running skip sum stay in fp32 registers in
// accumulator-fragment order, the 8 halo frames travel in fp32 through the same write-through stores + per-tile phase flags + sc1 loads, the
// head (skip projection, final projection, sampler update, next input projection) is the fp32 code of k_loop verbatim.  What changes:
//   * y = x + step and the gate tile are written to LDS as 16-bit planes, frame-major [plane][frame][264] (a lane's 4 consecutive
//     channels of a frame = one 8-byte write per plane; a fragment = 8 consecutive channels = one ds_read_b128 per plane);
//   * the weights stream in 32x32x16 fragment order through register stages, for WF = 0 / 2 in CONSUMPTION order ([layer][chunk][wave]), which
//     is what lets the workgroups of an XCD fetch the stream into their L2 ahead of themselves (L2Touch below);
//   * K order of the dilated conv: the 16 centre-tap chunks first (they need no halo), then the (-dil, +dil) pairs - the loop fetches its
//     neighbours' frames under the centre taps exactly like k_loop.
// DESIGN.md section 4b has the measurements: what bounds each stream, the shader clock under each, the road from 78 k to 130 k mel-frames/s.
#pragma once
#include "dsd_loop.hpp"
#include "dsd_split.hpp"

namespace dsd {

constexpr int kSpConvCentre = 16;           // centre-tap chunks (16 channels each) of the dilated conv
constexpr int kLoopTouchLds = 1024;         // 256 bytes per wave at the START of the LDS: where the L2 touches land (never read)
constexpr int kLoopSplitLdsBytes = kLoopTouchLds + (3 * kSpYPlane + 3 * kSpGPlane) * 2 + (kMPad * 32 + 2 * kC) * (int)sizeof(float);
static_assert(3 * kSpYPlane * 2 >= kC * 32 * 4 && 3 * kSpGPlane * 2 >= kC * 32 * 4, "the head reuses the plane regions as fp32 [256][32] tiles");

// B functors over the bf16 plane tiles (plane 0; plane pl at + pl * bplane elements)
struct SConvB {
    const su16* yc; int dilrow;             // yc = this lane's (frame row kHalo + j, channel 8 h); dilrow = dil * kSpRS
    __device__ __forceinline__ const su16* at(int kc) const {
        const int idx = kc - kSpConvCentre;
        const int oc = kc * 16, oo = (idx >> 1) * 16 + ((idx & 1) ? dilrow : -dilrow);
        return yc + ((kc < kSpConvCentre) ? oc : oo);
    }
    __device__ __forceinline__ const su16* operator()(int it, int u) const { return at(6 * it + u); }
};
struct STileB {
    const su16* base; int n;
    __device__ __forceinline__ const su16* at(int kc) const { return base + ((kc < n) ? kc : n - 1) * 16; }
    __device__ __forceinline__ const su16* operator()(int it, int u) const { return at(6 * it + u); }
};

// L2 touch of the weight stream.  Every CU of an XCD walks the same 3.1 MB of planes per layer in near lock step, so what is in flight towards
// the XCD's L2 is ONE CU's register window (96 KiB) - the other 31 CUs ask for the same lines - and at ~2 us of miss latency that is what
// bounds the plane stream (the time of a layer phase does not move with the shader clock).  With the stream in
// consumption order a chunk of all four waves is 384 consecutive lines; the 128 waves of the XCD's 32 workgroups take turns fetching 64 of
// them each (one dword per line, `buffer_load_dword ... lds` into a 256-byte scratch per wave: no destination register to keep alive),
// `ahead` chunks in front of the chunk being multiplied: the window towards the L2 becomes ahead x 48 KiB for 6 instructions per chunk
// and XCD.  The instruction is invisible to hipcc's vmcnt bookkeeping, which only makes its waits stricter (an unknown younger load).
// Whose turn: instruction number n = per * chunk + t (t < per instructions per chunk) belongs to wave n mod nwx of the XCD's nwx waves
// (4 x its workgroups: blockIdx & 7 is the XCD); a wave keeps r = (q - per * chunk) mod nwx as a running counter - one subtraction and one
// wrap per step, any number of workgroups - and fetches when r < per.  Active with at least two workgroups per XCD and ahead > 0.
struct L2Touch {
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    i32x4_ rs;                   // raw buffer over all layers' stream
    int r, nwx, per;             // the running turn counter, the XCD's waves, instructions per chunk (0: off, r stays out of reach)
    unsigned ahead, gtot;        // chunks; chunks in the whole stream (64 per layer)
    unsigned lds, lane128;       // LDS byte address of this wave's scratch; lane * 128
    // the turn of the step that starts now (a fetch is due when it is < per); steps come in stream order, one call each
    __device__ __forceinline__ int next() {
        const int t = r;
        r -= per;
        if (r < 0) r += nwx;
        return t;
    }
    __device__ __forceinline__ void issue(unsigned soff) const {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"                                  // "m0 is a reserved register": hipcc keeps nothing in it across statements
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" : : "s"(lds), "v"(lane128), "s"(rs), "s"(soff) : "m0");
#pragma clang diagnostic pop
    }
};

// Operand pipeline of the split contractions with run(begin, end) like GemmPipe: STAGES register stages of the weight planes (chunk
// kc + STAGES - 1 requested while chunk kc is multiplied), B planes one chunk ahead, loads interleaved behind the first MFMAs of a step.
template <int NMB, int MB0, int STAGES, typename BOff>
struct SplitPipeR {
    static_assert(STAGES == 3 || STAGES == 6, "register rotation period is 6");
    static constexpr int P = 6;
    static constexpr int kChunkBytes = 4 * 12288;     // consumption order: a chunk of all four waves is 48 KiB of consecutive lines
    static constexpr int kTouchPer = kChunkBytes / 8192;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned aoff;
    int n;
    BOff bof;
    int bplane;
    L2Touch& tc;
    unsigned gq;                 // index of this pipe's chunk 0 in the stream of all layers' chunks (64 per layer) + the touch's lead
    uint4 a[STAGES][NMB][3];
    sbf16x8 b[2][3];

    static __device__ __forceinline__ const uint4* wbase(const uint4* p, int l, int w, int) { return p + ((size_t)l * (64 * 4) + w) * 768; }
    __device__ __forceinline__ SplitPipeR(const uint4* wave_base, int lane, int n_, BOff bof_, int bplane_, L2Touch& tc_, unsigned gbase_)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wave_base), 0, 0x7ffffff0, 0x00020000)), aoff((unsigned)lane * 16u), n(n_), bof(bof_),
          bplane(bplane_), tc(tc_), gq(gbase_ + tc_.ahead) {}
    // L2 touch (see L2Touch): wave q of the XCD fetches 64 lines of the chunk `ahead` steps in front of everybody, when it is its turn
    __device__ __forceinline__ void touch(int kc) {
        const int t = tc.next();
        if (t < kTouchPer) {
            unsigned g = gq + (unsigned)kc;
            if (g >= tc.gtot) g -= tc.gtot;                                  // (the counter runs on: every chunk stays covered exactly once)
            tc.issue(g * (unsigned)kChunkBytes + (unsigned)t * 8192u);
        }
    }
    __device__ __forceinline__ void lda(uint4 (&dst)[NMB][3], int kc) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int kcc = (kc < n) ? kc : n - 1;                             // prefetches past the end re-read the last chunk
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)aoff, kcc * kChunkBytes + ((MB0 + mb) * 3 + pl) * 1024, 0);
                dst[mb][pl] = make_uint4(v.x, v.y, v.z, v.w);
            }
    }
    __device__ __forceinline__ void ldb(sbf16x8 (&dst)[3], int it, int u) {
        const su16* bp = bof(it, u);
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
        ldb(b[0], 0, 0);
        DSD_SB();
    }
    template <int I>
    __device__ __forceinline__ void step(f32x16 (&acc)[NMB], int it) {
        touch(6 * it + I);
        lda(a[(I + STAGES - 1) % STAGES], 6 * it + I + STAGES - 1);
        ldb(b[(I + 1) & 1], it, I + 1);
        constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};      // smallest plane products first
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sbf16x8, a[I % STAGES][mb][TI[q]]), b[I & 1][TJ[q]], acc[mb], 0, 0, 0);
        pattern();
        DSD_SB();
    }
    template <int BEGIN, int END>
    __device__ __forceinline__ void run(f32x16 (&acc)[NMB]) { run(acc, BEGIN, END); }
    __device__ __forceinline__ void finish(f32x16 (&)[NMB]) {}
    // chunks [begin, end); begin a multiple of 6
    __device__ __forceinline__ void run(f32x16 (&acc)[NMB], int begin, int end) {
        for (int it = begin / 6; 6 * it < end; ++it) {
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

// PAIR format (WF = 2): every fp32 operand as TWO fp16 planes, x = h0 + 2^-11 h1 (sp_split2h: 11 + 11 mantissa bits, the second plane scaled so
// that it is a normal fp16), a product as h0 g0 + 2^-11 (h0 g1 + h1 g0) - three v_mfma_f32_32x32x16_f16 per 16-deep chunk instead of six bf16
// ones, 4 bytes per weight on the wire instead of 6, two LDS reads per chunk instead of three.  The products are exact in fp32 (11 x 11
// bits); what is dropped is h1 g1 (<= 2^-24 relative) and the operands' bits below 2^-22 - of the order of ONE rounding of the fp32 MFMA
// chain, which rounds after every one of its K accumulations (tests/test_gpu_split_loop.py measures both against an fp64 evaluation).
// The cross terms go to a second set of accumulators (their common factor 2^-11 is applied once, in finish()), so a wave holds
// 2 x NMB x 16 accumulator registers.  Four register stages of 8 KiB per wave; blocks in consumption order like the plane stream
// ([chunk][wave][mb][plane 2][lane] x 8 fp16: a chunk = 32 KiB = 256 lines = four L2-touch instructions).
typedef _Float16 sh8 __attribute__((ext_vector_type(8)));
template <int NMB, int MB0, typename BOff>
struct SplitPipeF {
    static constexpr int P = 4, S = 4;
    static constexpr int kBlockBytes = 8192, kChunkBytes = 4 * kBlockBytes, kTouchPer = kChunkBytes / 8192;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned aoff;
    int n;
    BOff bof;
    int bplane;
    L2Touch& tc;
    unsigned gq;
    uint4 a[S][NMB][2];
    sh8 b[2][2];
    f32x16 cross[NMB];           // h0 g1 + h1 g0, in units of 2^-11

    static __device__ __forceinline__ const uint4* wbase(const uint4* p, int l, int w, int) { return p + ((size_t)l * (64 * 4) + w) * (kBlockBytes / 16); }
    __device__ __forceinline__ SplitPipeF(const uint4* wave_base, int lane, int n_, BOff bof_, int bplane_, L2Touch& tc_, unsigned gbase_)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wave_base), 0, 0x7ffffff0, 0x00020000)), aoff((unsigned)lane * 16u), n(n_), bof(bof_),
          bplane(bplane_), tc(tc_), gq(gbase_ + tc_.ahead) {
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) cross[mb][r] = 0.f;
    }
    __device__ __forceinline__ void touch(int kc) {
        const int t = tc.next();
        if (t < kTouchPer) {
            unsigned g = gq + (unsigned)kc;
            if (g >= tc.gtot) g -= tc.gtot;
            tc.issue(g * (unsigned)kChunkBytes + (unsigned)t * 8192u);
        }
    }
    template <int ST>
    __device__ __forceinline__ void lda(int kc) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int kcc = (kc < n) ? kc : n - 1;
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)aoff, kcc * kChunkBytes + ((MB0 + mb) * 2 + p) * 1024, 0);
                a[ST][mb][p] = make_uint4(v.x, v.y, v.z, v.w);
            }
    }
    __device__ __forceinline__ void ldb(sh8 (&dst)[2], int kc) {
        const su16* bp = bof.at(kc);
#pragma unroll
        for (int p = 0; p < 2; ++p) dst[p] = __builtin_bit_cast(sh8, *reinterpret_cast<const uint4*>(bp + p * bplane));
    }
    __device__ __forceinline__ void pattern() {
#pragma unroll
        for (int i = 0; i < 2 * NMB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (3 * NMB - 2 * NMB - 2 > 0) __builtin_amdgcn_sched_group_barrier(0x008, 3 * NMB - 2 * NMB - 2, 0);
    }
    __device__ __forceinline__ void start_a() {
        lda<0>(0);
        lda<1>(1);
        lda<2>(2);
        DSD_SB();
    }
    __device__ __forceinline__ void start_b() {
        ldb(b[0], 0);
        DSD_SB();
    }
    template <int I>
    __device__ __forceinline__ void step(f32x16 (&acc)[NMB], int kc) {
        touch(kc);
        lda<(I + S - 1) % S>(kc + S - 1);
        ldb(b[(I + 1) & 1], kc + 1);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
            cross[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sh8, a[I % S][mb][1]), b[I & 1][0], cross[mb], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
            cross[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sh8, a[I % S][mb][0]), b[I & 1][1], cross[mb], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sh8, a[I % S][mb][0]), b[I & 1][0], acc[mb], 0, 0, 0);
        pattern();
        DSD_SB();
    }
    template <int I, int N>
    __device__ __forceinline__ void steps(f32x16 (&acc)[NMB], int kc0) {
        step<I>(acc, kc0 + I);
        if constexpr (I + 1 < N) steps<I + 1, N>(acc, kc0);
    }
    template <int BEGIN, int END>
    __device__ __forceinline__ void run(f32x16 (&acc)[NMB]) {
        static_assert(BEGIN % P == 0 && END > BEGIN, "a segment starts on a period");
        constexpr int kFull = (END - BEGIN) / P, kTail = (END - BEGIN) - kFull * P;
        if constexpr (kFull > 0)
            for (int kc0 = BEGIN; kc0 < BEGIN + kFull * P; kc0 += P) steps<0, P>(acc, kc0);
        if constexpr (kTail > 0) steps<0, kTail>(acc, BEGIN + kFull * P);
    }
    // acc = h0 g0 sum + 2^-11 x cross sum
    __device__ __forceinline__ void finish(f32x16 (&acc)[NMB]) {
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = fmaf(cross[mb][r], kPairInv, acc[mb][r]);
    }
};

// which pipe a loop instantiation streams its weights through: WF = 2 the pair format (4 stages), 0 the bf16 planes (3 stages)
template <int WF, int NMB, int MB0, typename BOff>
struct SplitPipeSel { typedef SplitPipeR<NMB, MB0, 3, BOff> type; };
template <int NMB, int MB0, typename BOff>
struct SplitPipeSel<2, NMB, MB0, BOff> { typedef SplitPipeF<NMB, MB0, BOff> type; };

// Cached loads through a buffer descriptor over a WAVE-UNIFORM base (SGPRs) + a 32-bit lane offset: no 64-bit per-lane address lives in
// VGPRs.  In this kernel the arch-VGPR file is full (three stages of weight planes), and a spilled pointer is a scratch reload = a vector
// memory op whose s_waitcnt vmcnt(0) would drain the whole weight prefetch in the middle of a contraction.
__device__ __forceinline__ float4 ld16_u(const void* base_uniform, int byte_off) {
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_uniform), 0, 0x7ffffff0, 0x00020000);
    const f32x4_ f = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ float ld4_u(const void* base_uniform, int byte_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_uniform), 0, 0x7ffffff0, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// four consecutive channels of one frame -> three 8-byte writes (one per plane)
__device__ __forceinline__ void sp_store4(su16* plane0, int plane_elems, int off, const float4& v) {
    typedef su16 su16x4 __attribute__((ext_vector_type(4)));
    su16 a0[4], a1[4], a2[4];
    sp_split3(v.x, a0[0], a1[0], a2[0]);
    sp_split3(v.y, a0[1], a1[1], a2[1]);
    sp_split3(v.z, a0[2], a1[2], a2[2]);
    sp_split3(v.w, a0[3], a1[3], a2[3]);
    *reinterpret_cast<su16x4*>(plane0 + off) = su16x4{a0[0], a0[1], a0[2], a0[3]};
    *reinterpret_cast<su16x4*>(plane0 + plane_elems + off) = su16x4{a1[0], a1[1], a1[2], a1[3]};
    *reinterpret_cast<su16x4*>(plane0 + 2 * plane_elems + off) = su16x4{a2[0], a2[1], a2[2], a2[3]};
}

// four consecutive channels of one frame in the pair format -> two 8-byte writes; `big` collects the largest magnitude stored (range guard)
constexpr unsigned kLoopRangeBit = 2u;      // in the loop's timeout word: an activation left fp16's range
__device__ __forceinline__ void sp_store4h(su16* plane0, int plane_elems, int off, const float4& v, float& big) {
    typedef su16 su16x4 __attribute__((ext_vector_type(4)));
    big = fmaxf(big, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    su16 a0[4], a1[4];
    sp_split2h(v.x, a0[0], a1[0]);
    sp_split2h(v.y, a0[1], a1[1]);
    sp_split2h(v.z, a0[2], a1[2]);
    sp_split2h(v.w, a0[3], a1[3]);
    *reinterpret_cast<su16x4*>(plane0 + off) = su16x4{a0[0], a0[1], a0[2], a0[3]};
    *reinterpret_cast<su16x4*>(plane0 + plane_elems + off) = su16x4{a1[0], a1[1], a1[2], a1[3]};
}
template <int WF>
__device__ __forceinline__ void sp_store4_wf(su16* plane0, int plane_elems, int off, const float4& v, float& big) {
    if constexpr (WF == 2) sp_store4h(plane0, plane_elems, off, v, big);
    else sp_store4(plane0, plane_elems, off, v);
}

struct LoopSplitParams {
    LoopParams lp;              // everything k_loop takes (w1p / w2p unused here)
    const uint4* w1c;           // ALL weights of the loop in consumption order [L][64 = 48 conv (centre taps first) + 16 out-projection chunks][w4][12 KiB of
                                // bf16 planes / 8 KiB of fp16 planes]
    const uint4* w2s;           // out-projection weights: w1c + 48 chunks
    unsigned wl_bytes;          // bytes of that buffer (the L2 touch's buffer bound)
    int touch_ahead;            // chunks the L2 touch runs in front (0 = off)
};

template <int MODE, int WF>
__global__ __launch_bounds__(kThreads, 1) void k_loop_split(const LoopSplitParams ps) {
    typedef typename SplitPipeSel<WF, 4, 0, SConvB>::type Pipe1;
    typedef typename SplitPipeSel<WF, 4, 0, STileB>::type Pipe2;
    typedef typename SplitPipeSel<WF, 2, 2, STileB>::type Pipe2L;
    constexpr int kSeg0 = Pipe1::P;            // chunks before the neighbour flags are tested (a multiple of the pipe's period)
    const LoopParams& p = ps.lp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    su16* yp = reinterpret_cast<su16*>(smem + kLoopTouchLds / 4);   // [3][48][264] y planes; head: scaled skip sum [256][32] fp32
    su16* gp = yp + 3 * kSpYPlane;                          // [3][32][264] gate planes; head: relu(skip_projection) [256][32] fp32
    float* xt = reinterpret_cast<float*>(gp + 3 * kSpGPlane);   // [96][32] spec tile of the in-projection
    float* dsbuf = xt + kMPad * 32;                         // [2][256]
    float* ytile = reinterpret_cast<float*>(yp);
    float* gtile = reinterpret_cast<float*>(gp);

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tl;
    {
        const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
        const int q = p.n_tiles >> 3, r = p.n_tiles & 7;
        tl = xcd * q + min(xcd, r) + k;
    }
    L2Touch tc;
    {
        const unsigned long long wb = (unsigned long long)ps.w1c;
        const int xcd = (int)(blockIdx.x & 7), nwx = 4 * ((p.n_tiles - xcd + 7) >> 3), q = 4 * (int)(blockIdx.x >> 3) + w;
        const bool en = Pipe1::kTouchPer > 0 && ps.touch_ahead > 0 && nwx >= 8;
        tc.rs = L2Touch::i32x4_{(int)(unsigned)wb, (int)(unsigned)((wb >> 32) & 0xffffu), (int)ps.wl_bytes, 0x00020000};
        tc.ahead = (unsigned)ps.touch_ahead;
        tc.nwx = nwx;
        tc.per = en ? Pipe1::kTouchPer : 0;
        if (en) {                                                // the first step multiplies chunk 0: its fetch is for chunk `ahead`
            int r0 = (q - Pipe1::kTouchPer * ps.touch_ahead) % nwx;
            tc.r = r0 < 0 ? r0 + nwx : r0;
        } else tc.r = 1 << 20;
        tc.gtot = (unsigned)p.L * 64u;
        tc.lds = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float*)smem + (unsigned)w * 256u;
        tc.lane128 = (unsigned)lane * 128u;
    }
    const int tile = p.tile_base + tl;
    const int b = tile / p.ntile32, tn = tile - b * p.ntile32, t0 = tn * 32;
    const bool has_left = tn > 0, has_right = tn + 1 < p.ntile32;
    const int M = p.head.M, T = p.T;
    const bool in_t = t0 + j < T;

    float4 xq[2][4], skp[2][4];
    const int ch0 = 64 * w + 4 * h;
    float big = 0.f;                    // pair format: the largest |activation| this lane has written as planes (checked once per evaluation)

    auto timed_out = [&]() -> bool { return __hip_atomic_load((gu32*)p.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; };

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
        __syncthreads();
    };

    for (int idx = tid; idx < kMPad * 32; idx += kThreads) {
        const int m = idx >> 5, t = t0 + (idx & 31);
        xt[idx] = (m < M && t < T) ? p.spec0[((size_t)b * M + m) * T + t] : 0.f;
    }
    dsbuf[tid] = p.ds_table[(size_t)p.eval_t[0] * p.L * kC + tid];
    __syncthreads();
    inproj_to_xq();

    auto publish_issue = [&](unsigned phase) {
        float* hb = p.halo + ((size_t)(phase & 1) * p.ntiles_total + tile) * (2 * kC * 8);
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(hb, 0, 0x7ffffff0, 0x00020000);
        if (j < 8 || j >= 24) {
            // (an opaque zero defined HERE keeps the offset arithmetic in this block: hoisted out of the loops, the eight offsets would be
            // loop invariants that live - spilled to scratch - across every contraction)
            int oz;
            asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
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
    auto publish_finish = [&](unsigned phase) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gu32*)(p.flags + tile), phase + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto publish = [&](unsigned phase) { publish_issue(phase); publish_finish(phase); };
    const bool stamp = p.dbg != nullptr;
#define LOOP_STAMP(i) do { if (stamp && ph == (unsigned)p.dbg_phase && lane == 0) p.dbg[((size_t)tl * 4 + w) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

    unsigned ph = 0;
    publish(0);
    for (int e = 0; e < p.n_evals; ++e) {
        const int t_e = p.eval_t[e];
        for (int l = 0; l < p.L; ++l, ++ph) {
            const bool last = (l == p.L - 1);
            const float* dsl = dsbuf + (ph & 1) * kC;
            LOOP_STAMP(0);
            // the shader clock under THIS instruction mix: s_memtime beside the constant 100 MHz s_memrealtime, one evaluation apart
            if (stamp && lane == 0 && (ph == (unsigned)p.dbg_phase || ph == (unsigned)p.dbg_phase + (unsigned)p.L)) {
                unsigned long long* d = p.dbg + ((size_t)tl * 4 + w) * 16 + (ph == (unsigned)p.dbg_phase ? 8 : 10);
                d[0] = __builtin_amdgcn_s_memtime();
                d[1] = __builtin_amdgcn_s_memrealtime();
            }
            // (c) weight planes of the conv: requested before anything of this phase exists
            const SConvB bof1{yp + (kHalo + j) * kSpRS + 8 * h, (int)p.dil[l] * kSpRS};
            Pipe1 pipe1(Pipe1::wbase(ps.w1c, l, w, 48), lane, 48, bof1, kSpYPlane, tc, (unsigned)l * 64u);
            pipe1.start_a();
            // (b) own frames of y = x + step_proj (zero at frames >= T) as planes: the lane's 32 channels of frame j
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = ch0 + 32 * mb + 8 * q;
                    const float4 d = *reinterpret_cast<const float4*>(dsl + c);
                    sp_store4_wf<WF>(yp, kSpYPlane, (kHalo + j) * kSpRS + c, fm_add_masked(xq[mb][q], d, in_t), big);
                }
            __syncthreads();
            LOOP_STAMP(1);
            // (d1) neighbour flags, tested behind the first centre chunks
            unsigned fv = 0xffffffffu;
            if (lane < 2) {
                const bool have = lane ? has_right : has_left;
                if (have) fv = __hip_atomic_load((const gu32*)(p.flags + tile + (lane ? 1 : -1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            DSD_SB();
            f32x16 acc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
            float4 cpv[4][4];
            pipe1.start_b();
            pipe1.template run<0, kSeg0>(acc);
            // (d2) both neighbours have published phase ph?
            if (fv < ph + 1u) {
                const gu32* f = (const gu32*)(p.flags + tile + (lane ? 1 : -1));
                for (int spins = 0;; ++spins) {
                    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ph + 1u) break;
                    if ((spins & 255) == 255 && timed_out()) break;
                    if (spins >= kLoopSpinLimit) { __hip_atomic_fetch_or((gu32*)p.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // (fetch_or: a concurrent range bit survives)
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            // (e1) the neighbours' frames: 8 frames x 256 channels per side, two float4 per thread and side (sc1 loads)
            float4 hv[2][2];
            {
                const float* hbase = p.halo + (size_t)(ph & 1) * p.ntiles_total * (2 * kC * 8);
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const bool have = side ? has_right : has_left;
                    const int off = (((tile + (side ? 1 : -1)) * 2 + (side ? 0 : 1)) * (8 * kC) + 4 * tid) * 4;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        hv[side][g] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (have) hv[side][g] = ld16_sc1(hbase, off + g * (4 * kC * 4));
                    }
                }
            }
            DSD_SB();
            pipe1.template run<kSeg0, 12>(acc);
            // (e2) halo rows of y as planes: float4 index tid + 256 g = (frame 4 g + tid / 64, channels 4 (tid % 64) ..)
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
                        sp_store4_wf<WF>(yp, kSpYPlane, ((side ? kHalo + 32 : 0) + f) * kSpRS + c, fm_add_masked(hv[side][g], d, have && t < T), big);
                    }
                }
            }
            __syncthreads();
            LOOP_STAMP(2);
            pipe1.template run<12, 24>(acc);
            {
                const float4* cpl = p.cp + (size_t)l * p.cp_lstride + ((size_t)tile * 4 + w) * (4 * 4 * 64);      // wave-uniform
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) cpv[mb][q] = ld16_u(cpl, ((mb * 4 + q) * 64 + lane) * 16);
            }
            DSD_SB();
            pipe1.template run<24, 48>(acc);
            pipe1.finish(acc);
            float ds_next = 0.f;
            {
                const bool more = !last || (e + 1 < p.n_evals);
                const int tn_ = last ? p.eval_t[min(e + 1, p.n_evals - 1)] : t_e, ln_ = last ? 0 : l + 1;
                if (more) ds_next = ld4_u(p.ds_table + ((size_t)tn_ * p.L + ln_) * kC, tid * 4);
            }
            const STileB bof2{gp + j * kSpRS + 8 * h, 16};
            // gate (net.py:73-74) in registers -> gate planes
            auto do_gate = [&]() {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float g4[4];
#pragma unroll
                        for (int ee = 0; ee < 4; ++ee) {
                            const int r = 4 * q + ee;
                            const float vg = f4at(cpv[pr][q], ee), vf = f4at(cpv[pr + 2][q], ee);
                            g4[ee] = sigmoid_f(acc[pr][r] + vg) * tanh_f(acc[pr + 2][r] + vf);
                        }
                        sp_store4_wf<WF>(gp, kSpGPlane, j * kSpRS + ch0 + 32 * pr + 8 * q, make_float4(g4[0], g4[1], g4[2], g4[3]), big);
                    }
            };
            LOOP_STAMP(3);
            const uint4* w2l = Pipe2::wbase(ps.w2s, l, w, 16);
            if (!last) {
                Pipe2 pipe2(w2l, lane, 16, bof2, kSpGPlane, tc, (unsigned)l * 64u + 48u);
                pipe2.start_a();
                do_gate();
                __syncthreads();
                LOOP_STAMP(4);
                f32x16 acc2[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;
                float4 bq[2][4];
                pipe2.start_b();
                pipe2.template run<0, kSeg0>(acc2);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[mb][q] = ld16_u(p.b2raw + (size_t)l * 2 * kC, (ch0 + 32 * mb + 8 * q) * 4);
                DSD_SB();
                pipe2.template run<kSeg0, 16>(acc2);
                pipe2.finish(acc2);
                LOOP_STAMP(5);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = get4(acc2[mb], q), x = xq[mb][q], bv = bq[mb][q];
                        constexpr float kInvSqrt2 = 1.0f / 1.41421354f;
                        xq[mb][q] = make_float4((x.x + (v.x + bv.x)) * kInvSqrt2, (x.y + (v.y + bv.y)) * kInvSqrt2,
                                                (x.z + (v.z + bv.z)) * kInvSqrt2, (x.w + (v.w + bv.w)) * kInvSqrt2);
                    }
                LOOP_STAMP(6);
                dsbuf[((ph + 1u) & 1u) * kC + tid] = ds_next;
                publish_issue(ph + 1u);
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = get4(acc2[2 + ms], q), s = skp[ms][q];
                        skp[ms][q] = (l == 0) ? a : make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                    }
                publish_finish(ph + 1u);
                LOOP_STAMP(7);
            } else {
                Pipe2L pipe2(w2l, lane, 16, bof2, kSpGPlane, tc, (unsigned)l * 64u + 48u);
                pipe2.start_a();
                do_gate();
                __syncthreads();
                f32x16 acc2[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;
                pipe2.start_b();
                pipe2.template run<0, 16>(acc2);
                pipe2.finish(acc2);
                dsbuf[((ph + 1u) & 1u) * kC + tid] = ds_next;
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = get4(acc2[ms], q), s = skp[ms][q];
                        skp[ms][q] = (l == 0) ? a : make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                    }
            }
        }

        // ---- head (net.py:126-129) + sampler epilogue + the next evaluation's input projection: the fp32 code of k_loop ---------------------
        if constexpr (WF == 2) {
            // range guard of the pair format: fp16 holds |x| <= 65504.  Loud like a timeout: the word makes every workgroup finish early and
            // return NaN tiles, the host reports DSD_ERR_RANGE (k_latch_tmo, check_sticky)
            if (!(big <= 65504.f)) __hip_atomic_fetch_or((gu32*)p.tmo, kLoopRangeBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        HeadParams hp = p.evals[e];
        const bool fuse = (e + 1 < p.n_evals);
        float* stile = ytile;               // [256][32]
        float* htile = gtile;               // [256][32]
        float* ptile = xt;                  // [96][32]
        __syncthreads();
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
        const float* hl = htile + 4 * h * 32 + j;
        GemmPipe<1, 1, 32, 192, 6, TileB> pipe_o(p.head.woutp + (size_t)min(w, 2) * 64, lane, 32, TileB{hl, 8 * 32, 32});
        if (w < 3) pipe_o.start_a();
        __syncthreads();
        if (w < 3) {
            f32x16 acc[1][1];
#pragma unroll
            for (int q = 0; q < 4; ++q) set4(acc[0][0], q, p.head.boutp[(w * 2 + h) * 4 + q]);
            pipe_o.start_b();
            pipe_o.run(acc, 0, 32);
            const int t = t0 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * w + frag_row(r, h);
                const bool ok = (m < M) && (t < T);
                const size_t idx = ((size_t)b * M + m) * T + t;
                float xn = 0.f;
                if (ok) {
                    HeadPre pre;
                    head_prefetch<MODE>(hp, idx, pre);
                    xn = head_apply<MODE>(hp, acc[0][0][r], idx, pre);
                }
                ptile[m * 32 + j] = ok ? xn : 0.f;
            }
        }
        __syncthreads();
        if (fuse) { inproj_to_xq(); publish(ph); }
    }
#undef LOOP_STAMP
    if (timed_out()) {
        float* xo = const_cast<float*>(p.spec0);
        for (int idx = tid; idx < M * 32; idx += kThreads) {
            const int m = idx >> 5, t = t0 + (idx & 31);
            if (t < T) xo[((size_t)b * M + m) * T + t] = __builtin_nanf("");
        }
    }
}

}  // namespace dsd
