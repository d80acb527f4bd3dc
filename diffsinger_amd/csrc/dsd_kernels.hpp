// dsd_kernels.hpp - hand-written gfx950 (CDNA4, wave64) kernels for the DiffNet denoiser + sampler epilogues.
//
// What is computed, and where the reference computes it (paths relative to the reference root):
//   k_layer       one ResidualBlock (usr/diff/net.py:66-78) for a tile of frames, fully fused:
//                 y = x + step_proj  ->  3-tap dilated conv as ONE K=768 contraction on fp32 MFMA, accumulators
//                 pre-loaded with the hoisted conditioner projection  ->  sigmoid*tanh gate in registers
//                 ->  1x1 output projection on fp32 MFMA  ->  (x + res)/sqrt(2) and skip accumulation.
//   k_condproj    conditioner_projection of every layer (net.py:68), hoisted out of the K-step loop.
//   k_inproj      input_projection + ReLU (net.py:116-118).
//   k_head        sum(skip)/sqrt(L) -> skip_projection + ReLU -> output_projection (net.py:126-129) with the
//                 sampler arithmetic fused as epilogue: p_sample (usr/diff/shallow_diffusion_tts.py:134-166) or
//                 p_sample_plms (:168-204), and optionally the NEXT evaluation's input projection.
//   k_step_embed / k_small_gemm   SinusoidalPosEmb + MLP + per-layer diffusion_projection (net.py:37-44,
//                 94-98, 67; Mish usr/diff/diffusion.py:68-70) tabulated for every integer t once per model.
//   k_qsample / k_norm_spec / k_denorm_spec   shallow_diffusion_tts.py:206-211, :278-282 (+ layout change :252/:271)
//
// Design notes (details in DESIGN.md):
//   * all contractions use v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fmaf chain); D[i][j]: i = output
//     channel, j = frame.  A (weights) is pre-packed in "fragment order" so that one global_load_dwordx4 per
//     lane (1 KiB contiguous per wave) feeds 4 consecutive MFMAs; within an 8-deep k-chunk the lane half h
//     supplies k = 4h + s at MFMA s, for A and B alike (a consistent permutation of the reduction order).
//   * B (activations) lives in LDS as [channel][frame] with the frame axis contiguous: lanes 0-31 read 32
//     consecutive floats (conflict-free ds_read_b32), the dilated taps are plain column offsets into a tile
//     with an 8-frame zero halo on each side, exactly the conv's zero padding of y = x + step (net.py:69-71).
//   * a wave owns gate rows AND the matching filter rows, so the gate never leaves registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace dsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kC = 256;        // residual_channels == encoder_hidden on this build
constexpr int kHalo = 8;       // max dilation supported (dilation_cycle_length <= 4)
constexpr int kMPad = 96;      // mel bins padded to 3 MFMA row blocks
constexpr int kThreads = 256;  // 4 waves, one per SIMD
constexpr int kWeightSlack = 8192;   // float4 of slack behind every packed weight buffer (A prefetch overrun: <= 5 chunks x 4 KiB + row blocks)

// C/D fragment map of v_mfma_f32_32x32x2_f32: lane (j = lane & 31, h = lane >> 5), register r in [0,16)
// holds D[row = (r & 3) + 8 (r >> 2) + 4 h][col = j].
__device__ __forceinline__ int frag_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__device__ __forceinline__ void set4(f32x16& v, int q, float4 x) {
    v[4 * q + 0] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
}
__device__ __forceinline__ float4 get4(const f32x16& v, int q) {
    return make_float4(v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// One 8-deep k-chunk: 4 MFMA steps x NMB row blocks x NB frame blocks, operands already in registers.
template <int NMB, int NB>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[NMB][NB], const float4 (&a)[NMB], const float (&b)[4][NB]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) {
            const float av = (s == 0) ? a[mb].x : (s == 1) ? a[mb].y : (s == 2) ? a[mb].z : a[mb].w;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma32(av, b[s][nb], acc[mb][nb]);
        }
    }
}

// K loop over `n` chunks, software-pipelined by hand.  Left alone, hipcc sinks the prefetch loads down to their first
// use and exposes the L2 latency of every chunk, so every step is fenced with sched_barrier and, inside a step, the
// loads of the NEXT operands are interleaved one-by-one behind the first MFMAs with sched_group_barrier (an MFMA
// occupies the pipe for 64 cycles; the wave issues its global / LDS loads in that shadow).
//   A fragments stream straight from global/L2 into VGPRs (no LDS: a wave's weight rows are not shared with the
//   other waves), three register stages = two chunks (2 x 16*NB MFMAs >= 2048 cycles) ahead of use;
//   B fragments (4*NB ds_read_b32 per chunk) two register stages = one chunk ahead.
// ap: lane-adjusted pointer to chunk 0 / row block 0; ASTRIDE: float4 between consecutive chunks; row block mb
// at + mb*64.  bof(kc): this lane's LDS B pointer for chunk kc (row 4h of the chunk, column j); LD: LDS row stride.
// Prefetches past the end are clamped to the last chunk (valid addresses, results unused).
// run(acc, begin, end) may be called several times with begin a multiple of 6 (the register rotation period) so that
// unrelated loads can be issued between two parts of one K loop without draining the pipeline.
#define DSD_SB() __builtin_amdgcn_sched_barrier(0)
// STAGES = register stages of the A stream (3 or 6): chunk kc+STAGES-1 is requested while chunk kc is multiplied.
// All workgroups of an XCD walk the same weight stream in lock-step, so every chunk is a first touch of that L2:
// the latency to hide is the Infinity-Cache / HBM one (~1-2 us), not an L2 hit - hence 5 chunks (>= 5k cycles) of
// distance for the 16-MFMA chunks of the 32-frame kernels.
// MBS: distance (in 32-row blocks of the packed stream) between the NMB row blocks a wave multiplies - 1: consecutive blocks; 2: a gate
// block and its filter block / a residual block and its skip block (the row-split kernels of dsd_lat.hpp).
// BV: the B tile is FRAME-MAJOR ([frame][k], k contiguous; dsd_loop_fm.hpp): the four k values of a chunk a lane needs (k = 4h + s) are
// ONE aligned ds_read_b128 instead of four ds_read_b32 at a stride of LD.
// VOL (round 6, the vocoder's one-convolution kernel): the 4 NB ds_read_b32 of a chunk as VOLATILE accesses.  hipcc pairs plain ones into
// ds_read2_b32, whose 8-bit offsets do not reach across a tile row wider than 1 KiB: one v_add_u32 per row for a new base, 3-5 vector-ALU
// instructions per chunk - each paid in matrix time beside an fp32 MFMA (tools/mfma_filler_probe.hip).  Volatile accesses are not merged: the
// row offset sits in the 16-bit offset field of each ds_read_b32.  Same values, same order of arithmetic.
template <int NMB, int NB, int LD, int ASTRIDE, int STAGES, typename BOff, int MBS = 1, bool BV = false, bool VOL = false>
struct GemmPipe {
    static_assert(STAGES == 3 || STAGES == 6, "register rotation period is 6");
    static_assert(!BV || NB == 1, "frame-major B tiles are one 32-frame block wide");
    __amdgpu_buffer_rsrc_t rsrc;   // buffer descriptor over this wave's A stream (wave-uniform base = chunk 0 / row block 0)
    unsigned aoff;                 // this lane's byte offset inside a 1 KiB fragment row (lane * 16)
    int n;
    BOff bof;
    float4 a[STAGES][NMB];
    float b[2][4][NB];

    __device__ __forceinline__ GemmPipe(const float4* abase_uniform, int lane, int n_, BOff bof_)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(abase_uniform), 0, 0x7ffffff0, 0x00020000)),
          aoff((unsigned)lane * 16u), n(n_), bof(bof_) {}

    // the same pipeline over another A stream (k_condproj walks several layers: the next layer's first chunks are requested before this layer's
    // results are stored)
    __device__ __forceinline__ void rebase(const float4* abase_uniform) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(abase_uniform), 0, 0x7ffffff0, 0x00020000);
    }
    // buffer_load_dwordx4 v, voffset(lane), rsrc, soffset(chunk) offset:imm(row block): the chunk walk is one SALU
    // value, there is no per-lane 64-bit address arithmetic in the loop.  Prefetches run up to STAGES-1 chunks past
    // the end of the stream: the weight buffers carry that much slack (kWeightSlack) and the values are never used.
    __device__ __forceinline__ void lda(float4 (&dst)[NMB], int kc) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int soff = kc * (ASTRIDE * 16);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)aoff + mb * (MBS * 1024), soff, 0);
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 f = __builtin_bit_cast(f32x4, v);      // whole-vector cast (element-wise bit_cast of v.x miscompiles)
            dst[mb] = make_float4(f.x, f.y, f.z, f.w);
        }
    }
    // B fragment of chunk 6*it + u (u is a compile-time constant at every call site, so the functor can fold the
    // part of the address that depends on u into the ds_read immediate)
    __device__ __forceinline__ void ldb(float (&dst)[4][NB], int it, int u) {
        const float* bp = bof(it, u);
        if constexpr (BV) {
            const float4 v = *reinterpret_cast<const float4*>(bp);
            dst[0][0] = v.x; dst[1][0] = v.y; dst[2][0] = v.z; dst[3][0] = v.w;
        } else if constexpr (VOL) {
            typedef const volatile __attribute__((address_space(3))) float lds_cvf;
            lds_cvf* vp = (lds_cvf*)bp;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) dst[s][nb] = vp[s * LD + nb * 32];
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) dst[s][nb] = bp[s * LD + nb * 32];
        }
    }
    // interleave: {1 MFMA, 1 global load} x NMB, {1 MFMA, 1 LDS read} x 2NB, then the remaining MFMAs
    __device__ __forceinline__ void pattern() {
#pragma unroll
        for (int i = 0; i < NMB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        constexpr int NDS = BV ? 1 : 2 * NB;      // LDS reads of a chunk
#pragma unroll
        for (int i = 0; i < NDS; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, VOL ? 2 : 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NMB * NB - NMB - NDS, 0);
    }
    // The A prefetch depends on nothing the kernel computes: start_a() may be issued long before the B tile is
    // ready (before staging / before the gate), so the first-touch latency of the weight stream is off the critical path.
    template <int FROM = 0, int TO = STAGES - 1>
    __device__ __forceinline__ void start_a() {
#pragma unroll
        for (int i = FROM; i < TO; ++i) lda(a[i], i);
        DSD_SB();
    }
    __device__ __forceinline__ void start_b() {
        ldb(b[0], 0, 0);
        DSD_SB();
    }
    __device__ __forceinline__ void start() { start_a(); start_b(); }
    template <int I>
    __device__ __forceinline__ void step(f32x16 (&acc)[NMB][NB], int it) {
        lda(a[(I + STAGES - 1) % STAGES], 6 * it + I + STAGES - 1);
        ldb(b[(I + 1) & 1], it, I + 1);
        mma_chunk<NMB, NB>(acc, a[I % STAGES], b[I & 1]);
        pattern();
        DSD_SB();
    }
    // chunks [begin, end); begin must be a multiple of 6 (register rotation period); may be called repeatedly to
    // place unrelated loads between two parts of one K loop without draining the pipeline
    __device__ __forceinline__ void run(f32x16 (&acc)[NMB][NB], int begin, int end) {
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
    // The same walk for a RUN-TIME chunk count with the whole groups of six as ONE basic block (one branch per six chunks) and the tail
    // behind it.  With `end` unknown at compile time every `break` above ends a block; hipcc then sinks the prefetch loads of the steps
    // across the blocks (seen in the disassembly: the six A loads of an iteration issued together behind its fifth chunk, s_waitcnt
    // vmcnt(0) in front of chunks) and the fences inside a step no longer order anything.
    __device__ __forceinline__ void run_blocks(f32x16 (&acc)[NMB][NB], int end) {
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

// B functor of a plain [k][frame] LDS tile: chunk kc at base + kc * 8 rows (clamped: the prefetch of the chunk behind
// the last one must stay inside the tile)
struct TileB {
    const float* base; int row8, n;
    __device__ __forceinline__ const float* operator()(int it, int u) const {
        const int kc = 6 * it + u;
        return base + ((kc < n) ? kc : n - 1) * row8;
    }
};

template <int NMB, int NB, int LD, int ASTRIDE, typename BOff>
__device__ __forceinline__ void gemm_k(f32x16 (&acc)[NMB][NB], const float4* __restrict__ abase_uniform, int lane, int n, BOff bof) {
    GemmPipe<NMB, NB, LD, ASTRIDE, 6, BOff> pipe(abase_uniform, lane, n, bof);
    pipe.start();
    pipe.run(acc, 0, n);
}
template <int NMB, int NB, int LD, int ASTRIDE, bool VOL = false, typename BOff>
__device__ __forceinline__ void gemm_k_blocks(f32x16 (&acc)[NMB][NB], const float4* __restrict__ abase_uniform, int lane, int n, BOff bof) {
    GemmPipe<NMB, NB, LD, ASTRIDE, 6, BOff, 1, false, VOL> pipe(abase_uniform, lane, n, bof);
    pipe.start();
    pipe.run_blocks(acc, n);
}

// K order of the dilated conv (K = 3 taps x 256 channels = 96 chunks of 8 channels x 1 tap): CENTRE TAP FIRST.
//   chunk kc <  32: tap 1 (column offset 0) of channel group k8 = kc
//   chunk kc >= 32: channel group k8 = (kc - 32) / 2, tap 0 (offset -dil) for even kc - 32, tap 2 (offset +dil) for odd
// The centre tap reads no halo column, so the first 32 chunks (a third of the contraction, ~33 k MFMA cycles) need nothing from the
// neighbour tiles: the persistent loop fetches its neighbours' columns UNDER them instead of in front of the conv (dsd_loop.hpp), and
// within the centre region channels are consumed in ascending order (k_layer's progressive staging).  Every fp32 kernel and the weight
// packers share this order, so they stay bit-identical to each other.
constexpr int kConvCentre = 32;
__host__ __device__ __forceinline__ int conv_chunk(int k8, int tap) { return (tap == 1) ? k8 : kConvCentre + 2 * k8 + (tap >> 1); }

// B functor of the dilated conv over a y tile [channel][LD] whose column kHalo is frame 0 of the tile: yc = this lane's pointer to
// (row 4 h, column kHalo + j); kbase = first chunk of this wave's K range (a multiple of 6).
// BF (branch-free): with a kbase known only at run time (the K-half waves of k_lat_conv<8>) the two returns become a BRANCH per chunk; the
// prefetch loads of the GemmPipe steps are then sunk across the block boundaries to their first use and the A stream runs one chunk ahead
// instead of five (s_waitcnt vmcnt(0) in front of every chunk, found in the ISA at the end of round 2).  The select form keeps a chunk one block.
template <int LD, bool BF = false>
struct ConvB {
    const float* yc; int dil, kbase;
    __device__ __forceinline__ const float* operator()(int it, int u) const {
        const int kc = kbase + 6 * it + u;
        if constexpr (BF) {
            const int idx = kc - kConvCentre;
            const int oc = kc * (8 * LD), oo = (idx >> 1) * (8 * LD) + ((idx & 1) ? dil : -dil);
            return yc + ((kc < kConvCentre) ? oc : oo);
        } else {
            if (kc < kConvCentre) return yc + kc * (8 * LD);
            const int idx = kc - kConvCentre;
            return yc + (idx >> 1) * (8 * LD) + ((idx & 1) ? dil : -dil);
        }
    }
};

// ------------------------------------------------------------------------------------------------------------
// residual layer
// ------------------------------------------------------------------------------------------------------------
// x lives in a TILE-MAJOR internal layout [B * ntile32][C][32] (one 32 KiB block per 32-frame tile): a wave's 64
// rows of a tile are 8 KiB contiguous, so tile loads and stores are whole-line, 1 KiB-per-instruction accesses.
struct LayerParams {
    const float* x_in;      // [tiles][C][32]
    float* x_out;           // [tiles][C][32]
    const float4* w1p;      // dilated conv, packed [w4][kc96 = 3 * k8 + tap][mb4][lane64] float4
    const float4* w2p;      // output projection, packed [w4][kc32][mb4][lane64]
    const float* b2;        // output projection bias [2C] (residual half used here; the skip half is summed over
                            // layers once and added in the head)
    const float4* cp;       // hoisted conditioner projection (+ both biases) [tile32][w4][mb4][q4][lane64]
    float4* skip;           // running skip sum (without biases) [tile32][w4][mb2][q4][lane64]
    const float* ds;        // step-projection table for this layer: ds[t * ds_tstride + c]
    const int* t_dev;       // per-utterance step index, or nullptr -> t_uniform
    int t_uniform, ds_tstride;
    int T, TS, ntile32, tiles_per_utt, dil, first;
    int wt_stores;          // 1: x_out / skip leave through write-through (sc1) stores
    int xcd_q, xcd_r;       // XCD-aware workgroup map (1-D grid): total workgroups = 8 * xcd_q + xcd_r; xcd_q < 0 -> 2-D grid
    unsigned long long* dbg;   // optional per-wave phase timestamps [block][wave][8] (s_memtime), nullptr in production
};

template <int NB>
constexpr int layer_lds_bytes() { return (kC * (32 * NB + 2 * kHalo) + kC * 32 * NB) * (int)sizeof(float); }

// 16-byte store, optionally WRITE-THROUGH (sc1): the line leaves the XCD's L2 as it is written instead of staying dirty until
// the kernel boundary, where the runtime's agent-scope release would have to write back every dirty line of the launch first
// (16.8 MB of x / skip per layer launch = 2.8-3.8 us on top of the 1.45 us boundary, MI355X_MICROARCH.md "boundary" row).
// base: WAVE-UNIFORM pointer (it becomes the SGPR buffer descriptor), idx: this lane's float4 index behind it.
template <bool WT>
__device__ __forceinline__ void store16(float4* base_uniform, int idx, const float4& v) {
    if (WT) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const f32x4_ f = {v.x, v.y, v.z, v.w};
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7ffffff0, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, f), r, idx * 16, 0, 16);     // aux 16 = sc1
    } else {
        base_uniform[idx] = v;
    }
}

// A wave-uniform pointer, made PROVABLY uniform: where hipcc moves a scalar address chain to the vector ALU (k_trb_fused_w: `base + w * 4096`
// ended up in VGPRs) a buffer descriptor built from it is "divergent" and EVERY load through it becomes a waterfall loop - four readfirstlanes,
// two compares and a branch per load, 59 instead of 37 cycles per MFMA in that kernel's convolution (profiles/r5_25_trb_timeline.txt).
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

// 4-byte WRITE-THROUGH store (sc1) at float index idx behind a wave-uniform base: outputs of the training kernels that the NEXT kernel reads
// leave the L2 as they are written instead of in the release at the kernel boundary
__device__ __forceinline__ void store4_wt(float* base_uniform, int idx, float v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, idx * 4, 0, 16);      // aux 16 = sc1
}

__device__ __forceinline__ float f4at(const float4& v, int e) { return (e == 0) ? v.x : (e == 1) ? v.y : (e == 2) ? v.z : v.w; }

// What the TRAINING forward (csrc/train_kernels.hpp) keeps of a layer for its backward pass; the
// inference kernels instantiate the body with TRAIN = false and none of this exists in their code.
struct LayerSave {
    float* y_cm;            // y = x + step projection, channel-major rows of y_rs floats: y_cm[(b * C + row) * y_rs + t], zero for t >= T (B operand
                            // of the conv weight gradient; the rows carry zero pads on both sides so that a tap shift never leaves a row)
    float4* a_frag;         // pre-activation of the gate (conv + conditioner projection + biases) in accumulator-fragment order, the layout of `cp`
    int y_rs;
};

template <int NB, bool LAST, bool TRAIN>
__device__ __forceinline__ void layer_body(const LayerParams& p, const LayerSave& sv) {
    constexpr int LD = 32 * NB + 2 * kHalo;   // y tile row stride (floats), 16-byte multiple
    constexpr int GLD = 32 * NB;              // gate tile row stride
    constexpr int TILE = kC * 32;             // floats per 32-frame x tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem;
    float* gtile = smem + kC * LD;

    const int tid = threadIdx.x;
    // Workgroup -> tile map.  Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  With the
    // plain map neighbouring tiles land on different XCDs and every halo line (the neighbour's first / last 8 columns, one
    // cache line per channel row) is fetched from the fabric a second time by the other L2 (~17 MB per launch at 8192
    // frames, PMC FETCH_SIZE).  XCD-aware map: XCD x owns the CONTIGUOUS tile range [x q + min(x, r), ...), so both
    // neighbours of a tile sit behind the same L2.  Placement is a speed matter only (results do not depend on it).
    int b, tn;
    if (p.xcd_q >= 0) {
        const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
        const int tl = xcd * p.xcd_q + min(xcd, p.xcd_r) + k;
        b = tl / p.tiles_per_utt;
        tn = tl - b * p.tiles_per_utt;
    } else {
        b = blockIdx.y; tn = blockIdx.x;     // grid = (tiles per utterance, utterances)
    }
    const int tile0 = b * p.ntile32 + tn * NB;
    const int ntv = min(NB, p.ntile32 - tn * NB);         // valid 32-frame tiles of this workgroup
    const float* __restrict__ xt = p.x_in + (size_t)tile0 * TILE;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, tsa = 0, tsb = 0;
    if (p.dbg) ts0 = __builtin_amdgcn_s_memtime();

    // 1. Progressive staging.  y = x + step_proj goes to LDS (zero outside [0,T): the conv's zero padding applies to y,
    //    net.py:69-71) in four 64-channel quarters; the conv's K order (chunk = 3 * k8 + tap) consumes the channels in
    //    that same order, so only the FIRST quarter (8 KiB of the tile + 4 KiB of halo) and the first weight chunks are on
    //    the critical path of the launch - the start of a launch is a chip-wide ingest burst (~10 B/cycle/CU) and the
    //    whole 48 KiB would cost ~9 k cycles before the first MFMA.  Quarter q+1 is written (one barrier) 6 chunks before
    //    the loop's B prefetch first touches it.
    //    Thread tid reads bytes [16 tid, 16 tid + 16) of every 4 KiB slab: the tile is one linear 32 KiB block
    //    (row = slab * 32 + tid / 8, columns 4 (tid & 7) .. +3); halo piece q covers rows 64 q + tid / 4.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xt), 0, 0x7ffffff0, 0x00020000);
    float4 xv[NB][8], hv[4];
    float dv[8], dh[4];
    auto load_x = [&](int slab) {
#pragma unroll
        for (int nbi = 0; nbi < NB; ++nbi) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, tid * 16, ((nbi < ntv) ? nbi : 0) * (TILE * 4) + slab * 4096, 0);
            const f32x4 f = __builtin_bit_cast(f32x4, v);
            xv[nbi][slab] = make_float4(f.x, f.y, f.z, f.w);
        }
    };
    load_x(0);
    load_x(1);
    DSD_SB();
    const int lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = tn * 32 * NB;
    const int tstep = p.t_dev ? p.t_dev[b] : p.t_uniform;
    const float* __restrict__ dsl = p.ds + (size_t)tstep * p.ds_tstride;
    const int hpart = tid & 3;
    const bool hleft = hpart < 2;
    const bool hhave = hleft ? (tn * NB > 0) : (tn * NB + NB < p.ntile32);
    auto load_h = [&](int q) {
        const int row = 64 * q + (tid >> 2);
        const float* src = hleft ? xt - TILE + row * 32 + 24 + 4 * hpart : xt + NB * TILE + row * 32 + 4 * (hpart - 2);
        hv[q] = *reinterpret_cast<const float4*>(hhave ? src : xt + row * 32);
        dh[q] = dsl[row];
        dv[2 * q] = dsl[(2 * q) * 32 + (tid >> 3)];
        dv[2 * q + 1] = dsl[(2 * q + 1) * 32 + (tid >> 3)];
    };
    load_h(0);
    DSD_SB();

    // conv operand pipeline.  K order of the dilated conv: centre tap first (ConvB above): chunk kc < 32 = centre tap of channel
    // group kc, then the two outer taps of every group.  Its weight stream does not depend on x: the first chunks are requested
    // right behind the first quarter.
    const int dil = p.dil;
    const ConvB<LD> bof1{ytile + 4 * h * LD + kHalo + j, dil, 0};
    GemmPipe<4, NB, LD, 256, (NB == 1 ? 6 : 3), ConvB<LD>> pipe1(p.w1p + (size_t)w * (96 * 256), lane, 96, bof1);
    constexpr int ST1 = (NB == 1 ? 6 : 3);
    pipe1.template start_a<0, 2>();
    if (p.dbg) tsa = __builtin_amdgcn_s_memtime();

    auto write_quarter = [&](int q) {
#pragma unroll
        for (int sl = 2 * q; sl < 2 * q + 2; ++sl)
#pragma unroll
            for (int nbi = 0; nbi < NB; ++nbi) {
                const int row = sl * 32 + (tid >> 3), c4 = tid & 7;
                const int t = t0 + 32 * nbi + 4 * c4;
                const bool ok = nbi < ntv;
                float4 v = xv[nbi][sl];
                const float d = dv[sl];
                v.x = (ok && t + 0 < p.T) ? v.x + d : 0.f;
                v.y = (ok && t + 1 < p.T) ? v.y + d : 0.f;
                v.z = (ok && t + 2 < p.T) ? v.z + d : 0.f;
                v.w = (ok && t + 3 < p.T) ? v.w + d : 0.f;
                *reinterpret_cast<float4*>(ytile + row * LD + kHalo + 32 * nbi + 4 * c4) = v;
                if (TRAIN && ok) store16<true>(reinterpret_cast<float4*>(sv.y_cm + (size_t)b * kC * sv.y_rs), (row * sv.y_rs + t) >> 2, v);
            }
        {
            const int row = 64 * q + (tid >> 2);
            const int t = hleft ? t0 - kHalo + 4 * hpart : t0 + 32 * NB + 4 * (hpart - 2);
            const int col = hleft ? 4 * hpart : kHalo + 32 * NB + 4 * (hpart - 2);
            float4 v = hv[q];
            const float d = dh[q];
            v.x = (hhave && t + 0 < p.T) ? v.x + d : 0.f;
            v.y = (hhave && t + 1 < p.T) ? v.y + d : 0.f;
            v.z = (hhave && t + 2 < p.T) ? v.z + d : 0.f;
            v.w = (hhave && t + 3 < p.T) ? v.w + d : 0.f;
            *reinterpret_cast<float4*>(ytile + row * LD + col) = v;
        }
    };
    write_quarter(0);
    if (p.dbg) tsb = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (p.dbg) ts1 = __builtin_amdgcn_s_memtime();

    // 2. dilated conv: K = 3 taps x 256 channels = 96 chunks of 8; tap k reads column offset (k-1)*dil.  The centre-tap region
    //    (chunks 0..31) consumes channel quarter q in chunks [8 q, 8 q + 8): quarter q + 1 is written at the 6-chunk boundary in
    //    front of its first chunk (the B prefetch runs one chunk ahead) and requested one segment earlier; the outer taps and the
    //    halo columns (chunks >= 32) find the whole tile in place.  The hoisted conditioner projection (+ conv bias + cond bias)
    //    is fetched halfway through the loop, when the memory system is idle, and added once the loop is done.
    f32x16 acc[4][NB];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
    float4 cpv[4][NB][4];
    {
        // every later quarter is requested one segment (18-24 chunks, ~20 k cycles) before it is written: few loads at a
        // time, none of them in front of the first MFMA
        auto load_quarter = [&](int q) { load_x(2 * q); load_x(2 * q + 1); load_h(q); DSD_SB(); };
        auto& pipe = pipe1;
        pipe.template start_a<2, ST1 - 1>();
        load_quarter(1);
        if (NB == 1) { load_quarter(2); load_quarter(3); }      // 32-frame tiles: a 6-chunk segment (6 k cycles) is too short to hide a
        pipe.start_b();                                         // quarter's latency at launch start - request all of them behind the first
        pipe.run(acc, 0, 6);
        write_quarter(1);
        __syncthreads();
        if (NB != 1) load_quarter(2);
        pipe.run(acc, 6, 12);
        write_quarter(2);
        __syncthreads();
        if (NB != 1) load_quarter(3);
        pipe.run(acc, 12, 18);
        write_quarter(3);
        __syncthreads();
        pipe.run(acc, 18, 48);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float4* cpl = p.cp + ((size_t)(tile0 + ((nb < ntv) ? nb : 0)) * 4 + w) * (4 * 4 * 64) + lane;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) cpv[mb][nb][q] = cpl[(mb * 4 + q) * 64];   // tiles past the end: unused columns
        }
        DSD_SB();
        pipe.run(acc, 48, 96);
    }
    if (p.dbg) ts2 = __builtin_amdgcn_s_memtime();

    // out-proj weight stream: first chunks requested before the gate arithmetic
    constexpr int NMB2 = LAST ? 2 : 4;
    constexpr int MB0 = LAST ? 2 : 0;
    const float* gl = gtile + 4 * h * GLD + j;
    const TileB bof2{gl, 8 * GLD, 32};
    GemmPipe<NMB2, NB, GLD, 256, (NB == 1 ? 6 : 3), TileB> pipe2(p.w2p + (size_t)w * (32 * 256) + MB0 * 64, lane, 32, bof2);
    pipe2.start_a();

    if (TRAIN) {
        // the backward pass differentiates the gate from the pre-activation a = conv + cp: saved in fragment order (1 KiB per instruction)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (nb >= ntv) continue;
            float4* al = sv.a_frag + ((size_t)(tile0 + nb) * 4 + w) * (4 * 4 * 64);      // wave-uniform; write-through like x_out / skip
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 c = cpv[mb][nb][q], a = get4(acc[mb][nb], q);
                    store16<true>(al, (mb * 4 + q) * 64 + lane, make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w));
                }
        }
    }
    // 3. gate in registers: rows [64w,64w+64) are gates, their partners (row blocks 2,3) the filters (net.py:73-74)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float vg = f4at(cpv[pr][nb][r >> 2], r & 3), vf = f4at(cpv[pr + 2][nb][r >> 2], r & 3);
                const float g = sigmoid_f(acc[pr][nb][r] + vg) * tanh_f(acc[pr + 2][nb][r] + vf);
                gtile[(64 * w + 32 * pr + frag_row(r, h)) * GLD + 32 * nb + j] = g;
            }
    __syncthreads();
    if (p.dbg) ts3 = __builtin_amdgcn_s_memtime();

    // 4. output projection (K = 256): row blocks 0,1 = residual rows, 2,3 = skip rows.  The last layer's
    //    residual half is dead (net.py:126 only reads the skips) and is not computed.  x (row layout of the
    //    transposed epilogue), the residual bias and the running skip sum are fetched behind the first chunks.
    f32x16 acc2[NMB2][NB];
#pragma unroll
    for (int m = 0; m < NMB2; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][nb][r] = 0.f;
    float4 xrow[NB][8], skp[2][NB][4];
    float brow[8];
    {
        auto& pipe = pipe2;
        pipe.start_b();
        pipe.run(acc2, 0, 6);
        if (!LAST) {
#pragma unroll
            for (int it = 0; it < 8; ++it) brow[it] = p.b2[64 * w + it * 8 + (lane >> 3)];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    xrow[nb][it] = reinterpret_cast<const float4*>(xt + ((nb < ntv) ? nb : 0) * TILE + 64 * w * 32)[it * 64 + lane];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) {
                const float4* sl = p.skip + (((size_t)(tile0 + ((nb < ntv) ? nb : 0)) * 4 + w) * 2 + ms) * (4 * 64) + lane;
                const bool keep = !p.first;     // layer 0 starts the sum (select, not multiply: the buffer may hold anything)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = sl[q * 64];
                    skp[ms][nb][q] = make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
                }
            }
        DSD_SB();
        pipe.run(acc2, 6, 32);
    }
    if (p.dbg) ts4 = __builtin_amdgcn_s_memtime();

    // 5. epilogue: the residual is transposed through this wave's slice of the (now free) y tile into row layout,
    //    x' = (x + (res + bias)) / sqrt(2)  (net.py:78) is formed there and the tile-major x_out is written as
    //    contiguous 1 KiB-per-instruction float4 stores; the skip sum is written back in fragment order.
    float* tw = ytile + w * (64 * 32 * NB);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (nb >= ntv) continue;
        if (!LAST) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tw[nb * 2048 + (32 * mb + frag_row(r, h)) * 32 + j] = acc2[mb][nb][r];
            __builtin_amdgcn_wave_barrier();
            float* __restrict__ xo = p.x_out + (size_t)(tile0 + nb) * TILE + 64 * w * 32;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const float4 v = reinterpret_cast<const float4*>(tw + nb * 2048)[it * 64 + lane];
                const float4 x = xrow[nb][it];
                const float bv = brow[it];
                // (x + residual) / sqrt(2) as a * (1/b) in fp32 - what torch's GPU kernel does for a scalar divisor
                // (the CPU kernel divides: <= 1 ulp apart); a correctly-rounded fp32 divide is ~10 VALU ops per element
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
            float4* sl = p.skip + (((size_t)(tile0 + nb) * 4 + w) * 2 + ms) * (4 * 64);      // wave-uniform
            const int m = (LAST ? 0 : 2) + ms;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = get4(acc2[m][nb], q), s = skp[ms][nb][q];
                const float4 sv = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                if (p.wt_stores) store16<true>(sl, q * 64 + lane, sv);
                else store16<false>(sl, q * 64 + lane, sv);
            }
        }
    }
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + (((size_t)b * p.tiles_per_utt + tn) * 4 + w) * 8;
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3; d[4] = ts4; d[5] = __builtin_amdgcn_s_memtime(); d[6] = tsa; d[7] = tsb;
    }
}

template <int NB, bool LAST>
__global__ __launch_bounds__(kThreads, 1) void k_layer(const LayerParams p) {
    layer_body<NB, LAST, false>(p, LayerSave{nullptr, nullptr, 0});
}
// the same layer in a training forward: additionally writes y and the gate pre-activation (LayerSave)
template <bool LAST>
__global__ __launch_bounds__(kThreads, 1) void k_tr_layer(const LayerParams p, const LayerSave sv) {
    layer_body<1, LAST, true>(p, sv);
}

// ------------------------------------------------------------------------------------------------------------
// hoisted conditioner projection: cp[l] = Wc_l * cond + bc_l + bd_l, written in accumulator-fragment order
// ------------------------------------------------------------------------------------------------------------
struct CondProjParams {
    const float* condT;     // [B][H=256][TS], zero for t >= T
    const float4* wcp;      // [L][w4][kc32][mb4][lane64]
    const float4* b1p;      // [L][w4][mb4][h2][q4]  (dilated_conv.bias + conditioner_projection.bias)
    float4* cp;             // [L][ntiles][w4][mb4][q4][lane64], or (wino) [L][ntiles][w4][accumulator 2][rb 8][lane64]
    int TS, ntile32, ntiles_total;
    int L;                  // layers
    int wino;               // 1: the INITIAL VALUES of the Winograd loop's two accumulator sets (dsd_loop_wino.hpp), in its accumulator order: for
                            // the pair p of a layer with dilation d, (cp[tE] + cp[tO]) / 2 and (cp[tE] - cp[tO]) / 2 - the loop's output transform
                            // (sum / difference of the sets) turns them back into cp[tE] and cp[tO]
    unsigned char dil[64];
};

// Grid (tiles, G): workgroup (tile, y) projects the tile for the layers y, y + G, y + 2 G, ... < L.  G = L is one layer per workgroup (small
// batches: as many workgroups as possible).  Round 6 (profiles/r6_42_train_glue_trace.txt: 430 us of a 5 ms training step at 0.65 of the
// matrix peak): for batches that fill the chip anyway the host launches G = the period of the dilation cycle, so that a workgroup stages its
// conditioner tile - and, Winograd order, re-lays it for the ONE dilation its layers share - once for L / G layers instead of once per layer
// (5 120 x 32 KiB of staging -> 1 024 x 32 KiB at L = 20), with the weight stream of the next layer requested while the stores of this one
// drain.  The same contraction per (layer, tile): the same bits.
__global__ __launch_bounds__(kThreads, 2) void k_condproj(const CondProjParams p) {
    constexpr int LD = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int b = tile / p.ntile32, t0 = (tile % p.ntile32) * 32;
    const float* src = p.condT + (size_t)b * kC * p.TS + t0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * kThreads + tid, row = idx >> 3, q = idx & 7;
        *reinterpret_cast<float4*>(smem + row * LD + 4 * q) =
            *reinterpret_cast<const float4*>(src + (size_t)row * p.TS + 4 * q);
    }
    // the bias rows of this workgroup's layers (2 KiB each: [w4][mb4][h2][q4] float4) behind the tile: the accumulators' initial values then come
    // from LDS, not from a global load in front of every contraction
    for (int i = tid, li = 0; ; i += kThreads) {
        li = i >> 7;
        const int l = (int)blockIdx.y + li * (int)gridDim.y;
        if (l >= p.L) break;
        reinterpret_cast<float4*>(smem + kC * LD)[i] = p.b1p[(size_t)l * 128 + (i & 127)];
    }
    __syncthreads();
    if (p.wino) {
        // the conditioner tile in the loop's pair order: column c < 16 = (cond[tE(c)] + cond[tO(c)]) / 2, column 16 + c the half difference - by
        // linearity the contraction then yields the half sum / half difference of the projection.  One thread per channel row.  (Every layer of
        // this workgroup has the dilation of its first one: the host's launch guarantees it.)
        const int e = __builtin_ctz((unsigned)p.dil[blockIdx.y]), d = 1 << e;
        float* row = smem + tid * LD;
        float v[32], o[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(row + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            // tE = ((c >> e) << (e + 1)) | (c & (d - 1)): a run-time dilation - select among the four compile-time forms
            const int tE0 = 2 * c, tE1 = ((c >> 1) << 2) | (c & 1), tE2 = ((c >> 2) << 3) | (c & 3), tE3 = ((c >> 3) << 4) | (c & 7);
            const float a = (e == 0) ? v[tE0] : (e == 1) ? v[tE1] : (e == 2) ? v[tE2] : v[tE3];
            const float b = (e == 0) ? v[tE0 + 1] : (e == 1) ? v[tE1 + 2] : (e == 2) ? v[tE2 + 4] : v[tE3 + 8];
            o[c] = 0.5f * (a + b);
            o[16 + c] = 0.5f * (a - b);
        }
        (void)d;
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(row + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        __syncthreads();
    }
    // (Winograd order: columns 16 .. 31 are differences of two frames - the biases cancel there)
    const float bsel = (p.wino && j >= 16) ? 0.f : 1.f;
    const float* cl = smem + 4 * h * LD + j;
    const float4* bias_l = reinterpret_cast<const float4*>(smem + kC * LD) + (w * 4) * 8 + h * 4;      // this lane's slice of layer slot 0
    const int G = (int)gridDim.y;
    GemmPipe<4, 1, LD, 256, 6, TileB> pipe(p.wcp + ((size_t)blockIdx.y * 4 + w) * (32 * 256), lane, 32, TileB{cl, 8 * LD, 32});
    pipe.start_a();
#pragma unroll 1
    for (int l = blockIdx.y, li = 0; l < p.L; l += G, ++li) {
        f32x16 acc[4][1];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = bias_l[li * 128 + mb * 8 + q];
                set4(acc[mb][0], q, make_float4(bv.x * bsel, bv.y * bsel, bv.z * bsel, bv.w * bsel));
            }
        pipe.start_b();
        pipe.run(acc, 0, 32);
        // the next layer's weight stream is requested HERE, in front of this layer's stores: its first-touch latency ran in front of every
        // contraction with both resident workgroups of a CU in the same phase (384 us for 279 us of matrix time, r6_45)
        if (l + G < p.L) {
            pipe.rebase(p.wcp + ((size_t)(l + G) * 4 + w) * (32 * 256));
            pipe.start_a();
        }
        if (p.wino) {
            // Winograd loop: lane (pair pr, k group g) of v_mfma_f32_16x16x4_f32 holds, for accumulator set i (0: half sum, 1: half difference) and row
            // block rb (16 rows; 0-3 gate, 4-7 filter), rows 4 g + {0..3}.  This lane's float4 (mb, q) = rows 32 (mb & 1) + 8 q + 4 h + {0..3} of the
            // wave's 64 gate (mb < 2) / filter rows in column j = 16 i + pr: rb = 2 (mb & 1) + (q >> 1) (+ 4), g = 2 (q & 1) + h.
            const int hf = j >> 4, pr = j & 15;
            float4* out = p.cp + (((size_t)l * p.ntiles_total + tile) * 4 + w) * (2 * 8 * 64) + (size_t)hf * (8 * 64) + pr;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) out[(2 * (mb & 1) + (q >> 1) + 4 * (mb >> 1)) * 64 + 16 * (2 * (q & 1) + h)] = get4(acc[mb][0], q);
        } else {
            float4* out = p.cp + (((size_t)l * p.ntiles_total + tile) * 4 + w) * (4 * 4 * 64) + lane;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) out[(mb * 4 + q) * 64] = get4(acc[mb][0], q);
        }
    }
}

// grid.y of a k_condproj launch: a multiple of the period of the dilation cycle (every layer of a workgroup then shares its dilation) when the
// batch fills the chip with that many workgroups per tile, else one layer per workgroup
constexpr int kCondProjMaxLayers = 10;      // layers per workgroup: 32 KiB of tile + 20 KiB of bias rows stay under the 64 KiB default limit
inline int condproj_groups(const unsigned char* dil, int L, int ntiles) {
    const char* e = getenv("DSD_CP_GROUPS");                        // "layer": one layer per workgroup whatever the batch (the A/B of the tests)
    if (e && e[0] == 'l') return L;
    const int forced = e ? atoi(e) : 0;                             // a number: that many groups if the dilation cycle allows it (measurements)
    int per = L;
    for (int c = 1; c <= 8 && c < L; c *= 2) {
        bool ok = true;
        for (int l = c; l < L && ok; ++l) ok = dil[l] == dil[l - c];
        if (ok) { per = c; break; }
    }
    // the smallest multiple of the period that gives every workgroup the same number of layers and the chip two workgroups per CU
    if (forced > 0) return (forced < L && forced % per == 0 && L / forced <= kCondProjMaxLayers) ? forced : L;
    for (int G = per; G < L; G += per)
        if (L % G == 0 && (long)ntiles * G >= 512 && L / G <= kCondProjMaxLayers) return G;
    return L;
}
// dynamic LDS of a k_condproj launch with G groups: the conditioner tile + 2 KiB of bias rows per layer of a workgroup
inline size_t condproj_lds(int L, int G) { return (size_t)kC * 32 * 4 + (size_t)((L + G - 1) / G) * 2048; }

// ------------------------------------------------------------------------------------------------------------
// input projection + ReLU on a [kMPad][32] tile held in LDS (rows >= M are zero)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inproj_tile(const float* ptile, const float4* __restrict__ winp,
                                            const float4* __restrict__ binp, int nk, float* __restrict__ xo_tile,
                                            int w, int lane) {
    const int j = lane & 31, h = lane >> 5;
    f32x16 acc[2][1];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) set4(acc[mb][0], q, binp[((w * 2 + mb) * 2 + h) * 4 + q]);
    const float4* ap = winp + (size_t)w * nk * 128;
    const float* pl = ptile + 4 * h * 32 + j;
    gemm_k<2, 1, 32, 128>(acc, ap, lane, nk, TileB{pl, 8 * 32, nk});
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            xo_tile[(64 * w + 32 * mb + frag_row(r, h)) * 32 + j] = fmaxf(acc[mb][0][r], 0.f);
}

struct InProjParams {
    const float* spec;      // [B][M][T]
    float* x_out;           // [tiles][C][32] (tile-major)
    const float4* winp;     // [w4][nk][mb2][lane64]
    const float4* binp;     // [w4][mb2][h2][q4]
    int nk, M, T, TS, ntile32;
};

__global__ __launch_bounds__(kThreads, 2) void k_inproj(const InProjParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.ntile32, t0 = (blockIdx.x % p.ntile32) * 32;
    for (int idx = tid; idx < kMPad * 32; idx += kThreads) {
        const int m = idx >> 5, t = t0 + (idx & 31);
        smem[idx] = (m < p.M && t < p.T) ? p.spec[((size_t)b * p.M + m) * p.T + t] : 0.f;
    }
    __syncthreads();
    inproj_tile(smem, p.winp, p.binp, p.nk, p.x_out + (size_t)blockIdx.x * (kC * 32), w, lane);
}

// ------------------------------------------------------------------------------------------------------------
// head: skip reduction scale -> skip_projection + ReLU -> output_projection -> sampler epilogue (-> next in-proj)
// ------------------------------------------------------------------------------------------------------------
// Counter-based N(0,1) draws for p_sample's `noise_like` (usr/diff/shallow_diffusion_tts.py:38-41,:165) when the caller passes no
// explicit noise: Philox4x32-10 keyed by the 64-bit seed, counter = (element index lo, hi, step, 0), Box-Muller on the first two
// output words.  One draw per (step, element): the value does not depend on tiling, chunking or which kernel evaluates it.
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned step, unsigned long long idx) {
    unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = step, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u1 = (float)((c0 >> 8) + 1u) * 5.9604644775390625e-8f;      // (0, 1]
    const float u2 = (float)(c1 >> 8) * 5.9604644775390625e-8f;             // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

enum HeadMode { HEAD_EPS = 0, HEAD_DDPM = 1, HEAD_PLMS = 2 };
enum PlmsOrder { PLMS_RAW = 0, PLMS_HEUN = 1, PLMS_AB2 = 2, PLMS_AB3 = 3, PLMS_AB4 = 4 };

struct HeadParams {
    const float4* skip;         // [tile32][w4][mb2][q4][lane64]
    const float4* wsp;          // skip_projection packed [w4][kc32][mb2][lane64]
    const float4* bsp;          // [w4][mb2][h2][q4]
    const float4* bskp;         // sum over layers of the skip-half output_projection biases, [w4][mb2][h2][q4]
    const float4* woutp;        // final projection packed [kc32][mb3][lane64]
    const float4* boutp;        // [mb3][h2][q4]
    const float4* winp;         // input projection (fused next-eval in-proj)
    const float4* binp;
    float* x_next;              // [tiles][C][32] tile-major (fused in-proj output)
    float sqrt_L;
    int nk_in, M, T, TS, ntile32;
    // mode-specific tensors, all [B][M][T]
    float* eps_out;             // HEAD_EPS: eps;  HEAD_PLMS: where to store this evaluation's eps (nullable)
    const float* x_base;        // DDPM / PLMS: x_t the update is applied to
    float* x_out;               // DDPM / PLMS: result (may alias x_base)
    const float* const* noise_cell;   // DDPM: *noise_cell + noise_off = this step's N(0,1) draw; *noise_cell == nullptr: Philox
    size_t noise_off;
    const unsigned long long* seed_cell;   // Philox seed (device cell: cached graphs / plans stay valid when it changes)
    unsigned step_id;                      // index of this p_sample call in the loop (Philox counter word)
    const float* e1; const float* e2; const float* e3;   // PLMS history (most recent first)
    // per-step scalars (fp32, computed on the host from the fp32 tables exactly like the reference's
    // [B,1,1,1] tensor arithmetic)
    float sa, sb, c1, c2, sigma;    // DDPM: sqrt_recip_ac, sqrt_recipm1_ac, coef1, coef2, [t != 0] * exp(0.5 logvar)
    float dA, cx, ce;               // PLMS get_x_pred: (a_prev - a_t), 1/(..), 1/(..)
    int order;
};

// The sampler arithmetic behind the final projection for ONE element (idx = (b * M + m) * T + t), shared by k_head and the row-split head
// of the latency path (dsd_lat.hpp): head_prefetch reads what the update needs besides eps - so that a kernel can issue it in front of a
// contraction - and head_apply does the arithmetic and the stores.  No FMA contraction: the reference rounds every product.
struct HeadPre { float x, z, e1, e2, e3; };

template <int MODE>
__device__ __forceinline__ void head_prefetch(const HeadParams& p, size_t idx, HeadPre& pre) {
    pre.x = pre.z = pre.e1 = pre.e2 = pre.e3 = 0.f;
    if (MODE == HEAD_DDPM) {
        pre.x = p.x_base[idx];
        const float* nzb = *p.noise_cell;
        pre.z = nzb ? nzb[p.noise_off + idx] : philox_normal(*p.seed_cell, p.step_id, idx);
    } else if (MODE == HEAD_PLMS) {
        if (p.order != PLMS_RAW) pre.e1 = p.e1[idx];
        if (p.order == PLMS_AB3 || p.order == PLMS_AB4) pre.e2 = p.e2[idx];
        if (p.order == PLMS_AB4) pre.e3 = p.e3[idx];
        pre.x = p.x_base[idx];
    }
}

template <int MODE>
__device__ __forceinline__ float head_apply(const HeadParams& p, float eps, size_t idx, const HeadPre& pre) {
    float xn = 0.f;
    if (MODE == HEAD_EPS) {
        p.eps_out[idx] = eps;
    } else if (MODE == HEAD_DDPM) {
        // p_mean_variance + p_sample (shallow_diffusion_tts.py:134-166)
        const float x = pre.x, z = pre.z;
        float x0 = __fsub_rn(__fmul_rn(p.sa, x), __fmul_rn(p.sb, eps));
        x0 = fminf(fmaxf(x0, -1.f), 1.f);
        const float mean = __fadd_rn(__fmul_rn(p.c1, x0), __fmul_rn(p.c2, x));
        xn = __fadd_rn(mean, __fmul_rn(p.sigma, z));
        p.x_out[idx] = xn;
    } else {
        // p_sample_plms (shallow_diffusion_tts.py:168-204)
        float ep;
        if (p.order == PLMS_RAW) {
            ep = eps;
        } else if (p.order == PLMS_HEUN) {
            ep = __fmul_rn(__fadd_rn(pre.e1, eps), 0.5f);                 // (first + prev) / 2
        } else if (p.order == PLMS_AB2) {
            ep = __fmul_rn(__fsub_rn(__fmul_rn(3.f, eps), pre.e1), 0.5f);
        } else if (p.order == PLMS_AB3) {
            ep = __fdiv_rn(__fadd_rn(__fsub_rn(__fmul_rn(23.f, eps), __fmul_rn(16.f, pre.e1)), __fmul_rn(5.f, pre.e2)), 12.f);
        } else {
            ep = __fdiv_rn(__fsub_rn(__fadd_rn(__fsub_rn(__fmul_rn(55.f, eps), __fmul_rn(59.f, pre.e1)), __fmul_rn(37.f, pre.e2)),
                                     __fmul_rn(9.f, pre.e3)), 24.f);
        }
        if (p.eps_out) p.eps_out[idx] = eps;
        const float x = pre.x;
        const float delta = __fmul_rn(p.dA, __fsub_rn(__fmul_rn(p.cx, x), __fmul_rn(p.ce, ep)));
        xn = __fadd_rn(x, delta);
        p.x_out[idx] = xn;
    }
    return xn;
}

template <int MODE, bool FUSE_INPROJ>
__global__ __launch_bounds__(kThreads, 2) void k_head(const HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* stile = smem;                    // [256][32] scaled skip sum
    float* htile = smem + kC * 32;          // [256][32] relu(skip_projection)
    float* ptile = smem + 2 * kC * 32;      // [96][32]  next x (fused in-proj)
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int b = tile / p.ntile32, t0 = (tile % p.ntile32) * 32;

    // x = sum(skip) / sqrt(L)   (net.py:126)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const float4* sl = p.skip + (((size_t)tile * 4 + w) * 2 + ms) * (4 * 64) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 s = sl[q * 64], bs = p.bskp[((w * 2 + ms) * 2 + h) * 4 + q];
            const float v[4] = {s.x + bs.x, s.y + bs.y, s.z + bs.z, s.w + bs.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                stile[(64 * w + 32 * ms + frag_row(4 * q + e, h)) * 32 + j] = __fdiv_rn(v[e], p.sqrt_L);
        }
    }
    __syncthreads();
    // skip_projection + ReLU (net.py:127-128)
    {
        f32x16 acc[2][1];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) set4(acc[mb][0], q, p.bsp[((w * 2 + mb) * 2 + h) * 4 + q]);
        const float4* ap = p.wsp + (size_t)w * (32 * 128);
        const float* sl = stile + 4 * h * 32 + j;
        gemm_k<2, 1, 32, 128>(acc, ap, lane, 32, TileB{sl, 8 * 32, 32});
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                htile[(64 * w + 32 * mb + frag_row(r, h)) * 32 + j] = fmaxf(acc[mb][0][r], 0.f);
    }
    __syncthreads();
    // output_projection (net.py:129): 96 padded rows on waves 0..2, then the sampler arithmetic on eps
    if (w < 3) {
        f32x16 acc[1][1];
#pragma unroll
        for (int q = 0; q < 4; ++q) set4(acc[0][0], q, p.boutp[(w * 2 + h) * 4 + q]);
        const float4* ap = p.woutp + (size_t)w * 64;
        const float* hl = htile + 4 * h * 32 + j;
        gemm_k<1, 1, 32, 192>(acc, ap, lane, 32, TileB{hl, 8 * 32, 32});
        const int t = t0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * w + frag_row(r, h);
            const bool ok = (m < p.M) && (t < p.T);
            const size_t idx = ((size_t)b * p.M + m) * p.T + t;
            const float eps = acc[0][0][r];
            float xn = 0.f;
            if (ok) {
                HeadPre pre;
                head_prefetch<MODE>(p, idx, pre);
                xn = head_apply<MODE>(p, eps, idx, pre);
            }
            if (FUSE_INPROJ) ptile[m * 32 + j] = ok ? xn : 0.f;
        }
    }
    if (FUSE_INPROJ) {
        __syncthreads();
        inproj_tile(ptile, p.winp, p.binp, p.nk_in, p.x_next + (size_t)tile * (kC * 32), w, lane);
    }
}
constexpr int kHeadLdsBytes = (2 * kC * 32 + kMPad * 32) * (int)sizeof(float);

// ------------------------------------------------------------------------------------------------------------
// weight packing (torch layouts -> fragment order).  One thread per packed float.
// ------------------------------------------------------------------------------------------------------------
// Generic A-operand pack: dst[(w, kc, mb, lane, s)] = src[row(w,mb,lane) * row_stride + col(kc,lane,s) * col_stride + tap]
// rows: split == 1: mb < nmb/2 -> base_lo + (nmb/2*32)*w + 32*mb + i ; else base_hi + (nmb/2*32)*w + 32*(mb-nmb/2) + i
//       split == 0: rows_per_wave*w + 32*mb + i
//       split == 2: block b = w * nmb + mb holds rows 16 b + (0..15) and hi_base + 16 b + (0..15)
struct PackParams {
    const float* src; float* dst;
    int nw, nkc, nmb;            // destination dims [nw][ntap*nkc][nmb][64][4]
    int ntap;                    // taps (dilated conv: 3, else 1); kc_total = ntap * nkc
    int split, hi_base;          // gate/filter or residual/skip split (hi rows start at hi_base)
    int rows_valid, cols_valid;  // zero outside (padding)
    int row_stride, col_stride;  // in floats: src[(row * row_stride) + col * col_stride + tap]
    int centre_first;            // ntap == 3: chunk order of the dilated conv (conv_chunk above) instead of "taps of one group consecutive"
};

__global__ void k_pack_a(const PackParams p) {
    const size_t n = (size_t)p.nw * p.ntap * p.nkc * p.nmb * 256;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int s = idx & 3, lane = (idx >> 2) & 63;
        size_t r = idx >> 8;
        const int mb = r % p.nmb; r /= p.nmb;
        const int kct = r % (p.ntap * p.nkc); r /= (p.ntap * p.nkc);
        const int w = (int)r;
        int tap = kct % p.ntap, kc = kct / p.ntap;            // taps of one 8-deep k group are consecutive chunks ...
        if (p.centre_first && p.ntap == 3) {                  // ... or the dilated conv's order: all centre taps, then the outer pairs
            if (kct < p.nkc) { tap = 1; kc = kct; }
            else { const int idx = kct - p.nkc; kc = idx >> 1; tap = (idx & 1) * 2; }
        }
        const int i = lane & 31, h = lane >> 5;
        int row;
        if (p.split == 2) {
            // one 32-row block = 16 low rows and THEIR 16 high rows (a gate half-block with its filter half-block: the G = 16 latency kernels)
            row = (i < 16) ? 16 * (w * p.nmb + mb) + i : p.hi_base + 16 * (w * p.nmb + mb) + (i - 16);
        } else if (p.split) {
            const int half = p.nmb / 2;
            row = (mb < half) ? (half * 32) * w + 32 * mb + i : p.hi_base + (half * 32) * w + 32 * (mb - half) + i;
        } else {
            row = (p.nmb * 32) * w + 32 * mb + i;
        }
        const int col = 8 * kc + 4 * h + s;
        float v = 0.f;
        if (row < p.rows_valid && col < p.cols_valid) v = p.src[(size_t)row * p.row_stride + (size_t)col * p.col_stride + tap];
        p.dst[idx] = v;
    }
}

// Bias pack into accumulator-fragment order: dst[(w, mb, h, r)] = a[row] (+ b[row]); row mapping as above.
struct PackBiasParams {
    const float* a; const float* b; float* dst;
    int nw, nmb, split, hi_base, rows_valid;
};

__global__ void k_pack_bias(const PackBiasParams p) {
    const int n = p.nw * p.nmb * 2 * 16;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int r = idx & 15, h = (idx >> 4) & 1;
        int q = idx >> 5;
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
        if (row < p.rows_valid) { v = p.a[row]; if (p.b) v += p.b[row]; }
        p.dst[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// step-embedding table
// ------------------------------------------------------------------------------------------------------------
// E[k][n]: sinusoidal embedding of integer step n (net.py:37-44), stored [dim][N] (N contiguous)
__global__ void k_step_embed(float* E, int dim, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (n >= N) return;
    const int half = dim / 2;
    const float scale = (float)(-(9.210340371976184 /* ln 1e4 */) / (double)(half - 1));
    const int kk = (k < half) ? k : k - half;
    const float f = expf((float)kk * scale);
    const float arg = (float)n * f;
    E[(size_t)k * N + n] = (k < half) ? sinf(arg) : cosf(arg);
}

// Y[m][n] = act(sum_k W[m][k] X[k][n] + bias[m]); X, Y with n contiguous unless out strides say otherwise.
// act: 0 none, 1 Mish (x * tanh(softplus(x)), softplus threshold 20 as torch's default)
__global__ void k_small_gemm(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ X,
                             float* __restrict__ Y, int M, int K, int N, int act, size_t out_stride_m, size_t out_stride_n) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    const float* wr = W + (size_t)m * K;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(wr[k], X[(size_t)k * N + n], acc);
    acc += bias[m];
    if (act == 1) {
        const float sp = (acc > 20.f) ? acc : log1pf(expf(acc));
        acc = acc * tanhf(sp);
    }
    Y[(size_t)m * out_stride_m + (size_t)n * out_stride_n] = acc;
}

// ------------------------------------------------------------------------------------------------------------
// layout / elementwise helpers
// ------------------------------------------------------------------------------------------------------------
// cond [B][H][T] with arbitrary element strides -> condT [B][H][TS] (frame axis contiguous, zero for t >= T)
__global__ void k_cond_layout(const float* __restrict__ cond, float* __restrict__ out, int H, int T, int TS,
                              int64_t sb, int64_t sh, int64_t st) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, h0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;     // 32 x 8
    const float* src = cond + (size_t)b * sb;
    if (st == 1 || sh != 1) {
        // frame axis already fastest (or generic): read with tx along t
        for (int k = ty; k < 32; k += 8) {
            const int hh = h0 + k, t = t0 + tx;
            tile[k][tx] = (hh < H && t < T) ? src[(int64_t)hh * sh + (int64_t)t * st] : 0.f;
        }
    } else {
        // channel axis fastest (the reference's transposed view of [B,T,H]): read with tx along h, transpose in LDS
        for (int k = ty; k < 32; k += 8) {
            const int t = t0 + k, hh = h0 + tx;
            tile[tx][k] = (hh < H && t < T) ? src[(int64_t)hh * sh + (int64_t)t * st] : 0.f;
        }
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int hh = h0 + k, t = t0 + tx;
        if (hh < H && t < TS) out[((size_t)b * H + hh) * TS + t] = tile[k][tx];
    }
}

// q_sample (shallow_diffusion_tts.py:206-211): out = a * x0 + b * z
__global__ void k_qsample(const float* __restrict__ x0, const float* __restrict__ z, float* __restrict__ out,
                          float a, float b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = __fadd_rn(__fmul_rn(a, x0[i]), __fmul_rn(b, z[i]));
}

// norm_spec + transpose: mel [B][T][M] -> x [B][M][T]   ((x - min) / (max - min) * 2 - 1, :278-279)
// p_mean_variance + p_sample (shallow_diffusion_tts.py:134-166) as a stand-alone element-wise kernel with PER-UTTERANCE coefficients
// coef[b] = {sqrt_recip_ac, sqrt_recipm1_ac, posterior_mean_coef1, posterior_mean_coef2, [t != 0] * exp(0.5 logvar)} at t[b];
// clip = clip_denoised; z_bstride = 0 repeats one [M][T] draw over the batch (repeat_noise).  Every product rounded like the
// reference's tensor ops (the file is compiled with -ffp-contract=off).
__global__ void k_psample_ex(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ z, const float* __restrict__ coef,
                             size_t per_utt, size_t n, int clip, size_t z_bstride) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per_utt, r = i - b * per_utt;
        const float* c = coef + 5 * b;
        const float xv = x[i];
        float x0 = __fsub_rn(__fmul_rn(c[0], xv), __fmul_rn(c[1], eps[i]));
        if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        const float mean = __fadd_rn(__fmul_rn(c[2], x0), __fmul_rn(c[3], xv));
        x[i] = __fadd_rn(mean, __fmul_rn(c[4], z[b * z_bstride + r]));
    }
}

__global__ void k_norm_spec(const float* __restrict__ mel, float* __restrict__ x, const float* __restrict__ smin,
                            const float* __restrict__ smax, int M, int T) {
    extern __shared__ float tile[];                 // [32][M+1]
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int nt = min(32, T - t0);
    const float* src = mel + ((size_t)b * T + t0) * M;
    for (int i = threadIdx.x; i < nt * M; i += blockDim.x) {
        const int tt = i / M, m = i % M;
        const float lo = smin[m], hi = smax[m];
        const float v = __fsub_rn(__fmul_rn(__fdiv_rn(__fsub_rn(src[i], lo), __fsub_rn(hi, lo)), 2.f), 1.f);
        tile[tt * (M + 1) + m] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < M * 32; i += blockDim.x) {
        const int m = i >> 5, tt = i & 31;
        if (tt < nt) x[((size_t)b * M + m) * T + t0 + tt] = tile[tt * (M + 1) + m];
    }
}

// denorm_spec + transpose: x [B][M][T] -> mel [B][T][M]   ((x + 1) / 2 * (max - min) + min, :281-282; * mask :273)
__global__ void k_denorm_spec(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ mel,
                              const float* __restrict__ smin, const float* __restrict__ smax, int M, int T) {
    extern __shared__ float tile[];                 // [32][M+1]
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int nt = min(32, T - t0);
    for (int i = threadIdx.x; i < M * 32; i += blockDim.x) {
        const int m = i >> 5, tt = i & 31;
        if (tt < nt) tile[tt * (M + 1) + m] = x[((size_t)b * M + m) * T + t0 + tt];
    }
    __syncthreads();
    float* dst = mel + ((size_t)b * T + t0) * M;
    for (int i = threadIdx.x; i < nt * M; i += blockDim.x) {
        const int tt = i / M, m = i % M;
        const float lo = smin[m], hi = smax[m];
        float v = __fadd_rn(__fmul_rn(__fmul_rn(__fadd_rn(tile[tt * (M + 1) + m], 1.f), 0.5f), __fsub_rn(hi, lo)), lo);
        if (mask) v = __fmul_rn(v, mask[(size_t)b * T + t0 + tt]);
        dst[i] = v;
    }
}

// bsum[c] = sum_l b2[l][C + c]: the skip halves of all output_projection biases (added once, in the head)
__global__ void k_sum_skip_bias(const float* __restrict__ b2all, float* __restrict__ bsum, int L) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= kC) return;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += b2all[(size_t)l * 2 * kC + kC + c];
    bsum[c] = acc;
}

__global__ void k_set_cell(const float** cell, const float* value) { *cell = value; }
__global__ void k_set_seed(unsigned long long* cell, unsigned long long value) { *cell = value; }

// out[i] = the Philox N(0,1) draw of element i at step `step` (tests: the explicit-noise loop fed with these == the seeded loop)
__global__ void k_philox_fill(float* out, size_t n, unsigned long long seed, unsigned step) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = philox_normal(seed, step, i);
}

}  // namespace dsd
