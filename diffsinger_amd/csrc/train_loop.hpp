// train_loop.hpp - the fused training stack as PERSISTENT kernels (gfx950): k_tr_stack_fwd, the forward (default where
// an utterance fits the co-resident grid), and - further down - k_trb_loop, the data-gradient chain of the backward pass (opt-in: measured equal
// to the per-layer launches).
//
// k_tr_stack_fwd:
// The 20 ResidualBlock.forward calls of DiffNet.forward (usr/diff/net.py:119-124; block :66-78) under GaussianDiffusion.p_losses
// (usr/diff/shallow_diffusion_tts.py:213-231), on the tile ownership and the neighbour exchange of the inference loop (dsd_loop.hpp:
// one workgroup owns a 32-frame tile, x and the running skip sum stay in registers from layer to layer, the 8 halo columns of a layer
// travel through write-through stores + a per-tile phase flag).  What differs from k_loop:
//   * no sampler head and no evaluations: one pass over the layers, the skip sum (+ the summed skip biases) leaves channel-major;
//   * the step projection is a row per UTTERANCE and layer (every utterance draws its own t, shallow_diffusion_tts.py:279), not a table row;
//   * every layer SAVES what its backward needs: y = x + step projection (channel-major, zero tail, rows padded for k_tr_wgrad) and the
//     gate pre-activation a = conv + hoisted conditioner projection (fragment order) - the same bytes k_tr_layer writes, as plain stores
//     (96 KiB per tile and layer; measured free: the step takes the same time with the saves switched off, profiles/r03t);
//   * x enters channel-major (the layout of train.py's input projection), read straight into the row layout of the registers.
// Arithmetic and its order are those of layer_body / k_loop: results are bit-identical to the per-layer launches (tests/test_gpu_train_fused.py).
#pragma once
#include "dsd_loop.hpp"
#include "train_kernels.hpp"

namespace dsd {

struct TrLoopParams {
    const float4* w1p;          // [L][w4][kc96: centre tap first][mb4][lane64]
    const float4* w2p;          // [L][w4][kc32][mb4][lane64]
    TrPtrs b2;                  // output_projection.bias of every layer, raw [2C] (the residual half is used here)
    const float4* cp;           // [L][tile][w4][mb4][q4][lane64] hoisted conditioner projection + both biases
    size_t cp_lstride;          // float4 between layers
    const float* step;          // [B][L][C] step projection rows
    const float* x0;            // [B][C][TS] input of layer 0 (channel-major)
    float* y_cm;                // layer 0 of the saved y: [L][B][C][y_rs] + pad offset applied by the host
    size_t y_lstride;           // floats between layers
    int y_rs;                   // row stride of the saved y
    float4* a_frag;             // [L][tile][w4][mb4][q4][lane64]
    size_t a_lstride;           // float4 between layers
    const float* bsum;          // [C] sum over layers of the skip-half output biases
    float* skip_out;            // [B][C][TS] skip sum + bsum, zero tail
    int L, T, TS, ntile32, ntiles_total;
    unsigned char dil[kTrMaxLayers];
    unsigned* flags;            // [ntiles_total] phase flags, zero at launch
    float* halo;                // [2][ntiles_total][2 sides][256][8]
    unsigned* tmo;              // sticky timeout word, zero at launch
    int tile_base, n_tiles;     // this launch covers tiles [tile_base, tile_base + n_tiles): whole utterances
};

__global__ __launch_bounds__(kThreads, 1) void k_tr_stack_fwd(const TrLoopParams p) {
    constexpr int LD = 32 + 2 * kHalo, GLD = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem;                    // [256][48]  conv input y = x + step_proj (+ halo)
    float* gtile = smem + kC * LD;          // [256][32]  gate tile
    float* xt = gtile + kC * 32;            // [256][32]  scratch of the residual transpose
    float* dsbuf = xt + kC * 32;            // [2][256]   step projection of layer l in dsbuf[l & 1], fetched one layer ahead

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tl = xcd_item(blockIdx.x, p.n_tiles >> 3, p.n_tiles & 7);         // neighbours behind one L2 (speed only)
    const int tile = p.tile_base + tl;
    const int b = tile / p.ntile32, tn = tile - b * p.ntile32, t0 = tn * 32;
    const bool has_left = tn > 0, has_right = tn + 1 < p.ntile32;
    const int T = p.T;

    float4 xreg[8];         // x tile, row layout: wave w owns rows [64w, 64w+64); xreg[it] = row 64w + 8 it + lane/8, cols 4 (lane%8)..+3
    float4 skp[2][4];       // running skip sum of this wave's skip rows, accumulator-fragment order
    const int xrow0 = 64 * w + (lane >> 3), xc4 = lane & 7;

    auto timed_out = [&]() -> bool { return __hip_atomic_load((gu32*)p.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; };

    {
        const float4* xin = reinterpret_cast<const float4*>(p.x0 + (size_t)b * kC * p.TS);
#pragma unroll
        for (int it = 0; it < 8; ++it) xreg[it] = xin[((xrow0 + 8 * it) * p.TS + t0 + 4 * xc4) >> 2];
    }
    dsbuf[tid] = p.step[(size_t)b * p.L * kC + tid];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int q = 0; q < 4; ++q) skp[ms][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    // halo of phase `phase` = this tile's first / last 8 columns of x: write-through stores, every storing wave drained, barrier, ONE relaxed
    // agent-scope flag store (the protocol of k_loop)
    auto publish_issue = [&](unsigned phase) {
        float* hb = p.halo + ((size_t)(phase & 1) * p.ntiles_total + tile) * (2 * kC * 8);
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(hb, 0, 0x7ffffff0, 0x00020000);
        if (xc4 < 2 || xc4 >= 6) {
            const int side = (xc4 >= 6) ? 1 : 0, c = xc4 & 1;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const f32x4_ f = {xreg[it].x, xreg[it].y, xreg[it].z, xreg[it].w};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, f), r, ((side * kC + xrow0 + 8 * it) * 8 + 4 * c) * 4, 0, 16);
            }
        }
    };
    auto publish_finish = [&](unsigned phase) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gu32*)(p.flags + tile), phase + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    publish_issue(0);
    publish_finish(0);
    for (int l = 0; l < p.L; ++l) {
        const unsigned ph = (unsigned)l;
        const bool last = (l == p.L - 1);
        const float* dsl = dsbuf + (l & 1) * kC;
        const int dil = p.dil[l];

        const ConvB<LD> bof1{ytile + 4 * h * LD + kHalo + j, dil, 0};
        GemmPipe<4, 1, LD, 256, 6, ConvB<LD>> pipe1(p.w1p + ((size_t)l * 4 + w) * (96 * 256), lane, 96, bof1);
        pipe1.template start_a<0, 5>();

        // own columns of y = x + step_proj (zero at frames >= T: the conv's zero padding applies to y, net.py:69-71), kept for the backward
        {
            float4* ysave = reinterpret_cast<float4*>(p.y_cm + (size_t)l * p.y_lstride + (size_t)b * kC * p.y_rs);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = xrow0 + 8 * it, t = t0 + 4 * xc4;
                const float d = dsl[row];
                float4 v = xreg[it];
                v.x = (t + 0 < T) ? v.x + d : 0.f;
                v.y = (t + 1 < T) ? v.y + d : 0.f;
                v.z = (t + 2 < T) ? v.z + d : 0.f;
                v.w = (t + 3 < T) ? v.w + d : 0.f;
                *reinterpret_cast<float4*>(ytile + row * LD + kHalo + 4 * xc4) = v;
                ysave[(row * p.y_rs + t) >> 2] = v;
            }
        }
        __syncthreads();
        // every wave reads the two neighbour flags now (lanes 0 / 1) and tests them behind chunk 12
        unsigned fv = 0xffffffffu;
        if (lane < 2) {
            const bool have = lane ? has_right : has_left;
            if (have) fv = __hip_atomic_load((const gu32*)(p.flags + tile + (lane ? 1 : -1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        DSD_SB();

        // dilated conv, K = 768, centre taps first: the exchange with the neighbour tiles runs under them
        f32x16 acc[4][1];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
        float4 cpv[4][4];
        pipe1.start_b();
        pipe1.run(acc, 0, 12);
        if (fv < ph + 1u) {
            const gu32* f = (const gu32*)(p.flags + tile + (lane ? 1 : -1));
            for (int spins = 0;; ++spins) {
                if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ph + 1u) break;
                if ((spins & 255) == 255 && timed_out()) break;
                if (spins >= kLoopSpinLimit) { __hip_atomic_store((gu32*)p.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        float4 hv[2][2];
        {
            const float* hbase = p.halo + (size_t)(ph & 1) * p.ntiles_total * (2 * kC * 8);
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const bool have = side ? has_right : has_left;
                // my left halo = left neighbour's LAST 8 columns (its side 1); my right halo = right neighbour's first 8 (side 0)
                const int off = (((tile + (side ? 1 : -1)) * 2 + (side ? 0 : 1)) * kC + tid) * 8 * 4;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    hv[side][g] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (have) hv[side][g] = ld16_sc1(hbase, off + 16 * g);
                }
            }
        }
        DSD_SB();
        pipe1.run(acc, 12, 30);
        {
            const float d = dsl[tid];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const bool have = side ? has_right : has_left;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float4 v = hv[side][g];
                    const int t = side ? t0 + 32 + 4 * g : t0 - kHalo + 4 * g;
                    v.x = (have && t + 0 < T) ? v.x + d : 0.f;
                    v.y = (have && t + 1 < T) ? v.y + d : 0.f;
                    v.z = (have && t + 2 < T) ? v.z + d : 0.f;
                    v.w = (have && t + 3 < T) ? v.w + d : 0.f;
                    *reinterpret_cast<float4*>(ytile + tid * LD + (side ? kHalo + 32 : 0) + 4 * g) = v;
                }
            }
        }
        __syncthreads();
        pipe1.run(acc, 30, 48);
        {
            const float4* cpl = p.cp + (size_t)l * p.cp_lstride + ((size_t)tile * 4 + w) * (4 * 4 * 64) + lane;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) cpv[mb][q] = cpl[(mb * 4 + q) * 64];
        }
        DSD_SB();
        pipe1.run(acc, 48, 96);
        float ds_next = 0.f;
        if (!last) ds_next = p.step[((size_t)b * p.L + l + 1) * kC + tid];

        const float* gl = gtile + 4 * h * GLD + j;
        const TileB bof2{gl, 8 * GLD, 32};
        // gate pre-activation a = conv + cp, saved for the backward (fragment order, 1 KiB per instruction), and the gate (net.py:73-74)
        auto save_a_and_gate = [&]() {
            float4* al = p.a_frag + (size_t)l * p.a_lstride + ((size_t)tile * 4 + w) * (4 * 4 * 64);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 c = cpv[mb][q], a = get4(acc[mb][0], q);
                    al[(mb * 4 + q) * 64 + lane] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
                }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float vg = f4at(cpv[pr][r >> 2], r & 3), vf = f4at(cpv[pr + 2][r >> 2], r & 3);
                    const float g = sigmoid_f(acc[pr][0][r] + vg) * tanh_f(acc[pr + 2][0][r] + vf);
                    gtile[(64 * w + 32 * pr + frag_row(r, h)) * GLD + j] = g;
                }
        };
        float* tw = xt + w * (64 * 32);
        if (!last) {
            GemmPipe<4, 1, GLD, 256, 6, TileB> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256), lane, 32, bof2);
            pipe2.start_a();
            save_a_and_gate();
            __syncthreads();
            f32x16 acc2[4][1];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[m][0][r] = 0.f;
            float brow[8];
            pipe2.start_b();
            pipe2.run(acc2, 0, 6);
            {
                const float* b2l = p.b2.p[l];
#pragma unroll
                for (int it = 0; it < 8; ++it) brow[it] = b2l[64 * w + it * 8 + (lane >> 3)];
            }
            DSD_SB();
            pipe2.run(acc2, 6, 32);
            // residual: accumulator fragments -> row layout through this wave's slice of the scratch; x' = (x + res + b) / sqrt(2)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tw[(32 * mb + frag_row(r, h)) * 32 + j] = acc2[mb][0][r];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const float4 v = reinterpret_cast<const float4*>(tw)[it * 64 + lane];
                const float4 x = xreg[it];
                const float bv = brow[it];
                float4 o;
                o.x = (x.x + (v.x + bv)) * kTrInvSqrt2;
                o.y = (x.y + (v.y + bv)) * kTrInvSqrt2;
                o.z = (x.z + (v.z + bv)) * kTrInvSqrt2;
                o.w = (x.w + (v.w + bv)) * kTrInvSqrt2;
                xreg[it] = o;
            }
            dsbuf[((l + 1) & 1) * kC + tid] = ds_next;          // visible behind the barrier inside publish_finish()
            publish_issue(ph + 1u);                             // the halo stores drain while the skip sum is updated
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = get4(acc2[2 + ms][0], q), s = skp[ms][q];
                    skp[ms][q] = (l == 0) ? a : make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                }
            publish_finish(ph + 1u);
        } else {
            // last layer: only the skip half (net.py:126 reads the skips; the residual is dead)
            GemmPipe<2, 1, GLD, 256, 6, TileB> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256) + 2 * 64, lane, 32, bof2);
            pipe2.start_a();
            save_a_and_gate();
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

    // skip sum + summed skip biases -> channel-major, zero tail (a wave store covers two 128-byte row segments); a wait that hit its spin
    // bound leaves garbage: make it LOUD - NaN instead
    const bool bad = timed_out();
    {
        const int t = t0 + j;
        float* so = p.skip_out + (size_t)b * kC * p.TS;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 64 * w + 32 * ms + frag_row(4 * q + e, h);
                    const float v = f4at(skp[ms][q], e) + p.bsum[row];
                    so[(size_t)row * p.TS + t] = bad ? __builtin_nanf("") : ((t < T) ? v : 0.f);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// the data-gradient chain of the backward pass as ONE persistent kernel
// ------------------------------------------------------------------------------------------------------------
// k_trb_gate<true> of the last layer, then for l = L-1 .. 0 the body of k_trb_fused (transposed dilated conv of layer l + output-projection
// data gradient and gate derivative of layer l - 1) on the same tile, in ONE launch per chunk of whole utterances.  What a kernel boundary
// did between the layers is done by the exchange protocol of k_loop / k_tr_stack_fwd:
//   * the transposed conv of layer l needs 8 columns of the two NEIGHBOUR tiles' da(l): a workgroup copies the first / last 8 columns of its
//     da tile into a halo buffer (2 x 16 KiB, write-through sc1 stores, double-buffered by phase parity), drains them, raises its phase flag,
//     and the neighbours read the columns with sc1 loads once the flag has reached the phase.  Every layer has its own da / g / dx slab (the
//     weight gradients run behind this kernel); those 128 KiB per tile and layer are plain stores - only later kernels read them;
//   * a tile's OWN da columns go from the gate epilogue's registers straight into the LDS tile of the next conv (they were re-read from
//     memory before), and the residual-path gradient dx' stays in registers from layer to layer;
//   * the centre taps of the conv (the first 32 chunks of either K half) read no halo column: the flag test and the halo loads run under them.
// Arithmetic and summation orders are those of k_trb_gate / k_trb_conv: results are bit-identical to the per-layer launches.
// 4-byte store at float index idx behind a wave-uniform base (buffer addressing: one VGPR of offset instead of a 64-bit address per store);
// plain - the line stays in the L2 until it is evicted or the kernel ends (outputs only LATER kernels read)
__device__ __forceinline__ void store4_buf(float* base_uniform, int idx, float v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, idx * 4, 0, 0);
}

struct TrbLoopParams {
    const float* dsk;           // gradient wrt the skip sum [B][256][TS]
    const float4* a_frag;       // saved gate pre-activations [L][ntiles][w4][mb4][q4][lane64]
    size_t a_lstride;           // float4 between layers
    const float4* wotp;         // [L] output_projection.weight transposed, packed [wr2][kc64][mb4][lane64]
    const float4* wdtp;         // [L] dilated_conv.weight flipped + transposed, packed [hk2][wr2][kc96][mb4][lane64]
    float* da;                  // da of layer l: da[b * da_bstride + l * da_lstride + row * TS + t], 512 rows
    long long da_bstride;
    size_t da_lstride;
    float* g;                   // gate outputs [L][B][256][TS]
    float* dx;                  // dx[l] = gradient wrt x_out of layer l, l in [0, L - 1): [L - 1][B][256][TS]
    size_t act;                 // B * 256 * TS: floats between the layers of g / dx
    float* dx0;                 // gradient wrt the input of layer 0 [B][256][TS]
    float* dds_part;            // [L][ntiles_total][256] per-tile row sums of dy
    int L, T, TS, ntile32, ntiles_total;
    unsigned char dil[kTrMaxLayers];
    unsigned* flags;            // [ntiles_total] phase flags, zero at launch
    float* halo;                // [2][ntiles_total][2 sides][512][8]: first / last 8 columns of a tile's da, by phase parity
    unsigned* tmo;              // sticky timeout word, zero at launch
    int tile_base, n_tiles;
};

__global__ __launch_bounds__(kThreads, 1) void k_trb_loop(const TrbLoopParams p) {
    constexpr int LD = kTrbConvLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // da tile [512][48] / dy2 tile [512][32]; exchange
    float* xbuf = smem + 2 * kC * LD;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wr = w & 1, wk = w >> 1;
    const int tile = p.tile_base + xcd_item(blockIdx.x, p.n_tiles >> 3, p.n_tiles & 7);
    const int b = tile / p.ntile32, tn = tile - b * p.ntile32, t0 = tn * 32;
    const bool has_left = tn > 0, has_right = tn + 1 < p.ntile32;
    const size_t brow = (size_t)b * kC * p.TS;
    // Everything derived from the lane id is RE-derived at the top of every layer from a value the compiler cannot see through: otherwise the
    // ~150 per-lane addresses / masks of a phase are hoisted out of the layer loop as loop invariants, the kernel runs out of registers and the
    // spilled values come back one scratch load + s_waitcnt vmcnt(0) at a time (measured: +12 us per layer).
    int tid, lane, j, h, t, sg, st;
    bool ok, m0, m1, m2, m3;
    auto derive = [&]() {
        int v = threadIdx.x;
        asm volatile("" : "+v"(v));
        tid = v; lane = v & 63; j = lane & 31; h = lane >> 5;
        t = t0 + j; ok = t < p.T;
        sg = v & 7; st = t0 + 4 * sg;
        m0 = st + 0 < p.T; m1 = st + 1 < p.T; m2 = st + 2 < p.T; m3 = st + 3 < p.T;
    };
    derive();

    auto timed_out = [&]() -> bool { return __hip_atomic_load((gu32*)p.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; };
    // The neighbours need the first / last 8 columns of this tile's da(l): every wave copies those columns of ITS 2 x 64 rows from the LDS tile
    // (it wrote them itself: no workgroup barrier) into the halo buffer of the phase's parity with write-through stores, drains, and after the
    // barrier ONE relaxed agent-scope flag store publishes the phase.  The bulk of da / g / dx is read by later kernels only: plain stores.
    auto publish = [&](unsigned value) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float* hb = p.halo + ((size_t)(value & 1) * p.ntiles_total + tile) * (2 * 2 * kC * 8);
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(hb, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int side = 0; side < 2; ++side)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int row = hk * kC + 128 * wr + 64 * wk + lane;
                    const float4 v = *reinterpret_cast<const float4*>(smem + row * LD + kHalo + (side ? 24 : 0) + 4 * g);
                    const f32x4_ f = {v.x, v.y, v.z, v.w};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, f), r, ((side * 2 * kC + row) * 8 + 4 * g) * 4, 0, 16);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gu32*)(p.flags + tile), value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // dskip rows of the dy2 tile (rows [256, 512)), eight float4 per thread
    float4 vs[8];
    auto request_dsk = [&]() {
#pragma unroll
        for (int it = 0; it < 8; ++it) vs[it] = *reinterpret_cast<const float4*>(p.dsk + brow + (size_t)(it * 32 + (tid >> 3)) * p.TS + st);
        DSD_SB();
    };
    auto write_dsk = [&]() {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 32 + (tid >> 3);
            const float4 s4 = vs[it];
            *reinterpret_cast<float4*>(smem + (kC + row) * 32 + 4 * sg) = make_float4(m0 ? s4.x : 0.f, m1 ? s4.y : 0.f, m2 ? s4.z : 0.f, m3 ? s4.w : 0.f);
        }
    };
    // gate derivative (net.py:73-74) of layer l from dg (fin) and the saved pre-activation: da / g to memory and the tile's own
    // da columns into the LDS tile of the next transposed conv
    auto gate_epilogue = [&](int l, int TSl, const f32x16 (&fin)[2], const float4 (&av)[4][4]) {
        float* dab = p.da + (size_t)b * p.da_bstride + (size_t)l * p.da_lstride;
        float* gb = p.g + (size_t)l * p.act + brow;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ag = f4at(av[mb][r >> 2], r & 3), af = f4at(av[mb + 2][r >> 2], r & 3);
                const float sg_ = 1.f / (1.f + expf(-ag)), th = tanhf(af);
                const float dg = fin[mb][r];
                const int row = 128 * wr + 64 * wk + 32 * mb + frag_row(r, h);
                const float dgate = ok ? dg * th * (sg_ * (1.f - sg_)) : 0.f, dfilt = ok ? dg * sg_ * (1.f - th * th) : 0.f;
                store4_buf(dab, row * TSl + t, dgate);
                store4_buf(dab, (kC + row) * TSl + t, dfilt);
                store4_buf(gb, row * TSl + t, ok ? sg_ * th : 0.f);
                smem[row * LD + kHalo + j] = dgate;
                smem[(kC + row) * LD + kHalo + j] = dfilt;
            }
    };
    auto load_a = [&](int l, float4 (&av)[4][4]) {
        const float4* al = p.a_frag + (size_t)l * p.a_lstride + ((size_t)tile * 4 + 2 * wr + wk) * (4 * 4 * 64) + lane;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) av[mb][qq] = al[(mb * 4 + qq) * 64];
        DSD_SB();
    };

    // ---- gate derivative of the last layer (k_trb_gate<true>: its x_out is dead, K = the 256 skip rows, split between the K halves) ----
    {
        const int l = p.L - 1;
        constexpr int NCH = 16;
        const int ch0 = 32 + NCH * wk;
        const TileB bof{smem + ch0 * 8 * 32 + 4 * h * 32 + j, 8 * 32, NCH};
        GemmPipe<4, 1, 32, 256, 6, TileB> pipe(p.wotp + (size_t)l * (2 * 64 * 256) + ((size_t)wr * 64 + ch0) * 256, lane, NCH, bof);
        request_dsk();
        pipe.start_a();
        float4 av[4][4];
        load_a(l, av);
        f32x16 acc[4][1];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
        write_dsk();
        __syncthreads();
        pipe.start_b();
        pipe.run(acc, 0, NCH);
        f32x16 fin[2];
        trb_exchange(acc, fin, xbuf, wr, wk, lane);
        gate_epilogue(l, p.TS, fin, av);
        publish(1u);
    }

    float rv[2][16];            // gradient wrt the output x of the layer at hand, at the fragment positions this wave finishes
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[mb][r] = 0.f;

    for (int l = p.L - 1; l >= 0; --l) {
        const unsigned need = (unsigned)(p.L - l);       // the neighbours' da(l) is complete when their flag has reached this value
        // the row stride as a value the compiler cannot prove loop-invariant: the 100 store offsets of a phase are then computed where they are
        // used (one v_mad each) instead of being hoisted out of the layer loop into registers the kernel does not have
        int TSl = p.TS;
        asm volatile("" : "+s"(TSl));
        derive();
        const ConvB<LD> bof{smem + (wk * kC + 4 * h) * LD + kHalo + j, (int)p.dil[l], 0};
        GemmPipe<4, 1, LD, 256, 6, ConvB<LD>> pipe(p.wdtp + (size_t)l * (4 * 96 * 256) + ((size_t)(wk * 2 + wr) * 96) * 256, lane, 96, bof);
        pipe.template start_a<0, 5>();
        f32x16 acc[4][1];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
        pipe.start_b();
        pipe.run(acc, 0, 12);
        // every wave reads the two neighbour flags (lanes 0 / 1) behind chunk 12 and tests them behind chunk 18: the neighbours finish their
        // gate epilogue (32 exp + 32 tanh per lane) about when this tile does - the first centre-tap chunks give their stores time to drain
        unsigned fv = 0xffffffffu;
        if (lane < 2) {
            const bool have = lane ? has_right : has_left;
            if (have) fv = __hip_atomic_load((const gu32*)(p.flags + tile + (lane ? 1 : -1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        DSD_SB();
        pipe.run(acc, 12, 18);
        asm volatile("" : "+v"(fv));        // keeps the comparison (and with it the wait for the flag load) HERE instead of right behind the load
        if (fv < need) {
            const gu32* f = (const gu32*)(p.flags + tile + (lane ? 1 : -1));
            for (int spins = 0;; ++spins) {
                if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
                if ((spins & 255) == 255 && timed_out()) break;
                if (spins >= kLoopSpinLimit) { __hip_atomic_store((gu32*)p.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        // the neighbours' columns of da(l): thread = row tid of either K half, 8 columns per side (sc1 loads: the producer stored write-through)
        float4 hv[2][2][2];
        {
            const float* hbase = p.halo + (size_t)(need & 1) * p.ntiles_total * (2 * 2 * kC * 8);
#pragma unroll
            for (int hk = 0; hk < 2; ++hk)
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const bool have = side ? has_right : has_left;
                    // my left halo = left neighbour's LAST 8 columns (its side 1); my right halo = right neighbour's first 8 (side 0)
                    const int off = ((((tile + (side ? 1 : -1)) * 2 + (side ? 0 : 1)) * 2 * kC + hk * kC + tid) * 8) * 4;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        hv[hk][side][g] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (have) hv[hk][side][g] = ld16_sc1(hbase, off + 16 * g);
                    }
                }
        }
        DSD_SB();
        pipe.run(acc, 18, 30);
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int side = 0; side < 2; ++side)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    *reinterpret_cast<float4*>(smem + (hk * kC + tid) * LD + (side ? kHalo + 32 : 0) + 4 * g) = hv[hk][side][g];
        __syncthreads();
        pipe.run(acc, 30, 48);
        constexpr int NCH = 32;
        const int ch0 = NCH * wk;
        const int lg = max(l - 1, 0);
        const TileB bofg{smem + ch0 * 8 * 32 + 4 * h * 32 + j, 8 * 32, NCH};
        GemmPipe<4, 1, 32, 256, 6, TileB> pipeg(p.wotp + (size_t)lg * (2 * 64 * 256) + ((size_t)wr * 64 + ch0) * 256, lane, NCH, bofg);
        if (l > 0) request_dsk();
        pipe.run(acc, 48, 96);
        f32x16 fin[2];
        trb_exchange(acc, fin, xbuf, wr, wk, lane);          // its barrier: every wave is done reading the da tile
        float4 av[4][4];
        if (l > 0) {
            pipeg.start_a();
            load_a(l - 1, av);
        }
        // conv gradient epilogue (k_trb_conv): dx = dx' / sqrt(2) + dy, the residual rows of the dy2 tile, per-tile row sums of dy
        {
            float* dxo = (l == 0) ? p.dx0 + brow : p.dx + (size_t)(l - 1) * p.act + brow;
            float* dds = p.dds_part + ((size_t)l * p.ntiles_total + tile) * kC;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 128 * wr + 64 * wk + 32 * mb + frag_row(r, h);
                    const float dy = ok ? fin[mb][r] : 0.f;
                    const float dx = ok ? rv[mb][r] * kTrInvSqrt2 + dy : 0.f;
                    store4_buf(dxo, row * TSl + t, dx);
                    if (l > 0) smem[row * 32 + j] = dx * kTrInvSqrt2;
                    rv[mb][r] = dx;
                    float s = dy;
                    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64);
                    if (j == 0) dds[row] = s;
                }
        }
        if (l == 0) break;
        write_dsk();
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
        trb_exchange(accg, fing, xbuf, wr, wk, lane);        // its barrier: every wave is done reading the dy2 tile
        gate_epilogue(l - 1, TSl, fing, av);
        publish(need + 1u);
    }
    // a wait that hit its spin bound leaves garbage: make it LOUD - poison this tile of dx0 with NaN
    if (timed_out()) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) p.dx0[brow + (size_t)(128 * wr + 64 * wk + 32 * mb + frag_row(r, h)) * p.TS + t] = __builtin_nanf("");
    }
}

}  // namespace dsd
