// train_loop.hpp - the forward of the fused training stack as ONE persistent kernel (gfx950): k_tr_stack_fwd, the
// default where an utterance fits the co-resident grid.  (The data-gradient chain of the backward pass had a persistent form in round 2,
// k_trb_loop: measured equal to the per-layer launches for 30 % more workspace, profiles/r03z - deleted in round 3.)
//
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

// y tile [256][48] + gate tile [256][32] + residual-transpose scratch [256][32] + step rows [2][256] (row-major tiles: this kernel keeps the layout
// of layer_body, whose bits it reproduces; the inference loop moved to frame-major tiles, dsd_loop.hpp)
constexpr int kTrStackLdsBytes = (kC * (32 + 2 * kHalo) + 2 * kC * 32 + 2 * kC) * (int)sizeof(float);

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

}  // namespace dsd
