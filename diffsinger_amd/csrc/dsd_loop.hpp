// dsd_loop.hpp - the WHOLE K-step reverse loop as ONE persistent kernel (gfx950).
//
// Replaces, for batches that fit the chip (<= one workgroup per CU), the hipGraph of 21 kernels per step
// (usr/diff/shallow_diffusion_tts.py:261-270 loop; per step DiffNet.forward usr/diff/net.py:107-130 + p_sample :159-166 /
// p_sample_plms :168-204).  Same arithmetic, same order, bit-identical results (tests/test_gpu_loop.py) - what changes is where
// the data lives between layers:
//   * a workgroup OWNS one 32-frame tile for the whole loop.  Its x tile (256 channels x 32 frames) and its running skip sum
//     stay in REGISTERS from layer to layer and from step to step; per layer only the hoisted conditioner projection (2 KiB /
//     frame) and the weight stream are read, nothing but the halo is written (the per-layer kernels move 6 KiB / frame and
//     pay a kernel boundary - 1.45 us + the write-back of 16.8 MB of dirty L2 lines - 21 times per step).
//   * the 3-tap dilated conv needs 8 frames of the two NEIGHBOUR tiles' x: each workgroup publishes its first / last 8
//     columns per layer (2 x 8 KiB, write-through sc1 stores), raises a per-tile phase flag, and reads its neighbours'
//     columns with sc1 loads once their flag has reached the phase (MI355X_MICROARCH.md "Inter-workgroup visibility": sc1
//     stores + every storing wave drained + relaxed agent-scope flag; sc1 loads on the consumer).  Halo buffers are double
//     buffered by phase parity: a neighbour can be at most one phase ahead.  No grid-wide barrier anywhere.
//   * all workgroups must be co-resident (they wait for each other): the host launches at most one workgroup per CU (112 KiB
//     of LDS each) and splits larger batches into chunks of whole utterances; every spin is bounded and a timeout is sticky
//     (the loop then finishes instead of hanging, poisons the result with NaN and the host can read the timeout word).
#pragma once
#include "dsd_kernels.hpp"

namespace dsd {

typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int kLoopMaxLayers = 64;
constexpr int kLoopSpinLimit = 1 << 21;     // ~2-4 s of polling: far beyond any legitimate skew, still bounded

struct LoopParams {
    const float4* w1p;          // [L][w4][kc96: centre tap first, conv_chunk()][mb4][lane64]
    const float4* w2p;          // [L][w4][kc32][mb4][lane64]
    const float* b2raw;         // [L][2C]
    const float4* cp;           // [L][tile][w4][mb4][q4][lane64]
    size_t cp_lstride;          // float4 between layers
    const float* ds_table;      // [t][L][C]
    int L, T, TS, ntile32, ntiles_total;
    unsigned char dil[kLoopMaxLayers];
    HeadParams head;            // head weights + geometry; the mode-specific fields come from evals[e]
    const HeadParams* evals;    // [n_evals] (device)
    const int* eval_t;          // [n_evals] step index of every denoiser evaluation (device)
    int n_evals;
    const float* spec0;         // [B][M][T] x at loop entry
    unsigned* flags;            // [ntiles_total] phase flags, zero at launch
    float* halo;                // [2][ntiles_total][2 sides][256][8]
    unsigned* tmo;              // sticky timeout word, zero at launch
    int tile_base, n_tiles;     // this launch covers tiles [tile_base, tile_base + n_tiles)
    unsigned long long* dbg;    // optional s_memtime stamps: [workgroup][wave][16] - 0..7 the layer phase dbg_phase, 8..15 the head of its evaluation
    int dbg_phase;
};

constexpr int kLoopLdsBytes = (kC * (32 + 2 * kHalo) + 2 * kC * 32 + 2 * kC) * (int)sizeof(float);      // y tile + gate tile + scratch + step rows [2]

__device__ __forceinline__ float4 ld16_sc1(const float* base_uniform, int byte_off) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, 0x7ffffff0, 0x00020000);
    const f32x4_ f = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));       // aux 16 = sc1
    return make_float4(f.x, f.y, f.z, f.w);
}

// MODE: HEAD_DDPM or HEAD_PLMS (the sampler arithmetic of the head epilogue)
template <int MODE>
__global__ __launch_bounds__(kThreads, 1) void k_loop(const LoopParams p) {
    constexpr int LD = 32 + 2 * kHalo, GLD = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem;                    // [256][48]  conv input y = x + step_proj (+ halo); head: scaled skip sum [256][32]
    float* gtile = smem + kC * LD;          // [256][32]  gate tile; head: relu(skip_projection)
    float* xt = gtile + kC * 32;            // [256][32]  scratch: residual transpose, spec tile of the in-projection
    float* dsbuf = xt + kC * 32;            // [2][256]   step projection of phase ph in dsbuf[ph & 1]: fetched from the table one phase ahead,
                                            //            so that staging y = x + step never waits for a global load

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware map (speed only): XCD x owns a contiguous range of this launch's tiles
    int tl;
    {
        const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
        const int q = p.n_tiles >> 3, r = p.n_tiles & 7;
        tl = xcd * q + min(xcd, r) + k;
    }
    const int tile = p.tile_base + tl;
    const int b = tile / p.ntile32, tn = tile - b * p.ntile32, t0 = tn * 32;
    const bool has_left = tn > 0, has_right = tn + 1 < p.ntile32;
    const int M = p.head.M, T = p.T;

    float4 xreg[8];         // x tile, row layout: wave w owns rows [64w, 64w+64); xreg[it] = row 64w + 8 it + lane/8, cols 4 (lane%8)..+3
    float4 skp[2][4];       // running skip sum of this wave's skip rows, accumulator-fragment order
    const int xrow0 = 64 * w + (lane >> 3), xc4 = lane & 7;

    auto timed_out = [&]() -> bool { return __hip_atomic_load((gu32*)p.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; };

    // in-projection of the tile in xt (as [kMPad][32]) -> xreg, through the (free) y tile region as [256][32]
    auto inproj_to_xreg = [&]() {
        inproj_tile(xt, p.head.winp, p.head.binp, p.head.nk_in, ytile, w, lane);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) xreg[it] = reinterpret_cast<const float4*>(ytile + 64 * w * 32)[it * 64 + lane];
        __syncthreads();    // every wave has its rows before the region becomes the y tile again
    };

    for (int idx = tid; idx < kMPad * 32; idx += kThreads) {
        const int m = idx >> 5, t = t0 + (idx & 31);
        xt[idx] = (m < M && t < T) ? p.spec0[((size_t)b * M + m) * T + t] : 0.f;
    }
    dsbuf[tid] = p.ds_table[(size_t)p.eval_t[0] * p.L * kC + tid];       // phase 0 = (evaluation 0, layer 0)
    __syncthreads();
    inproj_to_xreg();

    // publish this tile's first / last 8 columns of x (xreg) as the halo of phase `phase`: write-through stores, EVERY storing
    // wave drained, barrier, ONE relaxed agent-scope flag store.  Called as soon as x is known (behind the residual transpose of a
    // layer / behind the input projection); the skip-sum update, the next phase's weight prefetch and own-column staging
    // overlap the hop.
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
    auto publish = [&](unsigned phase) { publish_issue(phase); publish_finish(phase); };
    // debug stamps go straight to memory (held in registers they would cost 20 VGPRs for the whole kernel)
    const bool stamp = p.dbg != nullptr;
#define LOOP_STAMP(i) do { if (stamp && ph == (unsigned)p.dbg_phase && lane == 0) p.dbg[((size_t)tl * 4 + w) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define HEAD_STAMP(i) do { if (stamp && e == p.dbg_phase / p.L && lane == 0) p.dbg[((size_t)tl * 4 + w) * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

    unsigned ph = 0;
    publish(0);
    for (int e = 0; e < p.n_evals; ++e) {
        const int t_e = p.eval_t[e];
        for (int l = 0; l < p.L; ++l, ++ph) {
            const bool last = (l == p.L - 1);
            const float* dsl = dsbuf + (ph & 1) * kC;
            const int dil = p.dil[l];
            LOOP_STAMP(0);

            // (c) the weight stream does not depend on anything computed here: request its first chunks now
            const ConvB<LD> bof1{ytile + 4 * h * LD + kHalo + j, dil, 0};
            GemmPipe<4, 1, LD, 256, 6, ConvB<LD>> pipe1(p.w1p + ((size_t)l * 4 + w) * (96 * 256), lane, 96, bof1);
            pipe1.template start_a<0, 5>();

            // (b) own columns of y = x + step_proj (zero at frames >= T: the conv's zero padding applies to y, net.py:69-71).  They
            //     need nothing from the neighbours, and neither does the first third of the contraction: the K order of the dilated
            //     conv starts with the CENTRE tap of every channel group (ConvB, chunks 0..31), which reads no halo column.
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
            }
            __syncthreads();
            LOOP_STAMP(1);
            // (d1) EVERY wave reads the two neighbour flags now (lanes 0 / 1), without waiting: the value returns under the first chunks
            //      and is tested behind chunk 12.  The neighbours published at the end of their previous phase, so it is normally current
            //      already - no poll round trip with the matrix pipe idle, and, every wave having seen the flags itself, no barrier
            //      between the test and the halo loads.
            unsigned fv = 0xffffffffu;
            if (lane < 2) {
                const bool have = lane ? has_right : has_left;
                if (have) fv = __hip_atomic_load((const gu32*)(p.flags + tile + (lane ? 1 : -1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            DSD_SB();

            // (g) dilated conv, K = 768 (one contraction, taps are column offsets).  The exchange with the neighbour tiles runs UNDER
            //     the centre-tap chunks: flags read at chunk 0 and tested behind chunk 12 (the neighbours published at the end of their
            //     previous phase), their columns requested and in flight during chunks 12..29, written to the y tile in front
            //     of chunk 30; the outer taps (chunks >= 32) are the first to read them.  The hoisted conditioner projection is
            //     fetched half way.
            f32x16 acc[4][1];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
            float4 cpv[4][4];
            pipe1.start_b();
            pipe1.run(acc, 0, 12);
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
            // (e1) request the neighbours' columns (thread = channel row; sc1 loads: the producer stored write-through)
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
            // (e2) halo columns of the y tile
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
            LOOP_STAMP(2);
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
            // step projection of the NEXT phase (next layer, or layer 0 of the next evaluation): requested now, parked in LDS behind the
            // output projection, read by the next phase's staging
            float ds_next = 0.f;
            {
                const bool more = !last || (e + 1 < p.n_evals);
                const int tn_ = last ? p.eval_t[min(e + 1, p.n_evals - 1)] : t_e, ln_ = last ? 0 : l + 1;
                if (more) ds_next = p.ds_table[((size_t)tn_ * p.L + ln_) * kC + tid];
            }

            const float* gl = gtile + 4 * h * GLD + j;
            const TileB bof2{gl, 8 * GLD, 32};
            // gate (net.py:73-74) in registers -> gate tile (called behind the out-proj weight prefetch)
            auto do_gate = [&]() {
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
            LOOP_STAMP(3);
            if (!last) {
                // output projection, all four row blocks (0,1 residual, 2,3 skip) in one pass.  (Splitting it - residual rows
                // first, halo published, skip rows behind - hides the hop but costs more in pipeline restart + second B pass
                // than the hop itself: measured 135.3 vs 133.0 ms per 100-step loop, profiles/r01h.)
                GemmPipe<4, 1, GLD, 256, 6, TileB> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256), lane, 32, bof2);
                pipe2.start_a();
                do_gate();
                __syncthreads();
                LOOP_STAMP(4);
                f32x16 acc2[4][1];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[m][0][r] = 0.f;
                float brow[8];
                pipe2.start_b();
                pipe2.run(acc2, 0, 6);
#pragma unroll
                for (int it = 0; it < 8; ++it) brow[it] = p.b2raw[(size_t)l * 2 * kC + 64 * w + it * 8 + (lane >> 3)];
                DSD_SB();
                pipe2.run(acc2, 6, 32);
                LOOP_STAMP(5);
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
                    constexpr float kInvSqrt2 = 1.0f / 1.41421354f;
                    float4 o;
                    o.x = (x.x + (v.x + bv)) * kInvSqrt2;
                    o.y = (x.y + (v.y + bv)) * kInvSqrt2;
                    o.z = (x.z + (v.z + bv)) * kInvSqrt2;
                    o.w = (x.w + (v.w + bv)) * kInvSqrt2;
                    xreg[it] = o;
                }
                LOOP_STAMP(6);
                dsbuf[((ph + 1u) & 1u) * kC + tid] = ds_next;       // visible behind the barrier inside publish_finish()
                publish_issue(ph + 1u);                             // the halo stores drain while the skip sum is updated
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = get4(acc2[2 + ms][0], q), s = skp[ms][q];
                        skp[ms][q] = (l == 0) ? a : make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                    }
                publish_finish(ph + 1u);
                LOOP_STAMP(7);
            } else {
                // last layer: only the skip half (net.py:126 reads the skips; the residual is dead)
                GemmPipe<2, 1, GLD, 256, 6, TileB> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256) + 2 * 64, lane, 32, bof2);
                pipe2.start_a();
                do_gate();
                __syncthreads();
                f32x16 acc2[2][1];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[m][0][r] = 0.f;
                pipe2.start_b();
                pipe2.run(acc2, 0, 32);
                dsbuf[((ph + 1u) & 1u) * kC + tid] = ds_next;       // visible behind the barriers of the head
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = get4(acc2[ms][0], q), s = skp[ms][q];
                        skp[ms][q] = (l == 0) ? a : make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                    }
            }
            // the y tile is rewritten at the top of the next phase: every wave passed the barrier behind the gate, i.e. is
            // done reading it; the gate tile is rewritten only behind the next phase's barriers
        }

        // ---- head (net.py:126-129) + sampler epilogue for this tile, then the next evaluation's input projection -----------
        HeadParams hp = p.evals[e];
        const bool fuse = (e + 1 < p.n_evals);
        float* stile = ytile;               // [256][32]
        float* htile = gtile;               // [256][32]
        float* ptile = xt;                  // [96][32]
        HEAD_STAMP(0);
        __syncthreads();                    // all waves are out of the last layer's out-proj (gate tile reads)
        // weight streams of the head GEMMs are requested ahead of the barriers that gate their B tiles
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
            const int t = t0 + j;
            // sampler arithmetic (p_sample :134-166 / p_sample_plms :168-204).  All global reads of the 16 elements are issued
            // first (one workgroup per CU: nothing else would hide their latency), then the element-wise math, then the stores.
            // (Issuing the reads in FRONT of the final projection was tried in round 2: vmcnt counts in order, so the GEMM's operand
            // waits then cover the 32 cold reads too - the projection went from 9.3 k to 17.3 k cycles for 4 k saved here,
            // profiles/r02e_loop_timeline.txt.)
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
        if (fuse) { inproj_to_xreg(); publish(ph); }
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
