// dsd_loop.hpp - the WHOLE K-step reverse loop as ONE persistent kernel (gfx950), frame-major LDS tiles, x resident in accumulator-fragment order.
//
// Replaces, for batches that fit the chip (<= one workgroup per CU), the hipGraph of 21 kernels per step
// (usr/diff/shallow_diffusion_tts.py:261-270 loop; per step DiffNet.forward usr/diff/net.py:107-130 + p_sample :159-166 /
// p_sample_plms :168-204).  Same arithmetic, same order, bit-identical results to the per-layer kernels (tests/test_gpu_loop.py) -
// what changes is where the data lives between layers:
//   * a workgroup OWNS one 32-frame tile for the whole loop.  Its x tile (256 channels x 32 frames) and its running skip sum
//     stay in REGISTERS from layer to layer and from step to step; per layer only the hoisted conditioner projection (2 KiB /
//     frame) and the weight stream are read, nothing but the halo is written (the per-layer kernels move 6 KiB / frame and
//     pay a kernel boundary - 1.45 us + the write-back of 16.8 MB of dirty L2 lines - 21 times per step).
//   * the 3-tap dilated conv needs 8 frames of the two NEIGHBOUR tiles' x: each workgroup publishes its first / last 8
//     frames per layer (2 x 8 KiB, write-through sc1 stores), raises a per-tile phase flag, and reads its neighbours'
//     frames with sc1 loads once their flag has reached the phase (MI355X_MICROARCH.md "Inter-workgroup visibility": sc1
//     stores + every storing wave drained + relaxed agent-scope flag; sc1 loads on the consumer).  Halo buffers are double
//     buffered by phase parity: a neighbour can be at most one phase ahead.  No grid-wide barrier anywhere.
//   * all workgroups must be co-resident (they wait for each other): the host launches at most one workgroup per CU and splits
//     larger batches into chunks of whole utterances; every spin is bounded and a timeout is sticky (the loop then finishes instead
//     of hanging, poisons the result with NaN, and the host latches the timeout word into pinned memory: dsd_check, include/dsd.h).
// Tile layout (the row-major form of rounds 1-2 was retired in round 3 after the whole GPU suite ran on this one, profiles/r05_*:
// bit-identical, 0.862 vs 0.850 of the fp32 MFMA peak inside one call):
//   * the conv input y and the gate tile are [frame][channel] with a row stride of 260 floats (conflict-free for the lane groups of
//     ds_read_b128 / ds_write_b128, MI355X_MICROARCH.md section LDS).  A lane of the 32x32x2 MFMA supplies B[k = 4h + s][frame j] for the
//     four steps s of an 8-deep chunk: ONE ds_read_b128 (a [channel][frame] tile needs four ds_read_b32 at a stride of one row).
//   * x lives in fragment order all the time - xq[mb][q] = channels 64 w + 32 mb + 8 q + 4 h + {0..3} of frame j, exactly what the
//     output projection's accumulators hold - so the residual update is register arithmetic, y = x + step is written to the
//     frame-major tile with 8 ds_write_b128, the gate tile likewise, and the frame mask is ONE predicate per lane.
//   * the halo buffers are [side][8 frames][256 channels]; a tile's first / last 8 frames are held by the lanes j < 8 / j >= 24.
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

__device__ __forceinline__ float4 ld16_sc1(const float* base_uniform, int byte_off) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, 0x7ffffff0, 0x00020000);
    const f32x4_ f = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));       // aux 16 = sc1
    return make_float4(f.x, f.y, f.z, f.w);
}

constexpr int kFmLDK = kC + 4;              // row stride of the frame-major tiles: 65 x 16 bytes, an odd number of 16-byte slots
constexpr int kFmY = (32 + 2 * kHalo) * kFmLDK, kFmG = 32 * kFmLDK;
constexpr int kLoopLdsBytes = (kFmY + kFmG + kC * 32 + 2 * kC) * (int)sizeof(float);
static_assert(kFmY >= kC * 32 && kFmG >= kC * 32, "the head reuses the two tiles as [256][32]");

// B functor of the dilated conv over the frame-major y tile: yc = this lane's pointer to (frame row kHalo + j, channel 4 h); a tap is a ROW offset
struct ConvBT {
    const float* yc; int dilrow;            // dil * kFmLDK
    __device__ __forceinline__ const float* operator()(int it, int u) const {
        const int kc = 6 * it + u;
        if (kc < kConvCentre) return yc + kc * 8;
        const int idx = kc - kConvCentre;
        return yc + (idx >> 1) * 8 + ((idx & 1) ? dilrow : -dilrow);
    }
};
// B functor of a frame-major [frame][k] tile: chunk kc at + 8 kc (clamped: the prefetch behind the last chunk stays inside the row)
struct TileBT {
    const float* base; int n;
    __device__ __forceinline__ const float* operator()(int it, int u) const {
        const int kc = 6 * it + u;
        return base + ((kc < n) ? kc : n - 1) * 8;
    }
};

__device__ __forceinline__ float4 fm_add_masked(const float4& x, const float4& d, bool ok) {
    return make_float4(ok ? x.x + d.x : 0.f, ok ? x.y + d.y : 0.f, ok ? x.z + d.z : 0.f, ok ? x.w + d.w : 0.f);
}

// MODE: HEAD_DDPM or HEAD_PLMS (the sampler arithmetic of the head epilogue)
template <int MODE>
__global__ __launch_bounds__(kThreads, 1) void k_loop(const LoopParams p) {
    constexpr int LDK = kFmLDK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem;                    // [48][260] conv input y = x + step_proj (+ halo rows), frame-major; head: scaled skip sum [256][32]
    float* gtile = smem + kFmY;             // [32][260] gate tile, frame-major; head: relu(skip_projection) [256][32]
    float* xt = gtile + kFmG;               // [256][32] scratch: spec tile of the in-projection
    float* dsbuf = xt + kC * 32;            // [2][256]  step projection of phase ph in dsbuf[ph & 1], fetched one phase ahead

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // publish this tile's first / last 8 frames of x as the halo of phase `phase`, [side][frame][channel]: the lanes that hold those frames
    // store their 8 float4 (write-through), every storing wave drains, barrier, ONE relaxed agent-scope flag store (the protocol of k_loop)
    auto publish_issue = [&](unsigned phase) {
        float* hb = p.halo + ((size_t)(phase & 1) * p.ntiles_total + tile) * (2 * kC * 8);
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(hb, 0, 0x7ffffff0, 0x00020000);
        if (j < 8 || j >= 24) {
            const int side = (j >= 24) ? 1 : 0, f = j & 7;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_ v = {xq[mb][q].x, xq[mb][q].y, xq[mb][q].z, xq[mb][q].w};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r, ((side * 8 + f) * kC + ch0 + 32 * mb + 8 * q) * 4, 0, 16);
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
#define HEAD_STAMP(i) do { if (stamp && e == p.dbg_phase / p.L && lane == 0) p.dbg[((size_t)tl * 4 + w) * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

    unsigned ph = 0;
    publish(0);
    for (int e = 0; e < p.n_evals; ++e) {
        const int t_e = p.eval_t[e];
        for (int l = 0; l < p.L; ++l, ++ph) {
            const bool last = (l == p.L - 1);
            const float* dsl = dsbuf + (ph & 1) * kC;
            LOOP_STAMP(0);

            // (c) the weight stream does not depend on anything computed here: request its first chunks now
            const ConvBT bof1{ytile + (kHalo + j) * LDK + 4 * h, (int)p.dil[l] * LDK};
            GemmPipe<4, 1, LDK, 256, 6, ConvBT, 1, true> pipe1(p.w1p + ((size_t)l * 4 + w) * (96 * 256), lane, 96, bof1);
            pipe1.template start_a<0, 5>();

            // (b) own frames of y = x + step_proj (zero at frames >= T: the conv's zero padding applies to y, net.py:69-71): the lane's 32
            //     channels of frame j as 8 ds_write_b128 into row kHalo + j
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = ch0 + 32 * mb + 8 * q;
                    const float4 d = *reinterpret_cast<const float4*>(dsl + c);
                    *reinterpret_cast<float4*>(ytile + (kHalo + j) * LDK + c) = fm_add_masked(xq[mb][q], d, in_t);
                }
            __syncthreads();
            LOOP_STAMP(1);
            // (d1) every wave reads the two neighbour flags now (lanes 0 / 1), tested behind chunk 12
            unsigned fv = 0xffffffffu;
            if (lane < 2) {
                const bool have = lane ? has_right : has_left;
                if (have) fv = __hip_atomic_load((const gu32*)(p.flags + tile + (lane ? 1 : -1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            DSD_SB();

            // (g) dilated conv, K = 768, centre taps first: the exchange with the neighbour tiles runs under them
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
            pipe1.run(acc, 12, 30);
            // (e2) halo rows of the y tile: float4 index tid + 256 g = (frame f = 4 g + tid / 64, channels 4 (tid % 64) ..)
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
                        *reinterpret_cast<float4*>(ytile + ((side ? kHalo + 32 : 0) + f) * LDK + c) = fm_add_masked(hv[side][g], d, have && t < T);
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
            // step projection of the NEXT phase (next layer, or layer 0 of the next evaluation)
            float ds_next = 0.f;
            {
                const bool more = !last || (e + 1 < p.n_evals);
                const int tn_ = last ? p.eval_t[min(e + 1, p.n_evals - 1)] : t_e, ln_ = last ? 0 : l + 1;
                if (more) ds_next = p.ds_table[((size_t)tn_ * p.L + ln_) * kC + tid];
            }

            const TileBT bof2{gtile + j * LDK + 4 * h, 32};
            // gate (net.py:73-74) in registers -> frame-major gate tile, 8 ds_write_b128 (called behind the out-proj weight prefetch)
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
                            g4[ee] = sigmoid_f(acc[pr][0][r] + vg) * tanh_f(acc[pr + 2][0][r] + vf);
                        }
                        *reinterpret_cast<float4*>(gtile + j * LDK + ch0 + 32 * pr + 8 * q) = make_float4(g4[0], g4[1], g4[2], g4[3]);
                    }
            };
            LOOP_STAMP(3);
            if (!last) {
                // output projection, all four row blocks (0,1 residual, 2,3 skip) in one pass
                GemmPipe<4, 1, LDK, 256, 6, TileBT, 1, true> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256), lane, 32, bof2);
                pipe2.start_a();
                do_gate();
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
                GemmPipe<2, 1, LDK, 256, 6, TileBT, 1, true> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256) + 2 * 64, lane, 32, bof2);
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
        if (fuse) { inproj_to_xq(); publish(ph); }
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
