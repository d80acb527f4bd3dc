// train_loop_wino.hpp - the forward of the fused training stack with the dilated convolution as WINOGRAD F(2,3) (gfx950; SURVEY.md section 8
// row f3): k_tr_stack_fwd_w = the layer body of the inference loop k_loop_wino (dsd_loop_wino.hpp: pair-ordered frame-major y tile,
// v_mfma_f32_16x16x4_f32 over 16 output pairs, transformed weights in consumption order with the per-period L2 touch, the conditioner
// projection as the accumulators' initial values, publication merged into the next layer's top) behind the interface of k_tr_stack_fwd
// (train_loop.hpp): one pass over the layers, no sampler head.
//
// The 20 ResidualBlock.forward calls of DiffNet.forward (usr/diff/net.py:119-124; block :66-78) under GaussianDiffusion.p_losses
// (usr/diff/shallow_diffusion_tts.py:213-231).  What this kernel adds to the loop's layer:
//   * x enters channel-major (train.py's input projection) and goes straight into the fragment order of the registers;
//   * the step projection is a row per UTTERANCE and layer (every utterance draws its own t, shallow_diffusion_tts.py:279);
//   * every layer SAVES what its backward needs, in the layouts k_tr_stack_fwd writes (the backward kernels do not know which forward ran):
//     y = x + step projection channel-major with padded rows (k_tr_wgrad's B operand), and the gate pre-activation a = conv + conditioner
//     projection in the 32x32 fragment order k_trb_fused reads - a lane of the 16x16 accumulator (pair p, k group g, row block rb) holds four
//     consecutive channels of frames tE(p) and tE(p) + d: ONE float4 of that order each;
//   * the skip sum (+ the summed skip biases) leaves channel-major, zero tail.
// Results differ from k_tr_stack_fwd by reduction order and the transforms' roundings (tests/test_gpu_train_fused.py: skip, saved y / a and
// every gradient against float64 autograd inside the same tolerances); k_tr_stack_fwd stays the bit-identity anchor of the per-layer kernels
// (dsf_set_stack_conv(0)).
#pragma once
#include "dsd_loop_wino.hpp"
#include "train_loop.hpp"

namespace dsd {

struct TrLoopWinoParams {
    TrLoopParams tp;            // everything k_tr_stack_fwd takes (w1p unused; cp in the Winograd accumulator order: CondProjParams::wino)
    const float4* w1w;          // transformed conv weights of all layers, consumption order [L][128 steps][w4][r4][lane64] (k_pack_wino_multi)
    unsigned wl_bytes;          // bytes of that buffer (the L2 touch's buffer bound)
    int touch_ahead;            // steps the L2 touch runs in front (0 = off)
};

// k_pack_wino over the layers of a pointer table (the weights change every optimiser step: packed per forward call).  One workgroup per
// (layer, wave w, row block rb): its 16 weight rows are 48 KiB contiguous in the source - read as float4 into LDS (row stride 772 floats: the 16
// rows of a fragment column land in different banks), written as the 64 fragment rows (1 KiB each) of the steps that row block takes part in.
// The arithmetic is k_pack_wino's (fp64 sums, one rounding): the same bits.
constexpr int kPackWinoLD = 772;
__global__ __launch_bounds__(256) void k_pack_wino_multi(const TrPtrs src, float* __restrict__ dst) {
    __shared__ __attribute__((aligned(16))) float wt[16 * kPackWinoLD];
    const int tid = threadIdx.x, w = blockIdx.x >> 3, rb = blockIdx.x & 7, hb = rb >> 2, r4 = rb & 3;
    const int row0 = (rb < 4) ? 64 * w + 16 * rb : kC + 64 * w + 16 * (rb - 4);
    const float4* s4 = reinterpret_cast<const float4*>(src.p[blockIdx.y] + (size_t)row0 * kC * 3);
    for (int i = tid; i < 16 * 192; i += 256) {
        const int r = i / 192, c4 = i - r * 192;
        *reinterpret_cast<float4*>(wt + r * kPackWinoLD + 4 * c4) = s4[i];
    }
    __syncthreads();
    float4* d = reinterpret_cast<float4*>(dst + (size_t)blockIdx.y * ((size_t)kWnSteps * 4 * 4 * 64 * 4));
    const int lane = tid & 63, nn = lane & 15, g = lane >> 4;
    for (int combo = tid >> 6; combo < 64; combo += 4) {            // combo = (half * 16 + c) * 2 + pos
        const int pos = combo & 1, c = (combo >> 1) & 15, half = combo >> 5;
        const float* wp = wt + nn * kPackWinoLD + (64 * g + 4 * c) * 3;
        const float4 a = *reinterpret_cast<const float4*>(wp), b = *reinterpret_cast<const float4*>(wp + 4), cc = *reinterpret_cast<const float4*>(wp + 8);
        const float t[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, cc.x, cc.y, cc.z, cc.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double g0 = t[3 * e], g1 = t[3 * e + 1], g2 = t[3 * e + 2];
            double u;
            if (half == 0) u = pos ? 0.5 * (g0 - g1 + g2) : 0.5 * (g0 + g1 + g2);
            else u = pos ? g2 : g0;
            o[e] = (float)u;
        }
        const int st = combo * 2 + hb;
        d[(((size_t)st * 4 + w) * 4 + r4) * 64 + lane] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

constexpr int kTrStackWinoLdsBytes = kLoopTouchLds + (kWnY + kFmG + 2 * kC) * (int)sizeof(float);

__global__ __launch_bounds__(kThreads, 1) void k_tr_stack_fwd_w(const TrLoopWinoParams pw) {
    constexpr int LDK = kFmLDK, S = 4;
    const TrLoopParams& p = pw.tp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem + kLoopTouchLds / 4;   // pair-ordered y tile: E rows [0, 24), O rows [-8, 16) at kWnOBase
    float* gtile = ytile + kWnY;               // [32][260] gate tile, frame-major, natural frame order
    float* dsbuf = gtile + kFmG;               // [2][256]  step projection of layer l in dsbuf[l & 1], fetched one layer ahead

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int pp = lane & 15, gg = lane >> 4;   // the 16x16x4 fragment's pair column and k group
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tl = xcd_item(blockIdx.x, p.n_tiles >> 3, p.n_tiles & 7);         // neighbours behind one L2
    L2TouchP tc;
    {
        const unsigned long long wb = (unsigned long long)pw.w1w;
        const int xcd = (int)(blockIdx.x & 7), nwx = 4 * ((p.n_tiles - xcd + 7) >> 3), q = 4 * (int)(blockIdx.x >> 3) + w;
        const bool en = pw.touch_ahead > 0 && nwx >= 8;
        tc.rs = L2Touch::i32x4_{(int)(unsigned)wb, (int)(unsigned)((wb >> 32) & 0xffffu), (int)pw.wl_bytes, 0x00020000};
        tc.ahead = (unsigned)(pw.touch_ahead + 7) / 8u;
        tc.nwx = nwx;
        tc.dec = en ? 16 % nwx : 0;
        tc.r = en ? q : 1 << 20;
        tc.gtot = (unsigned)p.L * (unsigned)(kWnSteps / 8);
        tc.lds = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float*)smem + (unsigned)w * 256u;
        tc.lane128 = (unsigned)lane * 128u;
    }
    const int tile = p.tile_base + tl;
    const int b = tile / p.ntile32, tn = tile - b * p.ntile32, t0 = tn * 32;
    const bool has_left = tn > 0, has_right = tn + 1 < p.ntile32;
    const int T = p.T;
    const bool in_t = t0 + j < T;

    float4 xq[2][4];        // x tile in fragment order: xq[mb][q] = channels 64 w + 32 mb + 8 q + 4 h + {0,1,2,3} of frame j
    float4 skp[2][4];       // running skip sum of this wave's skip rows, the same order
    const int ch0 = 64 * w + 4 * h;

    auto timed_out = [&]() -> bool { return __hip_atomic_load((gu32*)p.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; };

    {
        // channel-major x: a wave load covers the 32 frames of two channels (two 128-byte segments)
        const float* xin = p.x0 + ((size_t)b * kC + ch0) * p.TS + t0 + j;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* s = xin + (size_t)(32 * mb + 8 * q) * p.TS;
                xq[mb][q] = make_float4(s[0], s[p.TS], s[2 * p.TS], s[3 * p.TS]);
            }
    }
    dsbuf[tid] = p.step[(size_t)b * p.L * kC + tid];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int q = 0; q < 4; ++q) skp[ms][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    // the halo protocol of k_loop_wino: first / last 8 frames of x as write-through stores; drained, barrier and flag at the top of the NEXT layer
    auto publish_issue = [&](unsigned phase) {
        float* hb = p.halo + ((size_t)(phase & 1) * p.ntiles_total + tile) * (2 * kC * 8);
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(hb, 0, 0x7ffffff0, 0x00020000);
        if (j < 8 || j >= 24) {
            const int side = (j >= 24) ? 1 : 0, f = j & 7;
            const int vo = ((side * 8 + f) * kC + ch0) * 4;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_ v = {xq[mb][q].x, xq[mb][q].y, xq[mb][q].z, xq[mb][q].w};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r, vo + (32 * mb + 8 * q) * 4, 0, 16);
                }
        }
    };

    f32x4w acc[2][8];
    auto load_cp = [&](int l) {
        const float4* cpl = p.cp + (size_t)l * p.cp_lstride + ((size_t)tile * 4 + w) * (2 * 8 * 64);       // wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rb = 0; rb < 8; ++rb) {
                const float4 c = ld16_u(cpl, ((i * 8 + rb) * 64 + lane) * 16);
                acc[i][rb] = f32x4w{c.x, c.y, c.z, c.w};
            }
    };
    load_cp(0);

    publish_issue(0);
    for (int l = 0; l < p.L; ++l) {
        const unsigned ph = (unsigned)l;
        const bool last = (l == p.L - 1);
        const float* dsl = dsbuf + (l & 1) * kC;
        const int dil = (int)p.dil[l], de = __builtin_ctz((unsigned)dil);

        WinoPipe<S> pipe1(pw.w1w + (size_t)w * 256, lane, l, ytile + pp * LDK + 64 * gg, ytile + kWnOBase + (8 + pp) * LDK + 64 * gg, dil * LDK, tc);
        pipe1.template start_a<S - 1>();

        // own frames of y = x + step_proj (zero at frames >= T: net.py:69-71 pads the conv INPUT) -> the frame's row of the pair-ordered tile,
        // and channel-major to the backward's copy (32 dword stores per lane: a wave store covers two 128-byte row segments)
        {
            float* yrow = ytile + wn_row_of_frame(j, de);
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc(p.y_cm + (size_t)l * p.y_lstride + (size_t)b * kC * p.y_rs, 0, 0x7ffffff0, 0x00020000);
            const int yvo = (ch0 * p.y_rs + t0 + j) * 4;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = ch0 + 32 * mb + 8 * q;
                    const float4 d = *reinterpret_cast<const float4*>(dsl + c);
                    const float4 v = fm_add_masked(xq[mb][q], d, in_t);
                    *reinterpret_cast<float4*>(yrow + c) = v;
                    const int so = (32 * mb + 8 * q) * p.y_rs * 4;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.x), ry, yvo, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.y), ry, yvo, so + p.y_rs * 4, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.z), ry, yvo, so + p.y_rs * 8, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.w), ry, yvo, so + p.y_rs * 12, 0);
                }
        }
        // this tile's halo frames of phase ph are visible once every wave has drained; the barrier is the one the y tile needs anyway
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gu32*)(p.flags + tile), ph + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned fv = 0xffffffffu;
        if (lane < 2) {
            const bool have = lane ? has_right : has_left;
            if (have) fv = __hip_atomic_load((const gu32*)(p.flags + tile + (lane ? 1 : -1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        DSD_SB();

        // first half: M1 (acc[0]) and M2 (acc[1]) on top of the conditioner projection's halves, the tile's own frames only
        pipe1.start_b();
        pipe1.template run<1, 0, 0>(acc);
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
                const int off = (((tile + (side ? 1 : -1)) * 2 + (side ? 0 : 1)) * (8 * kC) + 4 * tid) * 4;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    hv[side][g] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (have) hv[side][g] = ld16_sc1(hbase, off + g * (4 * kC * 4));
                }
            }
        }
        DSD_SB();
        pipe1.template run<4, 0, 0>(acc);
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
                    float* dst = side ? ytile + (16 + f) * LDK + c : ytile + kWnOBase + f * LDK + c;
                    *reinterpret_cast<float4*>(dst) = fm_add_masked(hv[side][g], d, have && t < T);
                }
            }
        }
        __syncthreads();
        pipe1.template run<2, 0, 0>(acc);
        pipe1.template run<1, 0, 1>(acc);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
            const f32x4w m1 = acc[0][rb], m2 = acc[1][rb];
            acc[0][rb] = m1 + m2;
            acc[1][rb] = m1 - m2;
        }
        DSD_SB();
        pipe1.template run<8, 1, 1>(acc);
        float ds_next = 0.f;
        if (!last) ds_next = p.step[((size_t)b * p.L + l + 1) * kC + tid];

        const TileBT bof2{gtile + j * LDK + 4 * h, 32};
        // the gate pre-activation a (the accumulators: conv + conditioner projection + both biases) -> the backward's copy in the 32x32 fragment
        // order [tile][w4][mb4][q4][h2][j32] of float4: rows 16 rb' + 4 g + {0..3} of the wave's 64 gate (rb < 4) / filter rows = block
        // mb = 2 (rb >> 2) + (rb' >> 1), q = 2 (rb' & 1) + (g >> 1), h = g & 1; and the gate (net.py:73-74) -> frame-major gate tile
        auto save_a_and_gate = [&]() {
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            const int tE = wn_frame_of_pair(pp, de);
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                p.a_frag + (size_t)l * p.a_lstride + ((size_t)tile * 4 + w) * (4 * 4 * 64), 0, 0x7ffffff0, 0x00020000);
            const int avo = ((gg >> 1) * 64 + (gg & 1) * 32 + tE) * 16;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int vo = avo + (hf ? dil * 16 : 0);
#pragma unroll
                for (int rb = 0; rb < 8; ++rb) {
                    const int so = (((rb >> 2) * 2 + ((rb & 3) >> 1)) * 4 + 2 * (rb & 1)) * 64 * 16;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, acc[hf][rb]), ra, vo, so, 0);
                }
                float* grow = gtile + (tE + (hf ? dil : 0)) * LDK + 64 * w + 4 * gg;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    float g4[4];
#pragma unroll
                    for (int ee = 0; ee < 4; ++ee) g4[ee] = sigmoid_f(acc[hf][rb][ee]) * tanh_f(acc[hf][rb + 4][ee]);
                    *reinterpret_cast<float4*>(grow + 16 * rb) = make_float4(g4[0], g4[1], g4[2], g4[3]);
                }
            }
        };
        if (!last) {
            GemmPipe<4, 1, LDK, 256, 6, TileBT, 1, true> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256), lane, 32, bof2);
            pipe2.start_a();
            save_a_and_gate();
            load_cp(l + 1);                                     // into the (dead) accumulators, in front of the barrier (dsd_loop_wino.hpp)
            dsbuf[((l + 1) & 1) * kC + tid] = ds_next;
            __syncthreads();
            f32x16 acc2[4][1];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[m][0][r] = 0.f;
            float4 bq[2][4];
            pipe2.start_b();
            pipe2.run(acc2, 0, 6);
            {
                const float* b2l = p.b2.p[l];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[mb][q] = *reinterpret_cast<const float4*>(b2l + ch0 + 32 * mb + 8 * q);
            }
            DSD_SB();
            pipe2.run(acc2, 6, 32);
            // residual in place: x' = (x + res + b) / sqrt(2)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = get4(acc2[mb][0], q), x = xq[mb][q], bv = bq[mb][q];
                    xq[mb][q] = make_float4((x.x + (v.x + bv.x)) * kTrInvSqrt2, (x.y + (v.y + bv.y)) * kTrInvSqrt2,
                                            (x.z + (v.z + bv.z)) * kTrInvSqrt2, (x.w + (v.w + bv.w)) * kTrInvSqrt2);
                }
            publish_issue(ph + 1u);
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = get4(acc2[2 + ms][0], q), s = skp[ms][q];
                    skp[ms][q] = (l == 0) ? a : make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                }
        } else {
            GemmPipe<2, 1, LDK, 256, 6, TileBT, 1, true> pipe2(p.w2p + ((size_t)l * 4 + w) * (32 * 256) + 2 * 64, lane, 32, bof2);
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

    // skip sum + summed skip biases -> channel-major, zero tail; a wait that hit its spin bound leaves garbage: make it LOUD - NaN instead
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
