// dsd_lat_wino.hpp - the dilated convolution of the LATENCY kernels (dsd_lat.hpp: k_lat_conv<G>, one ResidualBlock of usr/diff/net.py:66-78 as two
// row-split kernels) as Winograd F(2,3) along the frame axis - the arithmetic of the persistent loop (dsd_loop_wino.hpp) for the batches that
// fill less than half of the chip: the reference's own inference shape, one utterance per device (configs/tts/fs2.yaml:70).
//
// k_lat_conv_w<G> replaces k_lat_conv<G> for G = 2 / 4 / 8 when the handle's convolution mode is Winograd (the default; at G = 16 the direct kernel
// is faster and stays: dsd.hip launch_layer); k_lat_out<G> and the head kernels are unchanged.  Same row ownership as the direct kernels in units of 16 rows: workgroup g of a tile computes 256 / G gate rows and THEIR filter
// rows, so the gate never leaves a lane:
//   G = 2: wave wv <- gate blocks 8 g + 2 wv, + 1 and their filter blocks, the whole K            (4 row blocks of 16)
//   G = 4: wave wv <- gate block 4 g + wv and its filter block, the whole K                         (2 row blocks)
//   G = 8: wave wv <- gate block 2 g + (wv & 1) + filter block, K half wv >> 1 (8 chunks of 16 channels)
//   G = 16: wave wv <- gate block g + filter block, K quarter wv (4 chunks)
// K partials (G = 8 / 16) meet in LDS AFTER the output transform - two values per (row, pair) instead of four - and are added in wave order.
// What is shared with the loop: the pair-ordered frame-major y tile (wn_row_of_frame), the fragment maps of v_mfma_f32_16x16x4_f32, the
// halo-free half first / output transform / second half on 2 accumulator sets, and the TRANSFORMED WEIGHT STREAM itself - every (half, chunk,
// product, row block) fragment this kernel needs is a 1 KiB piece of the loop's consumption-order stream w1w (k_pack_wino), read by offset.
// The hoisted conditioner projection stays in the 32x32 fragment order of the direct kernels (a float4 there = four consecutive rows of one
// frame = exactly what a lane of the 16x16 accumulator holds).  Results differ from the direct kernels by reduction order and the transforms'
// roundings (tests/test_gpu_wino.py: every G against the oracle's layer, K = 100 against the oracle).
#pragma once
#include "dsd_lat.hpp"
#include "dsd_loop_wino.hpp"

namespace dsd {

constexpr int kLatConvWLdsBytes = (kWnY + 4 * 1024) * (int)sizeof(float);      // pair-ordered y tile + K partials [4 waves][2 sets][2 blocks][4][64]

template <int G>
__global__ __launch_bounds__(kThreads, 2) void k_lat_conv_w(const LatParams p) {
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "row split (G = 16 is written and correct - tests/test_wino_model.py, profiles/r5_10 - but not instantiated: the direct kernel is faster there)");
    constexpr int LDK = kFmLDK, TILE = kC * 32;
    constexpr int NRB = (G == 2) ? 4 : 2;                      // row blocks of 16 per wave: NRB / 2 gate blocks + their filter blocks
    constexpr int NGB = NRB / 2;
    constexpr int NCW = (G == 16) ? 4 : (G == 8) ? 8 : 16;     // 16-channel chunks per wave and half
    constexpr int STG = 4;                                     // register stages of the A fragments (groups of 2 products x NRB blocks)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem;                    // pair-ordered y tile (dsd_loop_wino.hpp)
    float* red = smem + kWnY;               // K partials
    const int tid = threadIdx.x, lane = tid & 63, pp = lane & 15, gg = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile, g;
    if (!lat_map<G>(p.ntiles, tile, g)) return;
    const int b = tile / p.ntile32, tn = tile - b * p.ntile32, t0 = tn * 32;
    const bool has_left = tn > 0, has_right = tn + 1 < p.ntile32;
    const int T = p.T, dil = p.dil, de = __builtin_ctz((unsigned)dil);

    // role of this wave: first gate block (of 16 rows), first chunk of its K range
    const int B0 = (G == 2) ? 8 * g + 2 * wv : (G == 4) ? 4 * g + wv : (G == 8) ? 2 * g + (wv & 1) : g;
    const int c0 = (G == 8) ? 8 * (wv >> 1) : (G == 16) ? 4 * wv : 0;

    // A fragments: group gi = half * NCW + ci covers chunk c0 + ci of that half: [product 2][row block NRB] pieces of 1 KiB of the loop's stream -
    // step = ((half * 16 + c) * 2 + product) * 2 + (filter ? 1 : 0), piece (wave w = B >> 2, r4 = B & 3) of it
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(p.w1w), 0, 0x7ffffff0, 0x00020000);
    float4 a[STG][2][NRB];
    // byte offset of a piece = step * 16 KiB + (w * 4 + r4) * 1 KiB: the block's part goes into the lane offset (one register per gate block, kept),
    // the (half, chunk) part is ONE scalar per group, product and gate / filter are immediates
    int vo[NGB];
#pragma unroll
    for (int k = 0; k < NGB; ++k) {
        const int B = B0 + k;
        vo[k] = lane * 16 + ((B >> 2) * 4 + (B & 3)) * 1024;
        asm volatile("" : "+v"(vo[k]));
    }
    auto lda = [&](float4 (&dst)[2][NRB], int gi) {
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const int cb = c0 + gi + ((gi >= NCW) ? 16 - NCW : 0);          // half * 16 + chunk
        const int sg = cb * (4 * kWnStepBytes);
#pragma unroll
        for (int pos = 0; pos < 2; ++pos)
#pragma unroll
            for (int k = 0; k < NRB; ++k) {
                const f32x4_ v = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(rs, vo[k % NGB], sg + (pos * 2 + k / NGB) * kWnStepBytes, 0));
                dst[pos][k] = make_float4(v.x, v.y, v.z, v.w);
            }
    };
    constexpr int NG = 2 * NCW;             // groups per wave
#pragma unroll
    for (int i = 0; i < STG - 1; ++i) lda(a[i], i);
    DSD_SB();

    // stage y = x + step_proj in PAIR order (zero at frames outside [0, T): the conv's zero padding applies to y, net.py:69-71)
    const int tstep = p.t_dev ? p.t_dev[b] : p.t_uniform;
    const float* __restrict__ dsl = p.ds + (size_t)tstep * p.ds_tstride;
    const float* __restrict__ xt = p.x_in + (size_t)tile * TILE;
    {
        // own frames: 512 items (channel quad c4, frame quad q), two per thread: four 16-byte loads (rows 4 c4 + e of the tile-major x tile: the
        // eight lanes of a frame-quad row cover one 128-byte line), a 4 x 4 transpose in registers, four ds_write_b128 into the frames' rows.
        // halo: 256 items (side, frame quad, c4), one per thread, the same way.
        float4 xr[2][4], hr[4];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c4 = (tid >> 3) + 32 * it, q = tid & 7;
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[it][e] = *reinterpret_cast<const float4*>(xt + (4 * c4 + e) * 32 + 4 * q);
        }
        const int hside = tid >> 7, hq = (tid >> 6) & 1, hc4 = tid & 63;
        const bool hhave = hside ? has_right : has_left;
        {
            const float* src = hside ? xt + TILE + 4 * hq : xt - TILE + 24 + 4 * hq;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                hr[e] = hhave ? *reinterpret_cast<const float4*>(src + (4 * hc4 + e) * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c4 = (tid >> 3) + 32 * it, q = tid & 7;
            const float4 d = *reinterpret_cast<const float4*>(dsl + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = t0 + 4 * q + e < T;
                const float4 v = make_float4(f4at(xr[it][0], e) + d.x, f4at(xr[it][1], e) + d.y, f4at(xr[it][2], e) + d.z, f4at(xr[it][3], e) + d.w);
                *reinterpret_cast<float4*>(ytile + wn_row_of_frame(4 * q + e, de) + 4 * c4) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        {
            // left frame f (t = t0 - 8 + f) is row O[f - 8], right frame f (t = t0 + 32 + f) is row E[16 + f]
            const float4 d = *reinterpret_cast<const float4*>(dsl + 4 * hc4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 4 * hq + e, t = hside ? t0 + 32 + f : t0 - kHalo + f;
                const bool ok = hhave && t < T;
                const float4 v = make_float4(f4at(hr[0], e) + d.x, f4at(hr[1], e) + d.y, f4at(hr[2], e) + d.z, f4at(hr[3], e) + d.w);
                float* dst = hside ? ytile + (16 + f) * LDK + 4 * hc4 : ytile + kWnOBase + f * LDK + 4 * hc4;
                *reinterpret_cast<float4*>(dst) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __syncthreads();

    // the hoisted conditioner projection (+ both biases) of this wave's rows, direct (32x32 fragment) layout: block B = rows 16 B .. of the gate
    // (+ 256: filter) half = stream w = B >> 2, row block mb = ((B & 3) >> 1) (+ 2), quad q = 2 (B & 1) + (gg >> 1), lane half h = gg & 1; this
    // lane needs the float4 of frames tE and tO.  Requested now, used behind the contraction.
    const int tE = wn_frame_of_pair(pp, de), tO = tE + dil;
    float4 cpv[2][NRB];
    {
        const float4* cpl = p.cp + (size_t)tile * (4 * 4 * 4 * 64);
        // (with a K split only the lanes / registers a wave finishes are needed; loading all keeps the code uniform - 2 x NRB loads)
#pragma unroll
        for (int k = 0; k < NRB; ++k) {
            const int B = B0 + (k % NGB), f = k / NGB;
            const int base = (((B >> 2) * 4 + ((B & 3) >> 1) + 2 * f) * 4 + 2 * (B & 1) + (gg >> 1)) * 64 + 32 * (gg & 1);
            cpv[0][k] = cpl[base + tE];
            cpv[1][k] = cpl[base + tO];
        }
    }
    DSD_SB();

    // operand rows of pair pp: E[p], O[p], O[p - d], E[p + d]; k group gg <-> channels [64 gg, 64 gg + 64)
    const float* pE = ytile + pp * LDK + 64 * gg + 4 * c0;
    const float* pO = ytile + kWnOBase + (8 + pp) * LDK + 64 * gg + 4 * c0;
    const float* pOm = pO - dil * LDK;
    const float* pEp = pE + dil * LDK;
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    auto pk_sub = [](f32x2_ x, f32x2_ y) -> f32x2_ {
        f32x2_ d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
        return d;
    };
    // B operands of group (half, ci): product 0 / 1 of the half, four k steps
    auto ldb = [&](float (&v)[2][4], int half, int ci) {
        const int o = 4 * ci;
        float4 r0, r1, r2, r3;
        if (half == 0) {
            r0 = *reinterpret_cast<const float4*>(pE + o);
            r1 = *reinterpret_cast<const float4*>(pO + o);
#pragma unroll
            for (int s = 0; s < 4; s += 2) {
                const f32x2_ x0 = {f4at(r0, s), f4at(r0, s + 1)}, x1 = {f4at(r1, s), f4at(r1, s + 1)};
                const f32x2_ p0 = x0 + x1, p1 = pk_sub(x1, x0);                 // d1 + d2, d2 - d1
                v[0][s] = p0[0]; v[0][s + 1] = p0[1]; v[1][s] = p1[0]; v[1][s + 1] = p1[1];
            }
        } else {
            r0 = *reinterpret_cast<const float4*>(pOm + o);
            r1 = *reinterpret_cast<const float4*>(pO + o);
            r2 = *reinterpret_cast<const float4*>(pEp + o);
            r3 = *reinterpret_cast<const float4*>(pE + o);
#pragma unroll
            for (int s = 0; s < 4; s += 2) {
                const f32x2_ x0 = {f4at(r0, s), f4at(r0, s + 1)}, x1 = {f4at(r1, s), f4at(r1, s + 1)};
                const f32x2_ x2 = {f4at(r2, s), f4at(r2, s + 1)}, x3 = {f4at(r3, s), f4at(r3, s + 1)};
                const f32x2_ p0 = pk_sub(x0, x1), p1 = pk_sub(x2, x3);          // d0 - d2, d3 - d1
                v[0][s] = p0[0]; v[0][s + 1] = p0[1]; v[1][s] = p1[0]; v[1][s + 1] = p1[1];
            }
        }
    };

    f32x4w acc[2][NRB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < NRB; ++k) acc[i][k] = f32x4w{0.f, 0.f, 0.f, 0.f};
    float vb[2][2][4];
    ldb(vb[0], 0, 0);
    DSD_SB();
    // one group: request the fragments of group gi + STG - 1, the B operands of group gi + 1, multiply group gi
    auto group = [&](const float4 (&af)[2][NRB], float4 (&nxt)[2][NRB], const float (&v)[2][4], float (&vn)[2][4], int gi) {
        lda(nxt, min(gi + STG - 1, NG - 1));                            // (past the end: the last group once more - valid addresses, values unused)
        const int gn = min(gi + 1, NG - 1);
        ldb(vn, (gn >= NCW) ? 1 : 0, gn - ((gn >= NCW) ? NCW : 0));
#pragma unroll
        for (int pos = 0; pos < 2; ++pos)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int k = 0; k < NRB; ++k)
                    acc[pos][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4at(af[pos][k], s), v[pos][s], acc[pos][k], 0, 0, 0);
        // the group's loads one by one behind its first MFMAs, then the LDS reads of the next operands, the packed adds in one gap
#pragma unroll
        for (int i = 0; i < 2 * NRB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NRB, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * NRB - 2 * NRB - 4 - 2 * NRB, 0);
        DSD_SB();
    };
#pragma nounroll
    for (int g0 = 0; g0 < NCW; g0 += 4) {
        group(a[0], a[3], vb[0], vb[1], g0 + 0);
        group(a[1], a[0], vb[1], vb[0], g0 + 1);
        group(a[2], a[1], vb[0], vb[1], g0 + 2);
        group(a[3], a[2], vb[1], vb[0], g0 + 3);
    }
    // output transform, first part: t = M1 + M2 (frame tE), u = M1 - M2 (frame tO); the second half accumulates M0 onto t, M3 onto u
#pragma unroll
    for (int k = 0; k < NRB; ++k) {
        const f32x4w m1 = acc[0][k], m2 = acc[1][k];
        acc[0][k] = m1 + m2;
        acc[1][k] = m1 - m2;
    }
    DSD_SB();
#pragma nounroll
    for (int g0 = NCW; g0 < NG; g0 += 4) {
        group(a[0], a[3], vb[0], vb[1], g0 + 0);
        group(a[1], a[0], vb[1], vb[0], g0 + 1);
        group(a[2], a[1], vb[0], vb[1], g0 + 2);
        group(a[3], a[2], vb[1], vb[0], g0 + 3);
    }

    // epilogue: acc[hf][k]: frame tE (hf = 0) / tO (hf = 1), block k (k < NGB: gate block B0 + k, else its filter block), rows 4 gg + r
    float* gout = p.gbuf + (size_t)tile * TILE;
    if constexpr (G == 2 || G == 4) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int t = hf ? tO : tE;
#pragma unroll
            for (int k = 0; k < NGB; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gv = sigmoid_f(acc[hf][k][r] + f4at(cpv[hf][k], r)) * tanh_f(acc[hf][NGB + k][r] + f4at(cpv[hf][NGB + k], r));
                    gout[(16 * (B0 + k) + 4 * gg + r) * 32 + t] = gv;
                }
        }
    } else {
        // K partials (G = 8: two halves, G = 16: four quarters) through LDS, added in wave order; a wave finishes the registers of one frame half
        // (G = 8: set hf = wv >> 1 of its block pair) / two registers of one frame half (G = 16: set wv >> 1, registers 2 (wv & 1), + 1)
        float* mine = red + wv * 1024;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[((hf * 2 + k) * 4 + r) * 64 + lane] = acc[hf][k][r];
        __syncthreads();
        const int hf = wv >> 1, t = hf ? tO : tE;
        if constexpr (G == 8) {
            const float* pa = red + (wv & 1) * 1024;                    // the wave of my block pair with the first K half, + 2048: the second
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ag = pa[((hf * 2 + 0) * 4 + r) * 64 + lane] + pa[2048 + ((hf * 2 + 0) * 4 + r) * 64 + lane];
                const float af = pa[((hf * 2 + 1) * 4 + r) * 64 + lane] + pa[2048 + ((hf * 2 + 1) * 4 + r) * 64 + lane];
                const float gv = sigmoid_f(ag + f4at(hf ? cpv[1][0] : cpv[0][0], r)) * tanh_f(af + f4at(hf ? cpv[1][1] : cpv[0][1], r));
                gout[(16 * B0 + 4 * gg + r) * 32 + t] = gv;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 2 * (wv & 1) + q;
                const float* pg = red + ((hf * 2 + 0) * 4 + r) * 64 + lane;
                const float* pf = red + ((hf * 2 + 1) * 4 + r) * 64 + lane;
                const float ag = ((pg[0] + pg[1024]) + pg[2048]) + pg[3072];
                const float af = ((pf[0] + pf[1024]) + pf[2048]) + pf[3072];
                const float gv = sigmoid_f(ag + f4at(hf ? cpv[1][0] : cpv[0][0], r)) * tanh_f(af + f4at(hf ? cpv[1][1] : cpv[0][1], r));
                gout[(16 * B0 + 4 * gg + r) * 32 + t] = gv;
            }
        }
    }
}

}  // namespace dsd
