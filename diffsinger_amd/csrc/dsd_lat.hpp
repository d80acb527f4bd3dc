// dsd_lat.hpp - LATENCY MODE of the residual layer (gfx950): one ResidualBlock (usr/diff/net.py:66-78) as TWO row-split kernels.
//
// Why.  k_layer / k_loop give one workgroup (= one CU) a whole 32-frame tile: 8192 MFMAs per layer per tile.  The reference's own
// inference shape is ONE utterance per device (configs/tts/fs2.yaml:70 max_eval_sentences: 1): T = 512 frames are 16 tiles = 16 of 256
// CUs, and the K = 100 loop takes the same ~130 ms as a chip-filling batch.  Frames cannot be split finer than the MFMA's N = 32, so the
// OUTPUT ROWS of the two contractions are split over G workgroups per tile instead (G = 2, 4 or 8 -> up to 8 x more CUs per utterance):
//
//   k_lat_conv<G>   every workgroup stages the whole y = x + step tile (256 channels x (32 + 2 x 8) frames, zero outside [0, T):
//                   net.py:69-71) and computes 512 / G rows of the dilated K = 768 contraction - a gate block together with its filter
//                   block - adds the hoisted conditioner projection, applies sigmoid * tanh (net.py:73-74) and writes its 256 / G rows of
//                   the gate tile to HBM/L2.
//   k_lat_out<G>    every workgroup stages the whole gate tile (256 x 32) and computes 512 / G rows of the output projection (K = 256):
//                   residual rows -> x' = (x + res + b) / sqrt(2), skip rows -> running skip sum (net.py:76-78), same buffers and layouts
//                   as k_layer (tile-major x, fragment-order skip), so k_inproj / k_head / the sampler epilogues are shared.
//
// The all-gather a persistent formulation would need twice per layer (gate rows, then x' rows) is the kernel boundary here: both kernels
// are nodes of the K-step hipGraph (2 x 20 + 1 nodes per evaluation), ~1.5-2 us each - of the same order as a flag hop through L2.
//
// Row ownership follows the packed weight streams (dsd.hip pack_a): stream "wave" w4 in [0,4) holds gate / residual rows [64 w4, 64 w4 +
// 64) as row blocks 0, 1 and the matching filter / skip rows as row blocks 2, 3.
//   G = 2: workgroup g <- streams {2g, 2g+1}; wave = (stream, pair p): row blocks {p, p + 2}, the whole K            (NMB = 2)
//   G = 4: workgroup g <- stream g;           wave = row block; conv: the filter waves hand tanh(.) to the gate waves through LDS
//   G = 8: workgroup g <- stream g / 2, pair g % 2; conv: wave = (gate | filter, K half) - the two K halves are summed through LDS
//          (the ONLY place where the summation order differs from k_layer: (first half) + (second half) instead of one k-ordered chain);
//          out-proj: wave = (residual | skip block, K half), summed the same way.
//   G = 16 (round 3: one utterance of <= 512 frames = 16 tiles on ALL 256 CUs): conv: workgroup g <- ONE 32-row block of a second packing
//          (w1q) that holds 16 gate rows and THEIR 16 filter rows - register r (gate) and r + 8 (filter) of a lane are a pair, the gate
//          never leaves the lane; the four waves split K = 768 four ways (24 chunks = 96 MFMAs each, half the G = 8 kernel's matrix time),
//          partial blocks summed through LDS in wave order (lat_ksum4).  out-proj: workgroup g <- row block (stream g / 4, block g % 4), the
//          four waves split K = 256 four ways.
// G = 2 and G = 4 are bit-identical to k_layer; G = 8 and G = 16 agree to reduction-order noise (tests/test_gpu_latency.py).
#pragma once
#include "dsd_kernels.hpp"

namespace dsd {

struct LatParams {
    const float* x_in;      // [tiles][C][32] tile-major
    float* x_out;           // [tiles][C][32]
    float* gbuf;            // [tiles][C][32] gate tile (k_lat_conv -> k_lat_out)
    const float4* w1p;      // this layer's dilated conv, packed [w4][kc96: centre tap first, conv_chunk()][mb4][lane64]
    const float4* w1q;      // the same weights for G = 16: [b16][kc96][lane64], block b16 = gate rows [16 b16, +16) + their filter rows
    const float4* w2p;      // this layer's output projection, packed [w4][kc32][mb4][lane64]
    const float* b2;        // output projection bias [2C] (residual half)
    const float4* cp;       // this layer's hoisted conditioner projection (+ biases) [tile][w4][mb4][q4][lane64]
    float4* skip;           // running skip sum [tile][w4][mb2][q4][lane64]
    const float* ds;        // this layer's step-projection table: ds[t * ds_tstride + c]
    const int* t_dev;       // per-utterance step index, or nullptr -> t_uniform
    int t_uniform, ds_tstride;
    int T, ntile32, ntiles, dil, first, last;
    const float4* w1w;      // this layer's Winograd-transformed conv weights in the persistent loop's consumption order (k_lat_conv_w, dsd_lat_wino.hpp)
};

constexpr int kLatConvLdsBytes = (kC * (32 + 2 * kHalo) + 4 * 32 * 32) * (int)sizeof(float);     // y tile + K partials [2] (G = 8) / [4] (G = 16) / filter [1..2]
constexpr int kLatOutLdsBytes = (kC * 32 + 3 * 32 * 32) * (int)sizeof(float);                     // gate tile + K partials [2] (G = 8) / [3] (G = 16)

// workgroup -> (tile, g): the G workgroups of a tile read the same x / gate tile, so they are placed behind the same L2 (workgroups
// are dealt round-robin to the 8 XCDs by linear id): XCD x takes tiles x, x + 8, ...   grid = ceil(ntiles / 8) * 8 * G
template <int G>
__device__ __forceinline__ bool lat_map(int ntiles, int& tile, int& g) {
    const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
    tile = (k / G) * 8 + xcd;
    g = k % G;
    return tile < ntiles;
}
__host__ __device__ inline int lat_grid(int ntiles, int G) { return (ntiles + 7) / 8 * 8 * G; }

// G = 8 splits the K range between wave pairs: the first chunk of a wave is a run-time value, so its chunk -> pointer map is the
// branch-free form (ConvB<LD, true>; the branching one let hipcc sink the weight prefetch to its use: 39.4 -> 32.9 ms per 1 x 512 K = 100
// call, bit-identical, profiles/r05_fm_lat_bf_probe_1x512.json)
__device__ __forceinline__ void lat_ksum4(f32x16& acc, float* red, int wv, int j, int h);

template <int G>
__global__ __launch_bounds__(kThreads, 2) void k_lat_conv(const LatParams p) {
    constexpr bool BF = (G >= 8);
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "row split");
    constexpr int LD = 32 + 2 * kHalo, TILE = kC * 32;
    constexpr int NMB = (G == 2) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem;                    // [256][48]
    float* red = smem + kC * LD;            // [3][32][32]: G = 8: K-half partials of the gate / filter block, then tanh(filter); G = 4: tanh(filter) x 2
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile, g;
    if (!lat_map<G>(p.ntiles, tile, g)) return;
    const int b = tile / p.ntile32, tn = tile - b * p.ntile32, t0 = tn * 32;
    const bool has_left = tn > 0, has_right = tn + 1 < p.ntile32;
    const int T = p.T, dil = p.dil;

    // role of this wave
    int w4, mb0, kbeg;                      // packed stream, first row block, first chunk
    constexpr int NCH = (G == 16) ? 24 : (G == 8) ? 48 : 96; // chunks per wave
    if (G == 2) { w4 = 2 * g + (wv >> 1); mb0 = wv & 1; kbeg = 0; }
    else if (G == 4) { w4 = g; mb0 = wv; kbeg = 0; }
    else if (G == 8) { w4 = g >> 1; mb0 = (g & 1) + 2 * (wv & 1); kbeg = 48 * (wv >> 1); }
    else { w4 = g >> 2; mb0 = (g & 3) >> 1; kbeg = 24 * wv; }       // G = 16: (w4, mb0) = the standard gate block the 16 gate rows of block g lie in

    // the weight stream does not depend on x: its first chunks are requested before the tile is staged
    const ConvB<LD, BF> bof{ytile + 4 * h * LD + kHalo + j, dil, kbeg};
    constexpr int ASTR = (G == 16) ? 64 : 256;                      // float4 between consecutive chunks of the packed stream
    const float4* abase = (G == 16) ? p.w1q + ((size_t)g * 96 + kbeg) * 64 : p.w1p + (size_t)w4 * (96 * 256) + (size_t)kbeg * 256 + mb0 * 64;
    GemmPipe<NMB, 1, LD, ASTR, 6, ConvB<LD, BF>, 2> pipe(abase, lane, NCH, bof);
    pipe.start_a();

    // stage y = x + step_proj (zero at frames outside [0, T): the conv's zero padding applies to y, net.py:69-71)
    const int tstep = p.t_dev ? p.t_dev[b] : p.t_uniform;
    const float* __restrict__ dsl = p.ds + (size_t)tstep * p.ds_tstride;
    const float* __restrict__ xt = p.x_in + (size_t)tile * TILE;
    {
        float4 xv[8], hv[4];
#pragma unroll
        for (int it = 0; it < 8; ++it) xv[it] = reinterpret_cast<const float4*>(xt)[it * kThreads + tid];
        // halo: thread = channel row; my left halo = the left tile's columns 24..31, my right halo = the right tile's columns 0..7
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            hv[q] = has_left ? *reinterpret_cast<const float4*>(xt - TILE + tid * 32 + 24 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            hv[2 + q] = has_right ? *reinterpret_cast<const float4*>(xt + TILE + tid * 32 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 32 + (tid >> 3), c4 = tid & 7, t = t0 + 4 * c4;
            const float d = dsl[row];
            float4 v = xv[it];
            v.x = (t + 0 < T) ? v.x + d : 0.f;
            v.y = (t + 1 < T) ? v.y + d : 0.f;
            v.z = (t + 2 < T) ? v.z + d : 0.f;
            v.w = (t + 3 < T) ? v.w + d : 0.f;
            *reinterpret_cast<float4*>(ytile + row * LD + kHalo + 4 * c4) = v;
        }
        const float d = dsl[tid];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool right = q >= 2, have = right ? has_right : has_left;
            const int t = right ? t0 + 32 + 4 * (q - 2) : t0 - kHalo + 4 * q;
            float4 v = hv[q];
            v.x = (have && t + 0 < T) ? v.x + d : 0.f;
            v.y = (have && t + 1 < T) ? v.y + d : 0.f;
            v.z = (have && t + 2 < T) ? v.z + d : 0.f;
            v.w = (have && t + 3 < T) ? v.w + d : 0.f;
            *reinterpret_cast<float4*>(ytile + tid * LD + (right ? kHalo + 32 + 4 * (q - 2) : 4 * q)) = v;
        }
    }
    __syncthreads();

    f32x16 acc[NMB][1];
#pragma unroll
    for (int m = 0; m < NMB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;
    // hoisted conditioner projection (+ conv bias + cond bias) of this wave's row block(s), accumulator-fragment order: requested in
    // front of the contraction (behind it the load latency would sit between the last MFMA and the gate, once per kernel node)
    const float4* cpl = p.cp + ((size_t)tile * 4 + w4) * (4 * 4 * 64) + lane;
    float4 cv[4], cf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cv[q] = cf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.T > 0) {          // always true; a load in its own block cannot be sunk into the (conditional) blocks of its uses behind the MFMAs
        if (G == 16) {
            // gate row 16 g + i sits in the standard block (w4, mb0) at row 16 (g & 1) + i: register r of THIS lane there is r + 8 (g & 1) ->
            // quads q' = q + 2 (g & 1), q = 0, 1; the filter rows are the same quads of block mb0 + 2.  Wave wv finishes registers 2 wv, 2 wv + 1
            // (and their filter partners): quad wv >> 1, components 2 (wv & 1), + 1
            cv[0] = cpl[(mb0 * 4 + (wv >> 1) + 2 * (g & 1)) * 64];
            cf[0] = cpl[((mb0 + 2) * 4 + (wv >> 1) + 2 * (g & 1)) * 64];
        } else if (G == 8) {
            // wave wv finishes registers 4 wv .. 4 wv + 3 of the gate block (g & 1) and of its filter block: quad wv of both
            cv[0] = cpl[((g & 1) * 4 + wv) * 64];
            cf[0] = cpl[(((g & 1) + 2) * 4 + wv) * 64];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cv[q] = cpl[(mb0 * 4 + q) * 64];
                if (G == 2) cf[q] = cpl[((mb0 + 2) * 4 + q) * 64];
            }
        }
    }
    DSD_SB();
    pipe.start_b();
    pipe.run(acc, 0, NCH);

    float* gout = p.gbuf + (size_t)tile * TILE;
    if (G == 16) {
        // the four K partials meet in LDS; every wave adds them - in wave order ((p0 + p1) + p2) + p3, the order of lat_ksum4 - for TWO of the
        // eight gate registers and their filter partners and finishes those rows (sigmoid and tanh of all eight on wave 0 alone were
        // ~0.6 us of every layer)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wv * 1024 + frag_row(r, h) * 32 + j] = acc[0][0][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = 2 * wv + q;
            const float* pg = red + frag_row(r, h) * 32 + j;
            const float* pf = red + frag_row(r + 8, h) * 32 + j;
            const float ag = ((pg[0] + pg[1024]) + pg[2048]) + pg[3072];
            const float af = ((pf[0] + pf[1024]) + pf[2048]) + pf[3072];
            const float gv = sigmoid_f(ag + f4at(cv[0], r & 3)) * tanh_f(af + f4at(cf[0], r & 3));
            gout[(16 * g + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = gv;
        }
        return;
    }
    if (G == 2) {
        // a gate block and its filter block in the same wave: the gate never leaves registers (like k_layer)
        float4 (&cg)[4] = cv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float gv = sigmoid_f(acc[0][0][r] + f4at(cg[r >> 2], r & 3)) * tanh_f(acc[NMB - 1][0][r] + f4at(cf[r >> 2], r & 3));
            gout[(64 * w4 + 32 * mb0 + frag_row(r, h)) * 32 + j] = gv;
        }
        return;
    }
    if (G == 8) {
        // waves 0 / 1 hold the first K half of the gate / filter block, waves 2 / 3 the second: the four partial blocks meet in LDS and every
        // wave finishes FOUR registers of the pair - first half + second half (the order of the two-wave form this replaces, bit for bit),
        // sigmoid(gate) * tanh(filter) (net.py:73-74).  One barrier instead of two, the transcendental work on four waves instead of two.
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wv * 1024 + frag_row(r, h) * 32 + j] = acc[0][0][r];
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int row = frag_row(4 * wv + qq, h);
            const float* pr = red + row * 32 + j;
            const float ag = pr[0] + pr[2048], af = pr[1024] + pr[3072];
            const float gv = sigmoid_f(ag + f4at(cv[0], qq)) * tanh_f(af + f4at(cf[0], qq));
            gout[(64 * w4 + 32 * (g & 1) + row) * 32 + j] = gv;
        }
        return;
    }
    // G = 4: filter waves -> tanh(filter pre-activation) through LDS -> gate waves: sigmoid(gate pre-activation) * tanh(.)   (net.py:73-74)
    const bool is_filter = (wv >= 2);
    const bool is_gate = (wv < 2);
    float* fx = red + (wv & 1) * 1024;                                  // two filter blocks
    if (is_filter) {
#pragma unroll
        for (int r = 0; r < 16; ++r) fx[frag_row(r, h) * 32 + j] = tanh_f(acc[0][0][r] + f4at(cv[r >> 2], r & 3));
    }
    __syncthreads();
    if (is_gate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float gv = sigmoid_f(acc[0][0][r] + f4at(cv[r >> 2], r & 3)) * fx[frag_row(r, h) * 32 + j];
            gout[(64 * w4 + 32 * mb0 + frag_row(r, h)) * 32 + j] = gv;
        }
    }
}

template <int G>
__global__ __launch_bounds__(kThreads, 2) void k_lat_out(const LatParams p) {
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "row split");
    constexpr int TILE = kC * 32;
    constexpr int NMB = (G == 2) ? 2 : 1;
    constexpr int NCH = (G == 16) ? 8 : (G == 8) ? 16 : 32;     // chunks per wave: G = 8 / 16 split K = 256 over two / four waves per row block
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gtile = smem;                    // [256][32]
    float* red = smem + kC * 32;            // G = 8: [2][32][32] K-half partials of the residual / skip block
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile, g;
    if (!lat_map<G>(p.ntiles, tile, g)) return;

    int w4, mb0, kbeg = 0;
    if (G == 2) { w4 = 2 * g + (wv >> 1); mb0 = wv & 1; }
    else if (G == 4) { w4 = g; mb0 = wv; }
    else if (G == 8) { w4 = g >> 1; mb0 = (g & 1) + 2 * (wv & 1); kbeg = 16 * (wv >> 1); }
    else { w4 = g >> 2; mb0 = g & 3; kbeg = 8 * wv; }
    // the last layer's residual half is dead (net.py:126 reads the skips only)
    const bool do_res = !p.last && ((G == 2) || mb0 < 2);
    const bool do_skip = (G == 2) || mb0 >= 2;
    const bool active = (G == 2) || !(p.last && mb0 < 2);

    const float* gl = gtile + kbeg * (8 * 32) + 4 * h * 32 + j;
    GemmPipe<NMB, 1, 32, 256, 6, TileB, 2> pipe(p.w2p + (size_t)w4 * (32 * 256) + (size_t)kbeg * 256 + mb0 * 64, lane, NCH, TileB{gl, 8 * 32, NCH});
    // G = 16: a wave contracts over ONE quarter of the gate tile (8 chunks = 32 values per lane) and no other wave of the workgroup reads
    // that quarter - the B fragments come straight from global memory into registers, all requested at once behind the 8 A fragments;
    // there is no staging pass, no barrier in front of the contraction and nothing to wait for inside it
    float4 aq[(G == 16) ? 8 : 1][NMB];
    float bq[(G == 16) ? 8 : 1][4][1];
    if constexpr (G == 16) {
        if (active) {
            const float* __restrict__ gs = p.gbuf + (size_t)tile * TILE + kbeg * (8 * 32) + 4 * h * 32 + j;
#pragma unroll
            for (int c = 0; c < 8; ++c) pipe.lda(aq[c], c);
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) bq[c][s][0] = gs[(c * 8 + s) * 32];
        }
    } else {
        if (active) pipe.start_a();
        const float4* src = reinterpret_cast<const float4*>(p.gbuf + (size_t)tile * TILE);
        float4 v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) v[it] = src[it * kThreads + tid];
#pragma unroll
        for (int it = 0; it < 8; ++it) reinterpret_cast<float4*>(gtile)[it * kThreads + tid] = v[it];
        __syncthreads();
    }

    f32x16 acc[NMB][1];
#pragma unroll
    for (int m = 0; m < NMB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;
    // what the epilogue reads - x and the bias at this lane's residual rows, the running skip sum - is requested in front of the contraction
    // (the waves that only contribute a K half, G = 8, finish nothing)
    const bool fin = active && ((G == 16) ? (wv == 0) : (G != 8 || wv < 2));
    const float* __restrict__ xi = p.x_in + (size_t)tile * TILE;
    float4* sl = p.skip + (((size_t)tile * 4 + w4) * 2 + (mb0 & 1)) * (4 * 64) + lane;
    const bool keep = !p.first;             // layer 0 starts the sum (select, not multiply: the buffer may hold anything)
    float xv[16], bvv[16];
    float4 sk[4];
    if (fin && do_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 64 * w4 + 32 * (mb0 & 1) + frag_row(r, h);
            xv[r] = xi[row * 32 + j];
            bvv[r] = p.b2[row];
        }
    }
    if (fin && do_skip) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sk[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (keep) sk[q] = sl[q * 64];
        }
    }
    DSD_SB();
    if constexpr (G == 16) {
        if (active) {
#pragma unroll
            for (int c = 0; c < 8; ++c) mma_chunk<NMB, 1>(acc, aq[c], bq[c]);
        }
    } else if (active) {
        pipe.start_b();
        pipe.run(acc, 0, NCH);
    }
    if (G == 16) {
        if (!active) return;                 // uniform per workgroup: the whole block is dead
        lat_ksum4(acc[0][0], red, wv, j, h);
        if (wv > 0) return;
    }
    if (G == 8) {
        // sum the two K halves (first half + second half; the only difference from k_layer's single k-ordered chain)
        float* part = red + (wv & 1) * 1024;
        if (active && wv >= 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) part[frag_row(r, h) * 32 + j] = acc[0][0][r];
        }
        __syncthreads();
        if (wv >= 2) return;
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] = acc[0][0][r] + part[frag_row(r, h) * 32 + j];
        }
    }
    if (!active) return;

    if (do_res) {
        // x' = (x + (res + b)) / sqrt(2)   (net.py:78; same operation order as k_layer)
        float* __restrict__ xo = p.x_out + (size_t)tile * TILE;
        constexpr float kInvSqrt2 = 1.0f / 1.41421354f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 64 * w4 + 32 * (mb0 & 1) + frag_row(r, h);
            xo[row * 32 + j] = (xv[r] + (acc[0][0][r] + bvv[r])) * kInvSqrt2;
        }
    }
    if (do_skip) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = get4(acc[NMB - 1][0], q), s = sk[q];
            sl[q * 64] = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The head of an evaluation (net.py:126-129 + the sampler update + the next evaluation's input projection) for the G = 8 latency path
// ------------------------------------------------------------------------------------------------------------
// k_head gives a tile to one workgroup: with 16 tiles (one utterance of 512 frames) the head runs on 16 CUs for ~31 us per evaluation - as
// long as one and a half row-split layers.  Here its three contractions are split over the rows like the layers, three kernel nodes with the
// all-gathers at the boundaries:
//   k_lat_head_a   8 workgroups per tile: workgroup g = rows [32 g, +32) of relu(skip_projection(sum(skip) / sqrt(L))) -> hbuf
//   k_lat_head_b   3 workgroups per tile (of a G = 4 map): workgroup g = mel rows [32 g, +32) of the final projection + the sampler update
//                  (head_prefetch / head_apply of k_head) -> x, and the next x as a padded [96][32] tile -> pbuf
//   k_lat_head_c   8 workgroups per tile: rows [32 g, +32) of relu(input_projection(next x)) -> the x tile of the next evaluation
// In a and b the four waves of a workgroup split K = 256 four ways (8 chunks each, partial blocks summed through LDS in wave order: the
// summation order differs from k_head's single chain - like the G = 8 layer kernels, which this path accompanies); c is one short chain
// (K = 8 nk <= 96) on wave 0.
struct LatHeadParams {
    HeadParams hp;
    float* hbuf;            // [tiles][256][32] relu(skip projection)
    float* pbuf;            // [tiles][96][32] next x (mel rows padded with zeros)
    int ntiles;
};
constexpr int kLatHeadALdsBytes = (kC * 32 + 3 * 32 * 32) * (int)sizeof(float);
constexpr int kLatHeadBLdsBytes = (kC * 32 + 4 * 32 * 32) * (int)sizeof(float);
constexpr int kLatHeadCLdsBytes = kMPad * 32 * (int)sizeof(float);

// partial blocks of waves 1..3 -> wave 0: ((own + p1) + p2) + p3
__device__ __forceinline__ void lat_ksum4(f32x16& acc, float* red, int wv, int j, int h) {
    if (wv > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wv - 1) * 1024 + frag_row(r, h) * 32 + j] = acc[r];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] + red[q * 1024 + frag_row(r, h) * 32 + j];
    }
}

__global__ __launch_bounds__(kThreads, 2) void k_lat_head_a(const LatHeadParams q) {
    const HeadParams& p = q.hp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* stile = smem;                    // [256][32] scaled skip sum
    float* red = smem + kC * 32;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile, g;
    if (!lat_map<8>(q.ntiles, tile, g)) return;
    const int w4 = g >> 1, mb = g & 1;      // rows [32 g, +32) = packed stream w4, row block mb
    GemmPipe<1, 1, 32, 128, 6, TileB> pipe(p.wsp + (size_t)w4 * (32 * 128) + (size_t)(8 * wv) * 128 + mb * 64, lane, 8,
                                           TileB{stile + (8 * wv) * (8 * 32) + 4 * h * 32 + j, 8 * 32, 8});
    pipe.start_a();
    // x = sum(skip) / sqrt(L)   (net.py:126): wave wv stages the rows of packed stream wv, as k_head does
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const float4* sl = p.skip + (((size_t)tile * 4 + wv) * 2 + ms) * (4 * 64) + lane;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float4 s = sl[qq * 64], bs = p.bskp[((wv * 2 + ms) * 2 + h) * 4 + qq];
            const float v[4] = {s.x + bs.x, s.y + bs.y, s.z + bs.z, s.w + bs.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) stile[(64 * wv + 32 * ms + frag_row(4 * qq + e, h)) * 32 + j] = __fdiv_rn(v[e], p.sqrt_L);
        }
    }
    __syncthreads();
    f32x16 acc[1][1];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const float4 bz = (wv == 0) ? p.bsp[((w4 * 2 + mb) * 2 + h) * 4 + qq] : make_float4(0.f, 0.f, 0.f, 0.f);
        set4(acc[0][0], qq, bz);
    }
    pipe.start_b();
    pipe.run(acc, 0, 8);
    lat_ksum4(acc[0][0], red, wv, j, h);
    if (wv == 0) {
        float* out = q.hbuf + (size_t)tile * (kC * 32);
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(32 * g + frag_row(r, h)) * 32 + j] = fmaxf(acc[0][0][r], 0.f);
    }
}

template <int MODE>
__global__ __launch_bounds__(kThreads, 2) void k_lat_head_b(const LatHeadParams q) {
    const HeadParams& p = q.hp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* htile = smem;                    // [256][32] relu(skip projection)
    float* red = smem + kC * 32;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile, g;
    if (!lat_map<4>(q.ntiles, tile, g)) return;
    if (g >= 3) return;                     // 96 padded mel rows = three row blocks
    const int b = tile / p.ntile32, t0 = (tile - b * p.ntile32) * 32;
    GemmPipe<1, 1, 32, 192, 6, TileB> pipe(p.woutp + (size_t)(8 * wv) * 192 + g * 64, lane, 8,
                                           TileB{htile + (8 * wv) * (8 * 32) + 4 * h * 32 + j, 8 * 32, 8});
    pipe.start_a();
    {
        const float4* src = reinterpret_cast<const float4*>(q.hbuf + (size_t)tile * (kC * 32));
        float4 v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) v[it] = src[it * kThreads + tid];
#pragma unroll
        for (int it = 0; it < 8; ++it) reinterpret_cast<float4*>(htile)[it * kThreads + tid] = v[it];
    }
    __syncthreads();
    // what the sampler update reads besides eps, at the positions THIS wave finishes (registers 4 wv .. 4 wv + 3 of the block: the sampler
    // arithmetic - with the in-kernel Philox draw ~150 instructions per element - used to run on wave 0 alone, 16 elements per lane, and made
    // this kernel 12.5 us of a 258 us evaluation): requested in front of the contraction
    const int t = t0 + j;
    HeadPre pre[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int m = 32 * g + frag_row(4 * wv + qq, h);
        const bool ok = (m < p.M) && (t < p.T);
        const size_t idx = ((size_t)b * p.M + (ok ? m : 0)) * p.T + (ok ? t : 0);
        head_prefetch<MODE>(p, idx, pre[qq]);
    }
    DSD_SB();
    f32x16 acc[1][1];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const float4 bz = (wv == 0) ? p.boutp[(g * 2 + h) * 4 + qq] : make_float4(0.f, 0.f, 0.f, 0.f);
        set4(acc[0][0], qq, bz);
    }
    pipe.start_b();
    pipe.run(acc, 0, 8);
    // the four K partials meet in LDS; every wave adds them for ITS four registers in wave order ((p0 + p1) + p2) + p3 - the order of
    // lat_ksum4, bit for bit - and finishes those rows
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv * 1024 + frag_row(r, h) * 32 + j] = acc[0][0][r];
    __syncthreads();
    float* pt = q.pbuf + (size_t)tile * (kMPad * 32);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int row = frag_row(4 * wv + qq, h), m = 32 * g + row;
        const float* pr = red + row * 32 + j;
        const float eps = ((pr[0] + pr[1024]) + pr[2048]) + pr[3072];
        const bool ok = (m < p.M) && (t < p.T);
        const size_t idx = ((size_t)b * p.M + m) * p.T + t;
        float xn = 0.f;
        if (ok) xn = head_apply<MODE>(p, eps, idx, pre[qq]);
        pt[m * 32 + j] = ok ? xn : 0.f;
    }
}

__global__ __launch_bounds__(kThreads, 2) void k_lat_head_c(const LatHeadParams q) {
    const HeadParams& p = q.hp;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [96][32] next x
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile, g;
    if (!lat_map<8>(q.ntiles, tile, g)) return;
    const int w4 = g >> 1, mb = g & 1;
    GemmPipe<1, 1, 32, 128, 6, TileB> pipe(p.winp + (size_t)w4 * p.nk_in * 128 + mb * 64, lane, p.nk_in, TileB{smem + 4 * h * 32 + j, 8 * 32, p.nk_in});
    if (wv == 0) pipe.start_a();
    {
        const float4* src = reinterpret_cast<const float4*>(q.pbuf + (size_t)tile * (kMPad * 32));
#pragma unroll
        for (int it = 0; it < 3; ++it) reinterpret_cast<float4*>(smem)[it * kThreads + tid] = src[it * kThreads + tid];
    }
    __syncthreads();
    if (wv != 0) return;
    f32x16 acc[1][1];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) set4(acc[0][0], qq, p.binp[((w4 * 2 + mb) * 2 + h) * 4 + qq]);
    pipe.start_b();
    pipe.run(acc, 0, p.nk_in);
    float* out = p.x_next + (size_t)tile * (kC * 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(32 * g + frag_row(r, h)) * 32 + j] = fmaxf(acc[0][0][r], 0.f);
}

}  // namespace dsd
