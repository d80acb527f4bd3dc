// pwg_kernels.hpp - gfx950 kernels of the ParallelWaveGAN generator (the reference's other vocoder: vocoders/pwg.py,
// modules/parallel_wavegan/models/parallel_wavegan.py:21-177, layers/residual_block.py:39-129, layers/upsample.py:63-183; the default of
// configs/tts/base.yaml:88, overridden to HiFi-GAN by the DiffSpeech / DiffSinger YAMLs).
//
//   k_pwg_layer      one ResidualBlock (residual_block.py:96-129) per launch, a 32-sample tile of all channels per workgroup:
//                      a  = W_conv * [x(t - d); x(t); x(t + d)] + b_conv + W_aux * c(t)        128 gate rows, K = 3 * 64 + aux  (one MFMA contraction:
//                                                                                               the taps are ROWS of the staged B tile)
//                      z  = tanh(a[0:64]) * sigmoid(a[64:128])                                  through LDS (the halves sit in different waves)
//                      x' = (W_out z + b_out + x) * sqrt(0.5)        skip_sum (+)= W_skip z + b_skip      one more contraction, K = 64
//                    Dilations run to 512 samples (2 ** (layer % 10)): the three taps are three tiles of x fetched at their own offsets, there is
//                    no halo to stage - the reason the HiFi-GAN convolution kernel (taps within +-48 samples) cannot run these layers.
//   k_pwg_upsample   one stage of UpsampleNetwork (upsample.py:96-117): nearest-neighbour stretch by `scale` + the Conv2d(1, 1, (1, 2 scale + 1))
//                    smoothing filter shared by all channels, zero padded.
//   k_pwg_first      first_conv (parallel_wavegan.py:78): Conv1d1x1(1, 64) on the noise signal = an outer product.
#pragma once
#include "voc_kernels.hpp"

namespace dsd {

constexpr int kPwgRes = 64, kPwgGate = 128, kPwgMaxAux = 128;
constexpr int kPwgLD = 32;

struct PwgLayerParams {
    const float* x;         // [B][64][LS]
    const float* c;         // [B][naux][LS] upsampled conditioning, or nullptr (naux = 0)
    const float4* w1p;      // dsv_pack_weight of [128][192 + naux][1]: columns tap * 64 + ci (tap 0 = t - d), then the aux channels
    const float* b1;        // [128] bias of the dilated conv (conv1x1_aux has none), or nullptr
    const float4* w2p;      // dsv_pack_weight of [128][64][1]: rows 0..63 conv1x1_out, 64..127 conv1x1_skip
    const float* b2;        // [128] or nullptr
    float* x_out;           // [B][64][LS]
    float* skip;            // [B][64][LS] running sum of the skip outputs
    int L, LS, dil, naux, first;
};
// LDS: one [192 + 128][32] tile (40 KiB: four workgroups per CU - the kernel is a chain of short dependent phases, co-residency is what hides
// them).  Behind the first contraction its rows are reused: the gate pre-activations [128][32] take rows 128..255 (tap + 1 and the conditioning),
// the gated activations [64][32] rows 0..63 (tap - 1); rows 64..127 (the centre tap = x itself) stay for the residual connection.
constexpr int kPwgLayerLdsBytes = (3 * kPwgRes + kPwgMaxAux) * kPwgLD * (int)sizeof(float);

__global__ __launch_bounds__(kThreads, 4) void k_pwg_layer(const PwgLayerParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bt = smem;                                               // [192 + naux][32]: the B operand of the first contraction
    float* at = smem + 2 * kPwgRes * kPwgLD;                        // [128][32] gate pre-activations (over rows 128..255, after the contraction)
    float* zt = smem;                                               // [64][32] gated activations (over rows 0..63)
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * 32, b = blockIdx.y;
    const int K1 = 3 * kPwgRes + p.naux, nch1 = K1 / 8;
    const float* xb = p.x + (size_t)b * kPwgRes * p.LS;
    // the weight stream of this wave's row block does not depend on the tile: requested first
    GemmPipe<1, 1, kPwgLD, 64, 6, TileB> pipe1(p.w1p + (size_t)w * nch1 * 64, lane, nch1, TileB{bt + 4 * h * kPwgLD + j, 8 * kPwgLD, nch1});
    pipe1.start_a();
    // stage: rows tap * 64 + ci = x[ci][t0 + col + (tap - 1) * dil] (zero outside [0, L)), then the conditioning rows.  All the loads of the tile
    // are requested before the first LDS write (24 + naux / 8 per thread): one exposure of the memory latency instead of one per row
    {
        const int col = tid & 31, r0 = tid >> 5;                    // thread: column col of rows r0, r0 + 8, ...
        float xv[24], cv[kPwgMaxAux / 8];
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const int r = r0 + 8 * i, tap = r >> 6, ci = r & 63;
            const int t = t0 + col + (tap - 1) * p.dil;
            const bool ok = (t >= 0) && (t < p.L);
            const float v = xb[(size_t)ci * p.LS + (ok ? t : t0)];
            xv[i] = ok ? v : 0.f;
        }
        const float* cb = p.naux ? p.c + (size_t)b * p.naux * p.LS : nullptr;
        const int tc = t0 + col;
#pragma unroll
        for (int i = 0; i < kPwgMaxAux / 8; ++i) {
            const int r = r0 + 8 * i;
            const bool ok = (r < p.naux) && (tc < p.L);
            cv[i] = 0.f;
            if (ok) cv[i] = cb[(size_t)r * p.LS + tc];
        }
#pragma unroll
        for (int i = 0; i < 24; ++i) bt[(r0 + 8 * i) * kPwgLD + col] = xv[i];
#pragma unroll
        for (int i = 0; i < kPwgMaxAux / 8; ++i)
            if (r0 + 8 * i < p.naux) bt[(3 * kPwgRes + r0 + 8 * i) * kPwgLD + col] = cv[i];
    }
    __syncthreads();
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    pipe1.start_b();
    pipe1.run_blocks(acc, nch1);
    // second contraction's weights: requested before the gate
    GemmPipe<1, 1, kPwgLD, 64, 6, TileB> pipe2(p.w2p + (size_t)w * 8 * 64, lane, 8, TileB{zt + 4 * h * kPwgLD + j, 8 * kPwgLD, 8});
    pipe2.start_a();
    float b1v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b1v[r] = p.b1 ? p.b1[32 * w + frag_row(r, h)] : 0.f;
    __syncthreads();                                                // every wave is done reading the B tile: its rows may be overwritten
#pragma unroll
    for (int r = 0; r < 16; ++r) at[(32 * w + frag_row(r, h)) * kPwgLD + j] = acc[0][0][r] + b1v[r];
    __syncthreads();
    {
        // tanh(a) * sigmoid(g) (residual_block.py:120) on the hardware exponential: tanh(a) = 1 - 2 / (e^(2a) + 1) (exact limits at +-inf),
        // absolute error ~1e-7 per layer - the waveform of the 30-layer generator stays within 1e-6 of the reference's (tests/test_gpu_pwg.py)
        const int col = tid & 31, r0 = tid >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = r0 + 8 * i;
            const float xa = at[ch * kPwgLD + col], xg = at[(kPwgRes + ch) * kPwgLD + col];
            const float th = 1.f - 2.f / (__expf(2.f * xa) + 1.f);
            zt[ch * kPwgLD + col] = th * (1.f / (1.f + __expf(-xg)));
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    pipe2.start_b();
    pipe2.run_blocks(acc, 8);
    const int t = t0 + j;
    const bool tv = t < p.L;
    if (w < 2) {
        // residual connection (residual_block.py:126): x' = (conv1x1_out(z) + x) * sqrt(0.5)
        float* xo = p.x_out + (size_t)b * kPwgRes * p.LS;
        const float s = sqrtf(0.5f);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * w + frag_row(r, h);
            const float res = bt[(kPwgRes + ch) * kPwgLD + j];      // x[ch][t0 + j] is the centre tap's row
            const float v = ((acc[0][0][r] + (p.b2 ? p.b2[ch] : 0.f)) + res) * s;
            xo[(size_t)ch * p.LS + t] = tv ? v : 0.f;
        }
    } else {
        // skip connection (:123, parallel_wavegan.py:166 `skips += h`)
        float* sk = p.skip + (size_t)b * kPwgRes * p.LS;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * (w - 2) + frag_row(r, h);
            const float hv = acc[0][0][r] + (p.b2 ? p.b2[kPwgRes + ch] : 0.f);
            const size_t idx = (size_t)ch * p.LS + t;
            const float prev = p.first ? 0.f : sk[idx];
            sk[idx] = tv ? prev + hv : 0.f;
        }
    }
}

// out[r][t] = sum_j w[j] * in[r][(t + j - scale) / scale]   for 0 <= t + j - scale < L_in * scale   (rows r = b * C + c; the filter is shared)
__global__ __launch_bounds__(256) void k_pwg_upsample(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out, int L_in,
                                                      int LS_in, int scale, int LS_out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const size_t r = blockIdx.y;
    if (t >= LS_out) return;
    const int L_out = L_in * scale;
    float s = 0.f;
    if (t < L_out) {
        const float* row = in + r * LS_in;
        for (int jj = 0; jj <= 2 * scale; ++jj) {
            const int u = t + jj - scale;
            if (u >= 0 && u < L_out) s = fmaf(w[jj], row[u / scale], s);
        }
    }
    out[r * LS_out + t] = s;
}

// x[b][c][t] = w[c] * z[b][t] + bias[c]
__global__ __launch_bounds__(256) void k_pwg_first(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ out, int C, int L, int LS) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= LS) return;
    out[((size_t)b * C + c) * LS + t] = (t < L) ? fmaf(w[c], z[(size_t)b * LS + t], bias ? bias[c] : 0.f) : 0.f;
}

}  // namespace dsd
