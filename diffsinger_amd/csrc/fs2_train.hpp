// fs2_train.hpp - gfx950 kernels of the FastSpeech2 BACKWARD pass (SURVEY.md section 8 rows f1 x f3: the Opencpop e2e configuration trains
// FastSpeech2MIDI jointly with the denoiser - usr/diffsinger_task.py:60-64, :273-300; usr/configs/midi/e2e/opencpop/ds1000.yaml:18 `fs2_ckpt: ''`).
//
// The contractions of the backward pass are the convolution operators that already exist (data gradient = dsf_conv1d_dilated with the flipped,
// transposed weight; weight / bias gradient = dsf_conv1d_wgrad, csrc/train_kernels.hpp).  What autograd runs for the other two building blocks of
// the model - torch.nn.LayerNorm (modules/commons/common_layers.py:18-27, tts_modules.py:39-56) and F.multi_head_attention_forward
// (common_layers.py:243-263) - is here:
//   k_fs_ln_bwd          dx of the LayerNorm over the channel axis (+ the padding mask and the ReLU in front that k_fs_ln fuses), per-tile partial
//                        sums of dgamma / dbeta; k_fs_colsum adds the partials in a fixed order (deterministic)
//   k_fs_bmm             a small strided batched matrix product on the vector ALUs (LDS-tiled 32 x 32 x 32).  The encoder's attention runs at the
//                        PHONE rate (tens to a few hundred tokens; the mel-rate decoder is skipped in the diffusion training step,
//                        shallow_diffusion_tts.py:236 skip_decoder = True): five products per layer - S = q^T k, dP = dO^T v, dv = dO P, dq = k dS^T,
//                        dk = q dS - with S / P / dS materialised ([B heads][T][T]); not a roofline kernel, a correct and deterministic one
//   k_fs_softmax_rows    P = softmax(S + key_padding_mask) row by row, k_fs_softmax_bwd_rows  dS = P o (dP - rowsum(P o dP))
#pragma once
#include "fs2_kernels.hpp"

namespace dsd {

struct FsLnBwdParams {
    const float* x;         // [B][256][TS] the forward input
    const float* dy;        // [B][256][TS]
    const float* gamma;     // [256]
    const float* keep;      // [B][T] or nullptr
    float* dx;              // [B][256][TS]
    float* part;            // [B * TS / 32][2][256]: per-tile sums of dgamma, dbeta
    int T, TS;
    float eps;
    int relu_in;
};

// thread (tc, part): frame blockIdx.x * 32 + tc, channels [32 part, 32 part + 32) - the decomposition of k_fs_ln
__global__ __launch_bounds__(kThreads) void k_fs_ln_bwd(const FsLnBwdParams p) {
    __shared__ float red[2][8][32];
    const int tid = threadIdx.x, tc = tid & 31, part = tid >> 5;
    const int t = blockIdx.x * 32 + tc, b = blockIdx.y;
    const size_t base = ((size_t)b * kC + part * 32) * p.TS + t;
    const bool tv = t < p.T;
    float kp = tv ? 1.f : 0.f;
    if (p.keep && tv) kp = p.keep[(size_t)b * p.T + t];
    float v[32], g[32];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        float x = p.x[base + (size_t)i * p.TS];
        if (p.relu_in) x = fmaxf(x, 0.f);
        v[i] = x;
        s += x;
        g[i] = p.dy[base + (size_t)i * p.TS] * kp;              // gradient wrt the normalised, scaled output before the mask
    }
    red[0][part][tc] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) mean += red[0][q][tc];
    mean *= (1.f / kC);
    __syncthreads();
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { const float e = v[i] - mean; d += e * e; }
    red[0][part][tc] = d;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) var += red[0][q][tc];
    var *= (1.f / kC);
    const float rstd = 1.f / sqrtf(var + p.eps);
    __syncthreads();
    // xhat, gg = g * gamma; s1 = sum gg, s2 = sum gg * xhat over the 256 channels
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        v[i] = (v[i] - mean) * rstd;
        const float gg = g[i] * p.gamma[part * 32 + i];
        s1 += gg;
        s2 += gg * v[i];
    }
    red[0][part][tc] = s1;
    red[1][part][tc] = s2;
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { m1 += red[0][q][tc]; m2 += red[1][q][tc]; }
    m1 *= (1.f / kC);
    m2 *= (1.f / kC);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float gg = g[i] * p.gamma[part * 32 + i];
        float dx = rstd * (gg - m1 - v[i] * m2);
        if (p.relu_in && !(p.x[base + (size_t)i * p.TS] > 0.f)) dx = 0.f;
        p.dx[base + (size_t)i * p.TS] = tv ? dx : 0.f;
    }
    // dgamma_c += g * xhat, dbeta_c += g over the 32 frames of the tile: lanes tc = 0..31 of one half-wave hold one channel group
    float* pt = p.part + ((size_t)b * (p.TS / 32) + blockIdx.x) * 512;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        float a = g[i] * v[i], c = g[i];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 32); c += __shfl_xor(c, off, 32); }
        if (tc == 0) { pt[part * 32 + i] = a; pt[256 + part * 32 + i] = c; }
    }
}

// out[c] = sum_r part[r * stride + c] in row order (fixed summation order), c < ncol
__global__ __launch_bounds__(256) void k_fs_colsum(const float* __restrict__ part, float* __restrict__ out, int nrows, int ncol, int stride) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncol) return;
    float s = 0.f;
    for (int r = 0; r < nrows; ++r) s += part[(size_t)r * stride + c];
    out[c] = s;
}

// y[r][c] += bias[c]
__global__ __launch_bounds__(256) void k_fs_add_row_bias(float* __restrict__ y, const float* __restrict__ bias, int ncol) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < ncol) y[(size_t)blockIdx.y * ncol + c] += bias[c];
}

// ------------------------------------------------------------------------------------------------------------
// strided batched matrix product C[m][n] = alpha * sum_k A[m][k] B[k][n]
// ------------------------------------------------------------------------------------------------------------
struct FsBmmParams {
    const float* A; const float* B; float* C;
    int M, N, K;
    long long am, ak, bk, bn, cm, cn;           // element strides
    int inner;                                  // batch index z = outer * inner + in
    long long a_o, a_i, b_o, b_i, c_o, c_i;     // batch strides (outer, inner)
    float alpha;
};

__global__ __launch_bounds__(256) void k_fs_bmm(const FsBmmParams p) {
    __shared__ float As[32][33];                // [m][k]
    __shared__ float Bs[32][33];                // [k][n]
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int zo = blockIdx.z / p.inner, zi = blockIdx.z - zo * p.inner;
    const float* A = p.A + zo * p.a_o + zi * p.a_i;
    const float* B = p.B + zo * p.b_o + zi * p.b_i;
    float* C = p.C + zo * p.c_o + zi * p.c_i;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool a_m_fast = (p.am == 1), b_k_fast = (p.bk == 1);     // which index runs along the lanes when a tile is fetched
    for (int k0 = 0; k0 < p.K; k0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, hi = e >> 5, lo = e & 31;
            const int mr = a_m_fast ? lo : hi, kc = a_m_fast ? hi : lo;
            const int m = m0 + mr, k = k0 + kc;
            As[mr][kc] = (m < p.M && k < p.K) ? A[m * p.am + k * p.ak] : 0.f;
            const int kr = b_k_fast ? lo : hi, nc = b_k_fast ? hi : lo;
            const int kk = k0 + kr, n = n0 + nc;
            Bs[kr][nc] = (kk < p.K && n < p.N) ? B[kk * p.bk + n * p.bn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float bv = Bs[kk][tx];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(As[ty * 4 + i][kk], bv, acc[i]);
        }
        __syncthreads();
    }
    const int n = n0 + tx;
    if (n < p.N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ty * 4 + i;
            if (m < p.M) C[m * p.cm + n * p.cn] = p.alpha * acc[i];
        }
    }
}

__device__ __forceinline__ float fs_block_reduce(float v, float* red, bool is_max) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// S [BH][T][T] -> P in place: row (blockIdx.y, blockIdx.x); keys with key_pad[b][tk] != 0 get probability 0 (masked_fill(-inf))
__global__ __launch_bounds__(256) void k_fs_softmax_rows(float* S, const unsigned char* key_pad, int T, int heads) {
    __shared__ float red[4];
    const int bh = blockIdx.y, b = bh / heads;
    float* row = S + ((size_t)bh * T + blockIdx.x) * T;
    const unsigned char* kp = key_pad ? key_pad + (size_t)b * T : nullptr;
    float mx = -INFINITY;
    for (int k = threadIdx.x; k < T; k += 256)
        if (!(kp && kp[k])) mx = fmaxf(mx, row[k]);
    mx = fs_block_reduce(mx, red, true);
    float sum = 0.f;
    for (int k = threadIdx.x; k < T; k += 256) {
        const float e = (kp && kp[k]) ? 0.f : expf(row[k] - mx);
        row[k] = e;
        sum += e;
    }
    sum = fs_block_reduce(sum, red, false);
    const float inv = 1.f / sum;
    for (int k = threadIdx.x; k < T; k += 256) row[k] = row[k] * inv;
}

// dP [BH][T][T] -> dS in place: dS = P o (dP - sum_k P dP)
__global__ __launch_bounds__(256) void k_fs_softmax_bwd_rows(const float* P, float* dP, int T) {
    __shared__ float red[4];
    const size_t off = ((size_t)blockIdx.y * T + blockIdx.x) * T;
    const float* pr = P + off;
    float* dr = dP + off;
    float s = 0.f;
    for (int k = threadIdx.x; k < T; k += 256) s += pr[k] * dr[k];
    s = fs_block_reduce(s, red, false);
    for (int k = threadIdx.x; k < T; k += 256) dr[k] = pr[k] * (dr[k] - s);
}

}  // namespace dsd
