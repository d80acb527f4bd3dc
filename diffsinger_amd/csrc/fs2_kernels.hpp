// fs2_kernels.hpp - gfx950 kernels for the FastSpeech2 / FastSpeech2MIDI conditioner + aux decoder (SURVEY.md section 8 row
// f1: the caller of the diffusion hot path; it produces `cond` and the shallow-diffusion start).
//
// What is computed, and where the reference computes it (paths relative to the reference root):
//   k_fs_conv   every Conv1d / Linear of the model as ONE fp32-MFMA contraction over (input channel, tap):
//               MultiheadAttention in/out projections (modules/commons/common_layers.py:243-263 -> F.multi_head_attention_forward),
//               TransformerFFNLayer ffn_1 (k = 9, 'SAME') * k^-0.5 -> gelu and ffn_2 (common_layers.py:486-522), the residual
//               add + padding mask of EncSALayer (:565-588), the predictor convolutions + ReLU (modules/fastspeech/
//               tts_modules.py:84-97, :198-209), mel_out (modules/fastspeech/fs2.py:233-237)
//   k_fs_ln     LayerNorm over the channel axis (EncSALayer layer_norm1/2, FFTBlocks.layer_norm: eps 1e-5; predictor LayerNorm
//               (dim=1): eps 1e-12, tts_modules.py:39-56), optionally times the padding mask
//   k_fs_attn   softmax(q k^T + key_padding_mask) v per head (F.multi_head_attention_forward), flash-style: one workgroup per
//               (utterance, head, 32-query tile), its 4 waves split the keys; online softmax, S and P V on fp32 MFMA
//   k_fs_from_cm  internal channel-major [B][C][TS] -> the reference's [B,T,C]
//
// Activations live channel-major [B][C][TS] (frame axis contiguous, TS = T rounded up to 32, ZERO in [T,TS)): exactly the
// layout of the denoiser kernels, so every contraction reuses their operand pipeline (A = weights in MFMA-fragment order
// streamed from L2, B = an LDS tile [channel][frame] whose taps are column offsets).
#pragma once
#include "dsd_kernels.hpp"

namespace dsd {

constexpr int kFsHalo = 8;                 // conv taps reach +-8 frames at most (kernel <= 17)
constexpr int kFsLD = 32 + 2 * kFsHalo;    // LDS row stride of the staged input slab
constexpr int kFsSlab = 256;               // input channels staged per pass

enum FsAct { FS_ACT_NONE = 0, FS_ACT_RELU = 1, FS_ACT_GELU = 2, FS_ACT_MISH = 3 };

struct FsConvParams {
    const float* in;        // [B][Ci][TS]
    const float4* wp;       // packed [mtile][w4][chunk = ci8 * KT + tap][NMB][lane64] float4 (k_pack_a, ntap = KT)
    const float* bias;      // [Co] or nullptr
    float* out;             // [B][Co][TS]
    const float* res;       // residual [B][Co][TS] or nullptr (added after the activation)
    const float* keep;      // [B][T], 1 = frame valid, 0 = padding; nullptr = no mask
    int Ci, Co, KT, pad, dil, T, TS;     // pad = dil * (KT - 1) / 2 <= kFsHalo
    float scale;            // multiplied in before the activation (TransformerFFNLayer: kernel_size ** -0.5)
    int act;
};

// B-operand functor of GemmPipe: chunk kc = ci8 * KT + tap of the staged slab as a RUNNING pointer (see VocTapB, voc_kernels.hpp): GemmPipe asks
// for the chunks strictly in order, so the map is tap + 1 / wrap to the next 8-channel group / freeze on the last chunk - a handful of selects
// instead of a branch chain + constant-divisor division in front of every 4 NMB MFMAs; the K loop walks whole groups of six chunks as one
// basic block (GemmPipe::run_blocks).  -5 % on the FastSpeech2 forward (profiles/r05_fm_conv_inc_ab.jsonl).
struct FsTapB {
    const float* cur;       // the chunk handed out next (base = slab + 4 h LD + halo + j - pad)
    int KT, dil, left, tap;
    __device__ __forceinline__ FsTapB(const float* base, int KT_, int dil_, int n) : cur(base), KT(KT_), dil(dil_), left(n - 1), tap(0) {}
    __device__ __forceinline__ const float* operator()(int, int) {
        const float* r = cur;
        const bool adv = left > 0, wrap = (tap + 1 == KT);
        const int step = wrap ? 8 * kFsLD - (KT - 1) * dil : dil;
        cur += adv ? step : 0;
        tap = adv ? (wrap ? 0 : tap + 1) : tap;
        left -= adv ? 1 : 0;
        return r;
    }
};

// out rows [128 NMB mtile, 128 NMB (mtile + 1)) x 32 frames per workgroup: 4 waves x NMB row blocks of 32.  NMB = 4 (wide layers,
// Co >= 512): 16 MFMAs per 4 weight loads + 4 LDS reads like the denoiser's layer kernel, and the input slab is staged once per
// 512 output rows; NMB = 2 keeps two workgroups per CU for the narrow ones.
template <int NMB>
__global__ __launch_bounds__(kThreads, NMB == 4 ? 1 : 2) void k_fs_conv(const FsConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [kFsSlab][kFsLD]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * 32, b = blockIdx.y, mt = blockIdx.z;
    const int nchunk_total = (p.Ci / 8) * p.KT;
    f32x16 acc[NMB][1];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
    const float* inb = p.in + (size_t)b * p.Ci * p.TS;
    // Staging: channels [c0, c0 + nc) x frames [t0 - 8, t0 + 40) of a slab, 12 float4 per row, zero outside [0, TS).  All the loads of a slab
    // are requested at once (12 per thread for 256 channels) and the NEXT slab's loads are in flight during this slab's contraction: a
    // load -> write -> load chain exposes the memory latency once per float4 (12 times per slab - more than the 7 us of MFMA work a K = 256
    // convolution has per workgroup; A/B inside one GPU call: FastSpeech2 forward 3.94 -> 3.82 ms, profiles/r02x_fs_conv_staging_ab.jsonl)
    constexpr int NIT = kFsSlab * (kFsLD / 4) / kThreads;
    float4 sv[NIT];
    auto request = [&](int c0, int nc) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * kThreads + tid, row = idx / (kFsLD / 4), g = idx - row * (kFsLD / 4);
            const int t = t0 - kFsHalo + 4 * g;
            const bool ok = (row < nc) && (t >= 0) && (t < p.TS);
            const float4 v = *reinterpret_cast<const float4*>(inb + (size_t)(c0 + (ok ? row : 0)) * p.TS + (ok ? t : t0));
            sv[it] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        DSD_SB();
    };
    request(0, min(kFsSlab, p.Ci));
    for (int c0 = 0; c0 < p.Ci; c0 += kFsSlab) {
        const int nc = min(kFsSlab, p.Ci - c0);
        // the weight stream does not depend on the slab: its first chunks are requested BEFORE the slab is written, so that their first-touch
        // latency (every workgroup of a launch walks the stream in lock-step: each chunk is new to the L2) overlaps the staging
        const int nch = (nc / 8) * p.KT;
        const float4* ap = p.wp + (((size_t)mt * 4 + w) * nchunk_total + (size_t)(c0 / 8) * p.KT) * (NMB * 64);
        FsTapB bof(smem + 4 * h * kFsLD + kFsHalo + j - p.pad, p.KT, p.dil, nch);
        GemmPipe<NMB, 1, kFsLD, NMB * 64, 6, FsTapB> pipe(ap, lane, nch, bof);
        pipe.start_a();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * kThreads + tid, row = idx / (kFsLD / 4), g = idx - row * (kFsLD / 4);
            if (row < nc) *reinterpret_cast<float4*>(smem + row * kFsLD + 4 * g) = sv[it];
        }
        __syncthreads();
        if (c0 + kFsSlab < p.Ci) request(c0 + kFsSlab, min(kFsSlab, p.Ci - c0 - kFsSlab));
        pipe.start_b();
        pipe.run_blocks(acc, nch);
        __syncthreads();
    }
    const int t = t0 + j;
    const bool tv = t < p.T;
    float kp = 1.f;
    if (p.keep && tv) kp = p.keep[(size_t)b * p.T + t];
    // epilogue: all the bias / residual reads are issued first (clamped row index, no branches between them), then the
    // arithmetic, then the stores - two workgroups per CU are not enough to hide 32 dependent load -> store round trips
    float bv[NMB][16], rv[NMB][16];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (mt * 4 + w) * (32 * NMB) + 32 * mb + frag_row(r, h);
            const int rc = (row < p.Co) ? row : 0;
            bv[mb][r] = p.bias ? p.bias[rc] : 0.f;
            rv[mb][r] = p.res ? p.res[((size_t)b * p.Co + rc) * p.TS + t] : 0.f;
        }
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (mt * 4 + w) * (32 * NMB) + 32 * mb + frag_row(r, h);
            float v = (acc[mb][0][r] + bv[mb][r]) * p.scale;
            if (p.act == FS_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == FS_ACT_GELU) v = v * 0.5f * (1.f + erff(v * 0.70710678118654752440f));
            else if (p.act == FS_ACT_MISH) v = v * tanhf((v > 20.f) ? v : log1pf(expf(v)));        // x * tanh(softplus(x)), usr/diff/diffusion.py:68-70
            v = (v + rv[mb][r]) * kp;
            if (row < p.Co) p.out[((size_t)b * p.Co + row) * p.TS + t] = tv ? v : 0.f;
        }
}
constexpr int kFsConvLdsBytes = kFsSlab * kFsLD * (int)sizeof(float);

// The same convolution for SMALL GRIDS (the phone-rate encoder of a batch, everything of a single utterance): k_fs_conv gives a workgroup 256
// output rows x 32 frames and lets every wave walk the whole contraction - with 32 to 128 workgroups on 256 CUs the launch lasts as long as
// ONE such chain (K = 2304: 61 us of dependent MFMAs).  Here a workgroup owns 64 rows x 32 frames (blockIdx.z = the 64-row group, the packed
// weights are read where they lie) and its four waves split every slab's chunk range four ways; the partial blocks meet in LDS, are added in
// wave order (fixed: deterministic) and each wave finishes 8 rows of both 32-row blocks.  4 x the workgroups, a quarter of the chain each.
// Needs slabs whose channel count is a multiple of 32 (Ci % 32 == 0).
__global__ __launch_bounds__(kThreads, 2) void k_fs_conv_ks(const FsConvParams p) {
    constexpr int NMB = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [kFsSlab][kFsLD]; behind the contraction: [4 waves][2][16][64] partials
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * 32, b = blockIdx.y, rg = blockIdx.z;  // rg = 64-row group (mtile * 4 + w4 of k_fs_conv)
    const int nchunk_total = (p.Ci / 8) * p.KT;
    f32x16 acc[NMB][1];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = 0.f;
    const float* inb = p.in + (size_t)b * p.Ci * p.TS;
    constexpr int NIT = kFsSlab * (kFsLD / 4) / kThreads;
    float4 sv[NIT];
    auto request = [&](int c0, int nc) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * kThreads + tid, row = idx / (kFsLD / 4), g = idx - row * (kFsLD / 4);
            const int t = t0 - kFsHalo + 4 * g;
            const bool ok = (row < nc) && (t >= 0) && (t < p.TS);
            const float4 v = *reinterpret_cast<const float4*>(inb + (size_t)(c0 + (ok ? row : 0)) * p.TS + (ok ? t : t0));
            sv[it] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        DSD_SB();
    };
    request(0, min(kFsSlab, p.Ci));
    for (int c0 = 0; c0 < p.Ci; c0 += kFsSlab) {
        const int nc = min(kFsSlab, p.Ci - c0);
        const int g4 = nc / 32;                                        // 8-channel groups per wave
        const int nq = g4 * p.KT;                                      // chunks per wave
        const float4* ap = p.wp + ((size_t)rg * nchunk_total + (size_t)(c0 / 8) * p.KT + (size_t)w * nq) * (NMB * 64);
        FsTapB bof(smem + (w * g4 * 8 + 4 * h) * kFsLD + kFsHalo + j - p.pad, p.KT, p.dil, nq);
        GemmPipe<NMB, 1, kFsLD, NMB * 64, 6, FsTapB> pipe(ap, lane, nq, bof);
        pipe.start_a();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * kThreads + tid, row = idx / (kFsLD / 4), g = idx - row * (kFsLD / 4);
            if (row < nc) *reinterpret_cast<float4*>(smem + row * kFsLD + 4 * g) = sv[it];
        }
        __syncthreads();
        if (c0 + kFsSlab < p.Ci) request(c0 + kFsSlab, min(kFsSlab, p.Ci - c0 - kFsSlab));
        pipe.start_b();
        pipe.run_blocks(acc, nq);
        __syncthreads();
    }
    // partial blocks -> LDS [wave][mb][r][lane] (conflict-free: lanes contiguous), summed in wave order by the wave that finishes the rows
    float* part = smem;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[((w * NMB + mb) * 16 + r) * 64 + lane] = acc[mb][0][r];
    __syncthreads();
    const int t = t0 + j;
    const bool tv = t < p.T;
    float kp = 1.f;
    if (p.keep && tv) kp = p.keep[(size_t)b * p.T + t];
    float sum[NMB][4], bv[NMB][4], rv[NMB][4];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * w + q;
            const int row = rg * 64 + 32 * mb + frag_row(r, h);
            const int rc = (row < p.Co) ? row : 0;
            bv[mb][q] = p.bias ? p.bias[rc] : 0.f;
            rv[mb][q] = p.res ? p.res[((size_t)b * p.Co + rc) * p.TS + t] : 0.f;
            float s = part[((0 * NMB + mb) * 16 + r) * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) s += part[((ww * NMB + mb) * 16 + r) * 64 + lane];
            sum[mb][q] = s;
        }
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = rg * 64 + 32 * mb + frag_row(4 * w + q, h);
            float v = (sum[mb][q] + bv[mb][q]) * p.scale;
            if (p.act == FS_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == FS_ACT_GELU) v = v * 0.5f * (1.f + erff(v * 0.70710678118654752440f));
            else if (p.act == FS_ACT_MISH) v = v * tanhf((v > 20.f) ? v : log1pf(expf(v)));
            v = (v + rv[mb][q]) * kp;
            if (row < p.Co) p.out[((size_t)b * p.Co + row) * p.TS + t] = tv ? v : 0.f;
        }
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm over C = 256 channels of every frame column
// ------------------------------------------------------------------------------------------------------------
struct FsLnParams {
    const float* in;        // [B][256][TS]
    float* out;             // [B][256][TS]
    const float* gamma;     // [256]
    const float* beta;      // [256]
    const float* keep;      // [B][T] or nullptr
    int T, TS;
    float eps;
    int relu_in;            // apply ReLU to the input first (predictor: Conv1d -> ReLU -> LayerNorm)
};

__global__ __launch_bounds__(kThreads) void k_fs_ln(const FsLnParams p) {
    __shared__ float red[8][32];
    const int tid = threadIdx.x, tc = tid & 31, part = tid >> 5;
    const int t = blockIdx.x * 32 + tc, b = blockIdx.y;
    const float* src = p.in + ((size_t)b * kC + part * 32) * p.TS + t;
    float v[32];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        float x = src[(size_t)i * p.TS];
        if (p.relu_in) x = fmaxf(x, 0.f);
        v[i] = x;
        s += x;
    }
    red[part][tc] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) mean += red[q][tc];
    mean *= (1.f / kC);
    __syncthreads();
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { const float e = v[i] - mean; d += e * e; }
    red[part][tc] = d;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) var += red[q][tc];
    var *= (1.f / kC);
    const float rstd = 1.f / sqrtf(var + p.eps);
    const bool tv = t < p.T;
    float kp = 1.f;
    if (p.keep && tv) kp = p.keep[(size_t)b * p.T + t];
    float* dst = p.out + ((size_t)b * kC + part * 32) * p.TS + t;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = part * 32 + i;
        const float y = ((v[i] - mean) * rstd * p.gamma[c] + p.beta[c]) * kp;
        dst[(size_t)i * p.TS] = tv ? y : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Index / mask glue of the forward as FOUR kernels (+ two more further down: k_fs_token_masks, k_fs_pitch_coarse) (round 6; rounds 1-5 left it to ~110 torch launches = 0.5 ms of a 3.2 ms forward:
// profiles/r6_05_fs2_kernel_stats.txt).  Every value is the one the torch ops produce: the same operations in the same order, one
// rounding each (the library is built with -ffp-contract=off); positions, indices and masks are integers.
//   k_fs_positions   utils/__init__.py:145-157 make_positions: pos = cumsum(x != pad) * (x != pad) + pad along the frame axis, for a token
//                    tensor (FastspeechEncoder.forward_embedding, tts_modules.py:338-346) or for channel 0 of a float tensor
//                    (FFTBlocks.forward: embed_positions(x[..., 0]), tts_modules.py:291)
//   k_fs_input_cm    the front end of FFTBlocks.forward (tts_modules.py:288-296) and of FastspeechEncoder / FastspeechMIDIEncoder
//                    .forward_embedding (:338-346, diffsinger_midi/fs2.py:20-28): token embedding * sqrt(H) (+ up to three more
//                    embeddings) (+ sinusoidal positions [* alpha]), the padding mask (given, or `x.abs().sum(-1).eq(0)`), `* nonpadding`,
//                    and the [B,T,C] -> channel-major transposition - writes xc [B][C][TS], keep [B][T] and the u8 key-padding mask
//   k_fs_gather_frames   the length regulator's gather, fs2.py:128-134: decoder_inp = gather(pad(encoder_out), mel2ph) and
//                    pitch_inp = (decoder_inp + spk_embed_f0) * (mel2ph > 0)
//   k_fs_sum_embed   fs2.py:136-141: ((decoder_inp + pitch_embed[idx]) [+ energy_embed[idx]] + spk_embed) * (mel2ph > 0)
// ------------------------------------------------------------------------------------------------------------
struct FsPosParams {
    const long long* tok;   // [B][T] int64 tokens, or nullptr
    const float* x;         // [B][T][C] float (channel 0 is tested), used when tok == nullptr
    int* pos;               // [B][T]
    int T, C, pad;
};

// one workgroup per utterance; thread k owns the k-th contiguous piece of the frame axis
__global__ __launch_bounds__(256) void k_fs_positions(const FsPosParams p) {
    __shared__ int part[256];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int per = (p.T + 255) / 256;
    const int i0 = min(p.T, tid * per), i1 = min(p.T, i0 + per);
    auto nz = [&](int t) -> bool {
        return p.tok ? (p.tok[(size_t)b * p.T + t] != (long long)p.pad) : (p.x[((size_t)b * p.T + t) * p.C] != (float)p.pad);
    };
    int s = 0;
    for (int t = i0; t < i1; ++t) s += nz(t) ? 1 : 0;
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    int c = part[tid];
    for (int t = i0; t < i1; ++t) {
        const bool z = nz(t);
        c += z ? 1 : 0;
        p.pos[(size_t)b * p.T + t] = (z ? c : 0) + p.pad;
    }
}

constexpr int kFsInMaxC = 512;             // hidden size of the front end (LDS tile [C][33])
struct FsInputParams {
    const long long* tok;   // [B][T] tokens (mode "embedding") or nullptr (mode "tensor")
    const float* emb;       // [V][C] token embedding table
    float emb_scale;        // sqrt(C)
    const float* add[3];    // up to three [B][T][C] tensors added in this order (MIDI pitch / duration / slur embeddings), or nullptr
    const float* x;         // [B][T][C] input tensor (mode "tensor")
    const int* pos;         // [B][T] positions or nullptr (no positional embedding)
    const float* pos_tab;   // [n_pos][C] sinusoidal table
    const float* alpha;     // DEVICE scalar the positional embedding is multiplied by (pos_embed_alpha), or nullptr (no factor)
    const unsigned char* pad_in;   // [B][T] given padding mask (nonzero = padded) or nullptr: computed as "every channel of the frame is zero"
    float* xc;              // [B][C][TS]
    float* keep;            // [B][T]
    unsigned char* pad_out; // [B][T]
    int T, TS, C, pad;
    int mask_mode;          // 1: the padding mask is applied (xc = value * (1 - padding)); 0: no mask (keep = 1, pad_out = 0: PitchPredictor)
};

// grid (TS / 32, B); 4 waves, wave w builds the frames 8 w .. 8 w + 7 of the tile (a lane <-> four consecutive channels: 16-byte accesses, 1 KiB
// per wave instruction), then the tile leaves channel-major
__global__ __launch_bounds__(256) void k_fs_input_cm(const FsInputParams p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [C][33]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t0 = blockIdx.x * 32, b = blockIdx.y, C = p.C;
    const float alpha = p.alpha ? p.alpha[0] : 1.f;
    constexpr int NQ = kFsInMaxC / 256;
    // every load of the wave's eight frames is requested before the first is used (a frame at a time would pay the memory latency eight times)
    float4 v[8][NQ];
    bool padded[8];
#pragma unroll
    for (int ff = 0; ff < 8; ++ff) {
        const int t = t0 + 8 * w + ff;
        const bool tv = t < p.T;
        const size_t bt = (size_t)b * p.T + (tv ? t : 0);
        const long long tk = p.tok ? p.tok[bt] : 0;
        const int ps = p.pos ? p.pos[bt] : 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = 256 * q + 4 * lane;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tv && c < C) {
                if (p.tok) {
                    const float4 e = *reinterpret_cast<const float4*>(p.emb + (size_t)tk * C + c);
                    a = make_float4(p.emb_scale * e.x, p.emb_scale * e.y, p.emb_scale * e.z, p.emb_scale * e.w);
#pragma unroll
                    for (int e3 = 0; e3 < 3; ++e3)
                        if (p.add[e3]) { const float4 d = *reinterpret_cast<const float4*>(p.add[e3] + bt * C + c); a.x = a.x + d.x; a.y = a.y + d.y; a.z = a.z + d.z; a.w = a.w + d.w; }
                } else {
                    a = *reinterpret_cast<const float4*>(p.x + bt * C + c);
                }
            }
            v[ff][q] = a;
        }
        bool pd = !tv;
        if (tv && p.mask_mode) {
            if (p.pad_in) pd = p.pad_in[bt] != 0;
            else if (p.tok) pd = tk == (long long)p.pad;
            else {
                bool nzq = false;
#pragma unroll
                for (int q = 0; q < NQ; ++q) nzq = nzq || v[ff][q].x != 0.f || v[ff][q].y != 0.f || v[ff][q].z != 0.f || v[ff][q].w != 0.f;
                pd = __ballot(nzq) == 0ull;                 // x.abs().sum(-1).eq(0): a sum of magnitudes is zero iff every one is
            }
        }
        padded[ff] = pd;
        if (p.pos && tv) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = 256 * q + 4 * lane;
                if (c < C) {
                    const float4 pe = *reinterpret_cast<const float4*>(p.pos_tab + (size_t)ps * C + c);
                    float4& a = v[ff][q];
                    if (p.alpha) { a.x = a.x + alpha * pe.x; a.y = a.y + alpha * pe.y; a.z = a.z + alpha * pe.z; a.w = a.w + alpha * pe.w; }
                    else { a.x = a.x + pe.x; a.y = a.y + pe.y; a.z = a.z + pe.z; a.w = a.w + pe.w; }
                }
            }
        }
    }
#pragma unroll
    for (int ff = 0; ff < 8; ++ff) {
        const int f = 8 * w + ff, t = t0 + f;
        const bool tv = t < p.T;
        const float kp = padded[ff] ? 0.f : 1.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = 256 * q + 4 * lane;
            if (c < C) {
                const float4 a = v[ff][q];
                tile[(c + 0) * 33 + f] = tv ? a.x * kp : 0.f; tile[(c + 1) * 33 + f] = tv ? a.y * kp : 0.f;
                tile[(c + 2) * 33 + f] = tv ? a.z * kp : 0.f; tile[(c + 3) * 33 + f] = tv ? a.w * kp : 0.f;
            }
        }
        if (lane == 0 && tv) { p.keep[(size_t)b * p.T + t] = kp; p.pad_out[(size_t)b * p.T + t] = padded[ff] ? 1 : 0; }
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;                    // 32 frames x 8 channel rows per pass
    for (int c = ty; c < C; c += 8) p.xc[((size_t)b * C + c) * p.TS + t0 + tx] = tile[c * 33 + tx];
}

struct FsGatherParams {
    const float* enc;       // [B][Tp][C] encoder output
    const long long* mel2ph;// [B][T], 0 = padding frame, k = phone k - 1
    const float* spk;       // [B][C] speaker embedding added for out2, or nullptr (the reference adds the integer 0)
    float* out1;            // [B][T][C] decoder_inp = gather(pad(enc), mel2ph)
    float* out2;            // [B][T][C] (out1 + spk) * (mel2ph > 0), or nullptr
    int T, Tp, C;
};

// grid (ceil(T / 8), B); a wave per frame row, lanes <-> float4 of channels
__global__ __launch_bounds__(256) void k_fs_gather_frames(const FsGatherParams p) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    for (int f = w; f < 8; f += 4) {
        const int t = blockIdx.x * 8 + f;
        if (t >= p.T) return;
        const long long m = p.mel2ph[(size_t)b * p.T + t];
        const bool live = m > 0;
        const float* src = p.enc + ((size_t)b * p.Tp + (live ? (size_t)(m - 1) : 0)) * p.C;
        for (int c = 4 * lane; c < p.C; c += 256) {
            float4 v = live ? *reinterpret_cast<const float4*>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const size_t o = ((size_t)b * p.T + t) * p.C + c;
            *reinterpret_cast<float4*>(p.out1 + o) = v;
            if (p.out2) {
                float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.spk) s4 = *reinterpret_cast<const float4*>(p.spk + (size_t)b * p.C + c);
                const float kp = live ? 1.f : 0.f;
                v.x = (v.x + s4.x) * kp; v.y = (v.y + s4.y) * kp; v.z = (v.z + s4.z) * kp; v.w = (v.w + s4.w) * kp;
                *reinterpret_cast<float4*>(p.out2 + o) = v;
            }
        }
    }
}

struct FsSumEmbedParams {
    const float* dec;       // [B][T][C]
    const long long* idx1;  // [B][T] rows of tab1 (pitch), or nullptr
    const float* tab1;      // [n][C]
    const float* add1;      // [B][T][C] a ready embedding instead of (idx1, tab1), or nullptr
    const long long* idx2;  // [B][T] rows of tab2 (energy), or nullptr
    const float* tab2;
    const float* spk;       // [B][C] or nullptr (the reference adds the integer 0)
    const long long* mel2ph;// [B][T]
    float* out;             // [B][T][C]
    int T, C;
};

__global__ __launch_bounds__(256) void k_fs_sum_embed(const FsSumEmbedParams p) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    for (int f = w; f < 8; f += 4) {
        const int t = blockIdx.x * 8 + f;
        if (t >= p.T) return;
        const size_t bt = (size_t)b * p.T + t;
        const float kp = (p.mel2ph[bt] > 0) ? 1.f : 0.f;
        for (int c = 4 * lane; c < p.C; c += 256) {
            const size_t o = bt * p.C + c;
            float4 v = *reinterpret_cast<const float4*>(p.dec + o);
            if (p.idx1 || p.add1) {
                const float4 e = p.add1 ? *reinterpret_cast<const float4*>(p.add1 + o) : *reinterpret_cast<const float4*>(p.tab1 + (size_t)p.idx1[bt] * p.C + c);
                v.x = v.x + e.x; v.y = v.y + e.y; v.z = v.z + e.z; v.w = v.w + e.w;
            }
            if (p.idx2) {
                const float4 e = *reinterpret_cast<const float4*>(p.tab2 + (size_t)p.idx2[bt] * p.C + c);
                v.x = v.x + e.x; v.y = v.y + e.y; v.z = v.z + e.z; v.w = v.w + e.w;
            }
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.spk) s4 = *reinterpret_cast<const float4*>(p.spk + (size_t)b * p.C + c);
            v.x = (v.x + s4.x) * kp; v.y = (v.y + s4.y) * kp; v.z = (v.z + s4.z) * kp; v.w = (v.w + s4.w) * kp;
            *reinterpret_cast<float4*>(p.out + o) = v;
        }
    }
}

// The rest of the forward's glue (round 6, second half: profiles/r6_39_fs2_glue_trace.txt - 35 torch launches of 2-6 us were left between
// the encoder and the decoder, 22 of them the pitch quantisation):
//   k_fs_token_masks   the masks every stage derives from an int64 index tensor: (v > 0).float() (fs2.py:98, :127), v == 0 (fs2.py:157, :199)
//                      and (~(v == 0)).float() (DurationPredictor.forward, tts_modules.py:109-118) in one pass
//   k_fs_pitch_coarse  utils/pitch_utils.py:64-77 denorm_f0 (pitch_norm 'standard' / 'log', the uv and padding masks) followed by
//                      utils/pitch_utils.py:21-30 f0_to_coarse, every operation as the torch elementwise kernels evaluate it: one fp32 rounding
//                      per operation in the reference's order, a division by a Python scalar as the multiplication by its fp32 reciprocal
//                      (ATen div_true_kernel_cuda), pow / log through the device library (__ocml_pow_f32 / __ocml_log_f32, the functions ATen's
//                      pow_tensor_tensor_kernel and log_kernel_cuda call).  stage 1 / 2 split the kernel around the logarithm and pow_in takes the
//                      power from the caller: the forms the host falls back to if a torch build's device library ever rounds differently
//                      (tests/test_gpu_fs2.py compares every float of the working range).
struct FsTokMaskParams {
    const long long* v;     // [n] int64
    float* gt0;             // [n] (v > 0) as float, or nullptr
    unsigned char* eq0;     // [n] (v == 0), or nullptr
    float* ne0;             // [n] (v != 0) as float, or nullptr
    long long n;
};

__global__ __launch_bounds__(256) void k_fs_token_masks(const FsTokMaskParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const long long v = p.v[i];
    if (p.gt0) p.gt0[i] = (v > 0) ? 1.f : 0.f;
    if (p.eq0) p.eq0[i] = (v == 0) ? 1 : 0;
    if (p.ne0) p.ne0[i] = (v == 0) ? 0.f : 1.f;
}

struct FsPitchParams {
    const float* f0;                // [B][T] through (sb, st) element strides: normalised f0 (pow_in: the caller's 2 ** f0, contiguous)
    long long sb, st;
    const float* uv_f;              // [B][T] float (uv > 0 = unvoiced) or nullptr
    const unsigned char* uv_u8;     // [B][T] bool / u8 or nullptr
    const long long* mel2ph;        // [B][T]: frames with mel2ph == 0 are padding, or nullptr
    float* f0_denorm;               // [B][T] out (stage 0 / 1)
    float* tmp;                     // [B][T]: 1 + f0_denorm / 700 (stage 1: out; stage 2: its logarithm, in)
    long long* coarse;              // [B][T] out (stage 0 / 2)
    int T, norm, stage, pow_in;     // norm: 1 standard, 2 log
    long long n;
    float base, mean, std;          // 2 (a run-time value: the generic path of the device library's pow, as ATen runs it)
    float inv700, mel_min, scale, inv_range, top;
};

__global__ __launch_bounds__(256) void k_fs_pitch_coarse(const FsPitchParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    float c;
    if (p.stage != 2) {
        const long long b = i / p.T, t = i - b * p.T;
        const float x = p.f0[b * p.sb + t * p.st];
        float d = x;
        if (p.norm == 1) { d = x * p.std; d = d + p.mean; }
        else if (!p.pow_in) d = powf(p.base, x);
        bool z = false;
        if (p.uv_f) z = p.uv_f[i] > 0.f;
        if (p.uv_u8) z = z || (p.uv_u8[i] != 0);
        if (p.mel2ph) z = z || (p.mel2ph[i] == 0);
        if (z) d = 0.f;
        p.f0_denorm[i] = d;
        float a = d * p.inv700;
        a = a + 1.0f;
        if (p.stage == 1) { p.tmp[i] = a; return; }
        c = logf(a);
    } else {
        c = p.tmp[i];
    }
    float m = c * 1127.0f;
    float v = m - p.mel_min;
    v = v * p.scale;
    v = v * p.inv_range;
    v = v + 1.0f;
    m = (m > 0.f) ? v : m;
    m = (m <= 1.0f) ? 1.0f : m;
    m = (m > p.top) ? p.top : m;
    p.coarse[i] = (long long)(m + 0.5f);
}

// ------------------------------------------------------------------------------------------------------------
// self-attention core: one wave per (32-query tile, head, utterance)
// ------------------------------------------------------------------------------------------------------------
struct FsAttnParams {
    const float* qkv;               // [B][3C][TS]: q rows [0,C), k rows [C,2C), v rows [2C,3C); head hh = rows hh*HD..
    const unsigned char* key_pad;   // [B][T], nonzero = padded key (key_padding_mask)
    float* out;                     // [B][C][TS]
    int C, T, TS;
    float scale;                    // head_dim ** -0.5, applied to q like the reference (q * scaling before q k^T)
};

// Workgroup = (32-query tile, head, utterance), 4 waves; wave w takes the key tiles w, w+4, ... with its own online-softmax state
// and the four partial results are merged through LDS at the end.  Per key tile and wave:
//   S[tk][tq] = sum_d k[d][tk] q[d][tq]   - q lives in registers for the whole kernel (it is the B operand: lane (tq, h) needs
//               q[8c+4h+s][tq]), k goes global -> register directly (every element feeds exactly one MFMA: no reuse to stage for)
//   P = exp(S - m) stays in registers too: the accumulator fragment a lane holds (tk = frag_row(r, h)) is exactly the B fragment
//               the P V product wants from it (tk = 8c + 4h + s with r = 4c + s)
//   O[d][tq] += sum_tk v[d][tk] P[tk][tq] - v is the A operand read ACROSS rows: staged in this wave's LDS slice [128][33]
__device__ __forceinline__ float fs_ldg32(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float4 fs_ldg128(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const f32x4_ f = __builtin_bit_cast(f32x4_, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return make_float4(f.x, f.y, f.z, f.w);
}

template <int HD>
__global__ __launch_bounds__(kThreads, 2) void k_fs_attn(const FsAttnParams p) {
    constexpr int NMB = HD / 32, LDV = 33;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LDT = HD + 1;
    static_assert(32 * LDT <= HD * LDV, "the key-major V tile fits the wave's merge slice");
    float* vt = smem + w * (HD * LDV);                  // this wave's V tile (key-major [32][LDT]) / merge slice ([HD][LDV])
    float* vtw = vt + 4 * (lane & 7) * LDT + (lane >> 3);           // this lane's write base: key 4 (lane % 8), row lane / 8
    const float* vtr = vt + 4 * h * LDT + j;                        // this lane's read base: key 4 h, row j
    float* ml = smem + 4 * (HD * LDV);                  // [4 waves][32 queries][2] running max and sum
    const int tq0 = blockIdx.x * 32, hh = blockIdx.y, b = blockIdx.z;
    const float* qb = p.qkv + ((size_t)b * 3 * p.C + hh * HD) * p.TS;
    const float* kb = qb + (size_t)p.C * p.TS;
    const float* vb = kb + (size_t)p.C * p.TS;
    // buffer loads: wave-uniform descriptor + ONE per-lane offset + a scalar row offset per load - no per-load 64-bit address math
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qb), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kb), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vb), 0, 0x7ffffff0, 0x00020000);
    const int row_b = p.TS * 4;                                     // bytes per channel row
    const int lk = (4 * h * p.TS + j) * 4;                          // lane part of a q / k element address: row 4h, column j
    const int lv = ((lane >> 3) * p.TS + 4 * (lane & 7)) * 4;       // lane part of a v float4 address: row lane/8, column 4 (lane%8)
    float qr[HD / 2];                                   // q[8c + 4h + s][tq0 + j] * scale, index 4c + s
#pragma unroll
    for (int i = 0; i < HD / 2; ++i) qr[i] = fs_ldg32(rq, lk + tq0 * 4, (8 * (i >> 2) + (i & 3)) * row_b) * p.scale;
    // Online softmax, round 6 (the densest loop of the library by vector work until then: 3.76 vector-ALU instructions per MFMA, each ~8 cycles
    // of matrix time beside an fp32 MFMA - profiles/r5_32_isa_scan.txt):
    //   * P = exp2(fma(S, log2 e, -m log2 e)): one fma + v_exp_f32 per element instead of libm's expf (~12 instructions);
    //   * LAZY rescale: the running maximum m a query's P are taken against is raised only when a tile's maximum exceeds it by more than
    //     kLazy (P <= e^kLazy = 245: no range problem in fp32) - then, and only then, o and l are multiplied by exp(m_old - m_new).  After the
    //     first tile that is rare: the 64 multiplications of o per tile are gone from the common path (same value of O / L, other rounding);
    //   * the key mask (tail of T, key_padding_mask) is ONE scalar test per tile: a tile inside T without a padded key takes no per-element
    //     compare / select at all; the pad bytes of a tile are read once per wave (lane j <-> key tk0 + j) and turned into a ballot.
    constexpr float kL2E = 1.4426950408889634f, kLazy = 5.5f;
    float m = -INFINITY, l = 0.f;                       // m: the maximum the running sums are relative to (per query: both half-waves agree)
    f32x16 o[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mb][r] = 0.f;
    // k in groups of 16 loads, DOUBLE-BUFFERED (round 5): group g + 1 is requested before the MFMAs of group g are issued, and the first group of
    // the NEXT key tile travels under the P V product
    float kv[2][16];
    auto fetch_k = [&](int buf, int g16, int tk) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = 16 * g16 + u;
            kv[buf][u] = fs_ldg32(rk, lk + tk * 4, (8 * (i >> 2) + (i & 3)) * row_b);
        }
    };
    const unsigned char* kpb = p.key_pad ? p.key_pad + (size_t)b * p.T : nullptr;
    if (32 * w < p.T) fetch_k(0, 0, 32 * w);
    for (int tk0 = 32 * w; tk0 < p.T; tk0 += 128) {
        // V tile -> registers in two halves (row idx >> 3, float4 column idx & 7): the first is requested before the S product and
        // written to LDS behind it, the second is requested then and lands during the softmax arithmetic
        constexpr int NV = HD / 16;
        float4 vv[NV];
        auto fetch_v = [&](int half) {
#pragma unroll
            for (int it = 0; it < NV; ++it) {
                vv[it] = fs_ldg128(rv, lv + tk0 * 4, (half * NV + it) * 8 * row_b);
            }
        };
        // the V tile in LDS is KEY-major, vt[tk][LDT] with LDT = HD + 1 (round 6; rounds 2-5: [d][33]): every address of the writes below and of
        // the P V product's reads is ONE per-lane base + a compile-time offset (ds_read_b32 / ds_write_b32 take 16 bits of it) - the [d][33] tile
        // put a row block of 32 d 1 056 floats apart, beyond the 8-bit offsets of the paired LDS instructions, and cost an address addition per
        // access: 73 v_add_u32 per key tile.  Reads: lanes j <-> consecutive d: conflict-free.
        auto store_v = [&](int half) {
#pragma unroll
            for (int it = 0; it < NV; ++it) {
                float* d = vtw + (half * NV + it) * 8;            // row d = (half NV + it) 8 + lane / 8, keys 4 (lane % 8) + (0..3)
                d[0] = vv[it].x; d[LDT] = vv[it].y; d[2 * LDT] = vv[it].z; d[3 * LDT] = vv[it].w;
            }
        };
        unsigned padb = 0;
        if (kpb && tk0 + j < p.T) padb = kpb[tk0 + j];  // (requested in front of the S product, used behind it)
        fetch_v(0);
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g16 = 0; g16 < HD / 32; ++g16) {
            if (g16 + 1 < HD / 32) fetch_k((g16 + 1) & 1, g16 + 1, tk0);
            DSD_SB();
#pragma unroll
            for (int u = 0; u < 16; ++u) s = mfma32(kv[g16 & 1][u], qr[16 * g16 + u], s);
            DSD_SB();
        }
        __builtin_amdgcn_wave_barrier();                // the previous tile's LDS reads of this wave are done (in-order LDS)
        store_v(0);
        fetch_v(1);
        if (tk0 + 128 < p.T) fetch_k(0, 0, tk0 + 128); // the next tile's first k group: in flight under the softmax and the P V product
        const unsigned long long padm = __ballot(padb != 0u);
        const bool masked = (tk0 + 32 > p.T) || ((unsigned)padm != 0u);          // wave-uniform
        if (masked) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tk = frag_row(r, h);
                const bool dead = (tk0 + tk >= p.T) || (((unsigned)padm >> tk) & 1u);
                s[r] = dead ? -INFINITY : s[r];
            }
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // raise the reference maximum?  (first live tile: m = -inf, nothing accumulated yet; later: only by more than kLazy)
        const bool raise = (mx > m + kLazy) || (m == -INFINITY && mx > -INFINITY);
        if (__ballot(raise) != 0ull) {
            const float mn = raise ? mx : m;
            const float alpha = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m - mn) * kL2E);
            l *= alpha;
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[mb][r] *= alpha;
            m = mn;
        }
        const float m2 = (m == -INFINITY) ? 0.f : m * kL2E;      // (every key so far dead: the s are -inf, exp2(-inf - 0) = 0)
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], kL2E, -m2));
            ps += s[r];
        }
        l += ps;
        store_v(1);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // P V: 16 steps (key pairs of the k = 2 MFMA) x NMB row blocks; the A operands of step n + 1 are read from LDS while the MFMAs of step n
        // issue (left to itself hipcc put every read directly in front of its MFMAs: read, s_waitcnt lgkmcnt(0), two MFMAs - the LDS latency
        // exposed 32 times per tile)
        float va[2][NMB];
        auto lds_v = [&](float (&dst)[NMB], int n) {       // step n <-> key 8 (n / 4) + 4 h + n % 4: register s[n] of this lane
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) dst[mb] = vtr[(8 * (n >> 2) + (n & 3)) * LDT + 32 * mb];
        };
        lds_v(va[0], 0);
        DSD_SB();
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            if (n + 1 < 16) lds_v(va[(n + 1) & 1], n + 1);
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) o[mb] = mfma32(va[n & 1][mb], s[n], o[mb]);
            DSD_SB();
        }
    }
    // merge the four waves: O = sum_w O_w exp(m_w - M) / sum_w l_w exp(m_w - M)
    const float lt = l + __shfl_xor(l, 32, 64);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) vt[(32 * mb + frag_row(r, h)) * LDV + j] = o[mb][r];
    if (h == 0) { ml[(w * 32 + j) * 2] = m; ml[(w * 32 + j) * 2 + 1] = lt; }
    __syncthreads();
    float mw[4], sc[4], M = -INFINITY, L = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { mw[q] = ml[(q * 32 + j) * 2]; M = fmaxf(M, mw[q]); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sc[q] = (mw[q] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mw[q] - M) * 1.4426950408889634f);
        L += ml[(q * 32 + j) * 2 + 1] * sc[q];
    }
    const float inv = (L > 0.f) ? 1.f / L : 0.f;
    const int t = tq0 + j;
    // wave w finishes the d rows [32 w, 32 w + 32) (HD = 128: one row block per wave)
    float* ob = p.out + ((size_t)b * p.C + hh * HD) * p.TS + t;
#pragma unroll
    for (int mb = w; mb < NMB; mb += 4)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * mb + frag_row(r, h);
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += smem[q * (HD * LDV) + row * LDV + j] * sc[q];
            ob[(size_t)row * p.TS] = (t < p.T) ? acc * inv : 0.f;
        }
}
template <int HD>
constexpr int fs_attn_lds_bytes() { return (4 * HD * 33 + 4 * 32 * 2) * (int)sizeof(float); }

// One ancestral step x_{t-1} = p_sample(x_t, eps) (usr/diff/shallow_diffusion_tts.py:134-166) for a denoiser that is not the fused
// DiffNet (the `FFT` candidate decoder): the arithmetic of the DiffNet head epilogue as a stand-alone element-wise kernel.
__global__ void k_fs_p_sample(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ z, float sa, float sb,
                              float c1, float c2, float sigma, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float xv = x[i];
        float x0 = __fsub_rn(__fmul_rn(sa, xv), __fmul_rn(sb, eps[i]));
        x0 = fminf(fmaxf(x0, -1.f), 1.f);
        const float mean = __fadd_rn(__fmul_rn(c1, x0), __fmul_rn(c2, xv));
        x[i] = __fadd_rn(mean, __fmul_rn(sigma, z[i]));
    }
}

// The element-wise ends of GaussianDiffusion.p_losses (usr/diff/shallow_diffusion_tts.py:206-231) around the denoiser's training forward
// (round 6: profiles/r6_42_train_glue_trace.txt - q_sample was five torch launches, the L1 loss and its backward nine):
//   k_fs_q_sample_rows   x_noisy[b] = sqrt_alphas_cumprod[t_b] * x_start[b] + sqrt_one_minus_alphas_cumprod[t_b] * noise[b]  (:206-211; two
//                        products and one sum, each rounded once, as the tensor ops)
//   k_fs_l1_partial / k_fs_l1_final   mean |a - b| in a fixed order (per-thread strided sums, a tree per workgroup, one workgroup over the partials)
//   k_fs_l1_bwd          d mean|a - b| / d b = -(sign(a - b) * (g / N)), g the DEVICE scalar arriving from autograd
__global__ __launch_bounds__(256) void k_fs_q_sample_rows(const float* __restrict__ x0, const float* __restrict__ noise, const long long* __restrict__ t,
                                                          const float* __restrict__ tab_a, const float* __restrict__ tab_s, float* __restrict__ out, int per_row4,
                                                          int n_steps) {
    const int b = blockIdx.y;
    const long long tb = t[b];
    const bool ok = tb >= 0 && tb < n_steps;                        // a step outside the schedule: NaN rows (torch.gather raises; this fails loudly too)
    const float a = ok ? tab_a[tb] : __builtin_nanf(""), s = ok ? tab_s[tb] : __builtin_nanf("");
    const float4* x4 = reinterpret_cast<const float4*>(x0) + (size_t)b * per_row4;
    const float4* n4 = reinterpret_cast<const float4*>(noise) + (size_t)b * per_row4;
    float4* o4 = reinterpret_cast<float4*>(out) + (size_t)b * per_row4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per_row4; i += gridDim.x * 256) {
        const float4 x = x4[i], n = n4[i];
        o4[i] = make_float4(__fadd_rn(__fmul_rn(a, x.x), __fmul_rn(s, n.x)), __fadd_rn(__fmul_rn(a, x.y), __fmul_rn(s, n.y)),
                            __fadd_rn(__fmul_rn(a, x.z), __fmul_rn(s, n.z)), __fadd_rn(__fmul_rn(a, x.w), __fmul_rn(s, n.w)));
    }
}

__device__ __forceinline__ float fs_block_sum256(float v, float* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] = __fadd_rn(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    return sh[0];
}

__global__ __launch_bounds__(256) void k_fs_l1_partial(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partial, size_t n) {
    __shared__ float sh[256];
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc = __fadd_rn(acc, fabsf(__fsub_rn(a[i], b[i])));
    const float tot = fs_block_sum256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_fs_l1_final(const float* __restrict__ partial, int nblk, float inv_n, float* __restrict__ out) {
    __shared__ float sh[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) acc = __fadd_rn(acc, partial[i]);
    const float tot = fs_block_sum256(acc, sh);
    if (threadIdx.x == 0) out[0] = __fmul_rn(tot, inv_n);
}

__global__ __launch_bounds__(256) void k_fs_l1_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g, float inv_n,
                                                   float* __restrict__ db, size_t n) {
    const float gn = __fmul_rn(g[0], inv_n);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = __fsub_rn(a[i], b[i]);
        const float sg = (d > 0.f) ? 1.f : (d < 0.f) ? -1.f : d;                    // torch.sign: 0 at 0, NaN stays NaN
        db[i] = -__fmul_rn(gn, sg);
    }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient of a Conv1d / Linear (training):
//     dW[co][ci][tap] = sum_b sum_t dy[b][co][t] * x[b][ci][t + tap * dil - pad]
// A contraction over FRAMES: both operands are activations.  A workgroup owns a 128 (co) x 64 (ci) x KT tile of dW and one
// of `nsplit` frame ranges; per 32-frame chunk it stages dy [128][32] and x [64][32 + 16] in LDS (odd row strides: the MFMA
// operands are read ACROSS rows), D[i = co][j = ci] += A[i][k = t] B[k][j] on v_mfma_f32_32x32x2_f32.  Partial tiles go to
// part[split][co][ci][tap]; k_fs_wgrad_reduce sums the splits in a fixed order (deterministic gradients, no float atomics).
// ------------------------------------------------------------------------------------------------------------
constexpr int kWgLdy = 33, kWgLdx = 49;
struct FsWgradParams {
    const float* dy;        // [B][Co][TS]
    const float* x;         // [B][Ci][TS]
    float* part;            // [nsplit][Co][Ci][KT]
    float* part_b;          // [nsplit][Co] bias-gradient partials, or nullptr
    int B, Ci, Co, dil, pad, T, TS, nsplit;
};

template <int KT>
__global__ __launch_bounds__(kThreads, 2) void k_fs_wgrad(const FsWgradParams p) {
    __shared__ float dyt[128 * kWgLdy];
    __shared__ float xt[64 * kWgLdx];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.x * 128, ci0 = blockIdx.y * 64, split = blockIdx.z;
    const int tiles_per_utt = p.TS / 32, ntile = p.B * tiles_per_utt;
    const int per = (ntile + p.nsplit - 1) / p.nsplit;
    const int tile_lo = split * per, tile_hi = min(ntile, tile_lo + per);
    f32x16 acc[KT][2];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][nb][r] = 0.f;
    // software pipeline: the global loads of tile i+1 are in flight while tile i is multiplied.
    // thread tid stages dy rows (tid >> 3) + 32 q, float4 column (tid & 7) [q = 0..3] and x float4 tid + 256 q [q = 0..2]
    float4 dv[4], xv[3];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};                  // bias gradient: row sums of the dy this thread stages
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_utt, t0 = (tile - b * tiles_per_utt) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = (tid >> 3) + 32 * q, g = tid & 7;
            dv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co0 + row < p.Co) {
                float4 v = *reinterpret_cast<const float4*>(p.dy + ((size_t)b * p.Co + co0 + row) * p.TS + t0 + 4 * g);
                const int t = t0 + 4 * g;                   // frames >= T carry no gradient (whatever the buffer holds there)
                v.x = (t + 0 < p.T) ? v.x : 0.f; v.y = (t + 1 < p.T) ? v.y : 0.f; v.z = (t + 2 < p.T) ? v.z : 0.f; v.w = (t + 3 < p.T) ? v.w : 0.f;
                dv[q] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + 256 * q, row = idx / 12, g = idx - row * 12;
            const int t = t0 - kFsHalo + 4 * g;
            xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci0 + row < p.Ci && t >= 0 && t < p.TS) xv[q] = *reinterpret_cast<const float4*>(p.x + ((size_t)b * p.Ci + ci0 + row) * p.TS + t);
        }
    };
    if (tile_lo < tile_hi) fetch(tile_lo);
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        __syncthreads();                                    // the previous tile's LDS reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* d = dyt + ((tid >> 3) + 32 * q) * kWgLdy + 4 * (tid & 7);
            d[0] = dv[q].x; d[1] = dv[q].y; d[2] = dv[q].z; d[3] = dv[q].w;
            bsum[q] += (dv[q].x + dv[q].y) + (dv[q].z + dv[q].w);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + 256 * q, row = idx / 12, g = idx - row * 12;
            float* d = xt + row * kWgLdx + 4 * g;
            d[0] = xv[q].x; d[1] = xv[q].y; d[2] = xv[q].z; d[3] = xv[q].w;
        }
        __syncthreads();
        if (tile + 1 < tile_hi) fetch(tile + 1);
        const float* ap = dyt + (32 * w + j) * kWgLdy + h;                        // A[i = co][k = t]: lane half h supplies t = 2 s + h
        const float* bp = xt + j * kWgLdx + kFsHalo - p.pad + h;                  // B[k = t][j = ci]
#pragma unroll 4
        for (int s2 = 0; s2 < 16; ++s2) {
            const float a = ap[2 * s2];
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[k][nb] = mfma32(a, bp[nb * 32 * kWgLdx + 2 * s2 + k * p.dil], acc[k][nb]);
        }
    }
    float* out = p.part + (size_t)split * p.Co * p.Ci * KT;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * w + frag_row(r, h), ci = ci0 + 32 * nb + j;
                if (co < p.Co && ci < p.Ci) out[((size_t)co * p.Ci + ci) * KT + k] = acc[k][nb][r];
            }
    // bias-gradient partials: the 8 threads of a row sit in consecutive lanes; only the ci-tile-0 workgroups report
    if (p.part_b && blockIdx.y == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sb = bsum[q];
            sb += __shfl_xor(sb, 1, 64); sb += __shfl_xor(sb, 2, 64); sb += __shfl_xor(sb, 4, 64);
            const int row = co0 + (tid >> 3) + 32 * q;
            if ((tid & 7) == 0 && row < p.Co) p.part_b[(size_t)split * p.Co + row] = sb;
        }
    }
}

__global__ void k_fs_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, size_t n, int nsplit, int accumulate) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = accumulate ? dw[i] : 0.f;
        // (the loads of eight splits in flight, the additions in split order: the bits do not depend on the unrolling)
#pragma unroll 8
        for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * n + i];
        dw[i] = s;
    }
}

// the same for a GROUP of taps: part [nsplit][nrow][gk] (gk taps computed, the first ntap of them valid) -> dw [nrow][KT] at taps tap0 .. tap0 + ntap
__global__ void k_fs_wgrad_reduce_taps(const float* __restrict__ part, float* __restrict__ dw, size_t nrow, int gk, int ntap, int KT, int tap0, int nsplit,
                                       int accumulate) {
    const size_t n = nrow * ntap;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / ntap;
        const int k = (int)(i - row * ntap);
        float* d = dw + row * KT + tap0 + k;
        float s = accumulate ? *d : 0.f;
#pragma unroll 8
        for (int sp = 0; sp < nsplit; ++sp) s += part[((size_t)sp * nrow + row) * gk + k];
        *d = s;
    }
}

// db[c] = sum_b sum_t dy[b][c][t] (bias gradient): one wave per channel, fixed summation order
__global__ void k_fs_bias_grad(const float* __restrict__ dy, float* __restrict__ db, int B, int C, int T, int TS, int accumulate) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* row = dy + ((size_t)b * C + c) * TS;
        for (int t = lane; t < T; t += 64) s += row[t];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) db[c] = (accumulate ? db[c] : 0.f) + s;
}

// ------------------------------------------------------------------------------------------------------------
// element-wise pieces of ResidualBlock.forward (usr/diff/net.py:66-78) for the training path, forward and backward, on
// channel-major [B][C][TS] tensors (float4 along the frame axis; frames >= T are written as zero: the zero-tail invariant)
// ------------------------------------------------------------------------------------------------------------
// y = x + step[b][c] (net.py:69: `x + diffusion_step`), zero tail
__global__ void k_tr_add_step(const float4* __restrict__ x, const float* __restrict__ step, float4* __restrict__ y, int C, int T, int TS, size_t n4) {
    const int q = TS / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / q;                       // b * C + c
        const int t = (int)(i - row * q) * 4;
        const float d = step[row];
        float4 v = x[i];
        v.x = (t + 0 < T) ? v.x + d : 0.f; v.y = (t + 1 < T) ? v.y + d : 0.f; v.z = (t + 2 < T) ? v.z + d : 0.f; v.w = (t + 3 < T) ? v.w + d : 0.f;
        y[i] = v;
    }
}
// out[row] = sum_{t < T} g[row][t]  (gradient of the broadcast step projection); one wave per row, fixed order
__global__ void k_tr_rowsum(const float* __restrict__ g, float* __restrict__ out, int T, int TS) {
    const size_t row = blockIdx.x;
    const int lane = threadIdx.x;
    float s = 0.f;
    for (int t = lane; t < T; t += 64) s += g[row * TS + t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[row] = s;
}
// g = sigmoid(a[:, :C]) * tanh(a[:, C:]) (net.py:73-74).  a [B][2C][TS] -> g [B][C][TS]
__global__ void k_tr_gate(const float4* __restrict__ a, float4* __restrict__ g, int C, int T, int TS, size_t n4) {
    const int q = TS / 4;
    const size_t per_b = (size_t)C * q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per_b, r = i - b * per_b;
        const int t = (int)(r % q) * 4;
        const float4 va = a[b * 2 * per_b + r], vf = a[b * 2 * per_b + per_b + r];
        float4 o;
        o.x = (t + 0 < T) ? (1.f / (1.f + expf(-va.x))) * tanhf(vf.x) : 0.f;
        o.y = (t + 1 < T) ? (1.f / (1.f + expf(-va.y))) * tanhf(vf.y) : 0.f;
        o.z = (t + 2 < T) ? (1.f / (1.f + expf(-va.z))) * tanhf(vf.z) : 0.f;
        o.w = (t + 3 < T) ? (1.f / (1.f + expf(-va.w))) * tanhf(vf.w) : 0.f;
        g[i] = o;
    }
}
// da[:, :C] = dg * tanh(f) * s (1 - s), da[:, C:] = dg * s * (1 - tanh(f)^2), s = sigmoid(gate)
__global__ void k_tr_gate_bwd(const float4* __restrict__ a, const float4* __restrict__ dg, float4* __restrict__ da, int C, int T, int TS, size_t n4) {
    const int q = TS / 4;
    const size_t per_b = (size_t)C * q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per_b, r = i - b * per_b;
        const int t = (int)(r % q) * 4;
        const float4 va = a[b * 2 * per_b + r], vf = a[b * 2 * per_b + per_b + r], d = dg[i];
        const float av[4] = {va.x, va.y, va.z, va.w}, fv[4] = {vf.x, vf.y, vf.z, vf.w}, dv[4] = {d.x, d.y, d.z, d.w};
        float oa[4], of[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sg = 1.f / (1.f + expf(-av[e])), th = tanhf(fv[e]);
            const bool ok = t + e < T;
            oa[e] = ok ? dv[e] * th * (sg * (1.f - sg)) : 0.f;
            of[e] = ok ? dv[e] * sg * (1.f - th * th) : 0.f;
        }
        da[b * 2 * per_b + r] = make_float4(oa[0], oa[1], oa[2], oa[3]);
        da[b * 2 * per_b + per_b + r] = make_float4(of[0], of[1], of[2], of[3]);
    }
}
// x' = (x + y[:, :C]) / sqrt(2), skip' = skip + y[:, C:] (net.py:76-78, :121-126); skip may be nullptr (first layer)
__global__ void k_tr_res_skip(const float4* __restrict__ x, const float4* __restrict__ y, const float4* __restrict__ skip, float4* __restrict__ xo,
                              float4* __restrict__ so, int C, int T, int TS, size_t n4) {
    const int q = TS / 4;
    const size_t per_b = (size_t)C * q;
    const float s2 = 1.41421356237309504880f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per_b, r = i - b * per_b;
        const int t = (int)(r % q) * 4;
        const float4 vx = x[i], vr = y[b * 2 * per_b + r], vs = y[b * 2 * per_b + per_b + r];
        float4 o, k = skip ? skip[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        o.x = (t + 0 < T) ? (vx.x + vr.x) / s2 : 0.f; o.y = (t + 1 < T) ? (vx.y + vr.y) / s2 : 0.f;
        o.z = (t + 2 < T) ? (vx.z + vr.z) / s2 : 0.f; o.w = (t + 3 < T) ? (vx.w + vr.w) / s2 : 0.f;
        k.x = (t + 0 < T) ? k.x + vs.x : 0.f; k.y = (t + 1 < T) ? k.y + vs.y : 0.f; k.z = (t + 2 < T) ? k.z + vs.z : 0.f; k.w = (t + 3 < T) ? k.w + vs.w : 0.f;
        xo[i] = o;
        so[i] = k;
    }
}
// backward: dx = dxo / sqrt(2); dy[:, :C] = dxo / sqrt(2); dy[:, C:] = dso   (dskip_in = dso is passed through by the caller)
__global__ void k_tr_res_skip_bwd(const float4* __restrict__ dxo, const float4* __restrict__ dso, float4* __restrict__ dx, float4* __restrict__ dy,
                                  int C, int T, int TS, size_t n4) {
    const int q = TS / 4;
    const size_t per_b = (size_t)C * q;
    const float s2 = 1.41421356237309504880f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per_b, r = i - b * per_b;
        const int t = (int)(r % q) * 4;
        const float4 d = dxo[i], ds = dso[i];
        float4 o, k;
        o.x = (t + 0 < T) ? d.x / s2 : 0.f; o.y = (t + 1 < T) ? d.y / s2 : 0.f; o.z = (t + 2 < T) ? d.z / s2 : 0.f; o.w = (t + 3 < T) ? d.w / s2 : 0.f;
        k.x = (t + 0 < T) ? ds.x : 0.f; k.y = (t + 1 < T) ? ds.y : 0.f; k.z = (t + 2 < T) ? ds.z : 0.f; k.w = (t + 3 < T) ? ds.w : 0.f;
        dx[i] = o;
        dy[b * 2 * per_b + r] = o;
        dy[b * 2 * per_b + per_b + r] = k;
    }
}

// internal [B][C][TS] -> [B][T][C] (the reference's layout), 32 x 32 tiles through LDS
__global__ void k_fs_from_cm(const float* __restrict__ in, float* __restrict__ out, int C, int T, int TS) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;     // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, t = t0 + tx;
        tile[k][tx] = (c < C && t < TS) ? in[((size_t)b * C + c) * TS + t] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int t = t0 + k, c = c0 + tx;
        if (t < T && c < C) out[((size_t)b * T + t) * C + c] = tile[tx][k];
    }
}

// ------------------------------------------------------------------------------------------------------------
// PitchExtractor pieces (modules/fastspeech/pe.py, SURVEY section 8 row f2: mel -> f0 for the NSF vocoder)
// ------------------------------------------------------------------------------------------------------------
// y = (x * a[c] + b[c]) * keep[b][t]: BatchNorm1d in eval mode (alpha = gamma / sqrt(var + eps), beta' = beta - mean * alpha, what aten's
// batch_norm_cpu_transform_input evaluates) followed by Prenet's `* nonpadding_mask` (pe.py:33-35); zero tail.
__global__ void k_fs_affine(const float4* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ keep,
                            float4* __restrict__ y, int C, int T, int TS, size_t n4) {
    const int q = TS / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / q;                       // bb * C + c
        const int t = (int)(i - row * q) * 4;
        const int c = (int)(row % C);
        const size_t bb = row / C;
        const float al = a[c], be = b[c];
        const float4 v = x[i];
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r = 0.f;
            if (t + e < T) {
                r = o[e] * al + be;
                if (keep) r *= keep[bb * T + t + e];
            }
            o[e] = r;
        }
        y[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__device__ __forceinline__ float fs_block_sum(float v, float* red) {     // 256 threads; every thread gets the total
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// nn.GroupNorm(G, C) over [B][C][T] (statistics over the C/G channels x T frames of a group - ALL T frames, ConvStacks applies no
// mask, pe.py:98-108), then ReLU and the residual `x + x_` of ConvStacks.forward; one workgroup per (group, utterance); zero tail.
__global__ __launch_bounds__(256) void k_fs_group_norm(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ res, float* __restrict__ y, int C, int G, int T, int TS, float eps,
                                                      int relu) {
    __shared__ float red[4];
    const int g = blockIdx.x, bb = blockIdx.y, tid = threadIdx.x;
    const int cg = C / G;
    const size_t base = ((size_t)bb * C + (size_t)g * cg) * TS;
    const int n = cg * T;
    float s = 0.f;
    for (int i = tid; i < n; i += 256) { const int c = i / T, t = i - c * T; s += x[base + (size_t)c * TS + t]; }
    const float mean = fs_block_sum(s, red) / (float)n;
    float d = 0.f;
    for (int i = tid; i < n; i += 256) { const int c = i / T, t = i - c * T; const float e = x[base + (size_t)c * TS + t] - mean; d += e * e; }
    const float var = fs_block_sum(d, red) / (float)n;
    const float rstd = 1.f / sqrtf(var + eps);
    const int nn = cg * TS;
    for (int i = tid; i < nn; i += 256) {
        const int c = i / TS, t = i - c * TS;
        const size_t o = base + (size_t)c * TS + t;
        float v = 0.f;
        if (t < T) {
            const int ch = g * cg + c;
            v = (x[o] - mean) * rstd * gamma[ch] + beta[ch];
            if (relu) v = fmaxf(v, 0.f);
            if (res) v = res[o] + v;
        }
        y[o] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// optimiser step of the training path
// ------------------------------------------------------------------------------------------------------------
// torch.optim.AdamW (usr/diffspeech_task.py:40-46; amsgrad off) on ONE flat fp32 range - a rank's shard of the flattened parameters
// after the gradient reduce-scatter - in one pass: 16 B read per moment / parameter / gradient, 12 B written back per element.
//   g' = g * gscale[0]                   (1/world of the summed gradients times the clip_grad_norm_ coefficient; a DEVICE scalar: no host sync)
//   p *= decay ; m += (g' - m) (1 - b1) ; v = v b2 + ((1 - b2) g') g' ; p += -step_size * (m / (sqrt(v) / bc2_sqrt + eps))
struct AdamWParams {
    float* p; const float* g; float* m; float* v;
    const float* gscale;
    size_t n;
    float decay, one_minus_b1, b2, one_minus_b2, bc2_sqrt, eps, neg_step_size;
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamWParams& a, float gs) {
    g = g * gs;
    p = p * a.decay;
    m = m + (g - m) * a.one_minus_b1;
    v = v * a.b2 + (a.one_minus_b2 * g) * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p + a.neg_step_size * (m / denom);
}

__global__ __launch_bounds__(256) void k_adamw(const AdamWParams a) {
    const float gs = a.gscale ? a.gscale[0] : 1.f;
    const size_t n4 = a.n / 4;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        adamw_one(p.x, g.x, m.x, v.x, a, gs); adamw_one(p.y, g.y, m.y, v.y, a, gs);
        adamw_one(p.z, g.z, m.z, v.z, a, gs); adamw_one(p.w, g.w, m.w, v.w, a, gs);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {               // tail of a range that is not a multiple of 4
        const size_t i = n4 * 4 + threadIdx.x;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        adamw_one(p, a.g[i], m, v, a, gs);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
}

}  // namespace dsd
