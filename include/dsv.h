/* dsv.h - C ABI of the HiFi-GAN / NSF-HiFi-GAN generator ops in libdsdenoise.so (MI355X, gfx950).
 *
 * the step AFTER the diffusion hot path - mel [B,80,T] (+ f0 [B,T]) -> waveform [B,1,T*hop].
 * The reference (MoonInTheRiver/DiffSinger) computes it with torch nn modules (modules/hifigan/hifigan.py:104-179,
 * modules/parallel_wavegan/models/source.py:7-137, :484-531, called from vocoders/hifigan.py:55-69) and has no FFI; these are
 * the operators its modules would bind.  Every entry point names the reference code it replaces (paths relative to the
 * reference root).  Conventions as in dsd.h: fp32 device pointers, `stream` a hipStream_t as void*, work is only ENQUEUED,
 * 0 on success / negative dsd_status otherwise with the message in dsd_last_error().
 *
 * Activation layout ("channel-major"): [B][C][LS], sample axis contiguous, LS = dsv_padded_samples(L) (L rounded up to a
 * multiple of 32); every op writes ZERO to the samples [L, LS) and expects that of its inputs.  Stateless: the caller owns
 * every buffer. */
#ifndef DSV_H
#define DSV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSV_ACT_NONE 0
#define DSV_ACT_TANH 1      /* torch.tanh after conv_post, modules/hifigan/hifigan.py:167-168 */

int32_t dsv_padded_samples(int32_t L);

/* Replaces the parameter storage of a Conv1d / ConvTranspose1d AFTER remove_weight_norm() (hifigan.py:171-179): a
 * convolution weight [rows][Ci][K] repacked into the MFMA A-operand fragment order dsv_conv1d streams (rows padded to 32,
 * Ci to 8, with zeros).  dsv_packed_floats = floats the packed buffer must hold (or -1).
 * For ConvTranspose1d(Ci, Co, k, stride u, padding (k-u)/2) the caller passes the polyphase form: rows = Co * u,
 * row co * u + r holds the taps that reach output phase r (see dsv_conv1d). */
int64_t dsv_packed_floats(int32_t rows, int32_t Ci, int32_t K);
int dsv_pack_weight(const float* w, int32_t rows, int32_t Ci, int32_t K, float* packed, void* stream);

/* [R][L] contiguous rows -> [R][LS] padded rows with a zero tail (the mel [B,80,T] entering conv_pre, hifigan.py:151). */
int dsv_pad_rows(const float* in, float* out, int64_t R, int32_t L, void* stream);

/* One convolution of the generator with its element-wise neighbours fused - Conv1d / ConvTranspose1d of
 * HifiGanGenerator.forward (hifigan.py:144-169), ResBlock1.forward (:54-61), ResBlock2.forward (:82-87):
 *     y[row][q] = sum_ci sum_k  W[row][ci][k] * leaky_relu(in[ci][q + k * dil - pad], pre_slope)      (zero outside [0, L_in))
 *     co = row / up, phase = row % up, n = q * up + phase                                              (up = 1: plain Conv1d)
 *     v = y + bias[co] ; v += residual[co][n] ; v = sum_in[co][n] + v ; v = v / divide ; v = act(v)  -> out[co][n]
 * in [B][Ci][LS(L_in)]; out / residual / sum_in [B][rows / up][LS(L_in * up)]; bias [rows / up]; residual, sum_in, bias may be
 * NULL; pre_slope 1 = no activation in front, divide 1 = none.  Taps must stay within +-48 samples (pad <= 48 and
 * (K-1) * dil - pad <= 48: kernel 11 at dilation 5 reaches 25 on the shipped generators, the official v3's kernel 7 at dilation 12
 * reaches 36; beyond +-28 a second instantiation of the kernel with a wider staged window runs).
 * Covers: `leaky_relu -> convs1[i]`, `leaky_relu -> convs2[i] -> + x` (residual), the last conv of resblock j adding into
 * the running `xs` (sum_in) and the last one also doing `/ num_kernels` (divide), `leaky_relu -> ups[i] (+ x_source)`
 * (up = stride, residual = the noise_convs output), conv_pre, and `leaky_relu(0.01) -> conv_post -> tanh`. */
int dsv_conv1d(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t rows, int32_t K,
               int32_t pad, int32_t dil, int32_t L_in, int32_t up, float pre_slope, const float* residual, const float* sum_in,
               float divide, int32_t act, void* stream);

/* `ngroups` (1 .. 3) INDEPENDENT convolutions of the same shape (B, Ci, rows, L_in, up, pre_slope) in ONE launch (round 6) - the
 * convolutions of the parallel resblocks of a stage, level by level (hifigan.py:161-166: `resblocks[i * num_kernels + j](x)` for j = 0, 1, 2
 * read the same x and depend on nothing of each other until `xs +=`).  Convolution g is exactly dsv_conv1d(d[g].in, d[g].wpacked, ...,
 * d[g].act) - kernel size, padding, dilation, residual, running sum, divisor and activation are its own; the workgroups of d[0] are
 * dispatched first (pass the largest kernel first).  No output may be an operand of another convolution of the call.  On the 64-channel
 * stage of the shipped generator a convolution is one round of co-resident workgroups and a launch costs 15-19 us beyond its matrix time:
 * three convolutions per launch pay that once.  Same tiles, same chunk order as dsv_conv1d: bit-identical. */
typedef struct dsv_conv_desc {
    const float* in;
    const float* wpacked;
    const float* bias;
    float* out;
    const float* residual;
    const float* sum_in;
    int32_t K, pad, dil, act;
    float divide;
    int32_t reserved;
} dsv_conv_desc;
int dsv_conv1d_multi(int32_t ngroups, const dsv_conv_desc* d, int32_t B, int32_t Ci, int32_t rows, int32_t L_in, int32_t up, float pre_slope,
                     void* stream);

/* A/B switch of the measurement (round 6): the stride-2 transposed convolutions (up = 2, rows <= 32, taps within 4 samples) run on a lean build
 * of the same kernel - half the tile, LDS by the channel count, at most 128 registers: four and more workgroups per CU instead of two - the
 * same chunk order, the same bits; 0 = the standard build.  Process-wide, not thread-safe. */
int dsv_set_lean(int32_t on);

/* The same convolution (up = 1, 'same' padding pad = (K-1) * dil / 2, K odd) for the NARROW layers, Co <= 16 - the 16- and 8-channel
 * resblocks and conv_post of the shipped generator: F output samples are folded into the 32 MFMA rows so that no row multiplies
 * zeros.  dsv_fold_factor returns the F the library wants for such a layer (4: Co <= 8, 2: Co <= 16, 1: use dsv_conv1d; also 1
 * after dsv_set_fold(0) - the A/B switch of the measurement).  The caller packs, with
 * dsv_pack_weight(rows = Co * F, Ci, K + F - 1), the F shifted copies of the filter
 *     W'[co * F + e][ci][s] = w[co][ci][s - e]   (0 <= s - e < K, else 0)
 * and the kernel evaluates, with pos(c) = (c / dil) * F * dil + c % dil,
 *     out[co][pos(c) + e * dil] = sum_ci sum_s W'[co * F + e][ci][s] * leaky_relu(in[ci][pos(c) + s * dil - pad])   + the same fused tail
 * which is the convolution of dsv_conv1d sample for sample (every output sample is produced by exactly one (c, e)). */
int32_t dsv_fold_factor(int32_t Co, int32_t Ci, int32_t K, int32_t dil);
int dsv_set_fold(int32_t on);
int dsv_conv1d_folded(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t Co, int32_t K,
                      int32_t F, int32_t dil, int32_t L, float pre_slope, const float* residual, const float* sum_in, float divide,
                      int32_t act, void* stream);

/* Whole ResBlock1 chains in ONE launch, activations resident in LDS (csrc/voc_chain.hpp) - `xs = sum_r resblock_r(x)` of
 * HifiGanGenerator.forward (hifigan.py:161-166) with ResBlock1.forward (:54-61) inside:
 *     for r < nres:  y = x ; for q < npairs:  xt = conv[r][q][0](leaky_relu(y)) ; xt = conv[r][q][1](leaky_relu(xt)) ; y = xt + y
 *     out = (sum_in + y_0 + y_1 + ...) / divide            (sum_in may be NULL: the running `xs` of resblocks that ran before this call)
 * for C = 8, 16 or 32 channels (C * F == 32 with the fold F = dsv_chain_fold(C) = 4, 2, 1 of dsv_conv1d_folded).  in, out, sum_in:
 * [B][C][LS(L)], in != out, sum_in != out (out carries the running sum over the call's resblocks while it runs).  convs: HOST array [nres][npairs][2] - the first convolution of a pair at dilation `dil`, the second at 1
 * ('same' padding, K odd); w_offset = float offset of its packed weight inside `wpacked` (each piece = dsv_pack_weight(rows = 32,
 * Ci = C, K + F - 1) of the F shifted copies W'[co * F + e][ci][s] = w[co][ci][s - e], pieces at multiples of 256 floats, the buffer
 * ending with the slack dsv_packed_floats includes), bias_offset = float offset of its bias [C] inside `bias`.  A workgroup owns N output
 * samples plus a halo of the chain's receptive field; dsv_chain_supported returns that N (0: the chain does not fit the staged tile -
 * kernel / dilation too wide, too many convolutions - run it convolution by convolution with dsv_conv1d / dsv_conv1d_folded).  The
 * summation order of every output sample is the one of those operators: the results are bit-identical to theirs. */
typedef struct dsv_chain_conv {
    int64_t w_offset;
    int32_t bias_offset;
    int32_t K;
    int32_t dil;
    int32_t reserved;
} dsv_chain_conv;
int32_t dsv_chain_fold(int32_t C);
int32_t dsv_chain_supported(int32_t C, int32_t nres, int32_t npairs, const dsv_chain_conv* convs);
int dsv_resblock_chain(const float* in, const float* wpacked, const float* bias, float* out, const float* sum_in, int32_t B, int32_t C,
                       int32_t L, int32_t nres, int32_t npairs, const dsv_chain_conv* convs, float pre_slope, float divide, void* stream);
/* The parallel resblocks of a stage as TWO launches instead of one per resblock (round 6).  A chain launch of W workgroups on S co-resident
 * slots (256 CUs x 2 or 3) takes ceil(W / S) rounds, not W / S (profiles/r6_27_voc_tail_probe.jsonl): three dependent launches pay three
 * partial last rounds.  dsv_resblock_chain_multi runs `ngroups` (1 .. 3) INDEPENDENT single-resblock chains in one grid - group g =
 * convs[g][npairs][2] over the whole input, its raw y_g (no sum, no division, NOT zeroed in [L, LS)) to outs[g] (a HOST array of device
 * pointers, all different, none the input); the workgroups of group 0 are dispatched first: pass the longest chain first, so that the
 * short workgroups of a later group fill its last round.  dsv_resblock_chain_sum then runs ONE more resblock and forms the stage's result
 * in the order of `xs += resblock(x)` (hifigan.py:161-166): out = ((sum_in + y) + sum_in2) / divide - this resblock first or second of
 * three - or, own_last != 0, ((sum_in + sum_in2) + y) / divide; zero in [L, LS).  No workgroup waits for another; an ordinary kernel
 * boundary orders the two launches.  Same sums in the same order as dsv_resblock_chain: bit-identical.  Which resblock to leave for the
 * second launch is the caller's choice (diffsinger_amd/vocoder.py _merge_plan: the split with the fewest modelled rounds). */
int dsv_resblock_chain_multi(const float* in, const float* wpacked, const float* bias, float* const* outs, int32_t B, int32_t C, int32_t L,
                             int32_t ngroups, int32_t npairs, const dsv_chain_conv* convs, float pre_slope, void* stream);
int dsv_resblock_chain_sum(const float* in, const float* wpacked, const float* bias, float* out, const float* sum_in, const float* sum_in2,
                           int32_t own_last, int32_t B, int32_t C, int32_t L, int32_t npairs, const dsv_chain_conv* convs, float pre_slope,
                           float divide, void* stream);
/* A/B switch of the measurement: which instantiation of the chain kernel C = 8 / 16 / 32 channels run on - `nb` column blocks of 32 per wave
 * (a workgroup's window is 128 * nb * F samples; 2 or 4) and ONE LDS tile rewritten in place (in_place = 1: half the LDS, twice the workgroups
 * per CU or twice the window) or the two tiles of rounds 3-5 (in_place = 0).  Every variant evaluates the same sums in the same order: the
 * results do not depend on it.  dsv_chain_supported answers for the variant in force.  Process-wide, not thread-safe (a test / bench switch). */
int dsv_set_chain_variant(int32_t C, int32_t nb, int32_t in_place);
/* Measurement hook: DEVICE buffer of 18 * 4 * 4 uint64 (or NULL to switch it off) - the following dsv_resblock_chain launches record the
 * shader clock of ONE workgroup (the middle tile of utterance 0) per convolution n and wave w at [n][w][0..3] = {convolution start,
 * contraction done, epilogue done, barrier passed}. */
int dsv_debug_chain_timeline(uint64_t* device_stamps);

/* noise_convs[i] (hifigan.py:124-130, :158-160): the strided Conv1d(1 -> C, kernel K, stride, padding) over the harmonic
 * source.  har [B][LS(L_har)], w [C][K] (the torch weight [C][1][K]), bias [C] or NULL, out [B][C][LS(L_out)];
 * L_out must equal (L_har + 2 * pad - K) / stride + 1. */
int dsv_noise_conv(const float* har, const float* w, const float* bias, float* out, int32_t B, int32_t C, int32_t K, int32_t stride,
                   int32_t pad, int32_t L_har, int32_t L_out, void* stream);

/* torch.nn.Upsample(scale_factor = up) of f0 + SourceModuleHnNSF.forward (source.py:518-531) = SineGen.forward (:101-137,
 * _f02sine :45-77, flag_for_pulse False) -> Linear(H, 1) -> tanh.  The module's random draws are INPUTS (the caller draws them
 * with whatever generator it must match): rand_ini [B][H] uniform [0,1) (column 0 is ignored), noise [B][L][H] standard
 * normal, L = T * up.  f0 [B][T] Hz (<= voiced_threshold = unvoiced), lin_w [H], lin_b [1]; sines_ws: workspace of
 * B * H * L floats; har [B][LS(L)] = the merged harmonic source (the module's `noise` output is never used by the generator).
 * Both cumulative sums over the sample axis accumulate in fp64, as aten's CPU cumsum does for float tensors. */
int dsv_sine_source(const float* f0, const float* rand_ini, const float* noise, const float* lin_w, const float* lin_b, float* sines_ws,
                    float* har, int32_t B, int32_t T, int32_t up, int32_t H, float sample_rate, float sine_amp, float noise_std,
                    float voiced_threshold, void* stream);

/* ParallelWaveGAN generator (vocoders/pwg.py; modules/parallel_wavegan/models/parallel_wavegan.py:21-177) - residual_channels 64, gate_channels
 * 128, skip_channels 64, kernel_size 3 (the configuration the reference ships and trains); activations [B][C][LS(L)] like the HiFi-GAN ops.
 * dsv_pwg_first: first_conv, Conv1d1x1(1, C) on the noise z [B][LS]: out[b][c][t] = w[c] z[b][t] + bias[c].
 * dsv_pwg_upsample: one stage of UpsampleNetwork (layers/upsample.py:96-117) on `rows` = B * C rows: nearest-neighbour stretch by `scale`, then the
 *   (2 scale + 1)-tap filter the Conv2d(1, 1, (1, 2 scale + 1), padding (0, scale), bias=False) shares between all rows.
 * dsv_pwg_layer: one ResidualBlock (layers/residual_block.py:96-129):
 *     a = conv(x; kernel 3, dilation dil, 'same') + b1 + conv1x1_aux(c);  z = tanh(a[0:64]) * sigmoid(a[64:128])
 *     x_out = (conv1x1_out(z) + x) * sqrt(0.5);   skip = (first ? 0 : skip) + conv1x1_skip(z)
 *   w1_packed = dsv_pack_weight of the [128][3 * 64 + n_aux][1] matrix whose columns are tap * 64 + ci (tap 0 reads t - dil) followed by the aux
 *   channels; w2_packed = dsv_pack_weight of [128][64][1], rows 0..63 conv1x1_out, 64..127 conv1x1_skip; b1 [128] / b2 [128] may be NULL.
 *   Any dilation (the generator's run to 512 samples); x_out must not alias x. */
int dsv_pwg_first(const float* z, const float* w, const float* bias, float* out, int32_t B, int32_t C, int32_t L, void* stream);
int dsv_pwg_upsample(const float* in, const float* filter, float* out, int64_t rows, int32_t L_in, int32_t scale, void* stream);
int dsv_pwg_layer(const float* x, const float* c, const float* w1_packed, const float* b1, const float* w2_packed, const float* b2, float* x_out,
                  float* skip, int32_t B, int32_t L, int32_t n_aux, int32_t dil, int32_t first, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSV_H */
