/* dsf.h - C ABI of the FastSpeech2 / FastSpeech2MIDI conditioner ops in libdsdenoise.so (MI355X, gfx950).
 *
 * the step BEFORE the diffusion hot path - it produces `cond` (decoder_inp) and the aux-decoder
 * mel the shallow diffusion starts from.  The reference (MoonInTheRiver/DiffSinger) computes it with torch nn modules and
 * has no FFI; these are the operators its modules would bind.  Every entry point names the reference code it replaces
 * (paths relative to the reference root).  Conventions as in dsd.h: fp32 device pointers, `stream` a hipStream_t as void*,
 * work is only ENQUEUED, 0 on success / negative dsd_status otherwise with the message in dsd_last_error().
 *
 * Internal activation layout ("channel-major"): [B][C][TS], frame axis contiguous, TS = dsf_padded_frames(T) (T rounded
 * up to a multiple of 32); every op writes ZERO to the frames [T, TS).  Stateless: the caller owns every buffer. */
#ifndef DSF_H
#define DSF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activation codes of dsf_conv1d */
#define DSF_ACT_NONE 0
#define DSF_ACT_RELU 1
#define DSF_ACT_GELU 2      /* erf form, F.gelu default (modules/commons/common_layers.py:512-513) */
#define DSF_ACT_MISH 3      /* x * tanh(softplus(x)), usr/diff/diffusion.py:68-70 (the step MLP of the FFT candidate denoiser) */

int32_t dsf_padded_frames(int32_t T);

/* Replaces nn.Conv1d / nn.Linear parameter storage: a torch weight [Co][Ci][K] (Linear: K = 1) repacked into the MFMA
 * A-operand fragment order the conv kernel streams.  dsf_packed_floats = floats the packed buffer must hold (or -1). */
int64_t dsf_packed_floats(int32_t Co, int32_t Ci, int32_t K);
int dsf_pack_weight(const float* w, int32_t Co, int32_t Ci, int32_t K, float* packed, void* stream);

/* nn.Conv1d (stride 1, odd kernel K <= 17, 'SAME' zero padding) / nn.Linear with the element-wise tail fused:
 *     y = act(scale * (W * x + bias)) ; y += residual ; y *= keep[b][t]
 * covering MultiheadAttention's in/out projections (modules/commons/common_layers.py:243-263), TransformerFFNLayer
 * (ffn_1 * K**-0.5 -> gelu, ffn_2; :486-522), the residual + padding mask of EncSALayer (:565-588), the predictor
 * Conv1d + ReLU (modules/fastspeech/tts_modules.py:84-97, :198-209) and mel_out (modules/fastspeech/fs2.py:233-237).
 * in [B][Ci][TS], out / residual [B][Co][TS], bias [Co] or NULL, keep [B][T] (1 valid / 0 padding) or NULL.
 * Ci must be a multiple of 8.  `in` must be zero in [T, TS) (every dsf op guarantees that for its output). */
int dsf_conv1d(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t Co, int32_t K,
               int32_t T, float scale, int32_t act, const float* residual, const float* keep, void* stream);

/* nn.LayerNorm over the channel axis (C = 256 on this build): EncSALayer.layer_norm1/2 and FFTBlocks.layer_norm (eps 1e-5,
 * common_layers.py:70-77, tts_modules.py:276-281,:306-307), the predictor LayerNorm(dim=1) (eps 1e-12, tts_modules.py:39-56)
 * with the preceding ReLU (relu_in = 1).  out = LN(in) * gamma + beta, times keep[b][t] if given. */
int dsf_layer_norm(const float* in, const float* gamma, const float* beta, float* out, int32_t B, int32_t C, int32_t T, float eps,
                   int32_t relu_in, const float* keep, void* stream);

/* The attention core of F.multi_head_attention_forward as MultiheadAttention.forward calls it (common_layers.py:243-263):
 * per head softmax((q * head_dim**-0.5) k^T + key_padding_mask) v.  qkv [B][3C][TS] (q | k | v rows, head h = rows
 * [h*128, h*128+128) of each), key_pad [B][T] bytes (nonzero = padded key) or NULL, out [B][C][TS].  head_dim 128. */
int dsf_attention(const float* qkv, const uint8_t* key_pad, float* out, int32_t B, int32_t C, int32_t heads, int32_t T, void* stream);

/* BACKWARD of the two operators above - what torch autograd runs for nn.LayerNorm and F.multi_head_attention_forward when FastSpeech2 is trained
 * (the Opencpop e2e configuration trains it jointly with the denoiser: usr/diffsinger_task.py:60-64, :273-300; csrc/fs2_train.hpp).
 * dsf_layer_norm_bwd: x = the forward INPUT, dy = gradient wrt the forward output (times keep inside, like the forward) -> dx [B][C][TS]
 * (zero tail), dgamma [C], dbeta [C] (overwritten; per-tile partial sums in ws, added in a fixed order: deterministic);
 * ws: dsf_ln_bwd_workspace_floats(B, T) floats.
 * dsf_attention_bwd: dout [B][C][TS] -> dqkv [B][3C][TS] (zero tail); the probabilities are recomputed ([B heads][T][T], twice, in ws:
 * dsf_attention_bwd_workspace_floats(B, heads, T) floats) - the attention of the model that is trained runs at the phone rate.
 * LIMITS: the workspace is QUADRATIC in T (2 x B x heads x T x T floats: 33 MB at the phone rate of the e2e step, B = 8, T = 128 - but
 * 0.7-2 GB per layer for a mel-rate decoder, T = 1000-1500 with B in the tens: plain FastSpeech2 training at the mel rate should go layer by layer
 * with a reused workspace, or tile the query axis above this call); B x heads <= 65535 per call (DSD_ERR_INVALID beyond, split the batch). */
int64_t dsf_ln_bwd_workspace_floats(int32_t B, int32_t T);
int dsf_layer_norm_bwd(const float* x, const float* gamma, const float* dy, const float* keep, float* dx, float* dgamma, float* dbeta, float* ws,
                       int32_t B, int32_t C, int32_t T, float eps, int32_t relu_in, void* stream);
int64_t dsf_attention_bwd_workspace_floats(int32_t B, int32_t heads, int32_t T);
int dsf_attention_bwd(const float* qkv, const uint8_t* key_pad, const float* dout, float* dqkv, float* ws, int32_t B, int32_t C, int32_t heads,
                      int32_t T, void* stream);

/* torch.nn.Linear on a handful of rows - the step-embedding MLP and the layers' step projections of the denoiser under training
 * (usr/diff/net.py:94-98, :119-120, :67): y [rows][n_out] = x [rows][n_in] w[n_out][n_in]^T + bias (bias may be NULL), and the gradients
 * (each output may be NULL): dx = dy w, dw = dy^T x, db = column sums of dy in row order.  Plain row-major fp32, no packing.
 * ws: NULL, or dsf_linear_rows_workspace_floats(rows, n_in, n_out) floats - with it a long contraction behind few outputs is split over K
 * and the partial products are added in a fixed order. */
int64_t dsf_linear_rows_workspace_floats(int32_t rows, int32_t n_in, int32_t n_out);
int dsf_linear_rows(const float* x, const float* w, const float* bias, float* y, float* ws, int32_t rows, int32_t n_in, int32_t n_out, void* stream);
int dsf_linear_rows_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, float* ws, int32_t rows, int32_t n_in,
                        int32_t n_out, void* stream);

/* Layout changes at the boundary: the reference's [B,T,C] tensors (any element strides) <-> channel-major. */
int dsf_to_channel_major(const float* x, int64_t stride_b, int64_t stride_c, int64_t stride_t, float* out, int32_t B, int32_t C,
                         int32_t T, void* stream);
int dsf_from_channel_major(const float* in, float* out /* [B][T][C] contiguous */, int32_t B, int32_t C, int32_t T, void* stream);

/* The index / mask glue of FastSpeech2.forward as four operators (round 6).  Every value is the one the reference's tensor ops produce -
 * the same operations in the same order, one rounding each; positions, indices and masks are integers.
 * dsf_positions      utils/__init__.py:145-157 make_positions along the frame axis: pos = cumsum(v != pad) * (v != pad) + pad, v = the int64
 *                    tokens [B][T] (FastspeechEncoder.forward_embedding, tts_modules.py:338-346) or, tokens NULL, channel 0 of the float
 *                    tensor x [B][T][C] (FFTBlocks.forward: embed_positions(x[..., 0]), tts_modules.py:291).  pos [B][T] int32.
 * dsf_input_cm       the front end of FFTBlocks.forward (tts_modules.py:288-296) and of FastspeechEncoder / FastspeechMIDIEncoder
 *                    .forward_embedding: value = emb_scale * emb[token] (+ add0 + add1 + add2, in this order; each [B][T][C] or NULL)  -  or the
 *                    tensor x [B][T][C] (tokens NULL)  -  (+ pos_table[pos], times *alpha_dev when that DEVICE scalar is given); padding =
 *                    padding_mask [B][T] (nonzero = padded) if given, else token == padding_idx, else "every channel of x's frame is zero"
 *                    (x.abs().sum(-1).eq(0)); xc [B][C][TS] = value * (1 - padding) channel-major (zero in [T,TS)), keep [B][T] = 1 - padding,
 *                    pad_out [B][T] u8 = padding (the key_padding_mask of dsf_attention); mask_mode 0: no mask at all (keep = 1, pad_out = 0 - the
 *                    front end of PitchPredictor.forward, tts_modules.py:215-229).  C a multiple of 4, <= 512.
 * dsf_gather_frames  fs2.py:128-134: out [B][T][C] = gather(pad(enc [B][T_src][C]), mel2ph) (mel2ph [B][T] int64: 0 = padding frame, k = phone
 *                    k - 1) and, if out_masked is given, out_masked = (out + spk [B][C] or 0) * (mel2ph > 0).
 * dsf_sum_embed      fs2.py:136-141: out = (((dec + tab1[idx1] or add1) + tab2[idx2]) + spk or 0) * (mel2ph > 0); idx [B][T] int64, tab [n][C];
 *                    every optional operand may be NULL. */
int dsf_positions(const int64_t* tokens, const float* x, int32_t* pos, int32_t B, int32_t T, int32_t C, int32_t padding_idx, void* stream);
int dsf_input_cm(const int64_t* tokens, const float* emb, float emb_scale, const float* add0, const float* add1, const float* add2, const float* x,
                 const int32_t* pos, const float* pos_table, const float* alpha_dev, const uint8_t* padding_mask, float* xc, float* keep,
                 uint8_t* pad_out, int32_t B, int32_t T, int32_t C, int32_t padding_idx, int32_t mask_mode, void* stream);
int dsf_gather_frames(const float* enc, const int64_t* mel2ph, const float* spk, float* out, float* out_masked, int32_t B, int32_t T, int32_t T_src,
                      int32_t C, void* stream);
int dsf_sum_embed(const float* dec, const int64_t* idx1, const float* tab1, const float* add1, const int64_t* idx2, const float* tab2, const float* spk,
                  const int64_t* mel2ph, float* out, int32_t B, int32_t T, int32_t C, void* stream);

/* The rest of the forward's glue (round 6, second half; 35 torch launches of 2-6 us were left between the encoder and the decoder).
 * dsf_token_masks    the masks derived from an int64 index tensor v [n] (txt_tokens, mel2ph): gt0 = (v > 0) as float (fs2.py:98, :127),
 *                    eq0 = (v == 0) as u8 / bool (fs2.py:157, :199), ne0 = (~(v == 0)) as float (DurationPredictor.forward,
 *                    tts_modules.py:109-118); every output optional (NULL), one at least.
 * dsf_pitch_coarse   utils/pitch_utils.py:64-77 denorm_f0 followed by :21-30 f0_to_coarse on f0 [B][T] (element strides stride_b, stride_t):
 *                        d = f0 * f0_std + f0_mean (norm 1: pitch_norm 'standard') | 2 ** f0 (norm 2: 'log');   d = 0 where uv > 0 (uv_f float
 *                        or uv_u8 bool, [B][T] contiguous, at most one of them) or mel2ph == 0 (each optional)           -> f0_denorm [B][T]
 *                        m = 1127 * log(1 + d / 700);  m = (m - f0_mel_min) * (f0_bin - 2) / (f0_mel_max - f0_mel_min) + 1 where m > 0;
 *                        m = 1 where m <= 1;  m = f0_bin - 1 where m > f0_bin - 1;  coarse = int64(m + 0.5)                 -> coarse [B][T]
 *                    every operation in fp32 in this order, one rounding each, a division by a scalar as the multiplication by its fp32
 *                    reciprocal - the values of the reference's tensor ops on the device.  stage 0: all of it.  stage 1: up to tmp = 1 + d / 700
 *                    (writes f0_denorm and tmp); stage 2: from tmp = the caller's log(tmp) on (writes coarse); pow_in (norm 2): f0 already
 *                    holds 2 ** f0 - the split forms for a host whose tensor library's log / pow round differently from this library's. */
int dsf_token_masks(const int64_t* v, float* gt0, uint8_t* eq0, float* ne0, int64_t n, void* stream);
int dsf_pitch_coarse(const float* f0, int64_t stride_b, int64_t stride_t, const float* uv_f, const uint8_t* uv_u8, const int64_t* mel2ph,
                     float* f0_denorm, float* tmp, int64_t* coarse, int32_t B, int32_t T, int32_t norm, float f0_mean, float f0_std,
                     double f0_mel_min, double f0_mel_max, int32_t f0_bin, int32_t stage, int32_t pow_in, void* stream);

/* The element-wise ends of GaussianDiffusion.p_losses (usr/diff/shallow_diffusion_tts.py:206-231) around the denoiser's training forward (round 6).
 * dsf_q_sample_rows  :206-211 with a step per utterance: out[b] = sqrt_ac[t[b]] * x_start[b] + sqrt_1mac[t[b]] * noise[b]; x_start / noise / out
 *                    [B][per_row] contiguous, t [B] int64 on the device, the two schedule tables on the device; two products and a sum, each
 *                    rounded once (the values of the tensor expression).  per_row a multiple of 4; a step outside [0, n_steps) (the tables' length) gives a row of
 *                    NaN (the reference's gather raises).
 * dsf_l1_mean        :224-228 (loss_type 'l1', no mask): out[0] = mean |a - b| over n floats, summed in a fixed order (deterministic);
 *                    workspace of dsf_l1_workspace_floats() floats.
 * dsf_l1_mean_bwd    its gradient with respect to b: db = -(sign(a - b) * (grad_out[0] / n)), grad_out a DEVICE scalar. */
int dsf_q_sample_rows(const float* x_start, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac, int32_t n_steps,
                      float* out, int32_t B, int64_t per_row, void* stream);
int64_t dsf_l1_workspace_floats(void);
int dsf_l1_mean(const float* a, const float* b, float* workspace, float* out, int64_t n, void* stream);
int dsf_l1_mean_bwd(const float* a, const float* b, const float* grad_out, float* db, int64_t n, void* stream);

/* dsf_conv1d / dsf_conv1d_dilated pick their kernel by grid size: launches with at most one workgroup for every second CU (the phone-rate
 * encoder, everything of a single utterance) run 64-row workgroups whose waves split the contraction (k_fs_conv_ks; partial sums added in a
 * fixed order - results differ from the other kernel by summation order only).  mode: -1 by grid size (default), 0 never, 1 wherever the
 * shape allows it (Ci a multiple of 32).  Process-wide; for tests and A/B measurements. */
int dsf_set_conv_split(int32_t mode);

/* The same convolution with a dilation (kernel K odd, dil * (K-1)/2 <= 8): the DiffNet dilated_conv (usr/diff/net.py:62) as a
 * stand-alone operator of the TRAINING path, and the data gradient of any of these convolutions (= the convolution with the
 * flipped, transposed weight). */
int dsf_conv1d_dilated(const float* in, const float* wpacked, const float* bias, float* out, int32_t B, int32_t Ci, int32_t Co, int32_t K,
                       int32_t dil, int32_t T, void* stream);

/* Training: what autograd's conv1d / linear backward computes for the weight and the bias
 * (torch/csrc/autograd: ConvolutionBackward, AddmmBackward) -
 *     dw[co][ci][k] (+)= sum_b sum_t dy[b][co][t] * x[b][ci][t + k * dil - dil * (K-1)/2],   db[c] (+)= sum_b sum_t dy[b][c][t]
 * dy [B][Co][TS], x [B][Ci][TS] channel-major (zero in [T,TS)), dw in torch layout [Co][Ci][K] (K = 1 or 3), workspace of
 * dsf_wgrad_workspace_floats(Co, Ci, K) floats (split-K partials, reduced in a fixed order: deterministic). */
int64_t dsf_wgrad_workspace_floats(int32_t Co, int32_t Ci, int32_t K);
int dsf_conv1d_wgrad(const float* dy, const float* x, float* dw, float* db /* [Co] or NULL: fused bias gradient */, float* workspace, int32_t B,
                     int32_t Ci, int32_t Co, int32_t K, int32_t dil, int32_t T, int32_t accumulate, void* stream);
int dsf_bias_grad(const float* dy, float* db, int32_t B, int32_t C, int32_t T, int32_t accumulate, void* stream);

/* Element-wise pieces of ResidualBlock.forward (usr/diff/net.py:66-78) of the training path, forward and backward, on channel-major
 * [B][C][TS] tensors (frames >= T are written as zero):
 *   add_step      y = x + step[b][c]                                            (:69)   backward: dsf_train_rowsum over the frames
 *   gate          g = sigmoid(a[:, :C]) * tanh(a[:, C:]),  a [B][2C][TS]          (:73-74) backward: da from dg and the saved a
 *   res_skip      x' = (x + y[:, :C]) / sqrt(2),  skip' = skip + y[:, C:]         (:76-78, :121-126; skip NULL = first layer)
 *                 backward: dx = dx'/sqrt(2), dy = [dx'/sqrt(2) ; dskip'] */
int dsf_train_add_step(const float* x, const float* step, float* y, int32_t B, int32_t C, int32_t T, void* stream);
int dsf_train_rowsum(const float* g, float* out, int32_t rows, int32_t T, void* stream);
int dsf_train_gate(const float* a, float* g, int32_t B, int32_t C, int32_t T, void* stream);
int dsf_train_gate_bwd(const float* a, const float* dg, float* da, int32_t B, int32_t C, int32_t T, void* stream);
int dsf_train_res_skip(const float* x, const float* y, const float* skip, float* x_out, float* skip_out, int32_t B, int32_t C, int32_t T, void* stream);
int dsf_train_res_skip_bwd(const float* dx_out, const float* dskip_out, float* dx, float* dy, int32_t B, int32_t C, int32_t T, void* stream);

/* Sampler pieces for a denoise_fn that is not the fused DiffNet (the `FFT` candidate decoder, usr/diff/candidate_decoder.py:35-96;
 * SURVEY section 8 row f4): one p_sample update (usr/diff/shallow_diffusion_tts.py:134-166: x0 = a x - b eps, clamp, posterior
 * mean, + sigma z) in place on n contiguous floats with the fp32 table entries of step t passed by the host, and denorm_spec
 * (:281-282) fused with the [B][M][T] -> [B,T,M] transpose and the optional `mel2ph > 0` mask (:271-273); spec_min/max: DEVICE [M]. */
int dsf_p_sample(float* x, const float* eps, const float* noise, int64_t n, float sqrt_recip_ac, float sqrt_recipm1_ac, float coef1,
                 float coef2, float sigma, void* stream);
int dsf_denorm_spec(const float* x, const float* mask, float* mel, const float* spec_min, const float* spec_max, int32_t B, int32_t M,
                    int32_t T, void* stream);

/* Training, the optimiser step: torch.optim.AdamW as the tasks build it (usr/diffspeech_task.py:40-46, tasks/tts/tts.py:43-49; amsgrad off)
 * on ONE flat, 16-byte aligned fp32 range of n elements - the whole flattened parameter vector, or a rank's shard of it after the
 * gradient reduce-scatter (diffsinger_amd/train_dist.py) - fused into one pass.  `step` is the 1-based step count (bias corrections
 * 1 - beta^step are formed in double like torch does); grad_scale: DEVICE pointer to one float the gradient is multiplied by first
 * (1/world * the clip_grad_norm_ coefficient, utils/pl_utils.py:1165-1168) or NULL. */
int dsf_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step, const float* grad_scale, void* stream);

/* Training, the FUSED residual stack: the 20 ResidualBlocks of DiffNet.forward (usr/diff/net.py:66-78, :121-126)
 * forward and backward as what torch autograd would run for them, for residual_channels = encoder_hidden = 256 and L <= 32 layers.
 *   forward   x0 [B][256][TS] (relu(input_projection), net.py:116-118), cond [B][256][TS], step [B][L][256] (diffusion_projection_l of the
 *             step embedding, :67) -> skip [B][256][TS] = sum_l skip_l (the tensor :126 divides by sqrt(L)).  One conditioner-projection
 *             launch for all layers + the layers as ONE persistent launch per chunk of whole utterances (the tile ownership and
 *             neighbour exchange of the inference loop, csrc/train_loop.hpp) when an utterance fits one workgroup per CU and the chunks
 *             fill the chip, else the inference layer kernel per layer (dsf_set_stack_mode picks; bit-identical with dsf_set_stack_conv(0));
 *             either saves y = x + step and the gate pre-activation of every layer into `save_ws` for the backward pass.  A neighbour
 *             wait that hits its (seconds-long) bound poisons `skip` with NaN instead of hanging.
 *   backward  dskip [B][256][TS] -> dx0, dstep [B][L][256], every weight / bias gradient of the stack (torch layouts, OVERWRITTEN),
 *             per layer: output-projection data gradient + gate derivative, transposed dilated conv + residual path, and ONE launch for
 *             the layer's three weight gradients (contraction over frames, split-K partials reduced in a fixed order: deterministic).
 *             da_all: NULL, or [B][L*512][TS] to keep every layer's gradient wrt the gate pre-activation (rows [512 l, 512 l + 512) of an
 *             utterance) - the operand of the conditioner gradient dcond = sum_l Wc_l^T da_l, one dsf_conv1d over 512 L input channels.
 * The weight tables are HOST arrays of L device pointers in torch layouts: dilated_conv [512][256][3], conditioner / output projection
 * [512][256][1], biases [512].  Workspaces are caller-owned: dsf_stack_workspace_floats(B, T, L, 0) floats for save_ws (written by forward,
 * read by backward), (.., 1) for the backward scratch; contents are private to the library (dsf_stack_offsets: test access). */
typedef struct dsf_stack_weights {
    const float* const* dilated_conv_w; const float* const* dilated_conv_b;
    const float* const* cond_w;         const float* const* cond_b;
    const float* const* out_w;          const float* const* out_b;
    const int32_t* dilations;           /* [L], each in {1, 2, 4, 8} */
} dsf_stack_weights;
typedef struct dsf_stack_grads {
    float* const* dilated_conv_w; float* const* dilated_conv_b;
    float* const* cond_w;         float* const* cond_b;
    float* const* out_w;          float* const* out_b;
    float* dx0;                   /* [B][256][TS] */
    float* dstep;                 /* [B][L][256] */
} dsf_stack_grads;
int64_t dsf_stack_workspace_floats(int32_t B, int32_t T, int32_t L, int32_t which);
/* How dsf_stack_forward launches the layers (process-wide): 1 automatic (default: one persistent launch per chunk of whole utterances where
 * that fills the chip, else one launch per layer), 0 per-layer launches always (the A/B switch of the measurement), 2 persistent wherever an
 * utterance fits the co-resident grid (tests of the chunked form).  The workspace sizes do not depend on it. */
int dsf_set_stack_mode(int32_t mode);
/* The dilated convolution of the persistent forward (process-wide): 1 (default) Winograd F(2,3) along the frame axis - the inference loop's
 * form (include/dsd.h dsd_set_conv_mode; csrc/train_loop_wino.hpp): exact-fp32 MFMA on 2/3 of the multiplications, used where every dilation
 * is 1, 2, 4 or 8, results within the transforms' roundings of the direct form; 0 the direct form (csrc/train_loop.hpp: bit-identical to the
 * per-layer launches).  The saved tensors keep their layouts; the workspace sizes do not depend on it. */
int dsf_set_stack_conv(int32_t mode);
int dsf_get_stack_conv(void);
/* The weight gradient of the dilated convolution inside dsf_stack_backward (process-wide): 1 (default; in force where the stack's convolution
 * runs as Winograd, see dsf_set_stack_conv) the Winograd F(2,3) DUAL - for a frame pair (tE, tO = tE + d) with e = da[tE], f = da[tO] and
 * d0..d3 = y[tE - d], y[tE], y[tO], y[tO + d] the three tap gradients are Q0 + Q1 + Q2, Q1 - Q2, Q1 + Q2 + Q3 of the FOUR products
 * Q0 = e (d0 - d2), Q1 = (e + f)(d1 + d2) / 2, Q2 = (e - f)(d2 - d1) / 2, Q3 = f (d3 - d1) contracted over the pairs: 2/3 of the multiplications of
 * usr/diff/net.py:61's weight gradient, fp32 throughout (one add per operand, exact-fp32 MFMA), within the tests' 3e-6 of float64 autograd;
 * 0 the three tap products over the frames (rounds 2-5).  The workspace sizes do not depend on it. */
int dsf_set_wgrad_dual(int32_t on);
/* Developer hook (tools/trb_timeline.py): s_memtime stamps of the Winograd data-gradient kernel of layer 1, [workgroup][wave 4][8] uint64 per launch
 * (NULL switches it off).  Not part of the operator surface. */
int dsf_debug_trb_timeline(uint64_t* device_stamps);
int dsf_stack_offsets(int32_t B, int32_t T, int32_t L, int32_t which, int64_t* out, int32_t n);
int dsf_stack_forward(const float* x0, const float* cond, const float* step, const dsf_stack_weights* w, int32_t B, int32_t T, int32_t L,
                      float* save_ws, float* skip_out, void* stream);
int dsf_stack_backward(const float* dskip, const float* cond, const dsf_stack_weights* w, int32_t B, int32_t T, int32_t L, const float* save_ws,
                       float* bwd_ws, const dsf_stack_grads* grads, float* da_all, void* stream);

/* Measurement hook of the fused stack's dominant kernel: with the probe on, every weight-gradient launch of dsf_stack_backward is bracketed by
 * two events on its stream; ..._read waits for them and returns the summed kernel time, the number of launches and their algorithmic FLOP
 * (2 x 128 x 256 x tiles x B x T), then resets.  bench.py --row train reports its roofline from this. */
int dsf_wgrad_probe(int32_t on);
int dsf_wgrad_probe_read(double* total_ms, int64_t* launches, double* flops);

/* The weight-gradient kernel of the fused stack as a stand-alone operator with dsf_conv1d_wgrad's contract (K = 1 or 3, Co a multiple of
 * 128, Ci a multiple of 256): dw[co][ci][k] = sum dy[b][co][t] x[b][ci][t + (k - (K-1)/2) dil], db[co] = sum dy.  workspace:
 * dsf_wgrad2_workspace_floats(Co, Ci, K) floats. */
int64_t dsf_wgrad2_workspace_floats(int32_t Co, int32_t Ci, int32_t K);
int dsf_conv1d_wgrad2(const float* dy, const float* x, float* dw, float* db, float* workspace, int32_t B, int32_t Ci, int32_t Co, int32_t K,
                      int32_t dil, int32_t T, void* stream);

/* PitchExtractor (modules/fastspeech/pe.py:119-148; SURVEY section 8 row f2: mel -> f0 for the NSF vocoder) beyond the operators above:
 *   channel_affine  y = (x * a[c] + b[c]) * keep[b][t]: nn.BatchNorm1d in eval mode folded to a = gamma / sqrt(var + eps),
 *                   b = beta - mean * a, and Prenet's `* nonpadding_mask` (pe.py:12-17, :33-35); keep may be NULL
 *   group_norm      nn.GroupNorm(groups, C) of ConvBlock (pe.py:55-56, :72-75; statistics over C/groups channels x all T frames), then
 *                   ReLU (relu = 1) and ConvStacks' residual `x + x_` (pe.py:105-106; residual may be NULL)
 * x, y, residual channel-major [B][C][TS]; a, b, gamma, beta [C]. */
int dsf_channel_affine(const float* x, const float* a, const float* b, const float* keep, float* y, int32_t B, int32_t C, int32_t T, void* stream);
int dsf_group_norm(const float* x, const float* gamma, const float* beta, const float* residual, float* y, int32_t B, int32_t C, int32_t groups,
                   int32_t T, float eps, int32_t relu, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSF_H */
