/* dsd.h - C ABI of libdsdenoise.so: the DiffSinger/DiffSpeech diffusion-denoiser hot path on MI355X (gfx950).
 *
 * The reference (MoonInTheRiver/DiffSinger) is pure Python/PyTorch and has no FFI of its own; this is the
 * boundary its hot path would bind if it had one.  Every entry point names the reference code it replaces
 * (paths relative to the reference root).  Plain pointers and sizes only - no torch types.
 *
 * Conventions
 *   - all tensors fp32, row-major contiguous unless strides are given; device pointers unless marked HOST
 *   - "spec" tensors are [B][M][T] (the reference's [B,1,M,T] with the unit dim dropped), T innermost
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every data-path call only ENQUEUES work
 *     on it and returns; results are ordered like any other stream work.  The exceptions, all outside the steady state:
 *     dsd_prepare synchronises the stream when (and only when) the workspace has to GROW (more tiles / utterances than any
 *     batch before: the old buffers may still be in use), the first call that needs a larger step table does (one-time
 *     table build), and the calls documented as measurement / debug hooks do.  Small HOST arrays (dsd_denoise's t[B]) are
 *     copied into a pinned staging ring inside the call: the caller's array is not referenced after return and nothing waits
 *   - return 0 on success, a negative dsd_status otherwise; dsd_last_error() gives the message (thread-local)
 *   - ownership: the caller owns every tensor it passes; the handle owns only its packed weights, tables,
 *     workspace and cached hipGraphs, all released by dsd_destroy()
 *   - threading: one handle = one device = one caller at a time (not re-entrant); different handles are
 *     independent (one per DP thread / DDP process: utils/pl_utils.py:81-164, :570)
 */
#ifndef DSD_H
#define DSD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 6): dsd_get_conv_mode also reports the Winograd conv node of the latency kernels; dsv_set_chain_variant; dsv_resblock_chain refuses
 * sum_in == out; dsf_positions / dsf_input_cm / dsf_gather_frames / dsf_sum_embed / dsf_token_masks / dsf_pitch_coarse / dsf_q_sample_rows / dsf_l1_mean / dsf_l1_mean_bwd; dsf_set_wgrad_dual; dsv_resblock_chain_multi / dsv_resblock_chain_sum / dsv_conv1d_multi / dsv_set_lean.  5 (round 5): dsd_set_conv_mode. */
#define DSD_ABI_VERSION 6

typedef struct dsd_handle dsd_handle;

typedef enum dsd_status {
    DSD_OK = 0,
    DSD_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    DSD_ERR_HIP = -2,          /* a HIP runtime call failed */
    DSD_ERR_STATE = -3,        /* call order violated (weights / schedule / prepare missing) */
    DSD_ERR_NOMEM = -4,
    DSD_ERR_TIMEOUT = -5,      /* a persistent K-step loop of an EARLIER call hit its inter-workgroup spin bound (see dsd_check) */
    DSD_ERR_RANGE = -6         /* split mode, pair format: an activation of an EARLIER call left fp16's range (|x| > 65504); reported like a timeout */
} dsd_status;

/* The hparams DiffNet reads at construction (usr/diff/net.py:85-90) + audio_num_mel_bins (:82). */
typedef struct dsd_config {
    int32_t mel_bins;               /* M, <= 96                                  */
    int32_t residual_channels;      /* C, must be 256 for this (fused) engine; other widths run on the generic operators of dsf.h */
    int32_t encoder_hidden;         /* H (cond width), must be 256 for this engine (diffsinger_amd/net.py DiffNet.fused())        */
    int32_t residual_layers;        /* L, 1..64                                  */
    int32_t dilation_cycle_length;  /* dilation of layer l = 2^(l % cycle), must stay <= 8 */
} dsd_config;

/* Device pointers to the DiffNet parameters in torch state_dict layout (usr/diff/net.py:91-105):
 * conv weights [out][in][k], linear weights [out][in].  Per-layer arrays are HOST arrays of L device pointers. */
typedef struct dsd_weights {
    const float* input_projection_w;            /* [C][M][1]  */
    const float* input_projection_b;            /* [C]        */
    const float* mlp0_w;                        /* [4C][C]    */
    const float* mlp0_b;                        /* [4C]       */
    const float* mlp2_w;                        /* [C][4C]    */
    const float* mlp2_b;                        /* [C]        */
    const float* const* dilated_conv_w;         /* L x [2C][C][3] */
    const float* const* dilated_conv_b;         /* L x [2C]       */
    const float* const* diffusion_projection_w; /* L x [C][C]     */
    const float* const* diffusion_projection_b; /* L x [C]        */
    const float* const* conditioner_projection_w; /* L x [2C][H][1] */
    const float* const* conditioner_projection_b; /* L x [2C]       */
    const float* const* output_projection_w;    /* L x [2C][C][1] */
    const float* const* output_projection_b;    /* L x [2C]       */
    const float* skip_projection_w;             /* [C][C][1]  */
    const float* skip_projection_b;             /* [C]        */
    const float* final_projection_w;            /* [M][C][1]  (DiffNet.output_projection) */
    const float* final_projection_b;            /* [M]        */
} dsd_weights;

int dsd_abi_version(void);
const char* dsd_last_error(void);
/* sha256 (64 hex digits) of the sources this binary was compiled from - every .hip / .hpp under csrc/, every .h under include/ and the compiler flags, as
 * diffsinger_amd/build.py source_hash() computes it; "unknown" for a build that did not go through build.py.  smoke() and every bench line
 * print it and compare it with the hash of the tree they run from: the binary proves its source. */
const char* dsd_build_id(void);

/* DiffNet.__init__ (usr/diff/net.py:82-105): validates the configuration and binds the handle to `device`. */
int dsd_create(const dsd_config* cfg, int device, dsd_handle** out);
void dsd_destroy(dsd_handle* h);

/* Replaces nn.Module parameter ownership / load_state_dict (utils/__init__.py:178-209): repacks the torch
 * tensors into MFMA-fragment order in handle-owned memory; the source tensors are not referenced afterwards. */
int dsd_load_weights(dsd_handle* h, const dsd_weights* w, void* stream);

/* GaussianDiffusion.__init__ schedule tables (usr/diff/shallow_diffusion_tts.py:87-123): float64 on the host,
 * cast to fp32 exactly like the reference's register_buffer calls.  betas: HOST array of n doubles. */
int dsd_set_schedule(dsd_handle* h, const double* betas, int32_t n);
/* Copies one of the twelve fp32 tables back (HOST out[n]) for state_dict/buffer parity checks; `which` in
 * registration order: 0 betas, 1 alphas_cumprod, 2 alphas_cumprod_prev, 3 sqrt_alphas_cumprod,
 * 4 sqrt_one_minus_alphas_cumprod, 5 log_one_minus_alphas_cumprod, 6 sqrt_recip_alphas_cumprod,
 * 7 sqrt_recipm1_alphas_cumprod, 8 posterior_variance, 9 posterior_log_variance_clipped,
 * 10 posterior_mean_coef1, 11 posterior_mean_coef2. */
int dsd_get_schedule_table(dsd_handle* h, int32_t which, float* out, int32_t n);

/* spec_min / spec_max buffers (shallow_diffusion_tts.py:125-126): HOST arrays of M floats. */
int dsd_set_spec_range(dsd_handle* h, const float* spec_min, const float* spec_max);

/* Binds a batch: sizes the workspace for B utterances of T frames and evaluates every layer's
 * conditioner_projection (usr/diff/net.py:68) ONCE - it does not depend on x or t, so it is hoisted out of
 * the K-step loop.  cond is [B][H][T] addressed with element strides (the reference passes a transposed view
 * of [B,T,H], shallow_diffusion_tts.py:238 -> strides (T*H, 1, H)).  cond is not referenced afterwards. */
int dsd_prepare(dsd_handle* h, int32_t B, int32_t T, const float* cond,
                int64_t stride_b, int64_t stride_h, int64_t stride_t, void* stream);

/* DiffNet.forward(spec, diffusion_step, cond) (usr/diff/net.py:107-130) for the prepared batch.
 * x, eps: [B][M][T]; t: HOST array of B step indices (may differ per utterance, as in p_losses :214-217). */
int dsd_denoise(dsd_handle* h, const float* x, const int32_t* t, float* eps, void* stream);

/* GaussianDiffusion.q_sample (shallow_diffusion_tts.py:206-211) with one t for the batch (:255):
 * out = sqrt_alphas_cumprod[t] * x_start + sqrt_one_minus_alphas_cumprod[t] * noise, all [B][M][T]. */
int dsd_q_sample(dsd_handle* h, const float* x_start, const float* noise, int32_t t, float* out, void* stream);

/* The DDPM loop `for i in reversed(range(0, k_step)): x = p_sample(x, i, cond)` (shallow_diffusion_tts.py:
 * 269-270, p_sample :159-166, p_mean_variance :149-157).  x [B][M][T] is updated in place from x_{k_step} to
 * x_0; noise is [k_step][B][M][T], slice j feeds the j-th call (t = k_step-1-j), t = 0 included (NULL: see dsd_set_noise_seed). */
int dsd_sample_ddpm(dsd_handle* h, float* x, const float* noise, int32_t k_step, void* stream);

/* noise == NULL in dsd_sample_ddpm / dsd_p_sample: the N(0,1) draws of `noise_like` (shallow_diffusion_tts.py:38-41) are generated
 * in the kernel instead of read from HBM - Philox4x32-10 keyed by the seed set here, counter = (element index in [B][M][T],
 * index of the p_sample call in the loop), Box-Muller on the first two output words.  The stream is a pure function of
 * (seed, call index, element): independent of tiling, chunking and of which kernel path runs.  dsd_philox_normal writes the
 * draws of one call index (out[n], element i <-> counter i) so that a caller / test can reproduce them. */
int dsd_set_noise_seed(dsd_handle* h, uint64_t seed);
int dsd_philox_normal(dsd_handle* h, uint64_t seed, int32_t step, float* out, int64_t n, void* stream);

/* One p_sample call (shallow_diffusion_tts.py:159-166) at step t: x [B][M][T] updated in place from x_t to
 * x_{t-1}; noise [B][M][T] is the N(0,1) draw of `noise_like` (ignored by the arithmetic when t == 0). */
int dsd_p_sample(dsd_handle* h, float* x, const float* noise, int32_t t, void* stream);
/* The same call with everything the reference's signature `p_sample(x, t, cond, clip_denoised=True, repeat_noise=False)` allows:
 * t = HOST array of B step indices (one per utterance, as the [B] tensor of the reference), clip_denoised 0 / 1 (the clamp of
 * p_mean_variance :153-154), repeat_noise 1: noise is ONE [M][T] draw used for every utterance (noise_like :38-41), else [B][M][T]. */
int dsd_p_sample_ex(dsd_handle* h, float* x, const float* noise, const int32_t* t, int32_t clip_denoised, int32_t repeat_noise,
                    void* stream);

/* The PNDM/PLMS loop `for i in reversed(range(0, k_step, interval)): x = p_sample_plms(x, i, interval, cond)`
 * (shallow_diffusion_tts.py:261-267, p_sample_plms :168-204), noise_list reset at entry (:262). */
int dsd_sample_plms(dsd_handle* h, float* x, int32_t k_step, int32_t interval, void* stream);

/* norm_spec / denorm_spec (shallow_diffusion_tts.py:278-282) fused with the layout change of :252 / :271:
 *   dsd_norm_spec:   mel [B][T][M] -> x [B][M][T],  (mel - min) / (max - min) * 2 - 1
 *   dsd_denorm_spec: x [B][M][T] -> mel [B][T][M],  (x + 1) / 2 * (max - min) + min, optionally multiplied by
 *                    mask [B][T] (the `mel2ph > 0` factor of :273; NULL = none). */
int dsd_norm_spec(dsd_handle* h, const float* mel, float* x, void* stream);
int dsd_denorm_spec(dsd_handle* h, const float* x, const float* mask, float* mel, void* stream);

/* Options: 0 = eager launches (default 1 = replay the K-step loop as a cached hipGraph). */
int dsd_set_use_graph(dsd_handle* h, int32_t enable);
/* How the K-step loops run (dsd_set_loop_mode):
 *   2 (default) automatic.  A batch that fills less than half of the chip (32-frame tiles x 2 <= CU count; the reference's own inference
 *     shape, one utterance per device: configs/tts/fs2.yaml:70) takes the LATENCY kernels: every residual layer as two kernels whose
 *     workgroups split the output rows of a tile G = 16 / 8 / 4 / 2 ways (the largest G that still gives each workgroup its own CU), nodes of
 *     the cached hipGraph - up to 8 x more CUs per utterance.  Larger batches take the PERSISTENT loop (below) unless its chunking in
 *     whole utterances would idle more of the chip than the per-layer kernels' grid quantisation (e.g. T = 5000: 157 of 256 CUs).
 *   1 the persistent loop whenever the prepared batch allows it (32-frame tiles, hipGraph mode on, one utterance's tiles <= the CU
 *     count): ONE kernel for the whole loop - every workgroup keeps its x tile and skip sum in registers across layers and steps and
 *     exchanges only the conv halo with its neighbours; larger batches run as chunks of whole utterances.
 *   0 one kernel per residual layer + head (a cached hipGraph, or eager launches, see above).
 *   3 the latency kernels regardless of the batch size (tests).
 * Modes 0 and 1 and the G = 2 / 4 latency kernels give bit-identical results; G = 8 / 16 sum the K halves / quarters of the contractions separately
 * (reduction-order noise, ~1e-6).  dsd_set_lat_split: G = -1 by batch size (default), 0 never, 2 / 4 / 8 / 16 forced where the latency path
 * applies; dsd_get_lat_split: the G the prepared batch runs with (0 = not on that path).  dsd_get_loop_mode: 1 if the prepared batch
 * would take the persistent path.
 * dsd_loop_timeouts: synchronises the stream and returns the sticky timeout word of the persistent loop (0 = every
 * inter-workgroup wait was satisfied; nonzero = a wait hit its spin bound; the affected tiles of x are then NaN).
 * The persistent kernel needs all its workgroups resident at once (<= one per CU), so only ONE such loop may run on a device at
 * a time.  Inside one process the library enforces that itself: a persistent launch waits on the device (hipStreamWaitEvent) for
 * the previous persistent launch of any handle on the same GPU that went to a different stream - two handles / two streams sampling
 * "concurrently" are serialised (each loop fills the chip anyway), never starved into the timeout.  Two PROCESSES sharing one GPU
 * cannot see each other: use mode 0 there (one process per device, as the reference's DDP runner does, is always safe).
 * dsd_loop_launches: k_loop launches per sampling call for the prepared batch (chunks of whole utterances; 0 = not on that path).
 * Environment (read at dsd_create, each the same choice as the setter named): DSD_LOOP=<mode> = dsd_set_loop_mode; DSD_CONV=direct|winograd =
 * dsd_set_conv_mode; DSD_SPLIT=1 = dsd_set_split_mode with DSD_SPLIT_W=2|0 (format) and DSD_SPLIT_TOUCH=<chunks> (its L2 touch) - the
 * split-precision experiment below.  These are all the environment switches. */
int dsd_set_loop_mode(dsd_handle* h, int32_t mode);
int dsd_loop_launches(dsd_handle* h);
int dsd_set_lat_split(dsd_handle* h, int32_t g);
int dsd_get_lat_split(dsd_handle* h);
/* How the PERSISTENT loop evaluates the 3-tap dilated convolution (usr/diff/net.py:61,71):
 *   1 (default) Winograd F(2,3) along the frame axis (csrc/dsd_loop_wino.hpp): for every output pair (t, t + d) four products with transformed
 *     weights instead of six - 2/3 of the convolution's fp32 multiplications, the same dtype (exact-fp32 MFMA, fp32 transforms: weights summed
 *     in fp64 and rounded once at load, inputs one fp32 add).  Results differ from the direct form by reduction order and those roundings
 *     (~1e-5 on a K = 100 loop against a 1e-4 budget; tests/test_gpu_wino.py).
 *   0 the direct K = 768 contraction (k_loop): bit-identical to loop mode 0 and to the G = 2 / 4 latency kernels of this mode.
 * The row-split latency kernels follow the same switch at G = 2 / 4 / 8 (their conv node is k_lat_conv_w, reading the loop's transformed
 * weights; round 5) - so dsd_denoise / dsd_p_sample / the sampling calls on a SMALL batch change with it too; G = 16, the per-layer kernels
 * (loop mode 0) and the training operators always evaluate the direct form, and with mode 0 every path does.  touch_ahead: steps
 * (16 KiB of the transformed-weight stream) the waves of an XCD fetch into their L2 ahead of themselves, 0 = off, -1 = keep (default 32) - a
 * tuning knob of tools/, results do not depend on it.
 * dsd_get_conv_mode: 1 if the prepared batch's convolution runs in the Winograd form - on the persistent loop (dsd_get_loop_mode = 1) or on the
 * latency kernels at G = 2 / 4 / 8 (dsd_get_lat_split) - else 0.  Environment: DSD_CONV=direct|winograd at dsd_create. */
int dsd_set_conv_mode(dsd_handle* h, int32_t mode, int32_t touch_ahead);
int dsd_get_conv_mode(dsd_handle* h);

/* EXPERIMENT (csrc/dsd_split.hpp, csrc/dsd_loop_split.hpp; default off, env DSD_SPLIT=1 turns it on at creation): the residual layers on
 * the 16-bit matrix pipe with fp32-class accuracy.  The persistent loop (k_loop_split: the same loop, x / skip sum / halo exchange / head /
 * sampler in fp32) takes every fp32 operand as TWO scaled fp16 planes, x = h0 + 2^-11 h1, and a product as h0 g0 + 2^-11 (h0 g1 + h1 g0),
 * accumulated in fp32 (the pair format; env DSD_SPLIT_W=2, the default) - or as three exact bf16 planes and the six products with i + j <= 2
 * (DSD_SPLIT_W=0: the planes on the wire - the cross-check stream of the tests), which is also what the
 * per-layer kernel path does (k_layer_split, 32-frame tiles).  The latency path is bypassed while the mode is on.  Never the
 * headline dtype: bench.py reports it as a labelled `secondary` line with its error against an fp64 evaluation of the oracle beside the
 * fp32 path's.  Enqueues the weight packing on `stream`. */
int dsd_set_split_mode(dsd_handle* h, int32_t on, void* stream);
int dsd_get_split_mode(dsd_handle* h);

/* Debug hook: ONE residual layer (usr/diff/net.py:66-78; the fp32 kernel, or the split-precision one while that mode is on) on a
 * caller-supplied input with the prepared batch's conditioner projection and step t, results in logical layout - layer-level parity
 * tests and error localisation.  x_in, x_out, skip_out: DEVICE [B][C][TS], TS = T rounded up to 32 (x_out may be NULL for the last
 * layer, which computes skips only).  Overwrites the handle's x / skip work buffers. */
int dsd_debug_layer(dsd_handle* h, int32_t layer, int32_t t, const float* x_in, float* x_out, float* skip_out, void* stream);
int dsd_get_loop_mode(dsd_handle* h);
int dsd_loop_timeouts(dsd_handle* h, void* stream);

/* Loud failures of the persistent loop (the reference's loop, usr/diff/shallow_diffusion_tts.py:261-270, cannot fail this way - ours must
 * not fail silently).  A one-thread kernel enqueued behind every persistent loop latches a raised timeout word into PINNED host memory.
 * dsd_check reads that word WITHOUT synchronising anything: DSD_OK, or DSD_ERR_TIMEOUT when a persistent loop that has finished since the
 * last report hit its spin bound (its result tiles are NaN).  Every data-path entry point (dsd_prepare, dsd_denoise, dsd_sample_*,
 * dsd_denorm_spec, ...) makes the same check first, so a timeout of call n surfaces at call n + 1 at the latest; a caller that wants it
 * at call n synchronises its stream (it does anyway before reading the mel) and calls dsd_check.  Reporting consumes the flag and PARKS
 * the handle on the hipGraph path (per-layer kernels, no co-residency requirement) so that the retry succeeds.  A parked handle returns to
 * the persistent path by itself after 16 sampling calls (a server starved once is not slower for the rest of its life), or at once with
 * dsd_set_loop_mode.  dsd_loop_parked: sampling calls left on the hipGraph path, 0 = not parked. */
int dsd_check(dsd_handle* h);
int dsd_loop_parked(dsd_handle* h);

/* Test hook: occupy `n_workgroups` compute units (one 64-thread workgroup with the whole 160 KiB of LDS each, so nothing else fits
 * beside it) for `milliseconds` of wall-clock time on `stream` - the "foreign kernel" the persistent loop's timeout exists for.
 * ctl: NULL, or two words the device can reach (device memory or pinned host memory): ctl[0] a counter every holder increments once it is resident (the caller polls it), ctl[1] a release
 * word - the holders leave as soon as it is non-zero (or after `milliseconds`, whichever comes first). */
int dsd_debug_hold_cus(int32_t device, int32_t n_workgroups, int32_t milliseconds, uint32_t* ctl, void* stream);

/* Test hook (host logic only, no device work): grid.y of the hoisted conditioner projection's launch for a stack of L layers with these
 * dilations and a batch of ntiles 32-frame tiles - a multiple of the dilation cycle's period such that every workgroup gets the same number
 * (at most 10) of layers of ONE dilation and the chip two workgroups per CU, else L (one layer per workgroup); *lds_bytes: the launch's
 * dynamic LDS.  The environment variable DSD_CP_GROUPS ("layer" / a number) overrides it, for A/B measurements.  -1: bad argument. */
int32_t dsd_debug_condproj_groups(const uint8_t* dilations, int32_t L, int32_t ntiles, int64_t* lds_bytes);

/* Frames per workgroup of the residual-layer kernel: 0 = choose from the batch size, 32 or 64. */
int dsd_set_layer_tile(dsd_handle* h, int32_t frames);

/* Measurement hook for bench.py's roofline object: average device time (ms, hipEvents on the stream the
 * kernels run on) per launch of `iters` residual-layer kernel launches (step t) on the prepared batch, replayed
 * as nodes of one hipGraph exactly like the sampling loop issues them.  layer >= 0: that layer only; layer < 0:
 * the non-last layers 0..L-2 in order, as in one denoiser evaluation.  The figure includes the inter-kernel
 * dispatch gap (~1.3 us).  Synchronises the stream (a measurement call, not part of the data path). */
int dsd_time_layer_kernel(dsd_handle* h, int32_t layer, int32_t t, int32_t iters, float* avg_ms, void* stream);

/* Debug hook: per-wave shader-clock stamps of one launch of layer `layer` (start, staged, conv done, gate done,
 * out-proj done, end): HOST out[blocks*4*8] u64; *n_blocks = grid size.  Synchronises the stream. */
int dsd_debug_layer_timeline(dsd_handle* h, int32_t layer, int32_t t, uint64_t* out, int32_t max_blocks, int32_t* n_blocks,
                             void* stream);

/* Debug hook of the persistent loop: runs the DDPM loop (arguments as dsd_sample_ddpm) with per-wave shader-clock stamps taken in
 * phase `phase` (= evaluation * L + layer, a non-last layer of an evaluation that is not the last): HOST out[n_wg*4*16] u64 =
 * [0..7] {phase start, neighbours' flags seen, y tile staged, conv done, gate done, x' ready, halo published, phase end},
 * [8..15] the head of that evaluation {last layer done, skip tile staged, skip projection done, ReLU tile visible, final projection
 * done, sampler update stored, barrier, next input projection + halo published}.  Single-launch batches only.  Synchronises. */
int dsd_debug_loop_timeline(dsd_handle* h, float* x, const float* noise, int32_t k_step, int32_t phase, uint64_t* out,
                            int32_t max_wg, int32_t* n_wg, void* stream);

/* Introspection for tests: bytes of device memory owned by the handle; frames/workgroup currently selected. */
int64_t dsd_device_bytes(dsd_handle* h);
int dsd_get_layer_tile(dsd_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* DSD_H */
