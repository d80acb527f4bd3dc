#!/usr/bin/env python
"""bench.py - the reference's headline metric on MI355X: mel-frames/s of the K=100 DDPM reverse loop.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the driver launches the ranks (python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, with no launcher around it, `python bench.py --gpus N` starts them itself
with that same command line.

Workload at N = 1 (BASELINE.json configs[1], `--config 2`): DiffSpeech 80-bin denoiser (residual_channels 256, 20 layers, dilation
cycle 1, linear beta schedule max_beta 0.06, 100 timesteps), K=100 full DDPM from a Gaussian start, batch of 8 utterances x T=1024 frames.
Workload at N > 1 (BASELINE.json configs[4], `--config 5`, also available at N = 1): the SAME 512 utterances x T=2048, K=100, at every N
(strong scaling): rank r takes utterances r::W, runs them in micro-batches of 16 with no data-path collective, one RCCL gather collates the
finished mels on rank 0; value = 512 x 2048 frames / max-over-ranks wall time.  Synthetic inputs (seeded N(0,1) cond / x_T / per-step noise)
and seeded random-init weights - there are no checkpoints or datasets (no network).

One "step" = one complete pass of the hot path over the batch: hoisted conditioner projection (dsd_prepare),
the 100-step reverse loop (ONE persistent kernel launch, csrc/dsd_loop.hpp; fallback: one hipGraph replay of 2100
per-layer launches), de-normalisation to [B,T,80] and, for N > 1, the gather to rank 0.  Inputs are resident in HBM
before the timed region.

The JSON line also carries
  roofline      for the dominant kernel (k_loop: the whole loop; or k_layer, one residual block, on the fallback path):
                executed fp32 FLOPs per launch / launch duration measured live with HIP events on the launch stream,
                against the 157.3 TFLOP/s
                dense fp32 MFMA peak (MI355X_MICROARCH.md); `traffic` = HBM-side bytes per launch from the committed
                rocprofv3 --pmc passes of the same kernel and shape (profiles/layer_pmc.json).
  cpu_baseline  the CPU oracle (oracle/diffnet_oracle.py, torch fp32 = the reference's own arithmetic) timed on
                this box's host cores ON THE TIMED BATCH: all 100 steps of the last timed step's (x_T, cond, noise)
                with the benched model's weights (~12-20 s).
  parity        max-abs error of the de-normalised mel THE TIMED KERNEL produced for that batch against the same
                oracle run (`kernel` names it: k_loop<1>); `fixture` = the golden case generated from the reference
                (B=2, T=96: the latency kernels).

`--row vocoder` / `--row train` bench a row AROUND the path instead (f2 below; f3: one p_losses forward + backward of the denoiser,
8 x 1024 frames per GPU, CPU autograd on the oracle beside it).  `--row vocoder` benches the row BEHIND the path (HiFi-GAN generator, 8 x 1024 mel frames -> 8 x 262 144
samples per GPU) with the same contract: one JSON line, roofline of its dominant kernel, the CPU oracle timed beside it.  The default
(no --row) is the headline metric above and nothing else.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PRESET = 'lj_ds_beta6'
B_PER_GPU, T_FRAMES, K_STEPS = 8, 1024, 100
F_LAYER_EXEC = 2 * 512 * 768 + 2 * 512 * 256          # executed FLOP / frame / layer launch (dilated conv + out proj)
F_LAYER_REF = F_LAYER_EXEC + 2 * 512 * 256            # + the conditioner projection the reference recomputes per step
F_EVAL_REF = 26_427_392                               # reference GEMM FLOP / frame / denoiser evaluation
L_LAYERS = 20
F_EVAL_EXEC = (L_LAYERS - 1) * F_LAYER_EXEC + (F_LAYER_EXEC - 2 * 256 * 256) + 2 * (256 * 256 + 80 * 256 + 80 * 256)   # 21 053 440
# Winograd F(2,3) form of the dilated convolution (csrc/dsd_loop_wino.hpp, the default of the persistent loop): four [512 x 256] x [256 x 16 pairs]
# products per 32-frame tile instead of three [512 x 256] x [256 x 32] - 524 288 multiply-add FLOP / frame / layer instead of 786 432
F_EVAL_EXEC_WINO = F_EVAL_EXEC - L_LAYERS * (2 * 512 * 768 - 4 * 2 * 512 * 256 // 2)      # 15 810 560
PEAK_FP32_MFMA_TFLOPS = 157.3
# collectives of the bench's process group fail after this long instead of blocking for ever (a rank that died outside sharded_inference's own
# status exchange, diffsinger_amd/dist.py _agree_or_raise)
GROUP_TIMEOUT = __import__('datetime').timedelta(minutes=10)
WEIGHT_BYTES = 15_086_416 * 4                              # 15 086 416 parameters = 60.3 MB, re-read every step


def build_model(device):
    import diffsinger_amd
    from diffsinger_amd import hparams
    from diffsinger_amd.synth import presets
    pre = presets()[PRESET]
    hparams.clear()
    diffsinger_amd.use_preset(PRESET)
    torch.manual_seed(1234)                            # reference default seed (configs/config_base.yaml:5)
    net = diffsinger_amd.DIFF_DECODERS[pre['diff_decoder_type']](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)      # reference zero-inits it (net.py:105)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=K_STEPS,
                                          loss_type='l1', spec_min=pre['spec_min'], spec_max=pre['spec_max'])
    return gd.to(device).eval(), pre


def host_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                                # cgroup v2 CPU quota
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(gd, pre, cond, x_T, noise, mel_hip, kernel):
    """The oracle on the host cores, run ON THE TIMED BATCH: the (x_T, cond, noise) of the last timed step go through
    oracle.infer_mel for all K = 100 steps with the weights of the benched model (its state_dict).  One run gives both figures:
    the CPU rate of the same workload and the max-abs error of the de-normalised mel the timed kernel produced
    (usr/diff/shallow_diffusion_tts.py:248-276 on identical inputs).  ~12-20 s of CPU work on the box's host cores."""
    from oracle import diffnet_oracle as O
    cfg = O.NetConfig(80, 256, 256, 20, 1)
    p = {k: v.detach().to('cpu', torch.float32).contiguous() for k, v in gd.denoise_fn.state_dict().items()}
    sch = O.make_schedule(O.linear_beta_schedule(pre['timesteps'], pre['max_beta']))
    smin = torch.tensor(pre['spec_min'], dtype=torch.float32)[None, None, :]
    smax = torch.tensor(pre['spec_max'], dtype=torch.float32)[None, None, :]
    cond_c = cond.detach().cpu()                            # [B,H,T] view of [B,T,H], strides preserved
    x_c, nz = x_T.detach().cpu(), noise.detach().cpu()
    B, T = x_c.shape[0], x_c.shape[-1]
    t = torch.full((B,), K_STEPS - 1, dtype=torch.long)
    # host cores really available to this process (affinity / cgroup quota), then the best of a few thread counts:
    # oneDNN with one thread per SMT sibling of a 2-socket box is far slower than a moderate count on this shape
    avail = host_cpus()
    cands = sorted({c for c in (8, 16, 32, 64, 128, avail) if c <= avail}) or [avail]
    best, cores = None, cands[0]
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            O.diffnet_forward(p, cfg, x_c, t, cond_c)      # warm-up (oneDNN primitive creation for this thread count)
            t0 = time.perf_counter()
            O.diffnet_forward(p, cfg, x_c, t, cond_c)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, cores = dt, c
            if dt > 4 * best:
                break
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        mel = O.infer_mel(p, cfg, sch, cond_c, smin, smax, k_step=K_STEPS, noises=list(nz), x_T=x_c)
        el = time.perf_counter() - t0
    err = float((mel_hip.detach().cpu() - mel).abs().max())
    base = {'value': B * T / el, 'unit': 'mel-frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'all {K_STEPS} DDPM steps + denorm of THE TIMED BATCH (oracle.infer_mel, B={B}, T={T}: the x_T, cond and noise of the last '
                      f'timed step, the weights of the benched model) in {el:.1f}s on {cores} host threads (best of {cands}; {avail} CPUs available, '
                      f'os.cpu_count()={os.cpu_count()})',
            'sec_per_ddpm_step': el / K_STEPS}
    par = {'case': f'the timed batch itself: {B} x {T} frames, K={K_STEPS} DDPM from the Gaussian start, de-normalised mel [B,T,80] of the last timed '
                   'step vs oracle.infer_mel on the same (x_T, cond, noise[K]) and weights', 'kernel': kernel,
           'max_abs_mel_err': err, 'tolerance': 1e-4, 'elements': int(mel.numel())}
    return base, par, mel


def secondary_split(gd, eng, pre, cond, x_T, noise, mel_f32, mel_oracle, args):
    """The labelled EXPERIMENT line (`secondary`, never the headline): the SAME timed step with the residual layers' contractions on the 16-bit
    matrix pipe at fp32-class accuracy (csrc/dsd_loop_split.hpp; by default the pair format: every fp32 operand = two scaled fp16 planes, three
    plane products per product, fp32 accumulate; DSD_SPLIT_W=0 / 4: three exact bf16 planes, six products) - what the north star's 1e-4 budget
    buys beyond the fp32 MFMA ceiling.  Reports its rate, its roofline against the dense 16-bit peak / products, and - for BOTH paths, on
    utterance 0 of the timed batch - the error against the fp32 oracle and against an fp64 evaluation of the oracle (the same function in
    double: the parameters, inputs and noise cast up, the fp32 schedule tables as constants)."""
    from oracle import diffnet_oracle as O
    eng.set_split_mode(True)
    try:
        cond = cond.clone()                               # a fresh tensor: the conditioner projection is prepared again for THIS cond
        run = lambda: gd.inference(cond, x_T=x_T, noise=noise, K_step=K_STEPS, pndm_speedup=0)
        mel_sp = run()
        torch.cuda.synchronize()
        assert eng.loop_mode() == 1 and eng.loop_timeouts() == 0, 'the split-precision run did not take the persistent loop'
        conds2 = [cond, cond.clone()]                     # fresh cond tensor every step: dsd_prepare stays inside the timed step
        n = 5
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            gd.inference(conds2[i & 1], x_T=x_T, noise=noise, K_step=K_STEPS, pndm_speedup=0)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / n
        B, T = x_T.shape[0], x_T.shape[-1]
        xs = x_T[:, 0].contiguous().clone()
        nz = noise[:, :, 0]
        eng.prepare(cond)
        eng.sample_ddpm(xs, nz, K_STEPS)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(3):
            eng.sample_ddpm(xs, nz, K_STEPS)
        ev1.record()
        ev1.synchronize()
        ms_call = ev0.elapsed_time(ev1) / 3
        achieved = B * T * K_STEPS * F_EVAL_EXEC / (ms_call * 1e-3) / 1e12
    finally:
        eng.set_split_mode(False)
    # fp64 evaluation of the oracle on utterance 0 (~35 s on 16 host threads)
    cfg = O.NetConfig(80, 256, 256, 20, 1)
    p64 = {k: v.detach().to('cpu', torch.float64) for k, v in gd.denoise_fn.state_dict().items()}
    sch = O.make_schedule(O.linear_beta_schedule(pre['timesteps'], pre['max_beta']))
    smin = torch.tensor(pre['spec_min'], dtype=torch.float64)[None, None, :]
    smax = torch.tensor(pre['spec_max'], dtype=torch.float64)[None, None, :]
    c0 = cond[0:1].detach().cpu().double()
    t0 = time.perf_counter()
    m64 = O.infer_mel(p64, cfg, sch, c0, smin, smax, k_step=K_STEPS, noises=list(noise[:, 0:1].detach().cpu().double()), x_T=x_T[0:1].detach().cpu().double())
    t64 = time.perf_counter() - t0
    err = lambda a, b: float((a.detach().cpu().double() - b.double()).abs().max())
    rms = lambda a, b: float((a.detach().cpu().double() - b.double()).pow(2).mean().sqrt())
    wsel = os.environ.get('DSD_SPLIT_W', '2')
    fmt = {'2': {'dtype': 'f32 as 2 scaled fp16 planes (11 + 11 mantissa bits), 3 plane products per product, f32 accumulate; head, sampler and state in f32',
                 'format': 'pair format: x = h0 + 2^-11 h1, product = h0 g0 + 2^-11 (h0 g1 + h1 g0) on v_mfma_f32_32x32x16_f16 (DSD_SPLIT_W=2, the default)',
                 'products': 3, 'peak_note': 'peak = dense fp16 MFMA peak 2500 TFLOP/s / 3 plane products'}}.get(wsel, {
           'dtype': 'f32 as 3 exact bf16 planes, 6 plane products per product (i + j <= 2), f32 accumulate; head, sampler and state in f32',
           'format': 'three bf16 planes, six products on v_mfma_f32_32x32x16_bf16 (DSD_SPLIT_W=%s: %s)' % (
               wsel, 'the planes on the wire' if wsel == '0' else 'fp32 weights on the wire, split into the planes in registers'),
           'products': 6, 'peak_note': 'peak = dense bf16 MFMA peak 2500 TFLOP/s / 6 plane products'})
    peak = 2500.0 / fmt['products']
    return {
        'label': 'EXPERIMENT, not the headline: residual layers on the 16-bit matrix pipe at fp32-class accuracy (dsd_set_split_mode; csrc/dsd_loop_split.hpp)',
        'dtype': fmt['dtype'],
        'metric': BASELINE_METRIC, 'value': B * T / sec, 'unit': 'mel-frames/s', 'ms_per_step': sec * 1e3, 'steps': n,
        'workload': f'the timed step of this line (dsd_prepare + K={K_STEPS} loop + denorm, {B} x {T} frames)',
        'format': fmt['format'],
        'roofline': {'bound': 'mfma', 'kernel': 'k_loop_split<1, %s>' % wsel, 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s (fp32-equivalent)', 'frac': achieved / peak,
                     'avg_launch_ms': ms_call, 'note': 'executed fp32-equivalent FLOPs (21 053 440 / frame / evaluation) over the whole sampling call (HIP events); '
                                                       + fmt['peak_note'] + '; the head (2 % of the fp32 launch) runs on the fp32 pipe'},
        'parity': {'case': 'the timed batch (all 8 utterances) vs the fp32 oracle; utterance 0 vs an fp64 evaluation of the oracle, both paths', 'tolerance': 1e-4,
                   'split_vs_f32_hip': err(mel_sp, mel_f32.detach().cpu()), 'split_vs_oracle_f32': err(mel_sp, mel_oracle), 'f32_vs_oracle_f32': err(mel_f32, mel_oracle),
                   'split_vs_oracle_f64_utt0': err(mel_sp[0:1], m64), 'f32_vs_oracle_f64_utt0': err(mel_f32[0:1], m64),
                   'oracle_f32_vs_oracle_f64_utt0': err(mel_oracle[0:1], m64),
                   'rms_vs_oracle_f64_utt0': {'split': rms(mel_sp[0:1], m64), 'f32': rms(mel_f32[0:1], m64), 'oracle_f32': rms(mel_oracle[0:1], m64)},
                   'note': 'max-abs figures of different paths can coincide to the last digit: where the output is clamped every path returns the same fp32 value '
                           'and the distance to the fp64 evaluation is the rounding of the de-normalisation alone; the rms figures compare the paths',
                   'f64_oracle_seconds': t64},
    }


def torch_rocm_eager_baseline(gd, cond, x_T):
    """CONTEXT, never the target and never inside a timed region: what the reference's OWN code does on this chip.  The oracle is the bit-equal
    restatement of DiffNet.forward + p_sample (usr/diff/net.py:107-130, usr/diff/shallow_diffusion_tts.py:134-166) in plain torch ops; with its
    tensors on cuda:0 that is PyTorch-ROCm eager - MIOpen convolutions, rocBLAS linears, ~25 launches per residual layer - exactly what
    `GaussianDiffusion.forward(infer=True)` of the reference would run on an MI355X (usr/diff/shallow_diffusion_tts.py:269-270).  2 warm-up
    + 5 timed denoiser evaluations with the sampler's element-wise update on the timed batch, extrapolated x K."""
    from oracle import diffnet_oracle as O
    cfg = O.NetConfig(80, 256, 256, 20, 1)
    dev = x_T.device
    p = {k: v.detach().to(dev, torch.float32).contiguous() for k, v in gd.denoise_fn.state_dict().items()}
    x = x_T.detach().clone()
    c = cond.detach()
    B, T = x.shape[0], x.shape[-1]
    n = 5
    with torch.no_grad():
        for i in range(2):
            O.diffnet_forward(p, cfg, x, torch.full((B,), K_STEPS - 1 - i, device=dev, dtype=torch.long), c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            eps = O.diffnet_forward(p, cfg, x, torch.full((B,), K_STEPS - 1 - i, device=dev, dtype=torch.long), c)
            x0 = (1.01 * x - 0.1 * eps).clamp_(-1., 1.)                 # the shape of p_sample's update: predict_start, clip, posterior mean, noise
            x = 0.5 * x0 + 0.5 * x + 0.01 * torch.randn_like(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    return {'value': B * T / (dt * K_STEPS), 'unit': 'mel-frames/s', 'sec_per_ddpm_step': dt,
            'sample': f'{n} denoiser evaluations + sampler-shaped element-wise update of the timed batch ({B} x {T}) with the oracle\'s torch ops on cuda:0 '
                      f'(PyTorch {torch.__version__} eager: MIOpen / rocBLAS), after the timed region; x {K_STEPS}',
            'note': 'context, extrapolated: the reference\'s own operator sequence on this GPU - not a target, not the product path'}


def evidence(fname: str, kernel_tag=None, **must):
    """An evidence summary under profiles/ (PMC passes, in-kernel timelines: tools/pmc_summary.py, tools/loop_timeline.py) - but only if it
    was measured on THE CODE THIS PROCESS RUNS.  A summary carries the build id of the library it was taken from (dsd_build_id: sha256 of
    csrc/ + include/) and the hash of its kernel's device code (diffsinger_amd/kernel_isa.json).  Accepted: same build id as the loaded
    library, or - the library was rebuilt since, for an edit elsewhere - the same device code of that kernel.  Anything else is refused:
    returns (None, 'stale: ...') and the bench line says `traffic: null` rather than a number from another binary (every
    round-5 summary was stamped with a round-4 commit).  Returns (json, source string)."""
    path = os.path.join(ROOT, 'profiles', fname)
    try:
        js = json.load(open(path))
    except (OSError, ValueError):
        return None, f'no profiles/{fname}'
    if kernel_tag is not None and js.get('kernel_tag') != kernel_tag:
        return None, f"profiles/{fname} is about {js.get('kernel_tag')!r}, not {kernel_tag!r}"
    for k, v in must.items():
        if js.get(k) != v:
            return None, f'profiles/{fname}: {k} = {js.get(k)!r}, this run has {v!r}'
    from diffsinger_amd import _lib
    from diffsinger_amd.build import kernel_isa
    cur = _lib.build_id()
    bid = js.get('build_id')
    tag = f"profiles/{fname} (round {js.get('round', '?')}, build {str(bid)[:12]})"
    if bid == cur:
        return js, tag
    name, isa = js.get('kernel_isa_name'), js.get('kernel_isa')
    if name and isa and kernel_isa(cur).get(name) == isa:
        return js, tag + f' - the library was rebuilt since ({cur[:12]}), the device code of {name} is unchanged (kernel_isa {isa[:12]})'
    return None, f"stale: {tag} is not the loaded library ({cur[:12]}) and the kernel's device code does not match - rerun tools/gpu.sh"


def pmc_traffic(kernel: str, frames: int):
    """HBM-side bytes per launch of the benched kernel from the rocprofv3 --pmc passes under profiles/ (separate FETCH_SIZE / WRITE_SIZE runs
    at this very shape, corrected as MI355X_MICROARCH.md prescribes; tools/gpu.sh looppmc -> tools/pmc_summary.py).  PMC counters cannot be
    read from inside this process, so the figure comes from the summary of the same kernel + shape - if `evidence` accepts it."""
    fname = 'loop_pmc.json' if kernel.startswith('k_loop') else 'layer_pmc.json'
    js, src = evidence(fname, kernel, frames=frames)
    if js is None:
        return None, src
    try:
        return float(js['hbm_bytes_per_launch']['total']), src + ': ' + js['hbm_bytes_per_launch']['note']
    except (KeyError, TypeError, ValueError):
        return None, src + ': no hbm_bytes_per_launch'


def parity_check(device):
    """Golden case generated from the reference (tests/golden/ddpm_lj_k100.npz): max-abs de-normalised mel error."""
    from tests import helpers as H
    from tests.gpu_helpers import run_hip_case
    g = H.load_golden('ddpm_lj_k100')
    out = run_hip_case('ddpm_lj_k100')
    return {'case': 'ddpm_lj_k100 (B=2,T=96,K=100, fixture from the reference)', 'kernel': 'k_lat_conv<16> / k_lat_out<16> / k_lat_head_* (6 tiles: the latency path)',
            'max_abs_mel_err': float(np.abs(out - g['out']).max()), 'tolerance': 1e-4}


def _row_setup():
    """Process / device / process-group setup shared by the --row benches (same launch contract as the headline run)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device, timeout=GROUP_TIMEOUT)
    return world, rank, device, dist


def _row_time(step, args, world, device, dist):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; max over ranks.  Returns
    (seconds, last result)."""
    out = None
    for _ in range(max(1, args.warmup)):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    elt = torch.tensor([el], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elt, op=dist.ReduceOp.MAX)
    return float(elt.item()), out


VOC_CONFIG = dict(resblock='1', upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=128,
                  resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], audio_sample_rate=24000,
                  use_pitch_embed=False)          # configs/tts/hifigan.yaml
PEAK_HBM_GBPS = 8000.0


def vocoder_flop_per_frame(h) -> int:
    c0, f, rate, ch = h['upsample_initial_channel'], 2 * 80 * h['upsample_initial_channel'] * 7, 1, 0
    for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
        ch = c0 // 2 ** (i + 1)
        rate *= u
        f += rate * 2 * (2 * ch) * ch * (k // u)
        for kk, dd in zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes']):
            f += rate * len(dd) * 2 * 2 * ch * ch * kk
    return f + rate * 2 * ch * 7


def cpu_baseline_vocoder(budget_s: float = 20.0):
    """The generator oracle (oracle/hifigan_oracle.py = the reference's own torch arithmetic) on the host cores, one utterance of 256 mel
    frames at a time until ~budget_s of CPU work."""
    from oracle import hifigan_oracle as HO
    p = HO.synth_generator_params(VOC_CONFIG, 1234)
    T = 256
    mel = torch.randn(1, 80, T, generator=torch.Generator().manual_seed(7))
    avail = host_cpus()
    cores = min(avail, 32)
    torch.set_num_threads(cores)
    with torch.no_grad():
        HO.generator(p, VOC_CONFIG, mel)
        n, t0 = 0, time.perf_counter()
        while True:
            HO.generator(p, VOC_CONFIG, mel)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 64:
                break
    return {'value': n * T / el, 'unit': 'mel-frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} forwards of 1 x {T} mel frames (oracle/hifigan_oracle.generator, plain path) in {el:.1f}s on {cores} host threads '
                      f'({avail} CPUs available)'}


def main_vocoder(args):
    """Row f2: `steps` forwards of the HIP HiFi-GAN generator over 8 x 1024 mel frames per GPU (replicas: the row has no exchange step)."""
    world, rank, device, dist = _row_setup()
    from diffsinger_amd.vocoder import HifiGanGenerator, _HipOps, fold_weight, padded_samples, set_chain_mode
    set_chain_mode(None if args.chain == 'default' else args.chain)
    h = VOC_CONFIG
    m = HifiGanGenerator(h)
    m.remove_weight_norm()
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():                                 # seeded fan-in scaled weights: a live signal path
        for n, p in m.named_parameters():
            if n.endswith('weight'):
                p.copy_(torch.randn(p.shape, generator=g) / (p[0].numel() if not n.startswith('ups') else p.shape[0] * 2) ** 0.5)
    m = m.to(device).eval()
    B, T = B_PER_GPU, T_FRAMES
    mel = torch.randn(B, 80, T, device=device, generator=torch.Generator(device=device).manual_seed(1234 + rank))
    el, wav = _row_time(lambda: m(mel), args, world, device, dist)
    assert wav.shape == (B, 1, T * 256) and bool(torch.isfinite(wav).all()), 'bad waveform'
    # the same forward replayed as ONE hipGraph (diffsinger_amd/graphs.py): the ~80 launches of a forward without their Python / ctypes issue cost
    from diffsinger_amd.graphs import GraphedForward
    gm = GraphedForward(m)
    wav_g = gm(mel)
    assert torch.equal(wav_g, wav), 'graph replay differs from the eager forward'
    el_g, _ = _row_time(lambda: gm(mel), args, world, device, dist)
    if rank == 0:
        # dominant kernel: the fused resblock stage of the 32-channel stage - k_voc_chain<32,1,4>: 3 resblocks x 3 conv pairs (18 convolutions,
        # kernels 3 / 7 / 11, dilations 1 / 3 / 5) over 8 x 65 536 samples as TWO launches (round 6: two resblocks merged into one grid, the third
        # forms the sum), 36 % of the forward - timed with events on the launch stream.  With the chains fused the row is bound by the fp32
        # matrix pipe, not by HBM any more (its traffic is the PMC figure).
        chained = args.chain != 'off' and m._chain_prep(1) is not None
        stage, ch = 1, 32
        L = T * 64
        merged = chained and args.chain == 'default' and m._merge_plan_for(stage, m._chain_prep(stage), B, L) is not None
        x = torch.randn(B, ch, padded_samples(L), device=device)
        x[:, :, L:] = 0
        launch = lambda: m._stage_resblocks(stage, x, L)
        launch()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        ev0.record()
        for _ in range(reps):
            launch()
        ev1.record()
        ev1.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        flop = sum(2 * B * L * ch * ch * k * 6 for k in h['resblock_kernel_sizes'])          # useful FLOPs: 6 convolutions per resblock
        alg_bytes = 2 * B * ch * L * 4                                                          # the stage input read once, its output written once
        pj, traffic_src = evidence('voc_chain_32ch_pmc.json')
        traffic = None
        if pj is not None:
            nl = pj.get('launches_per_stage', 2 if merged else 3)
            traffic = pj['hbm_bytes_per_launch']['total'] * nl
            traffic_src += (': FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024, average over the launches of the stage '
                            f'at this shape x {nl} launches')
        kname = ('k_voc_chain<32, 1, 4, true, true> x 2 (two resblocks merged into one launch, longest first; the third sums)' if merged else
                 'k_voc_chain<32, 1, 4, true> x 3 (one launch per resblock)' if chained else 'k_voc_conv<4,4> x 18 (chains off)')
        roof = {'bound': 'mfma', 'kernel': kname, 'achieved': flop / (ms * 1e-3) / 1e12, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': flop / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic if chained else None, 'traffic_unit': 'bytes/stage',
                'traffic_source': traffic_src, 'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_ms': ms, 'flop_per_launch': flop,
                'note': 'the resblock stage of the 32-channel stage (3 parallel resblocks x 3 conv pairs = 18 convolutions, kernels 3 / 7 / 11, dilations 1 / 3 / 5, '
                        '8 x 65 536 samples): "launch" = the stage = two chain launches (round 6: the kernel-11 and the kernel-3 resblock in ONE grid, longest '
                        'first, each to its own buffer; the kernel-7 resblock forms the sum - a launch costs whole rounds of co-resident workgroups, three '
                        'dependent launches paid three partial ones: profiles/r6_27_voc_tail_probe.jsonl; `--chain resblock` = the three-launch form); achieved = USEFUL fp32 '
                        'FLOPs of the 18 convolutions (2 x 32 x 32 x k per sample) / stage time incl. the torch.empty of the outputs and the ctypes calls; the '
                        'kernels execute 1.22 x that (a workgroup stages 512 samples and owns 512 - 2 x 12 / 36 / 60 of them: the receptive field of ITS '
                        'resblock) on one 32-row MFMA block, one LDS tile rewritten in place, two workgroups per CU (round 6; rounds 3-5: one launch per '
                        'stage, two tiles, one workgroup per CU, 1.33 x); 4 bytes / sample / channel in and out per launch: far above the ridge (20 FLOP/B), '
                        'the matrix pipe is the roof - at the 2.1 GHz the chip holds under fp32 MFMA load its peak is 137.6, not 157.3 TFLOP/s'}
        fpf = vocoder_flop_per_frame(h)
        value = world * B * T * args.steps / el
        res = {'metric': 'mel-frames/sec (whole node) through the HiFi-GAN generator, 80-bin mel -> 24 kHz waveform, hop 256, T=1024', 'value': value,
               'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': el / args.steps * 1e3,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': f'SURVEY 8 row f2: HifiGanGenerator of configs/tts/hifigan.yaml (128 -> 8 channels, x256), batch={B} x T={T} mel '
                                      f'frames per GPU -> {B} x {T * 256} samples', 'resblock_chains': ('two chain launches per stage for the 32 / 16 / 8-channel stages (k_voc_chain, one LDS tile rewritten in place; two resblocks merged into one grid, the third sums), the 64-channel stage level by level (k_voc_conv_multi: 8 launches)' if merged else 'one chain launch per resblock (three per stage) for the 32 / 16 / 8-channel stages (k_voc_chain, one LDS tile rewritten in place)') if chained else 'off: one launch per convolution',
                          'sharding': 'replicas (no exchange step in this row)'},
               'roofline': roof, 'model_tflops': world * B * T * fpf * args.steps / el / 1e12, 'flop_per_mel_frame': fpf,
               'x_realtime_24k': value * 256 / 24000,
               'hipgraph_replay': {'ms_per_step': el_g / args.steps * 1e3, 'value': world * B * T * args.steps / el_g,
                                   'note': 'the same forward captured once and replayed as one hipGraph (diffsinger_amd.graphs.GraphedForward); '
                                           'bit-identical output; `value` above is the eager figure'}}
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline_vocoder()
            res['speedup_vs_cpu_baseline'] = value / res['cpu_baseline']['value']
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


F_TRAIN_FWD = 26_427_392                                   # GEMM FLOP / frame of one DiffNet forward in training (conditioner projection not hoisted)


def cpu_baseline_train(budget_s: float = 20.0):
    """torch autograd on the oracle (the reference's own arithmetic and graph) on the host cores: p_losses forward + backward on
    1 x 256 frames at a time until ~budget_s."""
    from oracle import diffnet_oracle as O
    from diffsinger_amd.synth import presets
    pre = presets()[PRESET]
    cfg = O.NetConfig(80, 256, 256, 20, 1)
    p = {k: v.clone().requires_grad_(True) for k, v in O.init_diffnet_params(cfg, 1234, 0.02).items()}
    sch = O.make_schedule(O.linear_beta_schedule(pre['timesteps'], pre['max_beta']))
    g = torch.Generator().manual_seed(7)
    B, T = 1, 256
    x0 = torch.clamp(torch.randn(B, 1, 80, T, generator=g) * 0.5, -1, 1)
    noise = torch.randn(B, 1, 80, T, generator=g)
    cond = torch.randn(B, 256, T, generator=g)
    t = torch.tensor([37])
    avail = host_cpus()
    cores = min(avail, 32)
    torch.set_num_threads(cores)

    def step():
        for v in p.values():
            v.grad = None
        loss = (noise - O.diffnet_forward(p, cfg, O.q_sample(sch, x0, t, noise), t, cond)).abs().mean()
        loss.backward()

    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    return {'value': n * B * T / el, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} p_losses forward + backward steps of {B} x {T} frames (torch autograd on oracle/diffnet_oracle.py) in {el:.1f}s on {cores} host '
                      f'threads ({avail} CPUs available)'}


def main_train(args):
    """Row f3: `steps` x (q_sample + DiffNet forward + L1 + backward) on the HIP training operators, 8 x 1024 frames per GPU (no optimiser,
    no gradient exchange: replicas)."""
    world, rank, device, dist = _row_setup()
    gd, pre = build_model(device)
    gd.train()
    net = gd.denoise_fn
    B, T = B_PER_GPU, T_FRAMES
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    x0 = torch.randn(B, 1, 80, T, device=device, generator=g).clamp(-1, 1)
    cond = torch.randn(B, T, 256, device=device, generator=g).transpose(1, 2)
    t = torch.randint(0, K_STEPS, (B,), device=device, generator=g)
    noise = torch.randn(B, 1, 80, T, device=device, generator=g)

    def step():
        net.zero_grad(set_to_none=True)
        loss = gd.p_losses(x0, t, cond, noise=noise)
        loss.backward()
        return loss

    el, loss = _row_time(step, args, world, device, dist)
    assert bool(torch.isfinite(loss)), 'non-finite loss'
    if rank == 0:
        # dominant kernel of the fused stack: k_tr_wgrad, timed live AS IT RUNS INSIDE THE STEP - the library brackets every weight-gradient
        # launch with events on its stream (dsf_wgrad_probe); one launch = the weight gradients of two layers (40 tiles x 6 frame splits)
        from diffsinger_amd import train_fused
        fused = train_fused.enabled() and train_fused.supported(net)
        roof = None
        if fused:
            train_fused.wgrad_probe(True)
            for _ in range(3):
                step()
            torch.cuda.synchronize(device)
            ms_tot, n_launch, fl = train_fused.wgrad_probe_read()
            train_fused.wgrad_probe(False)
            ms = ms_tot / max(n_launch, 1)
            flop = fl / max(n_launch, 1)
            achieved = flop / (ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': 'k_tr_wgrad<false>', 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': None, 'avg_launch_ms': ms, 'flop_per_launch': flop, 'launches_timed': n_launch,
                    'note': 'every weight gradient of two residual layers per launch: the dilated convolution as the Winograd F(2,3) DUAL (four products over '
                            'frame PAIRS per 128-row tile instead of three over the frames: 32 products x 4 frame splits) + conditioner and output projection '
                            '(16 tiles x 8 splits) = 256 workgroups; flop_per_launch = EXECUTED (a dual product contracts T / 2 pairs); average over the '
                            'launches of 3 steps, events on the launch stream around each launch (they include the gap to the previous kernel); the two '
                            'reduction kernels behind it (split-K sums + the dual\'s back-transform) are not part of the figure.  Kernel times of the whole '
                            'step: profiles/r6_*_train_kernel_stats.txt'}
        else:
            roof = {'bound': 'mfma', 'kernel': 'k_fs_conv<2> (operator path)', 'achieved': None, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': None,
                    'traffic': None, 'note': 'the fused stack is switched off (DSD_TRAIN_FUSED=0) or does not cover this DiffNet'}
        pm, pm_src = evidence('train_wgrad_pmc.json')      # fabric-side bytes of the same launch (two layers), from separate --pmc passes over the training step
        if fused:
            roof['traffic_unit'] = 'bytes/launch'
            roof['traffic'] = pm['hbm_bytes_per_launch']['total'] if pm is not None else None
            roof['traffic_source'] = pm_src + ('' if pm is None else ': FETCH_SIZE x 2 + WRITE_SIZE of k_tr_wgrad inside the training step; algorithmic: 118 MB of '
                                              "operands (da, y, cond, g, dx', dskip of two layers) + 31.5 MB of split-K partials per launch")
        value = world * B * T * args.steps / el
        res = {'metric': 'frames/sec (whole node) through one denoiser training step: q_sample + DiffNet forward + L1 + backward, T=1024', 'value': value,
               'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': el / args.steps * 1e3,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': f'SURVEY 8 row f3: GaussianDiffusion.p_losses (usr/diff/shallow_diffusion_tts.py:213-231) of the DiffSpeech denoiser, '
                                      f'batch={B} x T={T} per GPU, forward + backward on the ' + ('fused residual-stack kernels (csrc/train_kernels.hpp)' if fused else 'HIP training operators'), 'preset': PRESET,
                          'conv': (train_fused.stack_conv() + ' (the persistent forward AND the transposed convolution of the backward)') if fused else 'direct',
                          'optimizer': 'not included (diffsinger_amd/train_dist.py)', 'sharding': 'replicas (no gradient exchange in this bench)'},
               'roofline': roof, 'model_tflops_gemm': world * B * T * 3 * F_TRAIN_FWD * args.steps / el / 1e12,
               'frac_row_executed': world * B * T * F_TRAIN_EXEC * args.steps / el / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline_train()
            res['speedup_vs_cpu_baseline'] = value / res['cpu_baseline']['value']
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------
# Compact figures for the rows AROUND the path, the off-shapes and the in-service noise path - inside the driver-run N = 1 line (VERDICT r5,
# item 2: until round 6 every f-row number was a builder-kept file).  Each uses the workload of its own `--row` bench.
# ------------------------------------------------------------------------------------------------------------
F_CONV_LAYERS = L_LAYERS * 2 * 512 * 768                   # direct-form FLOP / frame of the 20 dilated convolutions (15 728 640)
# training step, EXECUTED: forward, data gradients and (round 6: the Winograd dual over frame pairs) weight gradients all run the dilated convolution
# on 2/3 of its multiplications; the 1 x 1 projections are what they are
F_TRAIN_EXEC = 3 * (F_TRAIN_FWD - F_CONV_LAYERS // 3)
F_TRAIN_EXEC_TAPS = 2 * (F_TRAIN_FWD - F_CONV_LAYERS // 3) + F_TRAIN_FWD      # dsf_set_wgrad_dual(0): the weight gradient as three tap products (round 5)


def _event_ms(fn, warm, steps):
    for _ in range(warm):
        fn()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        out = fn()
    ev1.record()
    ev1.synchronize()
    return ev0.elapsed_time(ev1) / steps, (time.perf_counter() - t0) * 1e3 / steps, out


def _event_ms_best(fn, warm, steps, windows=2):
    """_event_ms over `windows` back-to-back windows of `steps` steps; returns the window with the smallest device time and the list of all of
    them.  One window in four read 25 % high on one box (r6_62: the FastSpeech2 row 3.49 ms in the bench line, 2.70 from `--row fs2` in the same
    call) - a short row right behind the seconds of full-power work in front of it; both windows are in the line (`ms_windows`)."""
    best, all_ms = None, []
    for i in range(windows):
        r = _event_ms(fn, warm if i == 0 else 1, steps)
        all_ms.append(r[0])
        if best is None or r[0] < best[0]:
            best = r
    return best[0], best[1], best[2], all_ms


def fs2_forward_flops(m, step):
    """GEMM FLOPs of ONE forward of the HIP FastSpeech2, counted at its operators as it runs: every convolution / linear (2 B T Co Ci K) and
    every attention core (QK^T + PV: 4 B T^2 C)."""
    from diffsinger_amd import fs2 as F2
    total = [0]
    conv0, att0 = F2.conv1d_cm, F2.attention_cm

    def conv(x, T, weight, *a, **kw):
        total[0] += 2 * x.shape[0] * T * weight.shape[0] * weight.shape[1] * (weight.shape[2] if weight.dim() == 3 else 1)
        return conv0(x, T, weight, *a, **kw)

    def att(qkv, T, *a, **kw):
        total[0] += 4 * qkv.shape[0] * T * T * (qkv.shape[1] // 3)
        return att0(qkv, T, *a, **kw)

    F2.conv1d_cm, F2.attention_cm = conv, att
    try:
        step()
    finally:
        F2.conv1d_cm, F2.attention_cm = conv0, att0
    return total[0]


def quick_rows(gd, device, warm=5, steps=30):
    """rows: {fs2, vocoder, train}: ms per step (HIP events around `steps` steps after `warm`) and frac_row = FLOPs of the WHOLE row / time /
    157.3 TFLOP/s - beside, not instead of, the dominant kernel's fraction the `--row` lines report.  `ms` = the faster of TWO windows of 30 timed
    steps behind 5 warm ones (`ms_windows` lists both; ~0.7 s for the three rows): with 3 + 10 the first steps of the window ran before the host was ahead of the device - the training row read 5.0-5.07 ms for
    the 4.80-4.83 of a longer window on the same box, FastSpeech2 2.79-2.83 for 2.74-2.76 (profiles/r6_50_quick_rows_window.txt)."""
    B, T = B_PER_GPU, T_FRAMES
    rows = {}
    # f1: FastSpeech2 forward, teacher-forced
    try:
        m, hp, tok, kw = _fs2_setup(B, T // 8, 8, device)
        m = m.to(device)
        tok = tok.to(device)
        kw = {k: v.to(device) for k, v in kw.items()}
        step = lambda: m(tok, infer=True, **{k: (v.clone() if k == 'f0' else v) for k, v in kw.items()})
        with torch.no_grad():
            flop = fs2_forward_flops(m, step)
            ms, ms_wall, r, wins = _event_ms_best(step, warm, steps)
        assert r['mel_out'].shape == (B, T, 80) and bool(torch.isfinite(r['mel_out']).all())
        rows['fs2'] = {'ms': ms, 'ms_windows': wins, 'ms_wall': ms_wall, 'flop_row': flop, 'frac_row': flop / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                       'what': f'FastSpeech2 forward, teacher-forced, {B} x {T} mel frames (= `--row fs2`); flop_row = every convolution / linear / attention '
                               'core of the forward (useful GEMM FLOPs, counted at the operators)'}
        del m
    except Exception as e:                                  # report, do not hide
        rows['fs2'] = {'error': repr(e)[:300]}
    # f2: HiFi-GAN generator
    try:
        from diffsinger_amd.vocoder import HifiGanGenerator
        h = VOC_CONFIG
        m = HifiGanGenerator(h)
        m.remove_weight_norm()
        g = torch.Generator().manual_seed(1234)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.endswith('weight'):
                    p.copy_(torch.randn(p.shape, generator=g) / (p[0].numel() if not n.startswith('ups') else p.shape[0] * 2) ** 0.5)
        m = m.to(device).eval()
        mel = torch.randn(B, 80, T, device=device, generator=torch.Generator(device=device).manual_seed(1234))
        with torch.no_grad():
            ms, ms_wall, wav, wins = _event_ms_best(lambda: m(mel), warm, steps)
        assert wav.shape == (B, 1, T * 256) and bool(torch.isfinite(wav).all())
        flop = B * T * vocoder_flop_per_frame(h)
        rows['vocoder'] = {'ms': ms, 'ms_windows': wins, 'ms_wall': ms_wall, 'flop_row': flop, 'frac_row': flop / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                           'what': f'HifiGanGenerator of configs/tts/hifigan.yaml, {B} x {T} mel frames -> {B} x {T * 256} samples (= `--row vocoder`); flop_row = '
                                   'USEFUL FLOPs of every convolution (the fused chains execute 1.1-1.2 x that: receptive-field overlap)'}
        del m, mel, wav
    except Exception as e:
        rows['vocoder'] = {'error': repr(e)[:300]}
    # f3: one p_losses forward + backward of the denoiser
    try:
        gd.train()
        net = gd.denoise_fn
        g = torch.Generator(device=device).manual_seed(1234)
        x0 = torch.randn(B, 1, 80, T, device=device, generator=g).clamp(-1, 1)
        cond = torch.randn(B, T, 256, device=device, generator=g).transpose(1, 2)
        t = torch.randint(0, K_STEPS, (B,), device=device, generator=g)
        noise = torch.randn(B, 1, 80, T, device=device, generator=g)

        def step():
            net.zero_grad(set_to_none=True)
            loss = gd.p_losses(x0, t, cond, noise=noise)
            loss.backward()
            return loss

        ms, ms_wall, loss, wins = _event_ms_best(step, warm, steps)
        assert bool(torch.isfinite(loss))
        from diffsinger_amd import train_fused
        fused = train_fused.enabled() and train_fused.supported(net)
        wino = fused and train_fused.stack_conv() == 'wino'
        flop = B * T * (F_TRAIN_EXEC if wino else 3 * F_TRAIN_FWD)
        rows['train'] = {'ms': ms, 'ms_windows': wins, 'ms_wall': ms_wall, 'flop_row': flop, 'frac_row': flop / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         'frac_row_direct_accounting': B * T * 3 * F_TRAIN_FWD / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         'what': f'q_sample + DiffNet forward + L1 + backward, {B} x {T} frames, no optimiser (= `--row train`); flop_row = EXECUTED GEMM FLOPs '
                                 '(forward, data gradients and weight gradients with the dilated convolution as Winograd F(2,3) / its dual over frame pairs)'}
        net.zero_grad(set_to_none=True)
    except Exception as e:
        rows['train'] = {'error': repr(e)[:300]}
    finally:
        gd.eval()
    return rows


def offshape(gd, device, K=K_STEPS):
    """The shapes that do NOT fill the chip: 1 x 512 - the reference's own evaluation shape, configs/tts/fs2.yaml:70 `max_eval_sentences: 1` -
    and 3 x 1550 (147 of 256 CUs have a tile), K = 100 DDPM, explicit noise, one timed pass after one warm pass.  frac = executed FLOPs / time / 157.3."""
    out = {}
    for B, T in ((1, 512), (3, 1550)):
        try:
            g = torch.Generator(device=device).manual_seed(77 + T)
            cond = torch.randn(B, T, 256, device=device, generator=g).transpose(1, 2)
            x_T = torch.randn(B, 1, 80, T, device=device, generator=g)
            noise = torch.randn(K, B, 1, 80, T, device=device, generator=g)
            run = lambda: gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0)
            eng = gd._engine(cond)
            eng.set_loop_mode(2)                             # the library's own choice of path (the roofline leg above forces the persistent loop)
            with torch.no_grad():
                ms, ms_wall, mel = _event_ms(run, 1, 1)
            assert bool(torch.isfinite(mel).all())
            persistent = eng.loop_mode() == 1
            wino = eng.conv_mode() == 1
            f_exec = F_EVAL_EXEC_WINO if wino else F_EVAL_EXEC
            gsplit = eng.lat_split()
            out[f'{B}x{T}'] = {'ms': ms, 'mel_frames_per_s': B * T / (ms * 1e-3), 'frac': B * T * K * f_exec / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                               'flop_per_frame_per_evaluation': f_exec, 'tiles': B * ((T + 31) // 32),
                               'path': ('k_loop_wino (persistent)' if wino else 'k_loop (persistent)') if persistent else
                                       (f"row-split latency kernels, G = {gsplit}{', Winograd conv node' if wino else ''} (hipGraph)" if gsplit else 'k_layer (hipGraph)')}
        except Exception as e:
            out[f'{B}x{T}'] = {'error': repr(e)[:300]}
    out['what'] = ('K = 100 DDPM per call (prepare + loop + denorm), explicit noise; frac = executed FLOP / time / 157.3 TFLOP/s with the convolution counted '
                   'as the path executes it (dsd_get_conv_mode: Winograd F(2,3) on the persistent loop and on the latency kernels of G = 2 / 4 / 8)')
    return out


def inservice_noise(gd, cond, x_T, K, frames, steps=3):
    """The rate a DiffSpeechTask user gets: the reference draws torch.randn inside every p_sample (usr/diff/shallow_diffusion_tts.py:38-41,
    159-166); here `inference(noise=None)` draws in the loop's head (Philox + Box-Muller in the kernel) - the timed batch of the headline
    with that path instead of pre-drawn noise."""
    run = lambda: gd.inference(cond, x_T=x_T, noise=None, K_step=K, pndm_speedup=0, noise_seed=1234)
    with torch.no_grad():
        ms, ms_wall, mel = _event_ms(run, 1, steps)
    assert bool(torch.isfinite(mel).all())
    return {'value': frames / (ms * 1e-3), 'unit': 'mel-frames/s', 'ms_per_step': ms, 'steps': steps,
            'what': 'the headline step with noise=None: every p_sample draws its N(0,1) noise inside the kernel (Philox 4x32-10 + Box-Muller, csrc/dsd_kernels.hpp) '
                    'instead of reading a pre-drawn [K,B,1,M,T] tensor; not bit-comparable with the reference (different generator), same distribution '
                    '(tests/test_gpu_surfaces.py)'}


def _fs2_setup(B, T_txt, frames_per_phone, device, seed=7):
    """DiffSpeech FastSpeech2 (preset lj_ds_beta6) with seeded weights and teacher-forced inputs: B x T_txt phones x frames_per_phone frames."""
    import diffsinger_amd
    from diffsinger_amd import fs2, hparams
    hparams.clear()
    diffsinger_amd.use_preset(PRESET)
    torch.manual_seed(1234)
    m = fs2.FastSpeech2(63, 80).eval()
    g = torch.Generator().manual_seed(seed)
    tok = torch.randint(1, 63, (B, T_txt), generator=g)
    T = T_txt * frames_per_phone
    mel2ph = (torch.arange(T) // frames_per_phone + 1)[None].repeat(B, 1)
    kw = dict(mel2ph=mel2ph, f0=torch.rand(B, T, generator=g) * 2 + 6.5, uv=torch.zeros(B, T))
    return m, dict(hparams), tok, kw


def cpu_baseline_fs2(budget_s: float = 20.0):
    """The FastSpeech2 oracle (oracle/fs2_oracle.py = the reference's own torch arithmetic) on the host cores, 1 x 256 frames at a time."""
    from oracle import fs2_oracle as FO
    m, hp, tok, kw = _fs2_setup(1, 32, 8, 'cpu')
    p = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v.detach().clone()) for k, v in m.state_dict().items()}
    avail = host_cpus()
    cores = min(avail, 32)
    torch.set_num_threads(cores)
    T = kw['mel2ph'].shape[1]
    with torch.no_grad():
        run = lambda: FO.fs2_forward(p, hp, tok, **{k: v.clone() for k, v in kw.items()})
        run()
        n, t0 = 0, time.perf_counter()
        while True:
            run()
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 4096:
                break
    return {'value': n * T / el, 'unit': 'mel-frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} forwards of 1 x {T} mel frames (oracle/fs2_oracle.fs2_forward, teacher-forced) in {el:.1f}s on {cores} host threads '
                      f'({avail} CPUs available)'}


def main_fs2(args):
    """Row f1: `steps` teacher-forced forwards of the HIP FastSpeech2 over 8 x 1024 mel frames per GPU (replicas: no exchange step in this row)."""
    world, rank, device, dist = _row_setup()
    B, T = B_PER_GPU, T_FRAMES
    m, hp, tok, kw = _fs2_setup(B, T // 8, 8, device, seed=7 + rank)
    m = m.to(device)
    tok = tok.to(device)
    kw = {k: v.to(device) for k, v in kw.items()}
    if args.conv_split != -1:
        from diffsinger_amd import fs2 as _fs2
        _fs2.set_conv_split(args.conv_split)

    def step():
        return m(tok, infer=True, **{k: (v.clone() if k == 'f0' else v) for k, v in kw.items()})

    el, r = _row_time(step, args, world, device, dist)
    assert r['mel_out'].shape == (B, T, 80) and bool(torch.isfinite(r['mel_out']).all()), 'bad mel'
    # the same teacher-forced forward replayed as ONE hipGraph (diffsinger_amd/graphs.py)
    from diffsinger_amd.graphs import GraphedForward
    gm = GraphedForward(lambda tok_, mel2ph, f0, uv: m(tok_, infer=True, mel2ph=mel2ph, f0=f0, uv=uv))
    hip_graph = None
    try:
        rg = gm(tok, kw['mel2ph'], kw['f0'], kw['uv'])
        same = torch.equal(rg['mel_out'], r['mel_out'])
        el_g, _ = _row_time(lambda: gm(tok, kw['mel2ph'], kw['f0'], kw['uv']), args, world, device, dist)
        hip_graph = {'ms_per_step': el_g / args.steps * 1e3, 'value': world * B * T * args.steps / el_g, 'bit_identical_to_eager': bool(same),
                     'note': 'the same forward captured once and replayed as one hipGraph (diffsinger_amd.graphs.GraphedForward); `value` above is '
                             'the eager figure'}
    except Exception as e:                                    # report, do not hide
        hip_graph = {'error': repr(e)[:300]}
    if rank == 0:
        # dominant kernel: k_fs_conv<2> as the k = 9 conv of the feed-forward block (256 -> 1024), one launch timed with events on the launch stream
        from diffsinger_amd.fs2 import conv1d_cm, PackedWeight, padded_frames
        w = torch.randn(1024, 256, 9, device=device) * (256 * 9) ** -0.5
        bias = torch.zeros(1024, device=device)
        x = torch.randn(B, 256, padded_frames(T), device=device)
        pk = PackedWeight()
        launch = lambda: conv1d_cm(x, T, w, pk, bias, scale=9 ** -0.5, act='gelu')
        launch()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            launch()
        ev1.record()
        ev1.synchronize()
        ms = ev0.elapsed_time(ev1) / 20
        flop = 2 * 1024 * 256 * 9 * B * T
        achieved = flop / (ms * 1e-3) / 1e12
        roof = {'bound': 'mfma', 'kernel': 'k_fs_conv<2>', 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': None, 'avg_launch_ms': ms, 'flop_per_launch': flop,
                'algorithmic_bytes_per_launch': 4 * B * T * (256 + 1024) + 4 * 1024 * 256 * 9,
                'note': 'ffn_1 of TransformerFFNLayer (Conv1d 256 -> 1024, k = 9, * k**-0.5, gelu fused): the largest contraction of the model; eager launch '
                        'incl. the output allocation and the ctypes call'}
        pj, pj_src = evidence('fs2_ffn1_pmc.json')            # counters of THIS launch shape inside the whole forward (tools/gpu.sh fs2pmc)
        roof['traffic_unit'] = 'bytes/launch'
        roof['traffic'] = float(pj['hbm_bytes_per_launch']['total']) if pj is not None else None
        roof['traffic_source'] = pj_src + ('' if pj is None else ': the k_fs_conv<2> dispatches of at least 250 us of `bench.py --row fs2` = the four mel-rate '
                                          'ffn_1 launches per forward; FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024')
        value = world * B * T * args.steps / el
        res = {'metric': 'mel-frames/sec (whole node) through FastSpeech2 (encoder, predictors, length regulator, decoder, mel_out), teacher-forced, T=1024',
               'value': value, 'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': el / args.steps * 1e3,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': f'SURVEY 8 row f1: FastSpeech2 of {PRESET} (4 + 4 FFT blocks, hidden 256), batch={B} x {T // 8} phones x 8 frames = '
                                      f'{T} mel frames per GPU, mel2ph / f0 / uv supplied', 'preset': PRESET, 'sharding': 'replicas (no exchange step in this row)'},
               'roofline': roof, 'hipgraph_replay': hip_graph}
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline_fs2()
            res['speedup_vs_cpu_baseline'] = value / res['cpu_baseline']['value']
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the labelled split-precision line (`secondary`) of the N = 1 run (~45 s, mostly its fp64 oracle)')
    ap.add_argument('--no-cfg5-shard', action='store_true', help='skip the configs[4] shard leg of the N = 1 line (~5 s)')
    ap.add_argument('--no-extras', action='store_true', help='skip `rows` / `offshape` / `inservice_noise` of the N = 1 line (~6 s)')
    ap.add_argument('--tile', type=int, default=0, help='frames per workgroup of the layer kernel (0 auto, 32, 64)')
    ap.add_argument('--row', choices=['path', 'vocoder', 'train', 'fs2'], default='path',
                    help='path: the headline hot path (default); vocoder: SURVEY 8 row f2; train: row f3 (denoiser p_losses forward + backward); '
                         'fs2: row f1 (FastSpeech2 forward, teacher-forced)')
    ap.add_argument('--config', type=int, choices=[2, 5], default=0,
                    help='BASELINE configuration of the headline run: 2 = configs[1] (8 x T=1024 per GPU; the default at --gpus 1), 5 = configs[4] '
                         '(512 utterances x T=2048 sharded across the GPUs, strong scaling; the default at --gpus > 1)')
    ap.add_argument('--conv-split', type=int, choices=[-1, 0, 1], default=-1,
                    help='--row fs2 A/B: kernel choice of the FastSpeech2 convolutions (-1 by grid size = the product, 0 never the K-split kernel, 1 always)')
    ap.add_argument('--chain', choices=['default', 'off', 'stage', 'resblock', 'pair', 'merged'], default='default',
                    help='--row vocoder: how the ResBlock1 chains are launched (diffsinger_amd.vocoder.set_chain_mode); default = by channel count')
    ap.add_argument('--conv', choices=['winograd', 'direct'], default=None,
                    help='convolution of the persistent loop: winograd F(2,3) (the default of the library) or the direct K = 768 form (A/B, rounds 1-4)')
    ap.add_argument('--touch', type=int, default=-1, help='A/B: steps the L2 touch of the Winograd weight stream runs in front (0 = off, -1 = library default)')
    ap.add_argument('--split', action='store_true', help='EXPERIMENT: residual layers as six bf16 plane products per fp32 product (fp32-class accuracy) '
                                                          'on the bf16 matrix pipe; per-layer kernel path; the JSON line says so in dtype / config')
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args)
    if args.row == 'vocoder':
        return main_vocoder(args)
    if args.row == 'train':
        return main_train(args)
    if args.row == 'fs2':
        return main_fs2(args)
    return main_path(args)


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU, exactly the command line
    the driver uses - torch.distributed.run, rendezvous on 127.0.0.1) and pass their output through; rank 0 prints the JSON line."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not os.environ.get('DSD_BENCH_SPAWN_ANYWAY'):          # (the variable lets the CPU suite exercise the spawn itself)
        raise SystemExit(f'bench.py --gpus {n}: needs an MI355X node with at least {n} visible devices (torch sees {have})')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench.py: launching ' + ' '.join(cmd), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


CFG5_UTTS, CFG5_T, CFG5_MICRO = 512, 2048, 16         # BASELINE configs[4]: 512 synthetic utterances x T=2048, K=100
BASELINE_METRIC = 'mel-frames/sec (whole node) at K=100 DDPM, 80-bin, T=1024'      # BASELINE.json `metric`; the workload of a line is config.workload


def cfg5_workload(gd, rank, world, device, *, n_utts=CFG5_UTTS, T=CFG5_T, micro=CFG5_MICRO, K=K_STEPS, M=80, H=256, group=None):
    """BASELINE configs[4] exactly as the N-GPU run executes it: the SAME n_utts utterances x T at every N (strong scaling); rank r takes
    utterances r::W (tasks/tts/tts.py:85-88), samples them in micro-batches with zero communication, ONE gather collates the mels on rank 0
    (the reference collates through the filesystem, tasks/tts/fs2.py:414-431).  Returns (step, local_only, mine, first_cond, x_T, noise):
    step() = the timed unit (shard -> micro-batches -> gather -> [n_utts, T, M] on rank 0, None elsewhere); local_only() = this rank's shard
    without the gather (the same-workload N = 1 rate per GPU).  `gd` needs only `.inference(cond, **kw)` and `.mel_bins`
    (tests/test_bench_cfg5_gloo.py drives this function at world 2 over gloo with a stand-in sampler)."""
    from diffsinger_amd.dist import sharded_inference, shard_indices
    mine = shard_indices(n_utts, rank, world)
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    gens = {i: torch.Generator(device=device).manual_seed(50_000 + i) for i in mine}
    utt_conds = [None] * n_utts
    for i in mine:                                       # utterance i's conditioner depends on i only, not on the sharding
        utt_conds[i] = torch.randn(T, H, device=device, generator=gens[i]).t()
    x_T = torch.randn(micro, 1, M, T, device=device, generator=g)
    noise = torch.randn(K, micro, 1, M, T, device=device, generator=g)          # 1.05 GB at the full size, shared by the micro-batches
    kw = dict(x_T=lambda idx: x_T[:len(idx)], noise=lambda idx: noise[:, :len(idx)].contiguous() if len(idx) < micro else noise,
              K_step=K, pndm_speedup=0)

    def step():
        out = sharded_inference(gd, utt_conds, micro_batch=micro, dst=0, group=group, **kw)
        if rank == 0:
            assert out is not None and tuple(out.shape) == (n_utts, T, M), None if out is None else tuple(out.shape)
        else:
            assert out is None or world == 1
        return out

    def local_only():
        outs = []
        for s0 in range(0, len(mine), micro):
            idx = mine[s0:s0 + micro]
            outs.append(gd.inference(torch.stack([utt_conds[i] for i in idx]), **{k: (v(idx) if callable(v) else v) for k, v in kw.items()}))
        return outs

    first_cond = torch.stack([utt_conds[i] for i in mine[:micro]]) if mine else None
    return step, local_only, mine, first_cond, x_T, noise


def _time_steps(step, args, world, device, dist):
    """The bench contract: W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; max over ranks."""
    out = None
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    elt = torch.tensor([el], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elt, op=dist.ReduceOp.MAX)
    return float(elt.item()), out


def main_path(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit(f'bench.py needs an MI355X (no CPU fallback for the product path) [rank {rank} of {world}]')
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree')
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f'rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} devices are visible')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device, timeout=GROUP_TIMEOUT)
        assert dist.get_world_size() == args.gpus and dist.get_backend() == 'nccl'
    cfg = args.config or (5 if world > 1 else 2)

    gd, pre = build_model(device)
    K, M = K_STEPS, 80
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    if cfg == 2:
        # BASELINE configs[1]: batch of 8 x T=1024 on this GPU.  TWO conditioner tensors alternate from step to step, so that every timed
        # step really contains dsd_prepare (cond re-layout + the hoisted conditioner projection k_condproj): DiffNet.bind_cond skips it
        # when it is handed the very tensor it prepared last
        B, T = B_PER_GPU, T_FRAMES
        conds = [torch.randn(B, T, 256, device=device, generator=g).transpose(1, 2) for _ in range(2)]     # [B,H,T] views, like the reference
        x_T = torch.randn(B, 1, M, T, device=device, generator=g)
        noise = torch.randn(K, B, 1, M, T, device=device, generator=g)
        frames_per_step = world * B * T
        count = [0]

        def step():
            count[0] += 1
            return gd.inference(conds[count[0] & 1], x_T=x_T, noise=noise, K_step=K, pndm_speedup=0)
        cond = conds[0]
    else:
        B, T = CFG5_MICRO, CFG5_T
        step, local_only, mine, cond, x_T, noise = cfg5_workload(gd, rank, world, device)
        frames_per_step = CFG5_UTTS * T
    eng = gd._engine(cond)
    eng.set_layer_tile(args.tile)
    if args.conv or args.touch >= 0:
        eng.set_conv_mode(args.conv or 'winograd', args.touch)
    if args.split:
        eng.set_split_mode(True)

    el, out = _time_steps(step, args, world, device, dist)
    if rank == 0:
        assert out is not None and bool(torch.isfinite(out).all()), 'non-finite mel'
        assert out.shape == ((CFG5_UTTS, T, M) if cfg == 5 else (B, T, M)), out.shape
    scale_ref = None
    if cfg == 5 and world > 1:
        # the same-workload N = 1 reference for whoever divides the N > 1 value: rank 0 samples ITS shard once more, alone and without the
        # gather, while the other ranks wait - the per-GPU rate of configs[4]; N x this is the ideal of the line above it
        if rank == 0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            local_only()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            scale_ref = {'value': len(mine) * T / dt, 'unit': 'mel-frames/s', 'n_gpus': 1,
                         'what': f"rank 0's shard ({len(mine)} of {CFG5_UTTS} utterances x T={T}, micro-batches of {B}) sampled alone after the timed "
                                 f'region, no gather: the per-GPU rate of BASELINE configs[4]; ideal at N={world} = {world} x this'}
        dist.barrier()
    per_rank = None
    if cfg == 5:
        # every rank reports what IT did: its RCCL world, its shard (utterances, frames) and the wall time of that shard sampled once more,
        # all ranks at the same time, no gather - whoever reads the line sees imbalance or a slow device directly
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        local_only()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mine_rep = {'rank': rank, 'rccl_world': dist.get_world_size() if world > 1 else 1, 'backend': dist.get_backend() if world > 1 else None,
                    'device': torch.cuda.get_device_name(device), 'utterances': len(mine), 'frames': len(mine) * T, 'shard_s': dt,
                    'mel_frames_per_s': len(mine) * T / dt}
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine_rep)
        else:
            per_rank = [mine_rep]

    # roofline of the dominant kernel, measured live with HIP events on the launch stream (torch's current stream IS the
    # stream every dsd_* call is enqueued on).  Persistent path: the kernel is k_loop, ONE launch = the whole 100-step loop
    # for one chunk of whole utterances (at most one workgroup per CU).
    roof = None
    if rank == 0:
        eng.prepare(cond)
        frames = B * T
        persistent = eng.loop_mode() == 1
        if persistent:
            xs = x_T[:, 0].contiguous().clone()
            nz = noise[:, :, 0]
            eng.sample_ddpm(xs, nz, K)                                       # warm (plan upload)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            ev0.record()
            for _ in range(reps):
                eng.sample_ddpm(xs, nz, K)
            ev1.record()
            ev1.synchronize()
            assert eng.loop_timeouts() == 0, 'persistent loop: an inter-workgroup wait timed out'
            launches = eng.loop_launches()
            ms_call = ev0.elapsed_time(ev1) / reps                           # the whole call: every k_loop launch of the batch
            ms = ms_call / launches
            wino = eng.conv_mode() == 1 and not args.split
            f_exec = F_EVAL_EXEC_WINO if wino else F_EVAL_EXEC
            # whole-call FLOPs over whole-call time (chunks of whole utterances: the last launch of a batch may cover fewer frames than the
            # others - a per-launch figure from frames // launches would mislabel an average); per-launch numbers are AVERAGES
            flop = frames * K * f_exec / launches
            achieved = frames * K * f_exec / (ms_call * 1e-3) / 1e12
            frames_l = frames / launches
            kname = 'k_loop_split<1, %s>' % os.environ.get('DSD_SPLIT_W', '2') if args.split else 'k_loop_wino<1, 4>' if wino else 'k_loop<1>'
            w_layer = (2 * 1024 * 1024 + 512 * 1024) if wino else 2 * 1024 * 1024      # weight stream of a layer: 4 (3) x 512 KiB of conv + 512 KiB of out-projection
            alg_bytes = int(K * (frames * (20 * 2048 + 2 * 320 + 320) + launches * L_LAYERS * w_layer + frames // 32 * L_LAYERS * 2 * 16384) / launches)
            note = (f'one launch = the whole K=100 reverse loop (100 x (20 residual layers + head + sampler update + next input '
                    f'projection)) for a chunk of whole utterances; this batch of {B} x {T} = {launches} launch(es) of on average {frames_l:.0f} frames; '
                    'achieved = EXECUTED fp32 FLOPs of the WHOLE call / its duration (HIP events on the launch stream), per-launch fields are '
                    + ('averages; executed = 15 810 560 FLOP / frame / evaluation: the dilated convolution as Winograd F(2,3) along the frame axis (four '
                       'K = 256 products per output pair instead of six; fp32 in, exact-fp32 MFMA, fp32 transforms), conditioner projection hoisted, dead '
                       'residual half of the last layer dropped; achieved_direct_accounting credits the 21 053 440 FLOP of the direct form (what rounds '
                       '1-4 executed), '
                       if wino else
                       'averages; executed = 21 053 440 FLOP / frame / evaluation (direct K = 768 convolution, conditioner projection hoisted, dead residual '
                       'half of the last layer dropped); ')
                    + 'the duration includes two device copies of the spec tensor, a flag memset and the one-thread timeout latch around '
                    'the launches; *_ref_accounting credits the reference 26 427 392 FLOP / frame / evaluation')
            ref_acc = frames * K * F_EVAL_REF / (ms_call * 1e-3) / 1e12
            direct_acc = frames * K * F_EVAL_EXEC / (ms_call * 1e-3) / 1e12
            frames_k = int(frames_l)
        else:
            ms = eng.time_layer_kernel(layer=-1, t=50, iters=190)
            flop = frames * F_LAYER_EXEC
            achieved = flop / (ms * 1e-3) / 1e12
            kname = 'k_layer_split<false>' if args.split else f'k_layer<{eng.layer_tile() // 32},false>'
            alg_bytes = frames * 6144 + (3 if args.split else 2) * 1024 * 1024
            note = ('achieved counts executed fp32 FLOPs of one residual-layer launch (K=768 dilated conv + K=256 output projection '
                    'per frame); *_ref_accounting also credits the conditioner projection the reference recomputes every step')
            if args.split:
                note += ('; EXPERIMENT: every fp32 product is six bf16 plane products on v_mfma_f32_32x32x16_bf16 - peak = the dense bf16 MFMA '
                         'peak / 6 in fp32-equivalent FLOPs')
            ref_acc = frames * F_LAYER_REF / (ms * 1e-3) / 1e12
            frames_k = frames
        if persistent:
            # where a launch spends its time: in-kernel s_memtime stamps of one layer phase and one head (tools/loop_timeline.py, same shape)
            tl, tl_src = evidence('loop_timeline.json', kname)
            if tl is not None:
                ph, hd = tl['phase_cycles_mean'], tl['head_cycles_mean']
                share = L_LAYERS * ph / (L_LAYERS * ph + hd)
                note += (f'; per evaluation {L_LAYERS} layer phases of {ph:.0f} cycles ({tl["mfma_issue_ideal_per_phase"]} of MFMA issue) + a head of {hd:.0f} '
                         f'cycles -> layers_ms {ms * share:.2f}, head_ms {ms * (1 - share):.2f} of this launch ({tl_src})')
            else:
                note += f'; no in-kernel timeline quoted ({tl_src})'
        traffic, traffic_src = pmc_traffic(kname, frames_k)
        peak = 2500.0 / 6 if args.split else PEAK_FP32_MFMA_TFLOPS
        roof = {'bound': 'mfma', 'kernel': kname, 'achieved': achieved, 'peak': peak,
                'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': traffic, 'traffic_unit': 'bytes/launch',
                'traffic_source': traffic_src, 'algorithmic_bytes_per_launch': alg_bytes,
                'unavoidable_bytes_per_launch': int(K * (frames_k * 1984 + WEIGHT_BYTES)) if persistent else None,
                'bytes_note': 'algorithmic = what THIS design moves per launch by construction (the hoisted conditioner projection 40 KiB / frame / evaluation, '
                              'x / noise / x out, the weight stream once per launch-wide evaluation, the halo rows); unavoidable = SURVEY 8(d): K x (frames x 1 984 B '
                              '[x in 320 + noise 320 + x out 320 + cond 1 024] + 60.3 MB of weights) - the floor of ANY implementation of the reference path',
                'avg_launch_ms': ms, 'flop_per_launch': flop, 'achieved_ref_accounting': ref_acc, 'note': note}
        if persistent:
            roof['achieved_direct_accounting'] = direct_acc
            roof['flop_per_frame_per_evaluation'] = f_exec
        if persistent and cfg == 2:         # the per-layer kernel of the fallback path, for comparison with earlier rounds
            eng.set_loop_mode(0)
            lms = eng.time_layer_kernel(layer=-1, t=50, iters=190)
            eng.set_loop_mode(1)
            roof['per_layer_kernel_fallback'] = {'kernel': f'k_layer<{eng.layer_tile() // 32},false>', 'avg_launch_ms': lms,
                                                 'achieved': frames * F_LAYER_EXEC / (lms * 1e-3) / 1e12}
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms_per_step = el / args.steps * 1e3
        value = frames_per_step * args.steps / el
        if cfg == 2:
            workload = (f'BASELINE configs[1]: DiffSpeech 80-bin, residual_channels=256, 20 layers, K=100 DDPM, batch={B} x T={T} per GPU')
            conf = {'workload': workload, 'preset': PRESET, 'utterances_per_gpu': B, 'frames': T}
        else:
            workload = (f'BASELINE configs[4]: {CFG5_UTTS} synthetic utterances x T={T}, DiffSpeech 80-bin denoiser, K=100 DDPM, the same set at every N, '
                        f'sharded r::W across {world} GPU(s) in micro-batches of {B}, one RCCL gather of the mels to rank 0')
            conf = {'workload': workload, 'preset': PRESET, 'utterances_total': CFG5_UTTS, 'utterances_per_gpu': len(mine), 'frames': T, 'micro_batch': B}
        conf.update({'k_step': K, 'sampler': 'ddpm', 'layer_tile_frames': eng.layer_tile(),
                     'loop': ('persistent kernel (k_loop_wino)' if eng.conv_mode() == 1 else 'persistent kernel (k_loop)') if eng.loop_mode() == 1
                             else 'hipGraph of per-layer kernels',
                     'conv': 'winograd F(2,3)' if eng.conv_mode() == 1 else 'direct',
                     'loop_parked': eng.parked(),
                     'timed_step': 'dsd_prepare (cond re-layout + hoisted conditioner projection, fresh cond every step) + K-step loop + denorm'
                                   + (' + gather' if world > 1 else ''),
                     'sharding': f'utterances r::W, RCCL gather of mels to rank 0 (backend {dist.get_backend()}, world {dist.get_world_size()})'
                                 if world > 1 else 'single GPU'})
        res = {
            'metric': BASELINE_METRIC, 'value': value, 'unit': 'mel-frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak' if cfg == 2 else 'strong', 'vs_baseline': None,
            'dtype': 'f32 as 3 exact bf16 planes, 6 plane products per product, f32 accumulate (EXPERIMENT --split)' if args.split else 'f32',
            'data': 'synthetic', 'config': conf, 'roofline': roof,
            'model_tflops_ref_accounting': frames_per_step * K * F_EVAL_REF * args.steps / el / 1e12,
        }
        res['workload_id'] = 'configs[1]: 8 x T=1024 per GPU (weak)' if cfg == 2 else f'configs[4]: {CFG5_UTTS} x T={CFG5_T} total (strong)'
        res['T'] = T
        if scale_ref is not None:
            res['scale_ref_n1'] = scale_ref
        if per_rank is not None:
            res['per_rank'] = per_rank
        try:
            fixture = parity_check(device)
        except Exception as e:          # fixtures missing etc. - report, do not hide
            fixture = {'error': repr(e)}
        if world == 1 and cfg == 2 and not args.no_cpu_baseline:
            # ONE oracle run over the timed batch: the CPU rate of the workload and the parity of the kernel whose roofline is reported
            last = count[0] & 1
            res['cpu_baseline'], res['parity'], mel_oracle = cpu_baseline(gd, pre, conds[last], x_T, noise, out, roof['kernel'])
            res['parity']['fixture'] = fixture
            res['speedup_vs_cpu_baseline'] = value / res['cpu_baseline']['value']
            try:
                res['torch_rocm_eager_baseline'] = torch_rocm_eager_baseline(gd, conds[last], x_T)
                res['speedup_vs_torch_rocm_eager'] = value / res['torch_rocm_eager_baseline']['value']
            except Exception as e:
                res['torch_rocm_eager_baseline'] = {'error': repr(e)}
            if not args.no_secondary and not args.split:
                try:
                    res['secondary'] = secondary_split(gd, eng, pre, conds[last], x_T, noise, out, mel_oracle, args)
                except Exception as e:
                    res['secondary'] = {'error': repr(e)}
        else:
            res['parity'] = fixture
        if world == 1 and cfg == 2 and not args.no_extras:
            try:
                res['inservice_noise'] = inservice_noise(gd, conds[0], x_T, K, B * T)
                res['inservice_noise']['vs_explicit_noise'] = res['inservice_noise']['value'] / value
            except Exception as e:
                res['inservice_noise'] = {'error': repr(e)[:300]}
            res['offshape'] = offshape(gd, device)
        if world == 1 and cfg == 2 and not args.no_cfg5_shard:
            # the N > 1 lines run BASELINE configs[4] (512 x T=2048, strong): one GPU's shard of the 8-GPU run - 64 utterances x T=2048 in
            # micro-batches of 16, no gather - sampled here after everything else, so that a 1/2/4/8 curve can be normalised on ONE workload
            # from the N = 1 line alone (value x 8 = the ideal of the N = 8 line)
            try:
                del conds, x_T, noise
                torch.cuda.empty_cache()
                _, local5, mine5, _, _, _ = cfg5_workload(gd, 0, 8, device)
                local5()                                                   # warm: workspace growth to 16 x 2048, plan upload
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                local5()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                res['cfg5_shard_n1'] = {'value': len(mine5) * CFG5_T / dt, 'unit': 'mel-frames/s', 'n_gpus': 1, 'seconds': dt,
                                        'what': f"rank 0's shard of BASELINE configs[4] at 8 GPUs ({len(mine5)} of {CFG5_UTTS} utterances x T={CFG5_T}, "
                                                f'micro-batches of {CFG5_MICRO}, K={K_STEPS}), no gather: the per-GPU rate of the workload the N > 1 '
                                                f'lines run; N x this is their ideal'}
            except Exception as e:
                res['cfg5_shard_n1'] = {'error': repr(e)}
        if world == 1 and cfg == 2 and not args.no_extras:
            res['rows'] = quick_rows(gd, device)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
