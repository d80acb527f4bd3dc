"""Developer tool (GPU): throughput of the hot path on EVERY BASELINE.json configuration that fits one MI355X (the bench.py
line is config 2 only; the others are parity-test cases whose speed is still worth knowing).  One JSON line per config:

  cfg2   DiffSpeech, K=100 DDPM, B=8  x T=1024, Gaussian start
  cfg3   shallow diffusion K=60 from the aux-decoder mel (q_sample at t=59), dilation cycle 4, B=16 x T=1024
  cfg4   PLMS on the 1000-step schedule, pndm_speedup 40 (26 evaluations) and 250 (5 evaluations), B=32 x T=1024
  cfg5   ONE GPU's shard of config 5: 64 utterances x T=2048, K=100 DDPM, in micro-batches of 16 (what each of 8 ranks does
         before the RCCL gather)

    python tools/bench_configs.py [reps] [--parity]
--parity: instead of the rates, the oracle over ALL rows of cfg3 (16 x 1024, K = 60) and cfg4 (32 x 1024, 26 evaluations)
on the box's host cores (1-2 min) against the batch the Winograd loop produced - one JSON line per config with the worst row.
A "pass" = dsd_prepare (hoisted conditioner projection) + the sampling graph + denorm, inputs resident in HBM."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import diffsinger_amd
from diffsinger_amd import hparams
from diffsinger_amd.synth import presets

F_EXEC = 21_053_440          # executed FLOP / frame / evaluation with the direct convolution
F_EXEC_WINO = 15_810_560     # ... with the Winograd F(2,3) convolution of the persistent loop (the default)
F_REF = 26_427_392           # reference FLOP / frame / evaluation


def build(preset, k_step):
    pre = presets()[preset]
    hparams.clear()
    diffsinger_amd.use_preset(preset)
    torch.manual_seed(1234)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=k_step, loss_type='l1',
                                          spec_min=pre['spec_min'], spec_max=pre['spec_max'])
    return gd.cuda().eval(), pre


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def run(name, preset, B, T, k_step, sampler, reps, interval=0, shallow=False, micro=None):
    gd, pre = build(preset, k_step)
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(99)
    nb = micro or B
    conds = [torch.randn(nb, T, 256, device=dev, generator=g).transpose(1, 2) for _ in range(B // nb)]
    x_T = torch.randn(nb, 1, 80, T, device=dev, generator=g)
    noise = torch.randn(k_step, nb, 1, 80, T, device=dev, generator=g) if sampler == 'ddpm' else None
    fs2 = qn = None
    if shallow:
        smin = torch.tensor(pre['spec_min'], device=dev)[None, None, :]
        smax = torch.tensor(pre['spec_max'], device=dev)[None, None, :]
        z = torch.clamp(torch.randn(nb, T, 80, device=dev, generator=g) * 0.5, -1, 1)
        fs2 = (z + 1) / 2 * (smax - smin) + smin
        qn = torch.randn(nb, 1, 80, T, device=dev, generator=g)

    def one_pass():
        out = None
        for cond in conds:
            if sampler == 'plms':
                out = gd.inference(cond, x_T=x_T, K_step=k_step, pndm_speedup=interval)
            elif shallow:
                out = gd.inference(cond, fs2_mels=fs2, q_noise=qn, noise=noise, K_step=k_step, pndm_speedup=0, gaussian_start=False)
            else:
                out = gd.inference(cond, x_T=x_T, noise=noise, K_step=k_step, pndm_speedup=0)
        return out

    sec = timed(one_pass, reps)
    out = one_pass()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    evals = k_step if sampler == 'ddpm' else len(range(0, k_step, interval)) + 1
    frames = B * T
    eng = gd.denoise_fn.engine()
    print(json.dumps({'config': name, 'preset': preset, 'B': B, 'T': T, 'micro_batch': nb, 'sampler': sampler, 'evaluations': evals,
                      'ms_per_pass': sec * 1e3, 'mel_frames_per_s': frames / sec,
                      'conv': 'winograd F(2,3)' if eng.conv_mode() == 1 else 'direct',
                      'tflops_executed': frames * evals * (F_EXEC_WINO if eng.conv_mode() == 1 else F_EXEC) / sec / 1e12,
                      'tflops_direct_accounting': frames * evals * F_EXEC / sec / 1e12, 'tflops_ref_accounting': frames * evals * F_REF / sec / 1e12,
                      'layer_tile_frames': eng.layer_tile(), 'device_bytes': eng.device_bytes()}), flush=True)
    del gd
    torch.cuda.empty_cache()


def parity(name, preset, B, T, k_step, sampler, interval=0, shallow=False):
    """Every row of the batch against the oracle on identical inputs (usr/diff/shallow_diffusion_tts.py:248-276): DDPM / shallow rows as
    oracle batches of 4, PLMS per utterance (the reference's PLMS is B = 1 only; graded relative to max|mel|, quirk 4)."""
    from oracle import diffnet_oracle as O
    from tests import helpers as H
    from diffsinger_amd.synth import make_inputs
    gd, pre = build(preset, k_step)
    cfg = H.net_config(pre)
    p = {k: v.detach().cpu().float().contiguous() for k, v in gd.denoise_fn.state_dict().items()}
    sch = O.make_schedule(H.betas_for(pre))
    smin = torch.tensor(pre['spec_min'], dtype=torch.float32)[None, None, :]
    smax = torch.tensor(pre['spec_max'], dtype=torch.float32)[None, None, :]
    inp = make_inputs(4242 + k_step + interval, B, T, n_noise=k_step if sampler == 'ddpm' else 0, with_fs2_mel=shallow, spec_min=pre['spec_min'],
                      spec_max=pre['spec_max'])
    cond = inp['cond'].transpose(1, 2).contiguous().cuda().transpose(1, 2)
    with torch.no_grad():
        if sampler == 'plms':
            out = gd.inference(cond, x_T=inp['x_T'].cuda(), K_step=k_step, pndm_speedup=interval)
        elif shallow:
            out = gd.inference(cond, fs2_mels=inp['fs2_mel'].cuda(), q_noise=inp['q_noise'].cuda(), noise=inp['noise'].cuda(), K_step=k_step,
                               pndm_speedup=0, gaussian_start=False)
        else:
            out = gd.inference(cond, x_T=inp['x_T'].cuda(), noise=inp['noise'].cuda(), K_step=k_step, pndm_speedup=0)
    out = out.cpu()
    eng = gd.denoise_fn.engine()
    assert eng.loop_mode() == 1 and eng.loop_timeouts() == 0
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    errs, t0 = [], time.perf_counter()
    step = 1 if sampler == 'plms' else 4
    with torch.no_grad():
        for b in range(0, B, step):
            sl = slice(b, b + step)
            if sampler == 'plms':
                want = O.infer_mel(p, cfg, sch, inp['cond'][sl], smin, smax, k_step=k_step, x_T=inp['x_T'][sl], pndm_interval=interval)
            elif shallow:
                want = O.infer_mel(p, cfg, sch, inp['cond'][sl], smin, smax, k_step=k_step, noises=list(inp['noise'][:, sl]), fs2_mel=inp['fs2_mel'][sl],
                                   q_noise=inp['q_noise'][sl])
            else:
                want = O.infer_mel(p, cfg, sch, inp['cond'][sl], smin, smax, k_step=k_step, noises=list(inp['noise'][:, sl]), x_T=inp['x_T'][sl])
            for i in range(want.shape[0]):
                scale = max(1.0, float(want[i].abs().max())) if sampler == 'plms' else 1.0
                errs.append(float((out[b + i] - want[i]).abs().max()) / scale)
    print(json.dumps({'config': name, 'preset': preset, 'B': B, 'T': T, 'sampler': sampler, 'rows_checked': len(errs), 'max_err_over_rows': max(errs),
                      'median_err': sorted(errs)[len(errs) // 2], 'worst_row': errs.index(max(errs)), 'tolerance': 1e-4,
                      'graded': 'relative to max|mel| of the row (PLMS has no clamp)' if sampler == 'plms' else 'max-abs de-normalised mel',
                      'kernel': 'k_loop_wino' if eng.conv_mode() == 1 else 'k_loop', 'oracle_seconds': time.perf_counter() - t0, 'pass': max(errs) <= 1e-4}), flush=True)
    del gd
    torch.cuda.empty_cache()


if __name__ == '__main__':
    if '--parity' in sys.argv:
        parity('cfg3 shallow K=60 (Opencpop cascade, cycle 4)', 'opencpop_ds60_rel', 16, 1024, 60, 'ddpm', shallow=True)
        parity('cfg4 PLMS speedup 40 (26 evals)', 'opencpop_ds1000', 32, 1024, 1000, 'plms', interval=40)
        sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    run('cfg2 DiffSpeech K=100 DDPM', 'lj_ds_beta6', 8, 1024, 100, 'ddpm', reps)
    run('cfg3 shallow K=60 (Opencpop cascade, cycle 4)', 'opencpop_ds60_rel', 16, 1024, 60, 'ddpm', reps, shallow=True)
    run('cfg3b shallow K=51 (PopCS, cycle 1)', 'popcs_ds_beta6', 16, 1024, 51, 'ddpm', reps, shallow=True)
    run('cfg4 PLMS speedup 40 (26 evals)', 'opencpop_ds1000', 32, 1024, 1000, 'plms', reps, interval=40)
    run('cfg4b PLMS speedup 250 (5 evals)', 'opencpop_ds1000', 32, 1024, 1000, 'plms', reps, interval=250)
    run('cfg5 one-GPU shard: 64 x T=2048, K=100, micro-batch 16', 'lj_ds_beta6', 64, 2048, 100, 'ddpm', max(1, reps // 3), micro=16)
