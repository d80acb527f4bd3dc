"""GPU: parity at BASELINE.json's FULL sizes.  The CPU oracle cannot run a whole config-2/3/4 batch in seconds, so the
full-size runs are checked through size-independent properties of the domain:

  * utterances are independent (no op mixes batch rows): row b of the full batch must equal - bit for bit - the same
    utterance run alone, and the oracle is affordable for ONE utterance at full T and full K, so selected rows of the
    full batch are compared with the oracle directly (<= 1e-4 on the de-normalised mel, the north-star tolerance);
  * with the reference's own initialisation (final projection weight = 0, usr/diff/net.py:105) eps_hat is the bias,
    the sampler becomes element-wise and the whole [B,80,T] x K result has a closed form that numpy evaluates in
    milliseconds - this checks the sampler epilogue, the noise indexing and the step order at full size;
  * graph replay == eager launches, run-to-run determinism, 32- == 64-frame tiles (bit for bit)."""
import numpy as np
import pytest
import torch

from oracle import diffnet_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _setup(preset, k_step):
    from tests.gpu_helpers import build_hip
    gd, cfg, pre = build_hip(preset, k_step=k_step)
    p = H.oracle_params(cfg)
    sch = O.make_schedule(H.betas_for(pre))
    smin = torch.tensor(pre['spec_min'], dtype=torch.float32)[None, None, :]
    smax = torch.tensor(pre['spec_max'], dtype=torch.float32)[None, None, :]
    return gd, cfg, pre, p, sch, smin, smax


def _dev_cond(cond):
    return cond.transpose(1, 2).contiguous().cuda().transpose(1, 2)


def test_config2_ddpm_k100_rows_vs_oracle_and_row_independence():
    """BASELINE configs[1]: DiffSpeech, B=8, T=1024, K=100 DDPM from a Gaussian start.  Rows 0 and 5 of the batch against the oracle (the
    session's shared oracle run, tests/gpu_helpers.lj_k100_case; bench.py's parity leg checks the whole timed batch)."""
    from tests.gpu_helpers import build_hip, lj_k100_case
    c = lj_k100_case()
    gd, _, _ = build_hip('lj_ds_beta6', k_step=100)
    B, T, K, cond, x_T, noise = c['B'], c['T'], c['K'], c['cond'], c['x_T'], c['noise']
    with torch.no_grad():
        full = gd.inference(_dev_cond(cond), x_T=x_T.cuda(), noise=noise.cuda(), K_step=K, pndm_speedup=0).cpu()
        again = gd.inference(_dev_cond(cond), x_T=x_T.cuda(), noise=noise.cuda(), K_step=K, pndm_speedup=0).cpu()
    assert torch.isfinite(full).all()
    assert torch.equal(full, again)                                     # deterministic
    for b, want in sorted(c['want'].items()):
        err = float((full[b:b + 1] - want).abs().max())
        print(f'config 2 row {b}: max-abs mel err vs oracle {err:.3e}')
        assert err <= 1e-4
    b = 5
    want = c['want'][b]
    with torch.no_grad():
        alone_default = gd.inference(_dev_cond(cond[b:b + 1]), x_T=x_T[b:b + 1].cuda(), noise=noise[:, b:b + 1].contiguous().cuda(),
                                     K_step=K, pndm_speedup=0).cpu()
        eng = gd.denoise_fn.engine()
        assert eng.lat_split() == 8 and eng.conv_mode() == 1             # one utterance of 32 tiles: the latency kernels, 8-way row split, Winograd conv node
        eng.set_loop_mode(1)                                             # ... and the same utterance on the kernel the batch of 8 ran on
        alone = gd.inference(_dev_cond(cond[b:b + 1]), x_T=x_T[b:b + 1].cuda(), noise=noise[:, b:b + 1].contiguous().cuda(),
                             K_step=K, pndm_speedup=0).cpu()
        eng.set_loop_mode(2)
    assert torch.equal(alone, full[b:b + 1])                             # batch rows never interact: bit-identical on the same kernels
    # the automatic choice runs a lone utterance on the G = 8 latency kernels - k_lat_conv_w: the Winograd convolution of the loop, but the two K
    # halves of a wave pair summed separately: the same mel to reduction-order noise (measured 1.0e-5 - 2.2e-5 over rounds 5 / 6; the
    # reference's own B = 1 vs B = 2 results differ by 9.5e-7), and as close to the oracle as the batch row is
    d = float((alone_default - full[b:b + 1]).abs().max())
    print(f'config 2 row {b}: alone on the latency kernels vs row of the batch: max-abs mel difference {d:.3e}; vs oracle '
          f'{float((alone_default - want).abs().max()):.3e}')
    assert d <= 5e-5 and float((alone_default - want).abs().max()) <= 1e-4


def test_default_mode_is_deterministic_and_the_latency_graph_replays_its_eager_launches():
    """The SHIPPED defaults (no conv / loop override; ADVICE r5: the bit-identity tests of tests/test_gpu_loop.py pin conv='direct'):
    the Winograd persistent loop (8 x 1024) run to run, and the Winograd latency kernels (one utterance) as cached hipGraph against their
    eager launches - the same bits.  (dsd_set_use_graph(0) takes a chip-filling batch OFF the persistent loop - onto the eager per-layer
    kernels, direct convolution: equal to the loop to the Winograd form's rounding, asserted at 3e-5.)"""
    from tests.gpu_helpers import build_hip
    gd, _, _ = build_hip('lj_ds_beta6', k_step=100)
    K = 6
    for B, T, persistent in ((8, 1024, True), (1, 1000, False)):
        g = torch.Generator(device='cuda').manual_seed(11 + B)
        cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
        x_T = torch.randn(B, 1, 80, T, device='cuda', generator=g)
        noise = torch.randn(K, B, 1, 80, T, device='cuda', generator=g)
        eng = gd._engine(cond)
        assert (eng.loop_mode() == 1) == persistent and eng.conv_mode() == 1
        outs = {}
        for graph in (True, False):
            eng.set_use_graph(graph)
            with torch.no_grad():
                a = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()
                b_ = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()
            assert torch.equal(a, b_) and bool(torch.isfinite(a).all())
            if persistent:
                assert eng.loop_mode() == (1 if graph else 0)
            outs[graph] = a
        eng.set_use_graph(True)
        d = float((outs[True] - outs[False]).abs().max())
        if persistent:
            assert d <= 3e-5, d
        else:
            assert torch.equal(outs[True], outs[False]), d


def test_config3_shallow_k60_row_vs_oracle():
    """BASELINE configs[2]: shallow diffusion K=60 from the aux-decoder mel, dilation cycle 4, B=16, T=1024."""
    from diffsinger_amd.synth import make_inputs
    gd, cfg, pre, p, sch, smin, smax = _setup('opencpop_ds60_rel', 60)
    B, T, K = 16, 1024, 60
    inp = make_inputs(303, B, T, n_noise=K, with_fs2_mel=True, spec_min=pre['spec_min'], spec_max=pre['spec_max'])
    with torch.no_grad():
        full = gd.inference(_dev_cond(inp['cond']), fs2_mels=inp['fs2_mel'].cuda(), q_noise=inp['q_noise'].cuda(),
                            noise=inp['noise'].cuda(), K_step=K, pndm_speedup=0, gaussian_start=False).cpu()
        b = 11
        want = O.infer_mel(p, cfg, sch, inp['cond'][b:b + 1], smin, smax, k_step=K, noises=list(inp['noise'][:, b:b + 1]),
                           fs2_mel=inp['fs2_mel'][b:b + 1], q_noise=inp['q_noise'][b:b + 1])
    err = float((full[b:b + 1] - want).abs().max())
    print(f'config 3 row {b}: max-abs mel err vs oracle {err:.3e}')
    assert torch.isfinite(full).all() and err <= 1e-4


def test_config4_plms_row_vs_oracle():
    """BASELINE configs[3]: Opencpop e2e, 1000-step schedule, PNDM/PLMS pndm_speedup=40 (26 evaluations), B=32, T=1024.
    PLMS has no clamp: graded relative to max|mel|."""
    gd, cfg, pre, p, sch, smin, smax = _setup('opencpop_ds1000', 1000)
    B, T = 32, 1024
    g = torch.Generator().manual_seed(404)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, generator=g)
    with torch.no_grad():
        full = gd.inference(_dev_cond(cond), x_T=x_T.cuda(), K_step=1000, pndm_speedup=40).cpu()
        b = 19
        want = O.infer_mel(p, cfg, sch, cond[b:b + 1], smin, smax, k_step=1000, x_T=x_T[b:b + 1], pndm_interval=40)
    scale = float(want.abs().max())
    err = float((full[b:b + 1] - want).abs().max())
    print(f'config 4 row {b}: max-abs mel err {err:.3e}, relative to max|mel|={scale:.1f}: {err / scale:.3e}')
    assert torch.isfinite(full).all() and err / scale <= 1e-4


def test_zero_final_projection_closed_form_full_size():
    """Reference initialisation (final projection weight 0): eps_hat == bias, the K-step DDPM loop is element-wise.
    Checks the sampler arithmetic, noise slice <-> step mapping and layout at B=8, T=1024, K=100 against numpy."""
    from tests.gpu_helpers import build_hip
    gd, cfg, pre = build_hip('lj_ds_beta6', k_step=100)
    with torch.no_grad():
        gd.denoise_fn.output_projection.weight.zero_()
        bias = gd.denoise_fn.output_projection.bias.detach().cpu().clone()
    sch = O.make_schedule(H.betas_for(pre))
    B, T, K = 8, 1024, 100
    g = torch.Generator().manual_seed(99)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    x = torch.randn(B, 1, 80, T, generator=g)
    noise = torch.randn(K, B, 1, 80, T, generator=g)
    with torch.no_grad():
        mel, x0 = gd.inference(_dev_cond(cond), x_T=x.cuda(), noise=noise.cuda(), K_step=K, pndm_speedup=0, return_x=True)
    eps = bias[None, None, :, None].expand(B, 1, 80, T)
    for j, t in enumerate(reversed(range(K))):                          # shallow_diffusion_tts.py:134-166, element-wise
        xr = sch['sqrt_recip_alphas_cumprod'][t] * x - sch['sqrt_recipm1_alphas_cumprod'][t] * eps
        xr = xr.clamp(-1., 1.)
        mean = sch['posterior_mean_coef1'][t] * xr + sch['posterior_mean_coef2'][t] * x
        x = mean + (0.0 if t == 0 else 1.0) * (0.5 * sch['posterior_log_variance_clipped'][t]).exp() * noise[j]
    err = float((x0.cpu() - x).abs().max())
    print(f'closed-form DDPM at full size: max-abs x_0 err {err:.3e}')
    assert err <= 1e-6
    smin = torch.tensor(pre['spec_min'])[None, None, :]
    smax = torch.tensor(pre['spec_max'])[None, None, :]
    assert float((mel.cpu() - O.denorm_spec(x[:, 0].transpose(1, 2), smin, smax)).abs().max()) <= 1e-5


def test_config5_shape_row_vs_oracle():
    """BASELINE configs[4]'s per-GPU micro-batch: 16 utterances x T=2048, K=100 DDPM (1024 tiles = four persistent launches of whole
    utterances).  One row of the batch against the oracle on identical (x_T, cond, noise[K]) - usr/diff/shallow_diffusion_tts.py:248-276."""
    gd, cfg, pre, p, sch, smin, smax = _setup('lj_ds_beta6', 100)
    B, T, K = 16, 2048, 100
    g = torch.Generator().manual_seed(5005)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, generator=g)
    b = 13
    gn = torch.Generator(device='cuda').manual_seed(77)
    noise = torch.randn(K, B, 1, 80, T, device='cuda', generator=gn)            # 1.05 GB: drawn on the device, row b copied back for the oracle
    with torch.no_grad():
        full = gd.inference(_dev_cond(cond), x_T=x_T.cuda(), noise=noise, K_step=K, pndm_speedup=0).cpu()
        eng = gd.denoise_fn.engine()
        assert eng.loop_mode() == 1 and eng.loop_launches() == 4 and eng.loop_timeouts() == 0
        nb = noise[:, b:b + 1].cpu()
        want = O.infer_mel(p, cfg, sch, cond[b:b + 1], smin, smax, k_step=K, noises=list(nb), x_T=x_T[b:b + 1])
    err = float((full[b:b + 1] - want).abs().max())
    print(f'config 5 shape (16 x 2048, K = 100) row {b}: max-abs mel err vs oracle {err:.3e}')
    assert torch.isfinite(full).all() and err <= 1e-4


def test_graph_eager_and_tiles_agree_at_config5_shape():
    """One GPU's micro-batch of BASELINE configs[4] (16 utterances x T=2048): graph == eager, 32- == 64-frame tiles,
    bit for bit (K shortened: the property is per step)."""
    from tests.gpu_helpers import build_hip
    gd, cfg, pre = build_hip('lj_ds_beta6', k_step=100)
    B, T, K = 16, 2048, 4
    g = torch.Generator(device='cuda').manual_seed(5)
    cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, device='cuda', generator=g)
    noise = torch.randn(K, B, 1, 80, T, device='cuda', generator=g)
    eng = gd._engine(cond)
    eng.set_conv_mode('direct')                      # the persistent loop's direct form is the one that is bit-identical to the per-layer kernels
    outs = []
    try:
        for graph, tile in ((True, 0), (False, 0), (True, 32), (True, 64)):
            eng.set_use_graph(graph)
            eng.set_layer_tile(tile)
            with torch.no_grad():
                outs.append(gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).cpu())
    finally:
        eng.set_use_graph(True)
        eng.set_layer_tile(0)
    assert torch.isfinite(outs[0]).all()
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
