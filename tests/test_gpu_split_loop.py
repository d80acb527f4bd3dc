"""GPU: the split-precision PERSISTENT loop (csrc/dsd_loop_split.hpp; EXPERIMENT, opt-in through dsd_set_split_mode, never the headline dtype):
k_loop with the two contractions of every residual layer as six bf16 plane products per fp32 product on the bf16 matrix pipe.

  * the reference-generated golden cases (DDPM, shallow, PLMS) on the split loop and next to the split per-layer kernels (different K order
    of the dilated conv: equal to reduction-order noise);
  * BASELINE configs[1] (8 x 1024, K = 100): against the fp32 oracle (<= 1e-4) and - utterance 0 - against an fp64 evaluation of the oracle
    for BOTH paths: the split path must not be further from the double-precision result than the fp32 MFMA chain is;
  * its rate next to the fp32 loop."""
import time

import numpy as np
import pytest
import torch

from oracle import diffnet_oracle as O
from tests import helpers as H
from tests.gpu_helpers import build_hip, run_hip_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,tol', [('ddpm_lj_k100', 2e-5), ('shallow_opencpop_k60', 2e-5), ('shallow_popcs_k51', 2e-5), ('plms_opencpop_i40', 2e-5),
                                      ('plms_opencpop_i250', 2e-5)])
def test_golden_cases_on_the_split_loop(name, tol):
    g = H.load_golden(name)
    loop = run_hip_case(name, split=True, loop_mode=1)
    layers = run_hip_case(name, split=True, loop_mode=0)
    scale = float(np.abs(g['out']).max()) if name.startswith('plms') else 1.0
    e_loop, e_lay = float(np.abs(loop - g['out']).max()) / scale, float(np.abs(layers - g['out']).max()) / scale
    d = float(np.abs(loop - layers).max()) / scale
    print(f'{name}: max-abs error vs the reference fixture (/ {scale:.3g}): split loop {e_loop:.3e}, split per-layer kernels {e_lay:.3e}; loop vs layers {d:.3e}')
    assert np.isfinite(loop).all() and e_loop < tol and d < tol


def test_config2_against_the_fp32_and_the_fp64_oracle_and_rate():
    gd, cfg, pre = build_hip('lj_ds_beta6', 100)
    B, T, K = 8, 1024, 100
    g = torch.Generator().manual_seed(2025)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, generator=g)
    noise = torch.randn(K, B, 1, 80, T, generator=g)
    dcond = cond.transpose(1, 2).contiguous().cuda().transpose(1, 2)
    dx, dn = x_T.cuda(), noise.cuda()
    eng = gd._engine(dcond)
    run = lambda: gd.inference(dcond, x_T=dx, noise=dn, K_step=K, pndm_speedup=0)

    def timed():
        out = run().clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) / 3 * 1e3
    try:
        f32, ms32 = timed()
        assert eng.loop_mode() == 1 and eng.split_mode() == 0
        eng.set_split_mode(True)
        sp, mssp = timed()
        assert eng.loop_mode() == 1 and eng.split_mode() == 1 and eng.loop_timeouts() == 0
        assert torch.equal(run(), sp)                                    # deterministic
    finally:
        eng.set_split_mode(False)
    p = H.oracle_params(cfg)
    sch = O.make_schedule(H.betas_for(pre))
    smin = torch.tensor(pre['spec_min'], dtype=torch.float32)[None, None, :]
    smax = torch.tensor(pre['spec_max'], dtype=torch.float32)[None, None, :]
    rows = (0, 5)
    for b in rows:
        want = O.infer_mel(p, cfg, sch, cond[b:b + 1], smin, smax, k_step=K, noises=list(noise[:, b:b + 1]), x_T=x_T[b:b + 1])
        e_sp, e_32 = float((sp[b:b + 1].cpu() - want).abs().max()), float((f32[b:b + 1].cpu() - want).abs().max())
        print(f'8 x 1024 K=100 row {b} vs the fp32 oracle: split loop {e_sp:.3e}, fp32 loop {e_32:.3e}')
        assert e_sp <= 1e-4 and e_32 <= 1e-4
    p64 = {k: v.double() for k, v in p.items()}
    m64 = O.infer_mel(p64, cfg, sch, cond[0:1].double(), smin.double(), smax.double(), k_step=K, noises=list(noise[:, 0:1].double()), x_T=x_T[0:1].double())
    e64_sp, e64_32 = float((sp[0:1].cpu().double() - m64).abs().max()), float((f32[0:1].cpu().double() - m64).abs().max())
    print(f'row 0 vs an fp64 evaluation of the oracle: split loop {e64_sp:.3e}, fp32 loop {e64_32:.3e}; '
          f'rate: split {mssp:.1f} ms per call = {B * T / mssp * 1e3:.0f} mel-frames/s, fp32 {ms32:.1f} ms = {B * T / ms32 * 1e3:.0f} mel-frames/s')
    assert e64_sp <= 1e-4 and e64_sp <= 2.0 * e64_32 + 1e-6


def test_the_weight_stream_variants_agree_bit_for_bit_and_their_rates(monkeypatch):
    """The loop streams its weights either as three bf16 planes (6 bytes per weight, DSD_SPLIT_W=0) or as fp32 split into the same planes in
    registers beside the MFMAs (4 bytes, DSD_SPLIT_W=4, the default).  Same planes (round to nearest even, exact residuals), same
    products in the same order per accumulator: the outputs must be IDENTICAL; only the time may differ."""
    B, T, K = 8, 1024, 100
    g = torch.Generator().manual_seed(77)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, generator=g)
    noise = torch.randn(K, B, 1, 80, T, generator=g)
    outs, ms = {}, {}
    for wsrc in ('0', '4'):
        monkeypatch.setenv('DSD_SPLIT_W', wsrc)
        gd, cfg, pre = build_hip('lj_ds_beta6', 100)
        dcond = cond.transpose(1, 2).contiguous().cuda().transpose(1, 2)
        dx, dn = x_T.cuda(), noise.cuda()
        eng = gd._engine(dcond)
        eng.set_split_mode(True)
        run = lambda: gd.inference(dcond, x_T=dx, noise=dn, K_step=K, pndm_speedup=0)
        out = run().clone()
        assert eng.loop_mode() == 1 and eng.split_mode() == 1 and eng.loop_timeouts() == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ms[wsrc] = (time.perf_counter() - t0) / 3 * 1e3
        outs[wsrc] = out.cpu()
        assert torch.isfinite(outs[wsrc]).all()
        del gd, eng
    print('8 x 1024, K = 100, split loop by weight stream: ' + ', '.join(f'DSD_SPLIT_W={k}: {v:.1f} ms = {B * T / v * 1e3:.0f} mel-frames/s' for k, v in ms.items()))
    assert torch.equal(outs['0'], outs['4'])
