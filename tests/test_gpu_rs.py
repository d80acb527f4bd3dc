"""GPU: the ROW-SPLIT PERSISTENT loop (csrc/dsd_loop_rs.hpp) - a batch that fills less than half of the chip (the reference's own inference
shape, one utterance per device: configs/tts/fs2.yaml:70) as ONE launch for the whole K-step loop (usr/diff/shallow_diffusion_tts.py:261-270),
the G workgroups of a tile exchanging gate rows and x' rows through sentinel-tagged rings instead of 43 kernel boundaries per evaluation.

  * EXACT equality with the kernels it replaces, for every G, DDPM and PLMS, dilations 1..8, ragged tails, several utterances: G = 2 / 4 ==
    the per-layer kernels == k_loop; G = 8 / 16 == the latency kernels with the same split (same summation order).  A stale, torn or
    early-read word of any exchange shows up here as a mismatch - every one of the ~4000 hops of a K = 100 loop feeds the next;
  * the reference-generated golden cases and ONE utterance of 512 / 800 / 1550 frames at K = 100 against the oracle (<= 1e-4);
  * run-to-run determinism over repeated replays;
  * a starved loop is LOUD (DSD_ERR_TIMEOUT) and the retry succeeds;
  * the point of it: K = 100 latency next to the latency kernels."""
import time

import numpy as np
import pytest
import torch

from diffsinger_amd.synth import make_inputs
from oracle import diffnet_oracle as O
from tests import helpers as H
from tests.gpu_helpers import build_hip, run_hip_case

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _paths(eng, G):
    """(name, setter) of the row-split loop with G and of the kernels it must equal bit for bit."""
    def rs():
        eng.set_loop_mode(2); eng.set_lat_split(-1); eng.set_rs_split(G)

    def lat():
        eng.set_loop_mode(3); eng.set_lat_split(G); eng.set_rs_split(0)

    def per_layer():
        eng.set_loop_mode(0); eng.set_lat_split(-1); eng.set_rs_split(0)
    return rs, lat, per_layer


def _restore(eng):
    eng.set_loop_mode(2); eng.set_lat_split(-1); eng.set_rs_split(0)


@pytest.mark.parametrize('G', [2, 4, 8, 16])
@pytest.mark.parametrize('preset,B,T,K', [('opencpop_ds60_rel', 2, 90, 7), ('lj_ds_beta6', 1, 201, 5), ('opencpop_ds60_rel', 3, 64, 3)])
def test_ddpm_equals_the_kernels_it_replaces(preset, B, T, K, G):
    gd, _, _ = build_hip(preset, K)
    inp = make_inputs(61 + G, B, T, n_noise=K)
    cond, x_T, noise = inp['cond'].to(DEV), inp['x_T'].to(DEV), inp['noise'].to(DEV)
    eng = gd._engine(cond)
    rs, lat, per_layer = _paths(eng, G)
    try:
        lat()
        want = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()
        assert eng.lat_split() == G and eng.rs_split() == 0
        rs()
        got = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()
        assert eng.rs_split() == G and eng.lat_split() == 0 and eng.loop_mode() == 0
        assert eng.loop_timeouts() == 0
        assert torch.isfinite(got).all()
        d = float((got - want).abs().max())
        print(f'{preset} {B} x {T} K={K} G={G}: row-split loop vs latency kernels max-abs mel difference {d:.3e}')
        assert torch.equal(got, want), d
        if G <= 4:
            per_layer()
            base = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()
            assert torch.equal(got, base)
        # Philox draws inside the kernel: the same stream as the latency kernels'
        lat()
        want_p = gd.inference(cond, x_T=x_T, K_step=K, pndm_speedup=0, noise_seed=99).clone()
        rs()
        got_p = gd.inference(cond, x_T=x_T, K_step=K, pndm_speedup=0, noise_seed=99).clone()
        assert torch.equal(got_p, want_p)
    finally:
        _restore(eng)


@pytest.mark.parametrize('G', [2, 4, 8, 16])
def test_plms_equals_the_kernels_it_replaces(G):
    gd, _, _ = build_hip('opencpop_ds1000', 1000)
    B, T = 2, 75
    inp = make_inputs(71, B, T)
    cond, x_T = inp['cond'].to(DEV), inp['x_T'].to(DEV)
    eng = gd._engine(cond)
    rs, lat, per_layer = _paths(eng, G)
    try:
        for interval in (250, 40):
            lat()
            want = gd.inference(cond, x_T=x_T, K_step=1000, pndm_speedup=interval).clone()
            rs()
            got = gd.inference(cond, x_T=x_T, K_step=1000, pndm_speedup=interval).clone()
            assert eng.rs_split() == G and eng.loop_timeouts() == 0
            assert torch.equal(got, want), (interval, float((got - want).abs().max()))
    finally:
        _restore(eng)


@pytest.mark.parametrize('G', [-1, 2, 8])
@pytest.mark.parametrize('name,tol', [('ddpm_lj_k100', 1e-4), ('shallow_opencpop_k60', 1e-4), ('shallow_popcs_k51', 1e-4), ('plms_opencpop_i40', 1e-4),
                                      ('plms_opencpop_i250', 1e-4)])
def test_golden_cases(name, tol, G):
    g = H.load_golden(name)
    out = run_hip_case(name, rs_split=G)
    scale = float(np.abs(g['out']).max()) if name.startswith('plms') else 1.0        # PLMS: no clamp, graded relative
    err = float(np.abs(out - g['out']).max()) / scale
    print(f'{name} row-split loop G={G}: max-abs error vs the reference fixture {err:.3e} (/ {scale:.3g})')
    assert err <= tol


@pytest.mark.parametrize('T,G', [(512, 16), (800, 8), (1550, 4)])
def test_one_utterance_k100_vs_oracle_latency_and_determinism(T, G):
    """The reference's inference shape: ONE utterance, K = 100 DDPM (BASELINE configs[0] is 1 x 512).  Oracle: ~1-3 s per utterance."""
    gd, cfg, pre = build_hip('lj_ds_beta6', 100)
    K = 100
    g = torch.Generator().manual_seed(900 + T)
    cond = torch.randn(1, T, 256, generator=g).transpose(1, 2)
    x_T = torch.randn(1, 1, 80, T, generator=g)
    noise = torch.randn(K, 1, 1, 80, T, generator=g)
    dcond = cond.transpose(1, 2).contiguous().to(DEV).transpose(1, 2)
    dx, dn = x_T.to(DEV), noise.to(DEV)
    eng = gd._engine(dcond)
    try:
        eng.set_rs_split(-1)
        run = lambda: gd.inference(dcond, x_T=dx, noise=dn, K_step=K, pndm_speedup=0)
        got = run().clone()
        assert eng.rs_split() == G and eng.loop_timeouts() == 0
        for _ in range(4):
            assert torch.equal(run(), got)                               # replays: bit-identical
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ms_rs = (time.perf_counter() - t0) / 5 * 1e3
        eng.set_rs_split(0)
        lat = run().clone()
        assert eng.lat_split() == G
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ms_lat = (time.perf_counter() - t0) / 5 * 1e3
        p = H.oracle_params(cfg)
        sch = O.make_schedule(H.betas_for(pre))
        smin = torch.tensor(pre['spec_min'], dtype=torch.float32)[None, None, :]
        smax = torch.tensor(pre['spec_max'], dtype=torch.float32)[None, None, :]
        want = O.infer_mel(p, cfg, sch, cond, smin, smax, k_step=K, noises=list(noise), x_T=x_T)
        err = float((got.cpu() - want).abs().max())
        d = float((got - lat).abs().max())
        print(f'1 x {T}, K = 100: row-split loop G={G} {ms_rs:.2f} ms per call (prepare + loop + denorm), latency kernels {ms_lat:.2f} ms; '
              f'max-abs mel err vs oracle {err:.3e}; vs latency kernels {d:.3e}')
        assert err <= 1e-4
        assert torch.equal(got, lat) if G >= 8 else d <= 2e-5
    finally:
        _restore(eng)


def test_starved_row_split_loop_is_loud_and_the_retry_succeeds():
    gd, _, _ = build_hip('lj_ds_beta6', 100)
    K, T = 20, 512
    g = torch.Generator(device=DEV).manual_seed(3)
    cond = torch.randn(1, T, 256, device=DEV, generator=g).transpose(1, 2)
    x_T = torch.randn(1, 1, 80, T, device=DEV, generator=g)
    eng = gd._engine(cond)
    try:
        eng.set_rs_split(-1)
        ok = gd.inference(cond, x_T=x_T, K_step=K, pndm_speedup=0, noise_seed=1).clone()
        assert eng.rs_split() == 16
        side = eng.hold_cus(64, 120000)                                 # 64 CUs held by a foreign kernel: 256 workgroups cannot be co-resident
        t0 = time.time()
        try:
            with pytest.raises(RuntimeError, match='spin bound'):
                gd.inference(cond, x_T=x_T, K_step=K, pndm_speedup=0, noise_seed=1, check=True)
            print(f'starved row-split loop reported after {time.time() - t0:.1f} s')
            assert eng.parked() > 0 and eng.rs_split() == 0              # parked on the hipGraph path ...
            again = gd.inference(cond, x_T=x_T, K_step=K, pndm_speedup=0, noise_seed=1, check=True)      # ... the retry, holders still resident
            assert eng.lat_split() == 16 and torch.equal(again, ok)      # G = 16 latency kernels: the same sums
        finally:
            eng.release_cus()
            side.synchronize()
        eng.set_loop_mode(2)                                             # re-arm
        assert eng.parked() == 0 and eng.rs_split() == 16
        assert torch.equal(gd.inference(cond, x_T=x_T, K_step=K, pndm_speedup=0, noise_seed=1, check=True), ok)
    finally:
        eng.release_cus()
        _restore(eng)
