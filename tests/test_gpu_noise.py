"""GPU: the in-kernel N(0,1) draws (noise = NULL in dsd_sample_ddpm / dsd_p_sample, include/dsd.h) - the device stream against
the numpy Philox oracle, and the seeded loop against the explicit-noise loop fed with that very stream (bit-identical, both
loop modes): the generator is a pure function of (seed, call index, element)."""
import numpy as np
import pytest
import torch

from oracle import philox_oracle as P
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _model(K=7):
    from tests.gpu_helpers import build_hip
    gd, _, _ = build_hip('opencpop_ds60_rel', K)
    return gd


def test_device_stream_matches_philox_oracle():
    gd = _model()
    cond = torch.randn(1, 40, 256, device='cuda').transpose(1, 2)
    eng = gd._engine(cond)
    for seed, step in ((0, 0), (1234, 7), (2 ** 63 + 12345, 99)):
        z = eng.philox_normal(seed, step, 50000).cpu().numpy()
        ref = P.philox_normal(seed, step, 50000)
        err = float(np.abs(z - ref).max())
        print(f'seed {seed} step {step}: max-abs diff vs numpy oracle {err:.3e}')
        assert err <= 2e-5                       # same bits in, float32 log / cos / sqrt differ by ulps between libm and ocml
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02


@pytest.mark.parametrize('B,T', [(2, 70), (5, 1024)])
def test_seeded_loop_equals_explicit_noise_loop(B, T):
    K, seed = 7, 987654321
    gd = _model(K)
    g = torch.Generator(device='cuda').manual_seed(3)
    cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, device='cuda', generator=g)
    eng = gd._engine(cond)
    noise = torch.stack([eng.philox_normal(seed, j, B * 80 * T).reshape(B, 1, 80, T) for j in range(K)])
    outs = {}
    eng.set_conv_mode('direct')                      # persistent and per-layer paths bit-identical (Winograd form: tests/test_gpu_wino.py)
    for mode in (1, 0):
        eng.set_loop_mode(mode)
        with torch.no_grad():
            outs[mode, 'seeded'] = gd.inference(cond, x_T=x_T, K_step=K, pndm_speedup=0, noise_seed=seed).cpu().numpy()
            outs[mode, 'explicit'] = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).cpu().numpy()
        assert eng.loop_timeouts() == 0
    ref = outs[1, 'explicit']
    for k, v in outs.items():
        np.testing.assert_array_equal(v, ref, err_msg=str(k))
    assert np.isfinite(ref).all()


def test_inference_without_noise_is_reproducible_under_manual_seed():
    gd = _model(5)
    cond = torch.randn(2, 64, 256, device='cuda').transpose(1, 2)
    x_T = torch.randn(2, 1, 80, 64, device='cuda')
    outs = []
    for s in (11, 11, 12):
        torch.manual_seed(s)
        with torch.no_grad():
            outs.append(gd.inference(cond, x_T=x_T, K_step=5, pndm_speedup=0).cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], outs[2])
