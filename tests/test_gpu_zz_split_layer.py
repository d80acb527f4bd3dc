"""GPU: the EXPERIMENTAL split-precision layer kernel (csrc/dsd_split.hpp, opt-in via dsd_set_split_mode) - the reference-generated golden
cases with the residual layers evaluated as six bf16 plane products per fp32 product, and the per-layer launch time next to the fp32 kernel.
NOT YET RUN ON HARDWARE (written after the round's GPU minutes were spent; its lane-level model is tests/test_split_layer_model.py):
xfail(strict=False) until it has."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.gpu_helpers import build_hip, run_hip_case

pytestmark = [pytest.mark.gpu, pytest.mark.xfail(reason='split-precision layer kernel not yet run on hardware', strict=False)]


@pytest.mark.parametrize('name,tol', [('denoise_lj', 2e-5), ('denoise_opencpop', 2e-5), ('ddpm_lj_k100', 2e-5), ('shallow_opencpop_k60', 2e-5),
                                      ('plms_opencpop_i40', 5e-5)])
def test_golden_cases_with_split_layers(name, tol):
    g = H.load_golden(name)
    fp32 = run_hip_case(name, use_graph=True)
    out = run_hip_case(name, use_graph=True, split=True)
    e_split, e_fp32 = float(np.abs(out - g['out']).max()), float(np.abs(fp32 - g['out']).max())
    print(f'{name}: max-abs error vs the reference fixture: split layers {e_split:.3e}, fp32 layers {e_fp32:.3e}')
    assert e_split < tol


def test_split_layer_launch_time_next_to_fp32():
    gd, _, _ = build_hip('lj_ds_beta6', 100)
    g = torch.Generator().manual_seed(1)
    cond = torch.randn(8, 1024, 256, generator=g).transpose(1, 2).cuda()
    eng = gd._engine(cond)
    eng.prepare(cond)
    eng.set_loop_mode(0)
    t32 = eng.time_layer_kernel(layer=3, t=50, iters=100)
    eng.set_split_mode(True)
    assert eng.split_mode() == 1
    tsp = eng.time_layer_kernel(layer=3, t=50, iters=100)
    eng.set_split_mode(False)
    fl = 8192 * (2 * 512 * 768 + 2 * 512 * 256)
    print(f'layer launch at 8 x 1024 frames: fp32 k_layer {t32 * 1e3:.1f} us ({fl / t32 / 1e9:.0f} TFLOP/s), split k_layer_split {tsp * 1e3:.1f} us '
          f'({fl / tsp / 1e9:.0f} fp32-equivalent TFLOP/s)')
    assert t32 > 0 and tsp > 0
