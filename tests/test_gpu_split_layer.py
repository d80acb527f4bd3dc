"""GPU: the OPT-IN split-precision layer kernel (csrc/dsd_split.hpp, dsd_set_split_mode; never the default, never the headline dtype) -
the reference-generated golden cases with the residual layers evaluated as six bf16 plane products per fp32 product, single layers
against the oracle's layer, and the per-layer launch time next to the fp32 kernel.  First run on the MI355X in round 2
(profiles/r02a_pytest_gpu_zz_split_layer.txt: errors equal to the fp32 kernels', 51 vs 72 us per launch); its lane-level model is
tests/test_split_layer_model.py."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.gpu_helpers import build_hip, run_hip_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,tol', [('denoise_lj', 2e-5), ('denoise_opencpop', 2e-5), ('ddpm_lj_k100', 2e-5), ('shallow_opencpop_k60', 2e-5),
                                      ('plms_opencpop_i40', 2e-5)])
def test_golden_cases_with_split_layers(name, tol):
    g = H.load_golden(name)
    fp32 = run_hip_case(name, use_graph=True)
    out = run_hip_case(name, use_graph=True, split=True)
    # PLMS has no clamp: with untrained weights x_0 grows to O(100), so - like every PLMS parity test here - the error
    # is graded relative to max|mel| of the fixture
    scale = float(np.abs(g['out']).max()) if name.startswith('plms') else 1.0
    e_split, e_fp32 = float(np.abs(out - g['out']).max()) / scale, float(np.abs(fp32 - g['out']).max()) / scale
    print(f'{name}: max-abs error vs the reference fixture (/ {scale:.3g}): split layers {e_split:.3e}, fp32 layers {e_fp32:.3e}')
    assert e_split < tol


@pytest.mark.parametrize('layer', [0, 3, 19])
def test_one_layer_against_the_oracle_layer_fp32_and_split(layer):
    """Layer-level parity through dsd_debug_layer: the verified fp32 kernel (this also validates the debug plumbing) and the split kernel
    against the oracle's residual layer on the same x, cond and step; on a mismatch the per-row-block / per-frame-column error table says
    where (which wave's rows, interior vs halo columns)."""
    from oracle import diffnet_oracle as O
    pre = H.presets()['lj_ds_beta6']
    cfg = H.net_config(pre)
    gd, _, _ = build_hip('lj_ds_beta6', 100)
    p = {k: v.detach().cpu() for k, v in gd.denoise_fn.state_dict().items()}
    g = torch.Generator().manual_seed(layer)
    B, T, t = 2, 100, 37
    x = torch.randn(B, 256, T, generator=g)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    with torch.no_grad():
        d_emb = O.step_mlp(p, cfg, torch.full((B,), t))
        want_x, want_skip = O.residual_layer(p, cfg, layer, x, cond, d_emb)
    want_skip = want_skip - p[f'residual_layers.{layer}.output_projection.bias'][256:, None]       # the kernels add the skip biases once, in the head
    eng = gd._engine(cond.cuda())
    eng.prepare(cond.cuda())
    eng.set_loop_mode(0)
    for split in (False, True):
        eng.set_split_mode(split)
        xo, sk = eng.debug_layer(layer, t, x.cuda())
        errs = {'skip': (sk.cpu() - want_skip).abs()}
        if xo is not None:
            errs['x_out'] = (xo.cpu() - want_x).abs()
        for name, e in errs.items():
            worst = float(e.max())
            per_rb = e.reshape(B, 8, 32, T).amax(dim=(0, 2, 3)).tolist()
            per_col = e.amax(dim=(0, 1))
            print(f'layer {layer} {"split" if split else "fp32 "} {name}: max {worst:.3e}; per 32-row block {[f"{v:.1e}" for v in per_rb]}; '
                  f'first/last tile columns {float(per_col[:8].max()):.1e} / {float(per_col[-8:].max()):.1e}')
            assert worst < 2e-5, (name, split, worst)
    eng.set_split_mode(False)


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
