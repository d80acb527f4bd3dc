"""GPU: the LATENCY kernels (csrc/dsd_lat.hpp: every residual layer as two kernels whose workgroups split a tile's output rows G = 2 / 4 / 8
ways - what a batch that fills less than half of the chip runs by default, e.g. the reference's one-utterance-per-device inference,
configs/tts/fs2.yaml:70).

  * single layers through dsd_debug_layer against the oracle's residual layer (usr/diff/net.py:66-78) for every G, dilations 1..8, ragged
    T, with a per-row-block error table; G = 2 / 4 bit-identical to k_layer;
  * the reference-generated golden cases with every G forced (the default suite already runs them with the automatic choice);
  * the whole loop: G = 2 / 4 bit-identical to the per-layer kernels; per-utterance t through dsd_denoise;
  * the point of it: K = 100 latency of ONE utterance of 512 frames next to the persistent loop."""
import time

import numpy as np
import pytest
import torch

from diffsinger_amd.synth import make_inputs
from oracle import diffnet_oracle as O
from tests import helpers as H
from tests.gpu_helpers import build_hip, run_hip_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _direct_convolution(monkeypatch):
    # this file compares the latency kernels with the persistent loop's DIRECT form (bit-identity for G = 2 / 4); the Winograd form of the
    # persistent loop is tests/test_gpu_wino.py
    monkeypatch.setenv('DSD_CONV', 'direct')
DEV = 'cuda:0'


@pytest.mark.parametrize('preset,layer', [('lj_ds_beta6', 0), ('lj_ds_beta6', 19), ('opencpop_ds60_rel', 2), ('opencpop_ds60_rel', 7)])
def test_one_layer_against_the_oracle_layer_every_split(preset, layer):
    pre = H.presets()[preset]
    cfg = H.net_config(pre)
    gd, _, _ = build_hip(preset, pre['K_step'])
    p = {k: v.detach().cpu() for k, v in gd.denoise_fn.state_dict().items()}
    g = torch.Generator().manual_seed(100 + layer)
    B, T, t = 2, 100, 37
    x = torch.randn(B, 256, T, generator=g)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    with torch.no_grad():
        d_emb = O.step_mlp(p, cfg, torch.full((B,), t))
        want_x, want_skip = O.residual_layer(p, cfg, layer, x, cond, d_emb)
    want_skip = want_skip - p[f'residual_layers.{layer}.output_projection.bias'][256:, None]       # the kernels add the skip biases once, in the head
    eng = gd._engine(cond.to(DEV))
    eng.prepare(cond.to(DEV))
    eng.set_loop_mode(0)
    base_x, base_skip = eng.debug_layer(layer, t, x.to(DEV))
    eng.set_loop_mode(3)
    last = layer == cfg.residual_layers - 1
    for G in (2, 4, 8, 16):
        eng.set_lat_split(G)
        assert eng.lat_split() == G
        xo, sk = eng.debug_layer(layer, t, x.to(DEV))
        errs = {'skip': (sk.cpu() - want_skip).abs()}
        if not last:
            errs['x_out'] = (xo.cpu() - want_x).abs()
        for name, e in errs.items():
            per_rb = e.reshape(B, 8, 32, T).amax(dim=(0, 2, 3)).tolist()
            print(f'{preset} layer {layer} (dilation {2 ** (layer % cfg.dilation_cycle_length)}) G={G} {name}: max {float(e.max()):.3e}; '
                  f'per 32-row block {[f"{v:.1e}" for v in per_rb]}')
            assert float(e.max()) < 2e-5, (name, G)
        if G < 8:                                            # same k-ordered sums as k_layer
            assert torch.equal(sk, base_skip), G
            assert last or torch.equal(xo, base_x), G
    eng.set_lat_split(-1)


@pytest.mark.parametrize('G', [2, 4, 8, 16])
@pytest.mark.parametrize('name,tol', [('denoise_lj', 1e-5), ('denoise_opencpop', 1e-5), ('ddpm_lj_k100', 1e-4), ('shallow_opencpop_k60', 1e-4),
                                      ('plms_opencpop_i40', 1e-4)])
def test_golden_cases_with_every_split(name, tol, G):
    g = H.load_golden(name)
    out = run_hip_case(name, loop_mode=3, lat_split=G)
    scale = float(np.abs(g['out']).max()) if name.startswith('plms') else 1.0        # PLMS: no clamp, graded relative
    err = float(np.abs(out - g['out']).max()) / scale
    print(f'{name} G={G}: max-abs error vs the reference fixture {err:.3e} (/ {scale:.3g})')
    assert err <= tol


def test_default_mode_picks_the_latency_kernels_for_small_batches_only():
    gd, _, _ = build_hip('lj_ds_beta6', 100)
    eng = None
    # (3 x 1550 = 147 tiles, 1 x 5000 = 157: the band above half the chip runs G = 8 on several grid waves; 1 x 5200 = 163 tiles is the loop's again)
    for (B, T), want in {(1, 512): 16, (1, 1000): 8, (1, 1550): 4, (4, 777): 2, (3, 1550): 8, (1, 5000): 8, (1, 5200): 0, (8, 1024): 0, (5, 1550): 0}.items():
        cond = torch.randn(B, T, 256, device=DEV).transpose(1, 2)
        eng = gd._engine(cond)
        assert eng.lat_split() == want, ((B, T), eng.lat_split())
        assert eng.loop_mode() == (1 if want == 0 else 0)
    cond = torch.randn(3, 5000, 256, device=DEV).transpose(1, 2)       # 157 tiles per utterance: one utterance per persistent launch = 61 % of the chip
    eng = gd._engine(cond)
    assert eng.lat_split() == 0 and eng.loop_mode() == 0                # -> per-layer kernels (471 tiles = 92 % of two grid waves)
    eng.set_loop_mode(1)
    assert eng.loop_mode() == 1 and eng.loop_launches() == 3
    eng.set_loop_mode(2)


def test_loop_with_latency_kernels_is_bit_identical_to_the_per_layer_kernels_for_g2_g4():
    K, B, T = 7, 2, 90
    gd, _, _ = build_hip('opencpop_ds60_rel', K)
    inp = make_inputs(61, B, T, n_noise=K)
    cond, x_T, noise = inp['cond'].to(DEV), inp['x_T'].to(DEV), inp['noise'].to(DEV)
    eng = gd._engine(cond)
    outs = {}
    for mode, G in ((0, -1), (3, 2), (3, 4), (3, 8), (3, 16), (1, -1)):
        eng.set_loop_mode(mode)
        eng.set_lat_split(G)
        outs[(mode, G)] = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()
        assert eng.lat_split() == (G if mode == 3 else 0)
    eng.set_loop_mode(2)
    eng.set_lat_split(-1)
    assert torch.equal(outs[(0, -1)], outs[(1, -1)])
    assert torch.equal(outs[(0, -1)], outs[(3, 2)]) and torch.equal(outs[(0, -1)], outs[(3, 4)])
    d8 = float((outs[(3, 8)] - outs[(0, -1)]).abs().max())
    d16 = float((outs[(3, 16)] - outs[(0, -1)]).abs().max())
    print(f'G=8 / G=16 vs per-layer kernels after {K} steps: max-abs mel difference {d8:.3e} / {d16:.3e}')
    assert d8 <= 2e-5 and d16 <= 2e-5
    # per-utterance step indices (dsd_denoise with t[B]) on the latency kernels
    t = torch.tensor([3, 55])
    want = O.diffnet_forward(H.oracle_params(H.net_config(H.presets()['opencpop_ds60_rel'])), H.net_config(H.presets()['opencpop_ds60_rel']),
                             inp['x_T'], t, inp['cond'])
    got = gd.denoise_fn(x_T, t.to(DEV), cond)
    assert eng.lat_split() == 16                             # 2 x 90 frames = 6 tiles: 16 workgroups per tile still find a CU each
    assert float((got.cpu() - want).abs().max()) <= 1e-5


def test_one_utterance_latency_next_to_the_persistent_loop():
    K, B, T = 100, 1, 512
    gd, _, _ = build_hip('lj_ds_beta6', K)
    inp = make_inputs(62, B, T, n_noise=K)
    cond, x_T, noise = inp['cond'].to(DEV), inp['x_T'].to(DEV), inp['noise'].to(DEV)
    eng = gd._engine(cond)
    res, outs = {}, {}
    for label, mode, G in (('persistent k_loop', 1, -1), ('latency G=2', 3, 2), ('latency G=4', 3, 4), ('latency G=8', 3, 8), ('latency G=16', 3, 16), ('default', 2, -1)):
        eng.set_loop_mode(mode)
        eng.set_lat_split(G)
        outs[label] = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()      # graph capture / plan upload
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0)
        torch.cuda.synchronize()
        res[label] = (time.perf_counter() - t0) / 3 * 1e3
    print('K=100 DDPM, 1 utterance x 512 frames, ms per sampling call: ' + ', '.join(f'{k} {v:.1f}' for k, v in res.items()))
    assert eng.lat_split() == 16
    ref = outs['persistent k_loop']
    for k, v in outs.items():
        assert float((v - ref).abs().max()) <= 2e-5, k
    assert res['default'] < 0.5 * res['persistent k_loop']



def test_latency_path_repeated_calls_and_many_tiles():
    """Repeated sampling calls (graph replays) and a shape with 49 tiles x G = 4 (196 of 256 CUs): deterministic, bit-identical to k_layer."""
    K, B, T = 6, 1, 1550
    gd, _, _ = build_hip('lj_ds_beta6', K)
    inp = make_inputs(63, B, T, n_noise=K)
    cond, x_T, noise = inp['cond'].to(DEV), inp['x_T'].to(DEV), inp['noise'].to(DEV)
    eng = gd._engine(cond)
    assert eng.lat_split() == 4
    outs = [gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone() for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    eng.set_loop_mode(0)
    ref = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0)
    eng.set_loop_mode(2)
    assert torch.equal(outs[0], ref)                         # G = 4: bit-identical to the per-layer kernels


def test_midsize_band_runs_g8_on_several_grid_waves_and_matches_the_persistent_loop():
    """129-160 tiles (here 3 x 1550 = 147): the automatic choice is G = 8 with 1176 workgroups on 256 CUs (they are ordinary launches: no
    co-residency requirement).  Same mel as the persistent loop and the per-layer kernels up to the reduction order of the K split; the
    oracle on one utterance."""
    K, B, T = 8, 3, 1550
    gd, _, _ = build_hip('lj_ds_beta6', K)
    inp = make_inputs(64, B, T, n_noise=K)
    cond, x_T, noise = inp['cond'].to(DEV), inp['x_T'].to(DEV), inp['noise'].to(DEV)
    eng = gd._engine(cond)
    assert eng.lat_split() == 8 and eng.loop_mode() == 0
    out = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()
    again = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0)
    assert torch.equal(out, again)
    refs = {}
    eng.set_conv_mode('direct')                      # the bit-identity anchor of the persistent path (the Winograd form: tests/test_gpu_wino.py)
    for mode in (1, 0):
        eng.set_loop_mode(mode)
        refs[mode] = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()
        assert eng.lat_split() == 0
    eng.set_loop_mode(2)
    assert torch.equal(refs[0], refs[1])
    d = float((out - refs[1]).abs().max())
    print(f'3 x 1550 (147 tiles), G = 8 vs the persistent loop after {K} steps: max-abs mel difference {d:.3e}')
    assert d <= 2e-5
