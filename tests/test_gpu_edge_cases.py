"""GPU: edge cases and host-logic checks of the HIP path through the C ABI, against the CPU oracle on the same
seeded inputs (ragged / tiny / long T, B = 1, strided cond, per-utterance t, masks, re-binding, error behaviour).
Tolerances as in test_gpu_parity.py: single evaluation <= 1e-5, sampled mel <= 1e-4."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import diffnet_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

PRESET = 'lj_ds_beta6'


@pytest.fixture(scope='module')
def lj():
    from tests.gpu_helpers import build_hip
    gd, cfg, pre = build_hip(PRESET, k_step=100)
    p = H.oracle_params(cfg)
    sch = O.make_schedule(H.betas_for(pre))
    smin = torch.tensor(pre['spec_min'], dtype=torch.float32)[None, None, :]
    smax = torch.tensor(pre['spec_max'], dtype=torch.float32)[None, None, :]
    return dict(gd=gd, cfg=cfg, pre=pre, p=p, sch=sch, smin=smin, smax=smax)


def _inputs(seed, B, T, n_noise=0):
    g = torch.Generator().manual_seed(seed)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    x = torch.randn(B, 1, 80, T, generator=g)
    noise = torch.randn(n_noise, B, 1, 80, T, generator=g) if n_noise else None
    return cond, x, noise


def _dev_cond(cond):
    """[B,H,T] view of a contiguous [B,T,H] device tensor - what the reference hands the denoiser (:238)."""
    return cond.transpose(1, 2).contiguous().cuda().transpose(1, 2)


@pytest.mark.parametrize('B,T', [(1, 1), (1, 7), (2, 31), (1, 32), (3, 33), (1, 63), (2, 65), (1, 1000)])
def test_denoise_ragged_lengths(lj, B, T):
    cond, x, _ = _inputs(1000 + 7 * T + B, B, T)
    t = torch.tensor([(13 * b + T) % 100 for b in range(B)])
    with torch.no_grad():
        want = O.diffnet_forward(lj['p'], lj['cfg'], x, t, cond).numpy()
        got = lj['gd'].denoise_fn(x.cuda(), t.cuda(), _dev_cond(cond)).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5


@pytest.mark.parametrize('tile', [32, 64])
def test_denoise_tile_sizes_ragged(lj, tile):
    cond, x, _ = _inputs(77, 2, 97)
    t = torch.tensor([99, 0])
    dc = _dev_cond(cond)
    eng = lj['gd'].denoise_fn.bind_cond(dc)
    eng.set_layer_tile(tile)
    try:
        with torch.no_grad():
            got = lj['gd'].denoise_fn(x.cuda(), t.cuda(), dc).cpu().numpy()
            want = O.diffnet_forward(lj['p'], lj['cfg'], x, t, cond).numpy()
    finally:
        eng.set_layer_tile(0)
    assert np.abs(got - want).max() <= 1e-5


def test_cond_layouts_agree_bitwise(lj):
    """cond as the reference's transposed view, as a contiguous [B,H,T] tensor and as a generic strided slice."""
    cond, x, _ = _inputs(5, 2, 70)
    t = torch.tensor([40, 41])
    net = lj['gd'].denoise_fn
    with torch.no_grad():
        a = net(x.cuda(), t.cuda(), _dev_cond(cond)).cpu()
        b = net(x.cuda(), t.cuda(), cond.contiguous().cuda()).cpu()
        big = torch.zeros(2, 300, 150, device='cuda')
        view = big[:, 10:266, 5:145:2]
        view.copy_(cond.cuda())
        c = net(x.cuda(), t.cuda(), view).cpu()
    assert torch.equal(a, b) and torch.equal(a, c)


def test_rebinding_batches_of_different_shape(lj):
    """Workspace regrow / shrink and graph cache keyed by shape: results do not depend on what ran before."""
    net = lj['gd'].denoise_fn
    outs = {}
    for rnd in range(2):
        for (B, T) in [(1, 40), (3, 130), (2, 64), (1, 40)]:
            cond, x, _ = _inputs(900 + B * 1000 + T, B, T)
            with torch.no_grad():
                y = net(x.cuda(), torch.full((B,), 17).cuda(), _dev_cond(cond)).cpu()
            key = (B, T)
            if key in outs:
                assert torch.equal(outs[key], y)
            outs[key] = y


def test_recycled_cond_address_is_not_mistaken_for_the_cached_cond(lj):
    """The caching allocator hands a freed conditioner's address to the next one (same pointer, version 0, same shape):
    the hoisted projections must follow the CONTENT."""
    net = lj['gd'].denoise_fn
    t = torch.tensor([30])
    ptrs = []
    for seed in (41, 42, 43):
        cond, x, _ = _inputs(seed, 1, 48)
        dc = _dev_cond(cond)
        ptrs.append(dc.data_ptr())
        with torch.no_grad():
            got = net(x.cuda(), t.cuda(), dc).cpu()
            want = O.diffnet_forward(lj['p'], lj['cfg'], x, t, cond)
        assert (got - want).abs().max() <= 1e-5, seed
        del dc
    print('cond addresses:', ptrs)


def test_single_steps_match_oracle(lj):
    """q_sample, p_sample (t > 0 and t == 0) and the stateful p_sample_plms API, step by step."""
    from collections import deque
    gd, p, cfg, sch = lj['gd'], lj['p'], lj['cfg'], lj['sch']
    B, T = 2, 45
    cond, x, noise = _inputs(31, B, T, n_noise=3)
    dc = _dev_cond(cond)
    gd.denoise_fn.bind_cond(dc)
    with torch.no_grad():
        got = gd.q_sample(x.cuda(), torch.tensor([70]).cuda(), noise=noise[0].cuda()).cpu()
        assert torch.equal(got, O.q_sample(sch, x, torch.tensor([70]), noise[0]))          # element-wise: exact
        for tt, z in ((63, noise[1]), (0, noise[2])):
            t = torch.full((B,), tt, dtype=torch.long)
            got = gd.p_sample(x.cuda(), t.cuda(), dc, noise=z.cuda()).cpu()
            want = O.p_sample(p, cfg, sch, x, t, cond, z)
            assert (got - want).abs().max() <= 2e-5
        gd.noise_list = deque(maxlen=4)
        hist = deque(maxlen=4)
        xg, xo = x.cuda(), x
        for i in reversed(range(0, 100, 20)):
            t = torch.full((B,), i, dtype=torch.long)
            xg = gd.p_sample_plms(xg, t.cuda(), 20, dc)
            xo = O.p_sample_plms(p, cfg, sch, xo, t, 20, cond, hist)
            scale = float(xo.abs().max())
            assert float((xg.cpu() - xo).abs().max()) <= 1e-4 * max(scale, 1.0), i


def test_inference_with_mask_and_fresh_noise_tensor(lj):
    """Shallow start + mel mask (:273), then the SAME cached graph replayed with a different noise tensor."""
    gd, p, cfg, sch, pre = lj['gd'], lj['p'], lj['cfg'], lj['sch'], lj['pre']
    B, T, K = 2, 50, 12
    from diffsinger_amd.synth import make_inputs
    for seed in (11, 12):
        inp = make_inputs(seed, B, T, n_noise=K, with_fs2_mel=True, spec_min=pre['spec_min'], spec_max=pre['spec_max'])
        mask = (torch.arange(T)[None, :] < torch.tensor([T, T - 9])[:, None]).float()
        dc = _dev_cond(inp['cond'])
        with torch.no_grad():
            got = gd.inference(dc, fs2_mels=inp['fs2_mel'].cuda(), q_noise=inp['q_noise'].cuda(), noise=inp['noise'].cuda().clone(),
                               K_step=K, pndm_speedup=0, gaussian_start=False, mel_mask=mask.cuda()).cpu()
            want = O.infer_mel(p, cfg, sch, inp['cond'], lj['smin'], lj['smax'], k_step=K, noises=list(inp['noise']),
                               fs2_mel=inp['fs2_mel'], q_noise=inp['q_noise'], mel_mask=mask)
        assert (got - want).abs().max() <= 1e-4
        assert float(got[1, T - 9:].abs().max()) == 0.0


def test_schedule_tables_equal_module_buffers_bitwise(lj):
    gd = lj['gd']
    cond, _, _ = _inputs(3, 1, 33)
    eng = gd._engine(_dev_cond(cond))
    names = ['betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod', 'sqrt_one_minus_alphas_cumprod',
             'log_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'posterior_variance',
             'posterior_log_variance_clipped', 'posterior_mean_coef1', 'posterior_mean_coef2']
    for i, n in enumerate(names):
        np.testing.assert_array_equal(eng.schedule_table(i), getattr(gd, n).cpu().numpy(), err_msg=n)
        np.testing.assert_array_equal(eng.schedule_table(i), lj['sch'][n].numpy(), err_msg=n)


def test_weight_update_is_picked_up(lj):
    """In-place parameter changes (load_state_dict / optimiser step) repack the weights on the next call."""
    from tests.gpu_helpers import build_hip
    gd, cfg, pre = build_hip(PRESET, k_step=100)
    cond, x, _ = _inputs(8, 1, 40)
    dc, t = _dev_cond(cond), torch.tensor([5])
    with torch.no_grad():
        a = gd.denoise_fn(x.cuda(), t.cuda(), dc).cpu()
        gd.denoise_fn.output_projection.bias.add_(1.0)
        b = gd.denoise_fn(x.cuda(), t.cuda(), dc).cpu()
    assert torch.allclose(b, a + 1.0, atol=1e-6)


def test_error_behaviour_through_the_c_abi():
    """Reference-style loud failures: bad arguments / call order return codes + messages, never a crash."""
    from diffsinger_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.DsdConfig(80, 256, 256, 20, 1)
    assert lib.dsd_create(C.byref(cfg), 0, C.byref(h)) == 0
    try:
        x = torch.zeros(1, 80, 8, device='cuda')
        assert lib.dsd_prepare(h, 1, 8, x.data_ptr(), 0, 0, 0, None) == -3 and b'dsd_load_weights' in lib.dsd_last_error()
        assert lib.dsd_sample_ddpm(h, x.data_ptr(), x.data_ptr(), 5, None) == -3
        assert lib.dsd_q_sample(h, x.data_ptr(), x.data_ptr(), 0, x.data_ptr(), None) == -3
        assert lib.dsd_set_schedule(h, None, 10) == -1
        assert lib.dsd_set_layer_tile(h, 48) == -1
        betas = (C.c_double * 4)(1e-4, 1e-3, 1e-2, 2e-2)
        assert lib.dsd_set_schedule(h, betas, 4) == 0
        out = (C.c_float * 4)()
        assert lib.dsd_get_schedule_table(h, 12, out, 4) == -1
        assert lib.dsd_get_schedule_table(h, 0, out, 5) == -3
        assert lib.dsd_get_schedule_table(h, 0, out, 4) == 0 and abs(out[3] - 2e-2) < 1e-9
    finally:
        lib.dsd_destroy(h)
    bad = _lib.DsdConfig(80, 256, 256, 20, 5)           # dilation 16 > supported halo
    assert lib.dsd_create(C.byref(bad), 0, C.byref(h)) == -1 and b'dilation' in lib.dsd_last_error()
    assert lib.dsd_create(C.byref(cfg), 99, C.byref(h)) == -1


def test_engine_rejects_wrong_shapes(lj):
    gd = lj['gd']
    cond, x, _ = _inputs(2, 2, 40)
    eng = gd.denoise_fn.bind_cond(_dev_cond(cond))
    with pytest.raises(ValueError):
        eng.denoise(torch.zeros(2, 80, 41, device='cuda'), 3)
    with pytest.raises(ValueError):
        eng.sample_ddpm(torch.zeros(2, 80, 40, device='cuda'), torch.zeros(3, 2, 80, 40, device='cuda'), 4)
    with pytest.raises(ValueError):
        eng.prepare(torch.zeros(2, 256, 40))            # CPU tensor
    with pytest.raises(RuntimeError):
        eng.sample_ddpm(torch.zeros(2, 80, 40, device='cuda'), torch.zeros(101, 2, 80, 40, device='cuda'), 101)   # > schedule
