"""GPU: the persistent K-step loop on frame-major LDS tiles with x resident in fragment order (csrc/dsd_loop_fm.hpp, opt-in DSD_LOOP_FM=1)
against the default persistent loop (csrc/dsd_loop.hpp) and the per-layer kernels.  Same arithmetic in the same order -> BIT-identical, no
timeout.  (Written without GPU time as tests/test_gpu_zz_*; first run on the hardware passed, profiles/r04c_pytest_gpu_loop_fm.txt.)"""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(name, loop_mode):
    from tests.gpu_helpers import build_hip
    case, pre, cfg, k_step, inp, smin, smax = H.case_setup(name)
    gd, _, _ = build_hip(case['preset'], k_step, legacy=bool(case.get('legacy')))
    cond = inp['cond'].transpose(1, 2).contiguous().cuda().transpose(1, 2)
    eng = gd._engine(cond)                                  # the engine reads DSD_LOOP_FM when it is created
    eng.set_loop_mode(loop_mode)
    with torch.no_grad():
        if case['kind'] == 'plms':
            out = gd.inference(cond, x_T=inp['x_T'].cuda(), K_step=k_step, pndm_speedup=case['interval'])
        elif case['gaussian']:
            out = gd.inference(cond, x_T=inp['x_T'].cuda(), noise=inp['noise'].cuda(), K_step=k_step, pndm_speedup=0)
        else:
            out = gd.inference(cond, fs2_mels=inp['fs2_mel'].cuda(), q_noise=inp['q_noise'].cuda(), noise=inp['noise'].cuda(),
                               K_step=k_step, pndm_speedup=0, gaussian_start=False)
    return out.cpu().numpy(), eng.loop_mode(), eng.loop_timeouts()


@pytest.mark.parametrize('name', ['ddpm_lj_k100', 'shallow_opencpop_k60', 'shallow_popcs_k51', 'plms_opencpop_i40', 'plms_opencpop_i250'])
def test_frame_major_loop_equals_the_verified_loop(name, monkeypatch):
    monkeypatch.setenv('DSD_LOOP_FM', '1')
    a, used_a, tmo_a = _run(name, 1)
    monkeypatch.setenv('DSD_LOOP_FM', '0')
    b, used_b, tmo_b = _run(name, 1)
    assert used_a == 1 and used_b == 1 and tmo_a == 0 and tmo_b == 0
    np.testing.assert_array_equal(a, b)
    g = H.load_golden(name)['out']
    scale = max(1.0, float(np.abs(g).max())) if 'plms' in name else 1.0
    assert float(np.abs(a - g).max()) / scale <= 1e-4


@pytest.mark.parametrize('B,T,K', [(8, 1024, 12), (5, 2048, 6), (3, 1000, 8), (2, 33, 5), (1, 5, 3)])
def test_frame_major_loop_full_width_and_edges(B, T, K, monkeypatch):
    """256 workgroups at once, chunks of whole utterances, ragged T, a one-tile utterance: against the per-layer kernels."""
    import diffsinger_amd
    from diffsinger_amd import hparams
    from diffsinger_amd.synth import presets
    pre = presets()['opencpop_ds60_rel']                       # dilation cycle 4: halos up to 8 frames
    monkeypatch.setenv('DSD_LOOP_FM', '1')
    hparams.clear()
    diffsinger_amd.use_preset('opencpop_ds60_rel')
    torch.manual_seed(3)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=K, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(5)
    cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, device='cuda', generator=g)
    noise = torch.randn(K, B, 1, 80, T, device='cuda', generator=g)
    eng = gd._engine(cond)
    outs = []
    for mode in (1, 0):
        eng.set_loop_mode(mode)
        with torch.no_grad():
            outs.append(gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).cpu().numpy())
        assert eng.loop_mode() == mode
        assert eng.loop_timeouts() == 0
    np.testing.assert_array_equal(outs[0], outs[1])
    assert np.isfinite(outs[0]).all()
