"""GPU, HELD BACK (tests/conftest.py: runs only with DSD_RUN_UNVERIFIED=1; the KERNEL has run - tools/lat_bf_probe.py, profiles/r04f / r04g:
bit-identical, 39.4 -> 32.3 ms - this module has not): the latency path with the branch-free K-half conv (DSD_LAT_BF=1: k_lat_conv<kLatG8BF>,
csrc/dsd_kernels.hpp ConvB<LD, true>) against the default k_lat_conv<8>.  The chunk -> pointer map is the same function: BIT-identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _loop(preset, B, T, K, bf, monkeypatch):
    import diffsinger_amd
    from diffsinger_amd import hparams
    from diffsinger_amd.synth import presets
    monkeypatch.setenv('DSD_LAT_BF', '1' if bf else '0')
    pre = presets()[preset]
    hparams.clear()
    diffsinger_amd.use_preset(preset)
    torch.manual_seed(3)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=K, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(5)
    cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, device='cuda', generator=g)
    noise = torch.randn(K, B, 1, 80, T, device='cuda', generator=g)
    eng = gd._engine(cond)                                  # the engine reads DSD_LAT_BF when it is created
    with torch.no_grad():
        out = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0)
        plms = gd.inference(cond, x_T=x_T, K_step=100, pndm_speedup=20) if K >= 20 else out
    return out.cpu().numpy(), plms.cpu().numpy(), eng.lat_split()


@pytest.mark.parametrize('preset,B,T,K', [('lj_ds_beta6', 1, 512, 20), ('opencpop_ds60_rel', 2, 300, 12), ('opencpop_ds60_rel', 1, 33, 5),
                                          ('opencpop_ds60_rel', 3, 70, 6), ('lj_ds_beta6', 1, 5, 3)])
def test_branch_free_latency_conv_equals_the_default(preset, B, T, K, monkeypatch):
    a, pa, ga = _loop(preset, B, T, K, False, monkeypatch)
    b, pb, gb = _loop(preset, B, T, K, True, monkeypatch)
    assert ga == 8 and gb == 8, 'these batches take the G = 8 latency kernels'
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(pa, pb)
    assert np.isfinite(b).all()
