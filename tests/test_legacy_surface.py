"""Row a15: `diffsinger_amd.legacy.GaussianDiffusion` mirrors usr/diff/diffusion.py::GaussianDiffusion - constructor without
K_step, cosine schedule unless betas are given (hparams['schedule_type'] ignored), same 14 buffers, loud failure without
a device.  (The numerical parity of its loop is the GPU case `ddpm_legacy_cosine`, fixture generated from the reference.)"""
import inspect

import numpy as np
import pytest
import torch

import diffsinger_amd
from diffsinger_amd import hparams
from tests import helpers as H


def _build(**kw):
    hparams.clear()
    diffsinger_amd.use_preset('lj_ds_beta6')             # schedule_type linear: must NOT be used by the legacy class
    pre = H.presets()['lj_ds_beta6']
    from diffsinger_amd.legacy import GaussianDiffusion
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    return GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], spec_min=pre['spec_min'], spec_max=pre['spec_max'], **kw), pre


def test_signature_and_schedule():
    from diffsinger_amd.legacy import GaussianDiffusion
    names = list(inspect.signature(GaussianDiffusion.__init__).parameters)
    assert names[:9] == ['self', 'phone_encoder', 'out_dims', 'denoise_fn', 'timesteps', 'loss_type', 'betas', 'spec_min', 'spec_max']
    gd, pre = _build()
    assert gd.num_timesteps == gd.K_step == pre['timesteps'] and gd.loss_type == 'l1'
    g = H.load_golden('ddpm_legacy_cosine')
    for k in ('betas', 'alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'posterior_log_variance_clipped', 'posterior_mean_coef2'):
        np.testing.assert_array_equal(getattr(gd, k).numpy(), g['sched_' + k], err_msg=k)      # the reference's own buffers
    assert len(list(gd.buffers())) == 14
    gd2, _ = _build(betas=np.linspace(1e-4, 0.06, 100))
    np.testing.assert_array_equal(gd2.betas.numpy(), np.linspace(1e-4, 0.06, 100).astype(np.float32))


def test_no_cpu_path_and_loud_failure_without_fastspeech2():
    gd, _ = _build()
    with pytest.raises(RuntimeError):
        gd.sample(torch.zeros(1, 256, 8))                # parameters on the CPU: the HIP engine refuses
    with pytest.raises(RuntimeError, match='no FastSpeech2 attached'):
        gd(torch.zeros(1, 4, dtype=torch.long), infer=False)      # both branches of forward need self.fs2 (tests/test_gpu_surfaces.py runs them)


def test_p_losses_says_so_for_an_inference_only_denoiser():
    """a denoiser without a training forward (neither DiffNet's fused stack nor a `forward_train` like the FFT candidate's, round 5)
    gets a clear NotImplementedError from p_losses, not an AttributeError."""
    import diffsinger_amd
    from diffsinger_amd import hparams
    from diffsinger_amd.synth import presets
    pre = presets()['lj_ds_beta6']
    hparams.clear()
    diffsinger_amd.use_preset('lj_ds_beta6')
    gd = diffsinger_amd.GaussianDiffusion(None, 80, torch.nn.Identity(), timesteps=10, K_step=10, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max'])
    with pytest.raises(NotImplementedError, match='no training forward'):
        gd.p_losses(torch.zeros(1, 1, 80, 8), torch.zeros(1, dtype=torch.long), torch.zeros(1, 256, 8))
