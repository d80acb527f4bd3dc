"""Row f4: the `FFT` candidate denoiser (usr/diff/candidate_decoder.py:35-96).  CPU: the oracle against the fixture generated from
the reference class; registry / state_dict surface of the HIP module.  GPU: the HIP module and the generic DDPM loop driven
by it against the same fixture."""
import numpy as np
import pytest
import torch

from tests import fft_helpers as FH


def _golden():
    return dict(np.load(FH.GOLDEN))


def test_oracle_matches_reference_fixture_bitwise():
    from oracle import fft_decoder_oracle as O
    from tests.fs2_helpers import oracle_params
    m, hp, params = FH.build_module()
    inp = FH.make_inputs()
    with torch.no_grad():
        eps = O.fft_forward(oracle_params(params), hp, inp['x'], inp['t'], inp['cond'])
    np.testing.assert_array_equal(eps.numpy(), _golden()['eps'])


def test_registry_and_surface():
    import diffsinger_amd
    from diffsinger_amd import hparams
    m, hp, params = FH.build_module()
    assert type(diffsinger_amd.DIFF_DECODERS['fft'](hparams)).__name__ == 'FFT'
    reg = {'wavenet': None, 'fft': None}
    diffsinger_amd.register(reg)
    assert reg['fft'] is diffsinger_amd.DIFF_DECODERS['fft'] and 'wavenet_hip' in reg
    keys = set(m.state_dict())
    for k in ('input_projection.weight', 'mlp.0.weight', 'mlp.2.bias', 'get_decode_inp.weight', 'get_mel_out.bias', 'pos_embed_alpha',
              'layers.3.op.ffn.ffn_1.weight', 'layer_norm.weight'):
        assert k in keys, k
    assert tuple(m.get_decode_inp.weight.shape) == (256, 768)
    with pytest.raises(RuntimeError, match='no CPU path'):
        inp = FH.make_inputs()
        m(inp['x'], inp['t'], inp['cond'])


@pytest.mark.gpu
def test_hip_fft_denoiser_and_ddpm_match_reference():
    import diffsinger_amd
    from diffsinger_amd.synth import presets
    g = _golden()
    m, hp, params = FH.build_module()
    inp = FH.make_inputs()
    d = torch.device('cuda', 0)
    m = m.to(d)
    with torch.no_grad():
        eps = m(inp['x'].to(d), inp['t'].to(d), inp['cond'].to(d)).cpu().numpy()
    err = float(np.abs(eps - g['eps']).max())
    print(f'FFT denoiser: max-abs eps err {err:.3e} (max|eps| {float(np.abs(g["eps"]).max()):.2f})')
    assert eps.shape == g['eps'].shape and err <= 2e-5
    pre = presets()[FH.PRESET]
    gd = diffsinger_amd.GaussianDiffusion(None, 80, m, timesteps=pre['timesteps'], K_step=FH.K, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).to(d).eval()
    with torch.no_grad():
        mel, x = gd.inference(inp['cond'].to(d), x_T=inp['x'].to(d), noise=inp['noise'].to(d), K_step=FH.K, pndm_speedup=0, return_x=True)
    e_x, e_m = float(np.abs(x.cpu().numpy() - g['x_final']).max()), float(np.abs(mel.cpu().numpy() - g['mel']).max())
    print(f'FFT-driven DDPM K={FH.K}: max-abs err x {e_x:.3e}, mel {e_m:.3e}')
    assert e_x <= 1e-4 and e_m <= 1e-4


@pytest.mark.gpu
def test_p_losses_through_the_fft_denoiser_every_gradient_vs_the_oracle_under_autograd():
    """The reference trains whatever DIFF_DECODERS returns (usr/diffsinger_task.py:23-27, shallow_diffusion_tts.py:213-231): p_losses with the FFT
    candidate as denoise_fn - q_sample, FFT.forward_train on the HIP operators (backward on HIP kernels), L1 - against the oracle (bit-equal to
    the reference class, `test_oracle_matches_reference_fixture_bitwise`) under torch autograd on the CPU: the loss and EVERY parameter gradient."""
    import diffsinger_amd
    from diffsinger_amd.synth import presets
    from oracle import diffnet_oracle as DO
    from oracle import fft_decoder_oracle as O
    m, hp, params = FH.build_module()
    inp = FH.make_inputs()
    d = torch.device('cuda', 0)
    m = m.to(d)
    pre = presets()[FH.PRESET]
    gd = diffsinger_amd.GaussianDiffusion(None, 80, m, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).to(d)
    gd.eval()                                                   # dropout off: the comparison is deterministic (the reference's dropout positions
    for p in m.parameters():                                    # are covered by the FastSpeech2 training tests, tests/test_gpu_fs2_train.py)
        p.requires_grad_(True)
    g = torch.Generator().manual_seed(91)
    x0 = torch.randn(FH.B, 1, 80, FH.T, generator=g).clamp_(-1, 1)
    noise = torch.randn(FH.B, 1, 80, FH.T, generator=g)
    t = torch.tensor([17, 3])
    loss = gd.p_losses(x0.to(d), t.to(d), inp['cond'].to(d), noise=noise.to(d))
    loss.backward()
    # the oracle: the same q_sample + denoiser + L1 in plain torch on the CPU
    sch = DO.make_schedule(H_betas(pre))
    po = {k: v.clone().requires_grad_(True) for k, v in params.items() if v.is_floating_point()}
    po.update({k: v for k, v in params.items() if not v.is_floating_point()})
    shape = (FH.B, 1, 1, 1)
    x_noisy = sch['sqrt_alphas_cumprod'][t].reshape(shape) * x0 + sch['sqrt_one_minus_alphas_cumprod'][t].reshape(shape) * noise
    want = (noise - O.fft_forward(po, hp, x_noisy, t, inp['cond'])).abs().mean()
    want.backward()
    print(f'p_losses through FFT: loss {float(loss.detach()):.7f} (oracle {float(want.detach()):.7f})')
    assert abs(float(loss.detach()) - float(want.detach())) <= 2e-6 * max(1.0, abs(float(want.detach())))
    worst = ('', 0.0)
    n = 0
    for k, p in m.named_parameters():
        if k not in po or po[k].grad is None:
            continue
        gw = po[k].grad
        assert p.grad is not None, k
        rel = float((p.grad.cpu() - gw).abs().max()) / max(float(gw.abs().max()), 1e-12)
        worst = max(worst, (k, rel), key=lambda kv: kv[1])
        n += 1
    print(f'{n} parameter gradients, worst relative max-abs error {worst[1]:.3e} ({worst[0]})')
    assert n >= 40 and worst[1] <= 2e-5


def H_betas(pre):
    from tests import helpers as H
    return H.betas_for(pre)
