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
