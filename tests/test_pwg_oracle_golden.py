"""CPU: the ParallelWaveGAN oracle (oracle/pwg_oracle.py) against the fixtures the live reference produced (oracle/make_golden_pwg.py asserts
bit equality when it writes them), and the host side of diffsinger_amd.pwg: module tree, weight-norm removal, registry."""
import numpy as np
import pytest
import torch

from oracle import pwg_oracle as PO
from oracle.pwg_cases import CASES
from tests import pwg_helpers as PH


@pytest.mark.parametrize('name', list(CASES))
def test_oracle_reproduces_the_reference_fixture(name):
    case, cfg, m, state, inp = PH.case_setup(name)
    g = PH.load_golden(name)
    with torch.no_grad():
        y, c_up = PO.generator_forward(PO.plain_params(state), cfg, inp['x'], inp['c'], inp.get('pitch'))
    assert y.shape == g['y'].shape
    assert float(np.abs(y.numpy() - g['y']).max()) <= 1e-6                 # bit-equal on the torch build that wrote the fixture
    assert float(np.abs(c_up[:, :, :64].numpy() - g['c_up_head']).max()) <= 1e-6
    assert abs(float(c_up.double().sum()) - float(g['c_up_checksum'][0])) <= 1e-3 * max(1.0, abs(float(g['c_up_checksum'][0])))


@pytest.mark.parametrize('name', list(CASES))
def test_module_tree_and_weight_norm_removal(name):
    case, cfg, m, state, inp = PH.case_setup(name)
    assert any(k.endswith('weight_g') for k in m.state_dict())
    m.remove_weight_norm()
    plain = PO.plain_params(state)
    sd = m.state_dict()
    assert set(sd) == set(plain)
    for k in sd:
        assert torch.equal(sd[k], plain[k]), k
    # a state saved after remove_weight_norm() loads into a fresh (weight-normed) module
    from diffsinger_amd.pwg import ParallelWaveGANGenerator
    m2 = ParallelWaveGANGenerator(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in case['gen'].items()})
    m2.load_state_dict(sd, strict=True)
    assert not any(k.endswith('weight_g') for k in m2.state_dict())


def test_unsupported_configurations_raise():
    from diffsinger_amd.pwg import ParallelWaveGANGenerator
    for kw in (dict(kernel_size=5), dict(residual_channels=32), dict(use_causal_conv=True), dict(upsample_net='UpsampleNetwork'),
               dict(upsample_params={'upsample_scales': [4, 4], 'nonlinear_activation': 'ReLU'})):
        with pytest.raises(NotImplementedError):
            ParallelWaveGANGenerator(**kw)


def test_registry_knows_pwg_and_there_is_no_cpu_path():
    from diffsinger_amd.vocoder import get_vocoder_cls
    from diffsinger_amd.pwg import PWG, ParallelWaveGANGenerator
    assert get_vocoder_cls({'vocoder': 'pwg'}) is PWG                       # configs/tts/base.yaml:88
    assert get_vocoder_cls({'vocoder': 'vocoders.pwg.PWG'}) is PWG
    m = ParallelWaveGANGenerator(layers=4, stacks=2)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 256), torch.zeros(1, 80, 5))
