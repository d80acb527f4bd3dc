"""Row f2 groundwork (no HIP vocoder yet): the HiFi-GAN / NSF-HiFi-GAN generator oracle (oracle/hifigan_oracle.py) bit-for-bit
against the live reference generator - with and without the NSF pitch source, before and after remove_weight_norm() - in the
build container (skipped where /root/reference is absent)."""
import os
import subprocess
import sys

import pytest

from oracle.ref_driver import reference_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from oracle.ref_driver import Reference
from oracle import hifigan_oracle as HO
ref = Reference('configs/tts/hifigan.yaml')
import scipy.signal, scipy.signal.windows
if not hasattr(scipy.signal, 'kaiser'):                   # modules/parallel_wavegan/layers/pqmf.py:12 imports the pre-1.13 name (unused here)
    scipy.signal.kaiser = scipy.signal.windows.kaiser
from modules.hifigan.hifigan import HifiGanGenerator
h = dict(ref.hparams)
h.update(use_pitch_embed=%(nsf)r, audio_sample_rate=24000, upsample_initial_channel=64, resblock=%(resblock)r)
torch.manual_seed(11)
m = HifiGanGenerator(h).eval()
g = torch.Generator().manual_seed(5)
with torch.no_grad():
    for v in m.parameters():                              # init_weights is N(0, 0.01): make the signal path non-trivial
        v.add_(0.05 * torch.randn(v.shape, generator=g))
from diffsinger_amd.vocoder import HifiGanGenerator as HipGen            # the HIP module's parameter tree == the reference module's
mine = HipGen(h)
assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == {k: tuple(v.shape) for k, v in m.state_dict().items()}
mine.load_state_dict(m.state_dict(), strict=True)
B, T = 2, 37
mel = torch.randn(B, 80, T, generator=g)
f0 = None
if %(nsf)r:
    f0 = torch.rand(B, T, generator=g) * 300 + 80
    f0[0, 10:15] = 0
    f0[1, 30:] = 0
for stage in ('weight_norm', 'plain'):
    if stage == 'plain':
        m.remove_weight_norm()
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        torch.manual_seed(99); a = m(mel, f0)
        torch.manual_seed(99); b = HO.generator(p, h, mel, f0)
    assert a.shape == (B, 1, T * 256), a.shape
    assert torch.equal(a, b), (stage, float((a - b).abs().max()))
print('HIFIGAN_EQUAL_OK', float(a.abs().max()))
'''


@pytest.mark.skipif(not reference_available(), reason='/root/reference not mounted')
@pytest.mark.parametrize('resblock', ['1', '2'])
@pytest.mark.parametrize('nsf', [False, True])
def test_hifigan_oracle_bit_equal_to_live_reference(nsf, resblock):
    res = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, nsf=nsf, resblock=resblock)], capture_output=True, text=True)
    assert 'HIFIGAN_EQUAL_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


@pytest.mark.parametrize('name', ['hifigan_plain', 'hifigan_nsf'])
def test_hifigan_oracle_matches_reference_fixture_bitwise(name):
    """Everywhere (no reference needed): the oracle against the fixture oracle/make_golden_hifigan.py recorded from the reference."""
    import numpy as np
    import torch
    from oracle import hifigan_oracle as HO
    from oracle.make_golden_hifigan import CASES, CONFIG, inputs
    case = CASES[name]
    h = dict(CONFIG, use_pitch_embed=case['nsf'])
    p = HO.synth_generator_params(h, case['seed'] + 1000)
    mel, f0 = inputs(case)
    with torch.no_grad():
        torch.manual_seed(case['seed'])
        wav = HO.generator(p, h, mel, f0)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))['wav']
    assert wav.shape == g.shape == (case['B'], 1, case['T'] * 256)
    np.testing.assert_array_equal(wav.numpy(), g)
    assert 0.01 < float(np.abs(g).max()) <= 1.0
