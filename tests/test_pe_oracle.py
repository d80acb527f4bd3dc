"""Row f2: the PitchExtractor oracle (oracle/pe_oracle.py) against the reference-generated fixture (everywhere) and bit-for-bit against
the live reference module on a second input (build container only)."""
import ast
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import pe_oracle as PO
from oracle.make_golden_pe import CASE
from oracle.ref_driver import reference_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def golden():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pe_opencpop.npz'))
    return g, ast.literal_eval(str(g['hp']))


def test_pe_oracle_matches_reference_fixture_bitwise():
    g, hp = golden()
    p = {k: (v.requires_grad_(True) if v.is_floating_point() else v) for k, v in PO.synth_extractor_params(hp, CASE['seed'] + 1000).items()}
    mel = PO.synth_mel(CASE['B'], CASE['T'], CASE['seed'])
    with torch.no_grad():
        r = PO.pitch_extractor(p, hp, mel)
    np.testing.assert_array_equal(r['pitch_pred'].numpy(), g['pitch_pred'])
    np.testing.assert_array_equal(r['f0_denorm_pred'].numpy(), g['f0_denorm_pred'])
    f0 = g['f0_denorm_pred']
    assert (f0[1, -5:] == 0).all() and (f0[2, -10:] == 0).all()            # padding frames
    assert 0.2 < (f0 > 0).mean() < 0.9 and f0.max() < 2000                  # voiced and unvoiced frames both present


CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import torch
from oracle.ref_driver import Reference
from oracle import pe_oracle as PO
from oracle.make_golden_pe import CASE, HP_KEYS
ref = Reference(CASE['config'])
hp = {k: ref.hparams[k] for k in HP_KEYS}
from modules.fastspeech.pe import PitchExtractor
m = PitchExtractor().eval()
p = PO.synth_extractor_params(hp, 4242)
m.load_state_dict(p, strict=True)
mel = PO.synth_mel(2, 83, 77)
with torch.no_grad():
    a = m(mel)
    b = PO.pitch_extractor({k: v for k, v in m.state_dict().items()}, hp, mel)
for k in ('pitch_pred', 'f0_denorm_pred'):
    assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))
print('PE_EQUAL_OK')
'''


@pytest.mark.skipif(not reference_available(), reason='/root/reference not mounted')
def test_pe_oracle_bit_equal_to_live_reference():
    res = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT)], capture_output=True, text=True)
    assert 'PE_EQUAL_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
