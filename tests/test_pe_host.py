"""CPU: the HIP PitchExtractor's parameter tree equals the reference's (names, shapes, buffers: strict load both ways) and the
module refuses to run without the device."""
import ast
import os

import numpy as np
import pytest
import torch

from diffsinger_amd import _lib, hparams
from oracle import pe_oracle as PO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hp():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pe_opencpop.npz'))
    return ast.literal_eval(str(g['hp']))


def build():
    hp = _hp()
    hparams.clear()
    hparams.update(hp, dur_loss='mse')
    from diffsinger_amd.pe import PitchExtractor
    return hp, PitchExtractor().eval()


def test_state_dict_layout_is_the_reference_layout():
    hp, m = build()
    want = PO.extractor_shapes(hp)                                    # asserted equal to the reference module's when the fixture is generated
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    m.load_state_dict(PO.synth_extractor_params(hp, 1), strict=True)


def test_no_cpu_path_and_eval_only():
    hp, m = build()
    with pytest.raises(RuntimeError, match='no CPU path'):
        m(torch.zeros(1, 8, 80))


def test_new_symbols_exported():
    lib = _lib.load()
    assert lib.dsf_channel_affine(None, None, None, None, None, 1, 1, 1, None) == -1
    assert lib.dsf_group_norm(None, None, None, None, None, 1, 16, 1, 1, 1e-5, 0, None) == -1
    assert b'dsf_group_norm' in lib.dsd_last_error()
