"""CPU: the FastSpeech2 oracle (oracle/fs2_oracle.py) against the fixtures generated from the reference's own modules
(oracle/make_golden_fs2.py), and the module-tree parity of the HIP FastSpeech2 (state_dict names / shapes are asserted
against the live reference when the fixtures are generated; here: strict load of the synthetic state_dict, no CPU path)."""
import numpy as np
import pytest
import torch

from oracle.fs2_cases import CASES
from tests import fs2_helpers as FH


@pytest.mark.parametrize('name', list(CASES))
def test_fs2_oracle_matches_reference_bitwise(name):
    g = FH.load_golden(name)
    out = FH.run_oracle(name)
    for k, ref in g.items():
        if k == 'torch_version':
            continue
        a = out[k].detach().numpy()
        assert a.shape == ref.shape, (k, a.shape, ref.shape)
        np.testing.assert_array_equal(a, ref, err_msg=f'{name}:{k}')


def test_hip_module_has_no_cpu_path():
    case, m, hp, params, inp = FH.case_setup('fs2_popcs_teacher')
    with pytest.raises(RuntimeError, match='no CPU path'):
        m(inp['txt_tokens'], mel2ph=inp['mel2ph'], f0=inp['f0'], uv=inp['uv'], infer=True)
    with pytest.raises(RuntimeError, match='no CPU path'):                   # the training forward (autograd on HIP operators) neither
        m(inp['txt_tokens'], mel2ph=inp['mel2ph'], f0=inp['f0'].clone(), uv=inp['uv'], infer=False, skip_decoder=True)


def test_oracle_gradients_equal_the_live_reference_bitwise():
    """oracle/check_fs2_grad.py: every parameter gradient of the reference's FastSpeech2(.MIDI) training forward (infer=False, skip_decoder=True,
    eval mode) equals torch autograd on the oracle - one process per case (hparams are process-global there); build container only."""
    import json
    import os
    import subprocess
    import sys
    from oracle.ref_driver import reference_available
    if not reference_available():
        pytest.skip('/root/reference is not mounted')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in [n for n, c in CASES.items() if c['mode'] == 'teacher']:
        r = subprocess.run([sys.executable, '-m', 'oracle.check_fs2_grad', name], cwd=root, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
        res = json.loads(line)
        assert r.returncode == 0 and not res['missing_in_oracle'] and res['worst_rel_err'] == 0.0, res
        assert res['bit_equal'] == res['parameters_with_gradient'] >= 60, res


def test_shared_embedding_and_key_set():
    case, m, hp, params, inp = FH.case_setup('fs2_midi_cascade_teacher')
    assert m.encoder.embed_tokens is m.encoder_embed_tokens
    keys = set(m.state_dict())
    for k in ('encoder.layers.3.op.self_attn.in_proj_weight', 'decoder.pos_embed_alpha', 'decoder.embed_positions._float_tensor',
              'dur_predictor.conv.4.3.bias', 'pitch_predictor.conv.0.1.weight', 'midi_dur_layer.weight', 'is_slur_embed.weight', 'mel_out.bias'):
        assert k in keys, k
    assert 'encoder.embed_positions._float_tensor' not in keys          # rel_pos: RelPositionalEncoding has no state
