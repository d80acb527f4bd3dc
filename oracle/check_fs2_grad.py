"""Pin the FastSpeech2 oracle UNDER AUTOGRAD to the live reference (build container only: needs /root/reference).

    python -m oracle.check_fs2_grad NAME        # one process per case (hparams are process-global in the reference)

The reference's FastSpeech2 / FastSpeech2MIDI (modules/fastspeech/fs2.py:93-149, modules/diffsinger_midi/fs2.py:55-118) runs its training
forward - infer=False, skip_decoder=True as GaussianDiffusion.forward calls it (usr/diff/shallow_diffusion_tts.py:236), eval() so that dropout
is the identity - on the seeded case, a fixed random linear functional of its outputs (decoder_inp, dur, pitch_pred / cwt + statistics) is
back-propagated, and every parameter gradient is compared with the gradients torch autograd gives on oracle/fs2_oracle.py for the same
functional.  Prints one JSON line; exit code 1 on a mismatch.  TEST INFRASTRUCTURE ONLY."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOSS_KEYS = ('decoder_inp', 'dur', 'pitch_pred', 'cwt', 'f0_mean', 'f0_std', 'energy_pred')


def loss_of(ret, seed):
    """sum_k <ret[k], R_k> with R_k ~ N(0, 1) drawn from a generator seeded per key: the same functional for every implementation."""
    import torch
    total = 0.0
    for i, k in enumerate(LOSS_KEYS):
        v = ret.get(k)
        if v is None or not torch.is_tensor(v) or not v.requires_grad:
            continue
        g = torch.Generator().manual_seed(seed * 100 + i)
        total = total + (v * torch.randn(v.shape, generator=g).to(v.device)).sum()
    return total


def main(name):
    import torch
    sys.path.insert(0, ROOT)
    from oracle import fs2_oracle as FO
    from oracle.fs2_cases import CASES, VOCAB
    from oracle.ref_driver import Reference
    from diffsinger_amd.synth import presets
    from tests import fs2_helpers as FH
    case, m_hip, hp_ours, params, inp = FH.case_setup(name)
    assert case['mode'] == 'teacher', 'the training forward is teacher-forced (mel2ph, f0, uv given)'
    ref = Reference(presets()[case['preset']]['source'], overrides=case.get('overrides'))
    hp = ref.hparams
    hp['cwt_scales'] = np.arange(10)
    enc = ref.TokenTextEncoder(None, vocab_list=[f'p{i}' for i in range(VOCAB - 3)], replace_oov=',')
    if hp.get('use_midi'):
        from modules.diffsinger_midi.fs2 import FastSpeech2MIDI as M
    else:
        from modules.fastspeech.fs2 import FastSpeech2 as M
    m = M(enc, 80).eval()
    m.load_state_dict(params, strict=True)
    kw = {k: v.clone() for k, v in inp.items() if k != 'txt_tokens'}
    r = m(inp['txt_tokens'], skip_decoder=True, infer=False, **kw)
    loss_of(r, case['seed']).backward()
    ref_grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}

    p = FH.oracle_params(params)
    kw = {k: v.clone() for k, v in inp.items() if k != 'txt_tokens'}
    o = FO.fs2_forward(p, hp_ours, inp['txt_tokens'], skip_decoder=True, **kw)
    loss_of(o, case['seed']).backward()
    ora_grads = {k: v.grad for k, v in p.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
    alias = {'encoder.embed_tokens.weight', 'encoder_embed_tokens.weight'}      # one tensor in the module, two keys in the state_dict / oracle
    worst, n_bitequal, missing = ('', 0.0), 0, []
    for k, g in ref_grads.items():
        if k in alias:                                   # the oracle may hold the shared embedding as two leaves: their gradients add up
            parts = [ora_grads[x] for x in alias if x in ora_grads]
            og = sum(parts) if parts else None
        else:
            og = ora_grads.get(k)
        if og is None:
            missing.append(k)
            continue
        e = float((og - g).abs().max() / max(float(g.abs().max()), 1e-30))
        n_bitequal += int(torch.equal(og, g))
        if e > worst[1]:
            worst = (k, e)
    res = {'case': name, 'parameters_with_gradient': len(ref_grads), 'bit_equal': n_bitequal, 'worst_rel_err': worst[1], 'worst_at': worst[0],
           'missing_in_oracle': missing, 'loss_ref': float(loss_of(r, case['seed'])), 'loss_oracle': float(loss_of(o, case['seed']))}
    print(json.dumps(res))
    return 0 if (not missing and worst[1] <= 1e-6) else 1


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
