"""Generate tests/golden/fs2_*.npz by running the REAL reference FastSpeech2 / FastSpeech2MIDI (build container only).

    python -m oracle.make_golden_fs2 [NAME ...]

One subprocess per case (hparams are process-global in the reference).  The reference module is built by the reference's own
constructor from the shipped YAML, must accept our synthetic state_dict with strict=True (this is also the check that the HIP
module tree has exactly the reference's parameter names and shapes), and its inference forward is recorded."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def run_case(name):
    import torch
    sys.path.insert(0, ROOT)
    from oracle.fs2_cases import CASES, VOCAB, OUT_KEYS
    from oracle.ref_driver import Reference
    from diffsinger_amd.synth import presets
    from tests import fs2_helpers as FH
    case, m_hip, hp_ours, params, inp = FH.case_setup(name)
    ref = Reference(presets()[case['preset']]['source'], overrides=case.get('overrides'))
    hp = ref.hparams
    hp['cwt_scales'] = np.arange(10)                 # only its length is used (utils/cwt.py:118-125); set by the binarizer normally
    for k, v in hp_ours.items():                     # the preset really is the YAML
        if k in hp and k not in ('spec_min', 'spec_max'):
            assert hp[k] == v, (k, hp[k], v)
    enc = ref.TokenTextEncoder(None, vocab_list=[f'p{i}' for i in range(VOCAB - 3)], replace_oov=',')
    assert len(enc) == VOCAB
    if hp.get('use_midi'):
        from modules.diffsinger_midi.fs2 import FastSpeech2MIDI as M
    else:
        from modules.fastspeech.fs2 import FastSpeech2 as M
    m = M(enc, 80).eval()
    ref_shapes = {k: (tuple(v.shape), v.dtype) for k, v in m.state_dict().items()}
    assert ref_shapes == FH.shapes_of(m_hip), set(ref_shapes) ^ set(FH.shapes_of(m_hip))
    m.load_state_dict(params, strict=True)
    kw = {k: v.clone() for k, v in inp.items() if k != 'txt_tokens'}
    with torch.no_grad():
        r = m(inp['txt_tokens'], infer=True, **kw)
    out = {}
    for k in OUT_KEYS:
        if k in r and r[k] is not None:
            out[k] = r[k].detach().numpy()
    out['encoder_out'] = None
    del out['encoder_out']
    out['torch_version'] = np.array(torch.__version__)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + '.npz'), **out)
    print(name, {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.shape})


def main(argv):
    sys.path.insert(0, ROOT)
    from oracle.fs2_cases import CASES
    if len(argv) >= 2 and argv[0] == '--child':
        run_case(argv[1])
        return
    for n in (argv or list(CASES)):
        subprocess.run([sys.executable, '-m', 'oracle.make_golden_fs2', '--child', n], cwd=ROOT, check=True)


if __name__ == '__main__':
    main(sys.argv[1:])
