"""Generate tests/golden/pe_opencpop.npz from the REAL reference PitchExtractor (build container only): python -m oracle.make_golden_pe
The reference module is built by its own constructor under the shipped opencpop e2e config (pe_enable: true), must accept the
synthetic state_dict with strict=True, and its eval forward is recorded."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = dict(config='usr/configs/midi/e2e/opencpop/ds100_adj_rel.yaml', B=3, T=47, seed=401)
HP_KEYS = ('hidden_size', 'predictor_hidden', 'predictor_kernel', 'ffn_padding', 'pitch_type', 'use_uv', 'pitch_norm')


def main():
    import torch
    sys.path.insert(0, ROOT)
    from oracle.ref_driver import Reference
    from oracle import pe_oracle as PO
    ref = Reference(CASE['config'])
    hp = {k: ref.hparams[k] for k in HP_KEYS}
    from modules.fastspeech.pe import PitchExtractor
    m = PitchExtractor().eval()
    p = PO.synth_extractor_params(hp, CASE['seed'] + 1000)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in PO.extractor_shapes(hp).items()}
    m.load_state_dict(p, strict=True)
    mel = PO.synth_mel(CASE['B'], CASE['T'], CASE['seed'])
    with torch.no_grad():
        r = m(mel)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'pe_opencpop.npz'), pitch_pred=r['pitch_pred'].numpy(),
                        f0_denorm_pred=r['f0_denorm_pred'].numpy(), hp=np.array(repr(hp)))
    print({k: tuple(v.shape) for k, v in r.items()}, 'f0 range', float(r['f0_denorm_pred'].min()), float(r['f0_denorm_pred'].max()),
          'voiced frac', float((r['f0_denorm_pred'] > 0).float().mean()), hp)


if __name__ == '__main__':
    main()
