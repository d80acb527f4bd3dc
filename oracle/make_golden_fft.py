"""Generate tests/golden/fft_*.npz from the REAL reference `FFT` candidate denoiser (usr/diff/candidate_decoder.py), build
container only.    python -m oracle.make_golden_fft
Weights: the HIP module's seeded synthetic state_dict (oracle/fs2_cases.synth_params), loaded strict=True into the reference
class (which also proves the module tree has the reference's names and shapes).  Recorded: one denoiser evaluation with
per-utterance t, and an 8-step DDPM loop of the reference GaussianDiffusion driven by it with injected noise."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def main():
    import torch
    sys.path.insert(0, ROOT)
    from oracle.ref_driver import Reference
    from diffsinger_amd.synth import presets
    from tests import fft_helpers as FH
    ref = Reference(presets()[FH.PRESET]['source'])
    hp = ref.hparams
    m_hip, hp_ours, params = FH.build_module()
    from usr.diff.candidate_decoder import FFT
    m = FFT(hp['hidden_size'], hp['dec_layers'], hp['dec_ffn_kernel_size'], hp['num_heads']).eval()
    ref_shapes = {k: (tuple(v.shape), v.dtype) for k, v in m.state_dict().items()}
    ours = {k: (tuple(v.shape), v.dtype) for k, v in m_hip.state_dict().items()}
    assert ref_shapes == ours, set(ref_shapes) ^ set(ours)
    m.load_state_dict(params, strict=True)
    inp = FH.make_inputs()
    out = {}
    with torch.no_grad():
        out['eps'] = m(inp['x'], inp['t'], inp['cond']).numpy()
        enc = ref.TokenTextEncoder(None, vocab_list=['a', 'b', 'c'], replace_oov=',')
        gd = ref.sdt.GaussianDiffusion(enc, 80, m, timesteps=hp['timesteps'], K_step=FH.K, loss_type='l1', spec_min=hp['spec_min'],
                                       spec_max=hp['spec_max']).eval()
        x = ref.sample_ddpm(gd, inp['x'], inp['cond'], list(inp['noise']), FH.K)
        out['x_final'] = x.numpy()
        out['mel'] = gd.denorm_spec(x[:, 0].transpose(1, 2)).numpy()
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'fft_decoder.npz'), **out)
    print({k: v.shape for k, v in out.items()}, 'max|eps|', float(np.abs(out['eps']).max()))


if __name__ == '__main__':
    main()
