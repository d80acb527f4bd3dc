"""Generate tests/golden/hifigan_*.npz from the REAL reference generator (build container only): python -m oracle.make_golden_hifigan
Seeded synthetic weights (oracle/hifigan_oracle.synth_generator_params, loaded strict=True into the reference module after
remove_weight_norm()), seeded mel / f0, torch.manual_seed for the source module's draws."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {'hifigan_plain': dict(nsf=False, B=2, T=24, seed=301), 'hifigan_nsf': dict(nsf=True, B=2, T=24, seed=302)}
CONFIG = dict(resblock='1', upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=128,
              resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], audio_sample_rate=24000)


def inputs(case):
    import torch
    g = torch.Generator().manual_seed(case['seed'])
    mel = torch.randn(case['B'], 80, case['T'], generator=g)
    f0 = None
    if case['nsf']:
        f0 = torch.rand(case['B'], case['T'], generator=g) * 300 + 80
        f0[0, 5:9] = 0
        f0[1, 20:] = 0
    return mel, f0


def main():
    import torch
    sys.path.insert(0, ROOT)
    from oracle.ref_driver import Reference
    from oracle import hifigan_oracle as HO
    ref = Reference('configs/tts/hifigan.yaml')
    import scipy.signal, scipy.signal.windows
    if not hasattr(scipy.signal, 'kaiser'):
        scipy.signal.kaiser = scipy.signal.windows.kaiser
    from modules.hifigan.hifigan import HifiGanGenerator
    for name, case in CASES.items():
        h = dict(CONFIG, use_pitch_embed=case['nsf'])
        for k in ('resblock', 'upsample_rates', 'upsample_kernel_sizes', 'upsample_initial_channel', 'resblock_kernel_sizes'):
            assert ref.hparams[k] == CONFIG[k], k                   # CONFIG is the shipped configs/tts/hifigan.yaml
        m = HifiGanGenerator(h).eval()
        m.remove_weight_norm()
        p = HO.synth_generator_params(h, case['seed'] + 1000)
        m.load_state_dict(p, strict=True)
        mel, f0 = inputs(case)
        with torch.no_grad():
            torch.manual_seed(case['seed'])
            wav = m(mel, f0)
        np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), wav=wav.numpy())
        print(name, wav.shape, float(wav.abs().max()), float(wav.std()))


if __name__ == '__main__':
    main()
