"""Generate tests/golden/pwg_*.npz by running the REAL reference ParallelWaveGANGenerator (build container only).

    python -m oracle.make_golden_pwg [NAME ...]

The reference module is built by its own constructor, must accept the synthetic weight-normed state with strict=True (the check that the HIP
module tree has exactly the reference's parameter names and shapes), gets remove_weight_norm() like vocoders/pwg.py:47 does, and its forward is
recorded together with the upsampled conditioning.  The oracle restatement must reproduce both BIT FOR BIT (asserted here)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
REFERENCE_ROOT = '/root/reference'


def main(argv):
    import torch
    sys.path.insert(0, ROOT)
    sys.dont_write_bytecode = True
    for n in ('librosa', 'pycwt'):
        sys.modules.setdefault(n, types.ModuleType(n))
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, 'kaiser'):                      # layers/pqmf.py:12 imports the pre-1.13 name (PQMF is not on this path)
        scipy.signal.kaiser = scipy.signal.windows.kaiser
    sys.path.insert(0, REFERENCE_ROOT)
    from modules.parallel_wavegan.models.parallel_wavegan import ParallelWaveGANGenerator as Ref
    from oracle import pwg_oracle as PO
    from oracle.pwg_cases import CASES, gen_config, make_inputs, synth_state
    from diffsinger_amd.pwg import ParallelWaveGANGenerator as Ours
    for name in (argv or list(CASES)):
        case = CASES[name]
        cfg = gen_config(case)
        m = Ref(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in case['gen'].items()})
        ours = Ours(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in case['gen'].items()})
        ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert ref_shapes == {k: tuple(v.shape) for k, v in ours.state_dict().items()}, set(ref_shapes) ^ set(ours.state_dict())
        state = synth_state(ref_shapes, case['seed'])
        m.load_state_dict(state, strict=True)
        m.remove_weight_norm()
        m = m.eval()
        inp = make_inputs(case, cfg)
        with torch.no_grad():
            y = m(inp['x'], inp['c'], inp.get('pitch'))
            c_in = inp['c']
            if cfg['use_pitch_embed']:
                c_in = m.c_proj(torch.cat([c_in.transpose(1, 2), m.pitch_embed(inp['pitch'])], -1)).transpose(1, 2)
            c_up = m.upsample_net(c_in)
            yo, co = PO.generator_forward(PO.plain_params(state), cfg, inp['x'], inp['c'], inp.get('pitch'))
        assert torch.equal(yo, y) and torch.equal(co, c_up), (name, float((yo - y).abs().max()), float((co - c_up).abs().max()))
        out = {'y': y.numpy(), 'c_up_checksum': np.array([float(c_up.double().sum()), float(c_up.double().abs().sum())]),
               'c_up_head': c_up[:, :, :64].numpy(), 'torch_version': np.array(torch.__version__)}
        os.makedirs(GOLDEN_DIR, exist_ok=True)
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + '.npz'), **out)
        print(name, 'y', tuple(y.shape), 'max|y|', float(y.abs().max()), 'std', float(y.std()), '- oracle bit-equal to the reference')


if __name__ == '__main__':
    main(sys.argv[1:])
