"""Shared plumbing of the ParallelWaveGAN tests: the HIP module of a case with its synthetic state, the inputs, the fixture."""
import os

import numpy as np
import torch

from oracle.pwg_cases import CASES, gen_config, make_inputs, synth_state

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def case_setup(name):
    from diffsinger_amd.pwg import ParallelWaveGANGenerator
    case = CASES[name]
    cfg = gen_config(case)
    m = ParallelWaveGANGenerator(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in case['gen'].items()})
    state = synth_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, case['seed'])
    m.load_state_dict(state, strict=True)
    return case, cfg, m, state, make_inputs(case, cfg)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + '.npz')))
