"""Shared plumbing of the FastSpeech2 (row f1) tests: build the HIP module for a case, its synthetic weights and inputs."""
import os

import numpy as np
import torch

import diffsinger_amd
from diffsinger_amd import hparams
from oracle.fs2_cases import CASES, VOCAB, make_inputs, synth_params

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def build_module(case_name):
    """The HIP FastSpeech2(.MIDI) for the case (on the CPU: construction needs no device) + its hparams dict."""
    case = CASES[case_name]
    hparams.clear()
    diffsinger_amd.use_preset(case['preset'])
    hparams.update(case.get('overrides', {}))
    from diffsinger_amd import fs2
    cls = fs2.FastSpeech2MIDI if hparams.get('use_midi') else fs2.FastSpeech2
    return cls(VOCAB, 80).eval(), dict(hparams)


def shapes_of(module):
    return {k: (tuple(v.shape), v.dtype) for k, v in module.state_dict().items()}


def case_setup(case_name):
    case = CASES[case_name]
    m, hp = build_module(case_name)
    params = synth_params(shapes_of(m), case['seed'] + 1000)
    m.load_state_dict(params, strict=True)
    return case, m, hp, params, make_inputs(case, bool(hp.get('use_midi')))


def oracle_params(params):
    """requires_grad=True on float tensors: torch's CPU linear picks its kernel by it, and the reference's parameters have it."""
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in params.items()}


def run_oracle(case_name):
    from oracle import fs2_oracle as FO
    case, m, hp, params, inp = case_setup(case_name)
    kw = {k: v.clone() for k, v in inp.items() if k != 'txt_tokens'}
    with torch.no_grad():
        return FO.fs2_forward(oracle_params(params), hp, inp['txt_tokens'], **kw)


def load_golden(case_name):
    return dict(np.load(os.path.join(GOLDEN_DIR, case_name + '.npz')))
