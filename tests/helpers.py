"""Shared test plumbing: rebuild a golden case's inputs/weights from seeds and run it through the oracle."""
import os

import numpy as np
import torch

from diffsinger_amd.synth import make_inputs, presets
from oracle import diffnet_oracle as O
from oracle.golden_cases import CASES, FINAL_PROJ_STD, WEIGHT_SEED

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
_PARAM_CACHE = {}


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + '.npz')))


def net_config(pre) -> O.NetConfig:
    return O.NetConfig(mel_bins=pre['audio_num_mel_bins'], residual_channels=pre['residual_channels'],
                       encoder_hidden=pre['hidden_size'], residual_layers=pre['residual_layers'],
                       dilation_cycle_length=pre['dilation_cycle_length'])


def oracle_params(cfg: O.NetConfig):
    key = (cfg, WEIGHT_SEED, FINAL_PROJ_STD)
    if key not in _PARAM_CACHE:
        _PARAM_CACHE[key] = O.init_diffnet_params(cfg, WEIGHT_SEED, FINAL_PROJ_STD)
    return _PARAM_CACHE[key]


def betas_for(pre, case=None):
    if case is not None and case.get('legacy'):      # usr/diff/diffusion.py:192-195: always cosine
        return O.cosine_beta_schedule(pre['timesteps'])
    if pre['schedule_type'] == 'linear':
        return O.linear_beta_schedule(pre['timesteps'], pre['max_beta'])
    return O.cosine_beta_schedule(pre['timesteps'])


def case_setup(name):
    case = CASES[name]
    pre = presets()[case['preset']]
    cfg = net_config(pre)
    k_step = case.get('k_step', pre['K_step'])
    kind = case['kind']
    inp = make_inputs(case['seed'], case['B'], case['T'], n_noise=(k_step if kind == 'ddpm' else 0),
                      with_fs2_mel=(kind == 'ddpm' and not case['gaussian']),
                      spec_min=pre['spec_min'], spec_max=pre['spec_max'])
    smin = torch.tensor(pre['spec_min'], dtype=torch.float32)[None, None, :pre['keep_bins']]
    smax = torch.tensor(pre['spec_max'], dtype=torch.float32)[None, None, :pre['keep_bins']]
    return case, pre, cfg, k_step, inp, smin, smax


def run_oracle_case(name):
    case, pre, cfg, k_step, inp, smin, smax = case_setup(name)
    p = oracle_params(cfg)
    sch = O.make_schedule(betas_for(pre, case))
    kind = case['kind']
    with torch.no_grad():
        if kind == 'denoise':
            return O.diffnet_forward(p, cfg, inp['x_T'], torch.tensor(case['t']), inp['cond']).numpy()
        if kind == 'ddpm':
            if case['gaussian']:
                return O.infer_mel(p, cfg, sch, inp['cond'], smin, smax, k_step=k_step, noises=list(inp['noise']),
                                   x_T=inp['x_T']).numpy()
            return O.infer_mel(p, cfg, sch, inp['cond'], smin, smax, k_step=k_step, noises=list(inp['noise']),
                               fs2_mel=inp['fs2_mel'], q_noise=inp['q_noise']).numpy()
        if kind == 'plms':
            return O.infer_mel(p, cfg, sch, inp['cond'], smin, smax, k_step=k_step, x_T=inp['x_T'],
                               pndm_interval=case['interval']).numpy()
    raise ValueError(kind)
