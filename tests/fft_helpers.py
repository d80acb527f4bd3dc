"""Shared plumbing of the `FFT` candidate-denoiser tests (row f4)."""
import os

import numpy as np
import torch

import diffsinger_amd
from diffsinger_amd import hparams
from oracle.fs2_cases import synth_params

PRESET = 'popcs_ds_beta6'
K = 8
B, T = 2, 72
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fft_decoder.npz')


def build_module():
    hparams.clear()
    diffsinger_amd.use_preset(PRESET)
    from diffsinger_amd.candidate_decoder import FFT
    m = FFT(hparams['hidden_size'], hparams['dec_layers'], hparams['dec_ffn_kernel_size'], hparams['num_heads']).eval()
    shapes = {k: (tuple(v.shape), v.dtype) for k, v in m.state_dict().items()}
    params = synth_params({**shapes, 'dur_predictor.linear.bias': ((1,), torch.float32)}, 4242)
    params.pop('dur_predictor.linear.bias')
    params['get_mel_out.weight'] = params['get_mel_out.weight'] * 0.3          # keep eps O(1) like a trained head
    m.load_state_dict(params, strict=True)
    return m, dict(hparams), params


def make_inputs():
    g = torch.Generator().manual_seed(77)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    cond[1, :, 60:] = 0                                         # a shorter utterance: decoder_inp is masked there
    return {'cond': cond, 'x': torch.randn(B, 1, 80, T, generator=g), 't': torch.tensor([5, 0]),
            'noise': torch.randn(K, B, 1, 80, T, generator=g)}
