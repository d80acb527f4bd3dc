"""CPU oracle of the `FFT` candidate denoiser (usr/diff/candidate_decoder.py:35-96, `diff_decoder_type: 'fft'`, SURVEY.md section
8 row f4) and of one DDPM step driven by it.  TEST INFRASTRUCTURE ONLY.

Functional torch-CPU fp32 restatement on a plain state_dict with the reference's parameter names; the transformer stack is
the FFTBlocks restatement of oracle/fs2_oracle.py.  Pinned by fixtures generated from the reference class
(oracle/make_golden_fft.py -> tests/golden/fft_*.npz) and against the live reference in the build container."""
import math

import torch
import torch.nn.functional as F

from oracle import fs2_oracle as FO


def mish(x):
    return x * torch.tanh(F.softplus(x))                       # usr/diff/diffusion.py:68-70


def step_embedding(t, dim):
    half = dim // 2                                            # candidate_decoder.py:19-26 (arange is int64 there: * -emb -> float)
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = t[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def fft_forward(p, hp, spec, diffusion_step, cond):
    """FFT.forward (candidate_decoder.py:50-96).  spec [B,1,M,T], diffusion_step [B] (long), cond [B,H,T] -> [B,1,M,T]."""
    dim = hp['residual_channels']
    x = F.conv1d(spec[:, 0], p['input_projection.weight'], p['input_projection.bias']).permute(0, 2, 1)
    d = step_embedding(diffusion_step, dim)
    d = F.linear(mish(F.linear(d, p['mlp.0.weight'], p['mlp.0.bias'])), p['mlp.2.weight'], p['mlp.2.bias'])
    c = cond.permute(0, 2, 1)
    te = d[:, None, :].repeat(1, c.shape[1], 1)
    x = F.linear(torch.cat([x, c, te], dim=-1), p['get_decode_inp.weight'], p['get_decode_inp.bias'])
    x = FO.fft_blocks(p, '', x, hp['dec_layers'], hp['num_heads'], hp['dec_ffn_kernel_size'], hp['ffn_act'])
    x = F.linear(x, p['get_mel_out.weight'], p['get_mel_out.bias']).permute(0, 2, 1)
    return x[:, None, :, :]
