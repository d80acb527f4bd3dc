"""Seeded synthetic inputs for the denoiser hot path (SURVEY.md section 8d "common synthetic inputs").

There are no datasets or checkpoints (no network; the reference ships none), so every benchmark and
parity case runs on: cond ~ N(0,1) produced as [B,T,H] and viewed transposed (what
GaussianDiffusion.forward hands the denoiser, usr/diff/shallow_diffusion_tts.py:238), x_T ~ N(0,1)
[B,1,M,T], explicit per-step noise [K,B,1,M,T], all from a CPU torch.Generator so that the GPU path,
the oracle and the golden generator see bit-identical inputs.
"""
from __future__ import annotations

import json
import os
from typing import Dict

import torch

_PRESETS = None


def presets() -> Dict[str, dict]:
    """Hot-path hyper-parameters + spec statistics of the reference's shipped configs (data extracted
    from the reference YAML trees by oracle/make_presets.py)."""
    global _PRESETS
    if _PRESETS is None:
        with open(os.path.join(os.path.dirname(__file__), 'presets.json')) as f:
            _PRESETS = json.load(f)
    return _PRESETS


def make_inputs(seed: int, B: int, T: int, *, mel_bins: int = 80, hidden: int = 256, n_noise: int = 0,
                with_fs2_mel: bool = False, spec_min=None, spec_max=None) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device='cpu').manual_seed(seed)
    out = {
        'cond': torch.randn(B, T, hidden, generator=g).transpose(1, 2),      # [B,H,T] view of [B,T,H]
        'x_T': torch.randn(B, 1, mel_bins, T, generator=g),
    }
    if n_noise:
        out['noise'] = torch.randn(n_noise, B, 1, mel_bins, T, generator=g)
    if with_fs2_mel:
        # a plausible aux-decoder mel: normalised value in [-1,1], de-normalised with the dataset stats
        z = torch.clamp(torch.randn(B, T, mel_bins, generator=g) * 0.5, -1., 1.)
        smin = torch.tensor(spec_min, dtype=torch.float32)[None, None, :]
        smax = torch.tensor(spec_max, dtype=torch.float32)[None, None, :]
        out['fs2_mel'] = (z + 1) / 2 * (smax - smin) + smin
        out['q_noise'] = torch.randn(B, 1, mel_bins, T, generator=g)
    return out
