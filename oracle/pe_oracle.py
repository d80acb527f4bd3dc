"""CPU oracle of the PitchExtractor (mel -> f0 for the NSF vocoder).  TEST INFRASTRUCTURE ONLY.

Functional torch-CPU fp32 restatement of the EVAL-mode forward of (paths relative to /root/reference):
  PitchExtractor.forward                     modules/fastspeech/pe.py:119-148
  Prenet (Conv1d k5 -> ReLU -> BatchNorm1d, x3, out_proj)    pe.py:8-41
  ConvStacks / ConvBlock (ConvNorm k5 -> GroupNorm(C/16) -> ReLU, residual)   pe.py:44-116
  PitchPredictor                             modules/fastspeech/tts_modules.py:192-235 (oracle/fs2_oracle.pitch_predictor)
  denorm_f0                                  utils/pitch_utils.py:63-76
on a plain state_dict with the reference's parameter names.  Pinned bit-for-bit against the live reference module
(tests/test_pe_oracle.py, build container) and against a reference-generated fixture (oracle/make_golden_pe.py ->
tests/golden/pe_opencpop.npz)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import fs2_oracle as FO


def prenet(p, pre, x_btc, n_layers=3, kernel=5):
    """Prenet.forward (pe.py:23-41), strides 1.  x [B,T,80] -> [B,T,H] (the second return value of the module)."""
    padding_mask = x_btc.abs().sum(-1).eq(0)
    keep = 1 - padding_mask.float()[:, None, :]
    x = x_btc.transpose(1, 2)
    for l in range(n_layers):
        x = F.conv1d(x, p[f'{pre}layers.{l}.0.weight'], p[f'{pre}layers.{l}.0.bias'], padding=kernel // 2)
        x = F.relu(x)
        x = F.batch_norm(x, p[f'{pre}layers.{l}.2.running_mean'], p[f'{pre}layers.{l}.2.running_var'], p[f'{pre}layers.{l}.2.weight'],
                         p[f'{pre}layers.{l}.2.bias'], False, 0.1, 1e-5)
        x = x * keep
    x = F.linear(x.transpose(1, 2), p[pre + 'out_proj.weight'], p[pre + 'out_proj.bias'])
    return x * keep.transpose(1, 2)


def conv_stacks(p, pre, x_btc, n_layers=2, kernel=5):
    """ConvStacks.forward (pe.py:98-116), norm 'gn', res=True, strides 1."""
    x = F.linear(x_btc, p[pre + 'in_proj.weight'], p[pre + 'in_proj.bias']).transpose(1, -1)
    for i in range(n_layers):
        w = p[f'{pre}conv.{i}.conv.conv.weight']
        x_ = F.conv1d(x, w, p[f'{pre}conv.{i}.conv.conv.bias'], padding=(kernel - 1) // 2)
        x_ = F.group_norm(x_, w.shape[0] // 16, p[f'{pre}conv.{i}.norm.weight'], p[f'{pre}conv.{i}.norm.bias'], 1e-5)
        x = x + F.relu(x_)
    return F.linear(x.transpose(1, -1), p[pre + 'out_proj.weight'], p[pre + 'out_proj.bias'])


def pitch_extractor(p, hp, mel_bt80, conv_layers=2):
    """PitchExtractor.forward.  Returns {'pitch_pred' [B,T,2], 'f0_denorm_pred' [B,T]}."""
    h = prenet(p, 'mel_prenet.', mel_bt80)
    if conv_layers > 0:
        h = conv_stacks(p, 'mel_encoder.', h, conv_layers)
    pitch_pred = FO.pitch_predictor(p, 'pitch_predictor.', h, 5, hp['predictor_kernel'])
    pitch_padding = mel_bt80.abs().sum(-1) == 0
    use_uv = hp['pitch_type'] == 'frame' and hp['use_uv']
    f0 = FO.denorm_f0(pitch_pred[:, :, 0], (pitch_pred[:, :, 1] > 0) if use_uv else None, hp, pitch_padding=pitch_padding)
    return {'pitch_pred': pitch_pred, 'f0_denorm_pred': f0}


def extractor_shapes(hp, conv_layers=2):
    H = hp['hidden_size']
    ph = hp['predictor_hidden'] if hp['predictor_hidden'] > 0 else H
    k = hp['predictor_kernel']
    s = {}
    cin = 80
    for l in range(3):
        s[f'mel_prenet.layers.{l}.0.weight'], s[f'mel_prenet.layers.{l}.0.bias'] = (H, cin, 5), (H,)
        for n in ('weight', 'bias', 'running_mean', 'running_var'):
            s[f'mel_prenet.layers.{l}.2.{n}'] = (H,)
        s[f'mel_prenet.layers.{l}.2.num_batches_tracked'] = ()
        cin = H
    s['mel_prenet.out_proj.weight'], s['mel_prenet.out_proj.bias'] = (H, H), (H,)
    if conv_layers > 0:
        s['mel_encoder.in_proj.weight'], s['mel_encoder.in_proj.bias'] = (H, H), (H,)
        for i in range(conv_layers):
            s[f'mel_encoder.conv.{i}.conv.conv.weight'], s[f'mel_encoder.conv.{i}.conv.conv.bias'] = (H, H, 5), (H,)
            s[f'mel_encoder.conv.{i}.norm.weight'], s[f'mel_encoder.conv.{i}.norm.bias'] = (H,), (H,)
        s['mel_encoder.out_proj.weight'], s['mel_encoder.out_proj.bias'] = (H, H), (H,)
    cin = H
    for i in range(5):
        s[f'pitch_predictor.conv.{i}.1.weight'], s[f'pitch_predictor.conv.{i}.1.bias'] = (ph, cin, k), (ph,)
        s[f'pitch_predictor.conv.{i}.3.weight'], s[f'pitch_predictor.conv.{i}.3.bias'] = (ph,), (ph,)
        cin = ph
    s['pitch_predictor.linear.weight'], s['pitch_predictor.linear.bias'] = (2, ph), (2,)
    s['pitch_predictor.pos_embed_alpha'] = (1,)
    s['pitch_predictor.embed_positions._float_tensor'] = (1,)
    return s


def synth_extractor_params(hp, seed, conv_layers=2):
    """Seeded synthetic state (there are no checkpoints): fan-in scaled weights, non-trivial norm statistics."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for k, shp in sorted(extractor_shapes(hp, conv_layers).items()):
        if k.endswith('num_batches_tracked'):
            p[k] = torch.tensor(7, dtype=torch.long)
        elif k.endswith('_float_tensor'):
            p[k] = torch.zeros(shp)
        elif k.endswith('pos_embed_alpha'):
            p[k] = 1 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith('running_var'):
            p[k] = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith('running_mean'):
            p[k] = 0.2 * torch.randn(shp, generator=g)
        elif len(shp) >= 2:
            fan = 1
            for d in shp[1:]:
                fan *= d
            p[k] = torch.randn(shp, generator=g) * (1.5 / fan ** 0.5)
        elif k.endswith('weight'):
            p[k] = 1 + 0.1 * torch.randn(shp, generator=g)
        else:
            p[k] = 0.1 * torch.randn(shp, generator=g)
    p['pitch_predictor.linear.bias'] = p['pitch_predictor.linear.bias'] + torch.tensor([7.5, 0.0])      # log2 f0 around 180 Hz
    return p


def synth_mel(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    mel = torch.randn(B, T, 80, generator=g) * 1.5 - 4
    for b in range(1, B):                                            # ragged: trailing frames of utterance b are padding (all-zero mel)
        mel[b, T - 5 * b:] = 0
    return mel
