"""CPU restatement (torch fp32) of the reference's ParallelWaveGAN generator forward - TEST INFRASTRUCTURE ONLY (tests/, never the product path).

Follows modules/parallel_wavegan/models/parallel_wavegan.py:139-177 (forward), layers/residual_block.py:96-129 (ResidualBlock.forward),
layers/upsample.py:96-117, :166-183 (UpsampleNetwork / ConvInUpsampleNetwork.forward).  Pinned against the live reference module by
oracle/make_golden_pwg.py (bit-equal on the committed fixtures, asserted there) and against the fixtures by tests/test_pwg_oracle_golden.py."""
import math
from typing import Dict

import torch
import torch.nn.functional as F


def plain_params(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """weight_g / weight_v pairs -> weight (torch.nn.utils.weight_norm, dim 0: what remove_weight_norm() leaves)."""
    out = {}
    for k, v in state.items():
        if k.endswith('weight_g'):
            base = k[:-len('weight_g')]
            out[base + 'weight'] = torch._weight_norm(state[base + 'weight_v'], v, 0)
        elif k.endswith('weight_v'):
            continue
        else:
            out[k] = v
    return out


def upsample_net(p, c, scales, ctx):
    """ConvInUpsampleNetwork.forward (upsample.py:166-183)."""
    c = F.conv1d(c, p['upsample_net.conv_in.weight'])                         # no padding, no bias: T' -> T' - 2 ctx
    c = c.unsqueeze(1)                                                        # (B, 1, C, T)
    for i, s in enumerate(scales):
        c = F.interpolate(c, scale_factor=(1, s), mode='nearest')            # Stretch2d
        c = F.conv2d(c, p[f'upsample_net.upsample.up_layers.{2 * i + 1}.weight'], padding=(0, s))
    return c.squeeze(1)


def residual_block(p, pre, x, c, dilation):
    """ResidualBlock.forward (residual_block.py:96-129), non-causal, dropout 0."""
    residual = x
    x = F.conv1d(x, p[pre + 'conv.weight'], p.get(pre + 'conv.bias'), padding=dilation, dilation=dilation)
    xa, xb = x.split(x.size(1) // 2, dim=1)
    if c is not None:
        c = F.conv1d(c, p[pre + 'conv1x1_aux.weight'])
        ca, cb = c.split(c.size(1) // 2, dim=1)
        xa, xb = xa + ca, xb + cb
    x = torch.tanh(xa) * torch.sigmoid(xb)
    s = F.conv1d(x, p[pre + 'conv1x1_skip.weight'], p.get(pre + 'conv1x1_skip.bias'))
    x = (F.conv1d(x, p[pre + 'conv1x1_out.weight'], p.get(pre + 'conv1x1_out.bias')) + residual) * math.sqrt(0.5)
    return x, s


def generator_forward(p, cfg, x, c, pitch=None):
    """ParallelWaveGANGenerator.forward (parallel_wavegan.py:139-177).  p: plain weights; cfg: layers, stacks, upsample_scales,
    aux_context_window, use_pitch_embed."""
    layers, per = cfg['layers'], cfg['layers'] // cfg['stacks']
    if cfg.get('use_pitch_embed'):
        pe = F.embedding(pitch, p['pitch_embed.weight'], 0)
        c = F.linear(torch.cat([c.transpose(1, 2), pe], -1), p['c_proj.weight'], p['c_proj.bias']).transpose(1, 2)
    c = upsample_net(p, c, cfg['upsample_scales'], cfg['aux_context_window'])
    assert c.size(-1) == x.size(-1), (c.size(-1), x.size(-1))
    x = F.conv1d(x, p['first_conv.weight'], p['first_conv.bias'])
    skips = 0
    for i in range(layers):
        x, h = residual_block(p, f'conv_layers.{i}.', x, c, 2 ** (i % per))
        skips = skips + h
    skips = skips * math.sqrt(1.0 / layers)
    x = F.relu(skips)
    x = F.conv1d(x, p['last_conv_layers.1.weight'], p['last_conv_layers.1.bias'])
    x = F.relu(x)
    x = F.conv1d(x, p['last_conv_layers.3.weight'], p['last_conv_layers.3.bias'])
    return x, c
