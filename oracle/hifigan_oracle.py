"""CPU oracle of the HiFi-GAN / NSF-HiFi-GAN generator (SURVEY.md section 8 row f2 - the step AFTER the diffusion hot path:
mel [B,80,T] (+ f0 [B,T]) -> waveform [B,1,T*hop]).  TEST INFRASTRUCTURE ONLY.  Groundwork for the next round: there is no HIP
vocoder yet; this restatement is pinned now so that the kernels can be checked against it from their first line.

Functional torch-CPU fp32 restatement of (paths relative to /root/reference):
  HifiGanGenerator.forward, ResBlock1 / ResBlock2          modules/hifigan/hifigan.py:30-92, 104-169
  SourceModuleHnNSF.forward, SineGen.forward / _f02sine    modules/parallel_wavegan/models/source.py:7-137, 484-531
on a plain state_dict with the reference's parameter names, either before `remove_weight_norm()` (weight_g / weight_v pairs,
what a checkpoint holds: vocoders/hifigan.py:17-32) or after it (plain weights).  The three random draws of the source
module (initial phases `torch.rand`, additive sine noise and the unused noise branch `torch.randn_like`) are taken from
torch's global generator in the reference's order, so `torch.manual_seed(s)` before either implementation gives the same
excitation.  Pinned bit-for-bit against the live reference (tests/test_hifigan_oracle.py) and a reference-generated fixture."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1


def _weight(p, name, transposed=False):
    """Plain weight, or g * v / ||v|| for a weight-normed layer (torch.nn.utils.weight_norm, dim=0)."""
    if name + '.weight' in p:
        return p[name + '.weight']
    g, v = p[name + '.weight_g'], p[name + '.weight_v']
    return torch._weight_norm(v, g, 0)                              # the ATen op behind torch.nn.utils.weight_norm: v * (g / ||v||_dim0)


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


def resblock1(p, pre, x, k, dils):
    for i, d in enumerate(dils[:3]):                               # hifigan.py:54-61; the module builds exactly three conv pairs (:34-50)
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _weight(p, f'{pre}convs1.{i}'), p[f'{pre}convs1.{i}.bias'], padding=get_padding(k, d), dilation=d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, _weight(p, f'{pre}convs2.{i}'), p[f'{pre}convs2.{i}.bias'], padding=get_padding(k, 1))
        x = xt + x
    return x


def resblock2(p, pre, x, k, dils):
    for i, d in enumerate(dils[:2]):                               # hifigan.py:82-87; the module builds exactly TWO convs, dilation[0] and [1] (:71-79)
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _weight(p, f'{pre}convs.{i}'), p[f'{pre}convs.{i}.bias'], padding=get_padding(k, d), dilation=d)
        x = xt + x
    return x


def sine_gen(f0, samp_rate, harmonic_num=8, sine_amp=0.1, noise_std=0.003, voiced_threshold=0):
    """SineGen.forward (source.py:101-137), flag_for_pulse=False.  f0 [B,L,1] -> (sine_waves [B,L,H+1], uv [B,L,1])."""
    B, L, _ = f0.shape
    dim = harmonic_num + 1
    f0_buf = torch.zeros(B, L, dim)
    f0_buf[:, :, 0] = f0[:, :, 0]
    for idx in np.arange(harmonic_num):
        f0_buf[:, :, idx + 1] = f0_buf[:, :, 0] * (idx + 2)
    rad = (f0_buf / samp_rate) % 1                                 # _f02sine :45-77
    rand_ini = torch.rand(B, dim)
    rand_ini[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + rand_ini
    over = torch.cumsum(rad, 1) % 1
    over_idx = (over[:, 1:, :] - over[:, :-1, :]) < 0
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over_idx * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi) * sine_amp
    uv = torch.ones_like(f0) * (f0 > voiced_threshold)
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3
    noise = noise_amp * torch.randn_like(sines)
    return sines * uv + noise, uv


def source_module(p, f0_up, samp_rate, harmonic_num=8, sine_amp=0.1):
    """SourceModuleHnNSF.forward (source.py:518-531): merged harmonic source [B,L,1] (the noise branch is drawn and unused)."""
    sine_wavs, uv = sine_gen(f0_up, samp_rate, harmonic_num, sine_amp)
    merged = torch.tanh(F.linear(sine_wavs, p['m_source.l_linear.weight'], p['m_source.l_linear.bias']))
    torch.randn_like(uv)                                            # `noise = torch.randn_like(uv) * sine_amp / 3` consumes the generator
    return merged


def generator(p, h, x, f0=None):
    """HifiGanGenerator.forward (hifigan.py:144-169).  x [B,80,T]; f0 [B,T] or None.  h: the config dict of the vocoder."""
    rates, ksz = h['upsample_rates'], h['upsample_kernel_sizes']
    nk = len(h['resblock_kernel_sizes'])
    har = None
    if f0 is not None:
        up = int(np.prod(rates))
        f0u = F.interpolate(f0[:, None], scale_factor=float(up), mode='nearest').transpose(1, 2)      # torch.nn.Upsample(scale_factor=prod)
        har = source_module(p, f0u, h['audio_sample_rate']).transpose(1, 2)
    x = F.conv1d(x, _weight(p, 'conv_pre'), p['conv_pre.bias'], padding=3)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, _weight(p, f'ups.{i}'), p[f'ups.{i}.bias'], stride=u, padding=(k - u) // 2)
        if har is not None:
            if i + 1 < len(rates):
                s = int(np.prod(rates[i + 1:]))
                xs = F.conv1d(har, p[f'noise_convs.{i}.weight'], p[f'noise_convs.{i}.bias'], stride=s, padding=s // 2)
            else:
                xs = F.conv1d(har, p[f'noise_convs.{i}.weight'], p[f'noise_convs.{i}.bias'])
            x = x + xs
        acc = None
        for j in range(nk):
            pre = f'resblocks.{i * nk + j}.'
            blk = resblock1 if h['resblock'] == '1' else resblock2
            y = blk(p, pre, x, h['resblock_kernel_sizes'][j], h['resblock_dilation_sizes'][j])
            acc = y if acc is None else acc + y
        x = acc / nk
    x = F.leaky_relu(x)                                             # default slope 0.01 here (hifigan.py:166)
    x = F.conv1d(x, _weight(p, 'conv_post'), p['conv_post.bias'], padding=3)
    return torch.tanh(x)


def generator_shapes(h, c_out=1):
    """State-dict names and shapes of HifiGanGenerator after remove_weight_norm() (hifigan.py:105-142)."""
    s = {}
    c0 = h['upsample_initial_channel']
    s['conv_pre.weight'], s['conv_pre.bias'] = (c0, 80, 7), (c0,)
    rates, ksz = h['upsample_rates'], h['upsample_kernel_sizes']
    nk = len(h['resblock_kernel_sizes'])
    if h['use_pitch_embed']:
        s['m_source.l_linear.weight'], s['m_source.l_linear.bias'] = (1, 9), (1,)
    ch = c0
    for i, (u, k) in enumerate(zip(rates, ksz)):
        ch = c0 // (2 ** (i + 1))
        s[f'ups.{i}.weight'], s[f'ups.{i}.bias'] = (ch * 2, ch, k), (ch,)
        if h['use_pitch_embed']:
            st = int(np.prod(rates[i + 1:])) if i + 1 < len(rates) else None
            s[f'noise_convs.{i}.weight'], s[f'noise_convs.{i}.bias'] = ((ch, 1, st * 2) if st else (ch, 1, 1)), (ch,)
        for j, (kk, dd) in enumerate(zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes'])):
            pre = f'resblocks.{i * nk + j}.'
            names = ('convs1', 'convs2') if h['resblock'] == '1' else ('convs',)
            for nm in names:
                for q in range(min(len(dd), 3 if h['resblock'] == '1' else 2)):
                    s[f'{pre}{nm}.{q}.weight'], s[f'{pre}{nm}.{q}.bias'] = (ch, ch, kk), (ch,)
    s['conv_post.weight'], s['conv_post.bias'] = (c_out, ch, 7), (c_out,)
    return s


def synth_generator_params(h, seed):
    """Seeded synthetic weights (there are no checkpoints): fan-in scaled so that the signal neither dies nor explodes."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for k, shp in sorted(generator_shapes(h).items()):
        if k.endswith('bias'):
            p[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan = 1
            for d in (shp[1:] if not k.startswith('ups.') else (shp[0], shp[2] // 2)):
                fan *= d
            p[k] = torch.randn(shp, generator=g) * (1.2 / max(fan, 1)) ** 0.5
    return p
