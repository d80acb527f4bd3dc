"""ParallelWaveGAN vocoder on the HIP operators of include/dsv.h - the reference's other vocoder (vocoders/pwg.py:17-117; the default of
configs/tts/base.yaml:88, every DiffSpeech / DiffSinger YAML overrides it to HiFi-GAN).

`ParallelWaveGANGenerator` mirrors modules/parallel_wavegan/models/parallel_wavegan.py:21-177 (same constructor arguments, same state_dict
keys with or without weight norm): first_conv -> 30 gated residual blocks with dilations 1 ... 512 conditioned on the mel upsampled by
ConvInUpsampleNetwork (layers/upsample.py:130-183) -> ReLU / 1x1 / ReLU / 1x1.  Every block is one launch of `k_pwg_layer`
(csrc/pwg_kernels.hpp); the upsampling stages, the first convolution and the pointwise convolutions are HIP kernels too; what torch does here is
index plumbing (embedding lookup, padding, cropping).  This build covers the configuration the reference ships and trains: kernel_size 3,
residual / gate / skip channels 64 / 128 / 64, non-causal, `ConvInUpsampleNetwork`; anything else raises.

`PWG` mirrors vocoders/pwg.py:54-117 (`spec2wav`)."""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .vocoder import _HipOps, padded_samples, register_vocoder


class _WN(nn.Module):
    """weight_norm(Conv1d / Conv2d) parameter holder: weight_g / weight_v (/ bias) as a checkpoint stores them, or weight after
    remove_weight_norm() (torch.nn.utils.weight_norm, dim 0)."""

    def __init__(self, wshape, bias: bool, weight_norm: bool):
        super().__init__()
        if weight_norm:
            self.weight_g = nn.Parameter(torch.ones(wshape[0], *([1] * (len(wshape) - 1))))
            self.weight_v = nn.Parameter(torch.randn(wshape) * 0.01)
        else:
            self.weight = nn.Parameter(torch.randn(wshape) * 0.01)
        if bias:
            self.bias = nn.Parameter(torch.zeros(wshape[0]))
        else:
            self.bias = None

    def remove_weight_norm(self):
        if hasattr(self, 'weight_g'):
            w = torch._weight_norm(self.weight_v.detach(), self.weight_g.detach(), 0)
            del self.weight_g
            del self.weight_v
            self.weight = nn.Parameter(w)

    def plain_weight(self) -> torch.Tensor:
        if hasattr(self, 'weight_g'):
            return torch._weight_norm(self.weight_v.detach(), self.weight_g.detach(), 0)
        return self.weight.detach()

    def tag(self):
        ps = [self.weight_g, self.weight_v] if hasattr(self, 'weight_g') else [self.weight]
        if self.bias is not None:
            ps.append(self.bias)
        return tuple((p.data_ptr(), p._version, p.device) for p in ps)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if prefix + 'weight' in state_dict and hasattr(self, 'weight_g'):     # a state saved after remove_weight_norm()
            self.remove_weight_norm()
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class _Block(nn.Module):
    """ResidualBlock (layers/residual_block.py:39-94): parameters only."""

    def __init__(self, k, res, gate, skip, aux, dilation, bias, wn):
        super().__init__()
        self.dilation = dilation
        self.conv = _WN((gate, res, k), bias, wn)
        self.conv1x1_aux = _WN((gate, aux, 1), False, wn) if aux > 0 else None
        self.conv1x1_out = _WN((res, gate // 2, 1), bias, wn)
        self.conv1x1_skip = _WN((skip, gate // 2, 1), bias, wn)


class _Upsample(nn.Module):
    """UpsampleNetwork (layers/upsample.py:63-127): up_layers[2 i] = Stretch2d (no parameters), up_layers[2 i + 1] = Conv2d(1, 1, (1, 2 s + 1))."""

    def __init__(self, scales, wn):
        super().__init__()
        layers = []
        for s in scales:
            layers += [nn.Identity(), _WN((1, 1, 1, 2 * s + 1), False, wn)]
        self.up_layers = nn.ModuleList(layers)


class _ConvInUpsample(nn.Module):
    def __init__(self, scales, aux, ctx, wn):
        super().__init__()
        self.aux_context_window = ctx
        self.conv_in = _WN((aux, aux, 2 * ctx + 1), False, wn)
        self.upsample = _Upsample(scales, wn)


class ParallelWaveGANGenerator(nn.Module):
    """modules/parallel_wavegan/models/parallel_wavegan.py:21-177.  forward(x [B,1,T] noise, c [B,aux,T'] mel (T' = T / hop + 2 * context
    window), pitch [B,T'] or None) -> [B,1,T]."""

    def __init__(self, in_channels=1, out_channels=1, kernel_size=3, layers=30, stacks=3, residual_channels=64, gate_channels=128,
                 skip_channels=64, aux_channels=80, aux_context_window=2, dropout=0.0, bias=True, use_weight_norm=True,
                 use_causal_conv=False, upsample_conditional_features=True, upsample_net='ConvInUpsampleNetwork',
                 upsample_params=None, use_pitch_embed=False):
        super().__init__()
        upsample_params = dict(upsample_params or {'upsample_scales': [4, 4, 4, 4]})
        bad = []
        if (in_channels, out_channels, kernel_size) != (1, 1, 3):
            bad.append('in_channels / out_channels / kernel_size other than 1 / 1 / 3')
        if (residual_channels, gate_channels, skip_channels) != (64, 128, 64):
            bad.append('residual / gate / skip channels other than 64 / 128 / 64')
        if aux_channels % 8 or not (8 <= aux_channels <= 128):
            bad.append('aux_channels not a multiple of 8 in [8, 128]')
        if use_causal_conv or not upsample_conditional_features or upsample_net != 'ConvInUpsampleNetwork':
            bad.append('causal convolutions / no upsampling network / an upsampling network other than ConvInUpsampleNetwork')
        extra = {k: v for k, v in upsample_params.items() if k not in ('upsample_scales', 'use_causal_conv', 'aux_channels', 'aux_context_window')
                 and not (k == 'nonlinear_activation' and v is None) and not (k == 'nonlinear_activation_params') and not (k == 'interpolate_mode' and v == 'nearest')
                 and not (k == 'freq_axis_kernel_size' and v == 1)}
        if extra:
            bad.append(f'upsample_params {sorted(extra)}')
        if bad:
            raise NotImplementedError('ParallelWaveGANGenerator on HIP covers the shipped configuration only: ' + '; '.join(bad))
        assert layers % stacks == 0
        self.in_channels, self.out_channels, self.aux_channels = in_channels, out_channels, aux_channels
        self.layers, self.stacks, self.kernel_size = layers, stacks, kernel_size
        self.upsample_scales = [int(s) for s in upsample_params['upsample_scales']]
        wn = bool(use_weight_norm)
        self.first_conv = _WN((residual_channels, in_channels, 1), True, wn)
        self.upsample_net = _ConvInUpsample(self.upsample_scales, aux_channels, aux_context_window, wn)
        per = layers // stacks
        self.conv_layers = nn.ModuleList([_Block(kernel_size, residual_channels, gate_channels, skip_channels, aux_channels, 2 ** (i % per), bias, wn)
                                          for i in range(layers)])
        self.last_conv_layers = nn.ModuleList([nn.ReLU(), _WN((skip_channels, skip_channels, 1), True, wn), nn.ReLU(),
                                               _WN((out_channels, skip_channels, 1), True, wn)])
        self.use_pitch_embed = use_pitch_embed
        if use_pitch_embed:
            self.pitch_embed = nn.Embedding(300, aux_channels, 0)
            self.c_proj = nn.Linear(2 * aux_channels, aux_channels)          # not weight-normed (apply_weight_norm touches Conv1d / Conv2d only)
        self._ops: Optional[_HipOps] = None
        self._cache = {}

    def remove_weight_norm(self):
        for m in self.modules():
            if isinstance(m, _WN):
                m.remove_weight_norm()

    # -- packed weights ------------------------------------------------------------------------------------------------
    def _packed(self, key, mods, build):
        """build() -> (matrix [rows][Ci][1] to pack, bias or None); cached until one of the source parameters changes."""
        tag = tuple(m.tag() if isinstance(m, _WN) else tuple((p.data_ptr(), p._version) for p in m.parameters()) for m in mods)
        hit = self._cache.get(key)
        if hit is None or hit[0] != tag:
            w, b = build()
            hit = self._cache[key] = (tag, self._ops.pack(w.contiguous()), None if b is None else b.contiguous())
        return hit[1], hit[2]

    def _layer_weights(self, i: int):
        blk = self.conv_layers[i]

        def first():
            w = blk.conv.plain_weight()                                      # [128][64][3] -> columns tap * 64 + ci
            cols = [w.permute(0, 2, 1).reshape(w.shape[0], -1)]
            if blk.conv1x1_aux is not None:
                cols.append(blk.conv1x1_aux.plain_weight()[:, :, 0])
            return torch.cat(cols, 1)[:, :, None], (blk.conv.bias.detach() if blk.conv.bias is not None else None)

        def second():
            w = torch.cat([blk.conv1x1_out.plain_weight()[:, :, 0], blk.conv1x1_skip.plain_weight()[:, :, 0]], 0)
            b = None
            if blk.conv1x1_out.bias is not None:
                b = torch.cat([blk.conv1x1_out.bias.detach(), blk.conv1x1_skip.bias.detach()])
            return w[:, :, None], b
        mods1 = [blk.conv] + ([blk.conv1x1_aux] if blk.conv1x1_aux is not None else [])
        return self._packed(('l1', i), mods1, first) + self._packed(('l2', i), [blk.conv1x1_out, blk.conv1x1_skip], second)

    # -- forward ---------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, c=None, pitch=None, **kwargs):
        """parallel_wavegan.py:139-177 (eval mode: dropout 0)."""
        if x.device.type != 'cuda':
            raise RuntimeError('ParallelWaveGANGenerator: the HIP path needs device tensors (there is no CPU path in this package)')
        if c is None:
            raise NotImplementedError('ParallelWaveGANGenerator without local conditioning')
        if self._ops is None:
            self._ops = _HipOps()
        ops, lib, dev = self._ops, self._ops.lib, x.device
        s = ops._s
        B, _, T = x.shape
        aux, ctx = self.aux_channels, self.upsample_net.aux_context_window
        c = c.to(torch.float32)
        if self.use_pitch_embed:                                             # :153-155
            p = self.pitch_embed(pitch)                                      # [B,T',aux]
            cat = torch.cat([c.transpose(1, 2), p], -1).transpose(1, 2).contiguous()        # [B,2 aux,T']
            wp, b = self._packed('c_proj', [self.c_proj], lambda: (self.c_proj.weight.detach()[:, :, None], self.c_proj.bias.detach()))
            Tc = cat.shape[2]
            c = ops.conv(ops.pad_rows(cat), Tc, wp, b, aux, 2 * aux, 1, 0, 1)[:, :, :Tc]
        Tc = c.shape[2]
        # ConvInUpsampleNetwork (upsample.py:166-183): conv_in without padding (T' -> T' - 2 ctx), then the stretch + smoothing stages
        wp, _ = self._packed('conv_in', [self.upsample_net.conv_in], lambda: (self.upsample_net.conv_in.plain_weight(), None))
        k_in = 2 * ctx + 1
        y = ops.conv(ops.pad_rows(c.contiguous()), Tc, wp, None, aux, aux, k_in, 0, 1)
        L = Tc - 2 * ctx
        y = ops.pad_rows(y[:, :, :L].contiguous())
        for i, sc in enumerate(self.upsample_scales):
            filt = self.upsample_net.upsample.up_layers[2 * i + 1].plain_weight().reshape(-1).contiguous()
            out = torch.empty(B, aux, padded_samples(L * sc), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.check(lib.dsv_pwg_upsample(y.data_ptr(), filt.data_ptr(), out.data_ptr(), B * aux, L, sc, s(dev)), 'dsv_pwg_upsample')
            y, L = out, L * sc
        assert L == T, (L, T)                                                # :157
        LS = padded_samples(T)
        # first_conv (:160)
        z = ops.pad_rows(x.to(torch.float32).contiguous())                   # [B,1,LS]
        w0, b0 = self.first_conv.plain_weight().reshape(-1).contiguous(), self.first_conv.bias.detach().contiguous()
        h = torch.empty(B, 64, LS, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.dsv_pwg_first(z.data_ptr(), w0.data_ptr(), b0.data_ptr(), h.data_ptr(), B, 64, T, s(dev)), 'dsv_pwg_first')
        h2 = torch.empty_like(h)
        skips = torch.empty_like(h)
        for i, blk in enumerate(self.conv_layers):                           # :161-166
            w1, b1, w2, b2 = self._layer_weights(i)
            with torch.cuda.device(dev):
                _lib.check(lib.dsv_pwg_layer(h.data_ptr(), y.data_ptr(), w1.data_ptr(), ops._p(b1), w2.data_ptr(), ops._p(b2), h2.data_ptr(),
                                             skips.data_ptr(), B, T, aux, blk.dilation, 1 if i == 0 else 0, s(dev)), 'dsv_pwg_layer')
            h, h2 = h2, h
        skips *= math.sqrt(1.0 / len(self.conv_layers))                      # :167
        # last_conv_layers (:170-172): ReLU -> 1x1 -> ReLU -> 1x1; the ReLU in front of a convolution is its fused pre-activation (slope 0)
        l1, l3 = self.last_conv_layers[1], self.last_conv_layers[3]
        wp, b = self._packed('last1', [l1], lambda: (l1.plain_weight(), l1.bias.detach()))
        o = ops.conv(skips, T, wp, b, 64, 64, 1, 0, 1, pre_slope=0.0)
        wp, b = self._packed('last3', [l3], lambda: (l3.plain_weight(), l3.bias.detach()))
        o = ops.conv(o, T, wp, b, 1, 64, 1, 0, 1, pre_slope=0.0)
        return o[:, :, :T]


def load_pwg_model(config_path, checkpoint_path, stats_path=None, device=None):
    """vocoders/pwg.py:17-51: (model, scaler, config, device).  Official checkpoints (['model']['generator'] + normalisation statistics) and
    the reference's own training checkpoints (['state_dict'] with the generator under 'model_gen.')."""
    import yaml
    with open(config_path) as f:
        config = yaml.safe_load(f)
    if device is None:
        device = torch.device('cuda')
    model = ParallelWaveGANGenerator(**config['generator_params'])
    from .ckpt import _torch_load
    ckpt = _torch_load(checkpoint_path, trusted=True)
    scaler = None
    if 'state_dict' not in ckpt:
        model.load_state_dict(ckpt['model']['generator'])
        if config.get('format') == 'npy':
            st = np.load(stats_path)
            scaler = (st[0].astype(np.float64), st[1].astype(np.float64))
        elif config.get('format') == 'hdf5':
            try:
                import h5py
            except ImportError as e:
                raise NotImplementedError('hdf5 normalisation statistics need h5py (not in this image): convert stats.h5 to the npy format') from e
            with h5py.File(stats_path, 'r') as f:
                scaler = (f['mean'][()].astype(np.float64), f['scale'][()].astype(np.float64))
        else:
            raise ValueError('support only hdf5 or npy format.')
    else:
        sd = {k[len('model_gen.'):]: v for k, v in ckpt['state_dict'].items() if k.startswith('model_gen.')}
        model.load_state_dict(sd, strict=False)                               # strict=False like the reference's fake_task load
    model.remove_weight_norm()
    model = model.eval().to(device)
    print(f'| Loaded model parameters from {checkpoint_path}.')
    print(f'| PWG device: {device}.')
    return model, scaler, config, device


@register_vocoder
class PWG:
    """vocoders/pwg.py:54-117.  PWG() discovers the checkpoint like the reference (hparams['vocoder_ckpt'], '' = ./wavegan_pretrained);
    PWG(model, config, scaler, device) wraps a loaded generator.  spec2wav(mel [T,80], f0=[T]) -> wav [T * hop] (numpy)."""

    def __init__(self, model: Optional[ParallelWaveGANGenerator] = None, config: Optional[dict] = None, scaler=None, device='cuda'):
        if model is None:
            import glob
            import re
            from .hparams import hparams
            if hparams['vocoder_ckpt'] == '':
                base_dir = 'wavegan_pretrained'
                ckpt = sorted(glob.glob(f'{base_dir}/checkpoint-*steps.pkl'), key=lambda x: int(re.findall(rf'{base_dir}/checkpoint-(\d+)steps.pkl', x)[0]))[-1]
            else:
                base_dir = hparams['vocoder_ckpt']
                ckpt = sorted(glob.glob(f'{base_dir}/model_ckpt_steps_*.ckpt'),
                              key=lambda x: int(re.findall(rf'{re.escape(base_dir)}/model_ckpt_steps_(\d+).ckpt', x)[0]))[-1]
            print('| load PWG: ', ckpt)
            model, scaler, config, device = load_pwg_model(f'{base_dir}/config.yaml', ckpt, f'{base_dir}/stats.h5')
            if hparams['vocoder_ckpt'] != '':
                scaler = None
        self.model, self.config, self.scaler, self.device = model.eval().to(device), config, scaler, torch.device(device)

    def spec2wav(self, mel, **kwargs):
        from .fs2 import f0_to_coarse
        config = self.config
        ctx = config['generator_params'].get('aux_context_window', 2)
        c = np.asarray(mel)
        if self.scaler is not None:
            c = (c - self.scaler[0]) / self.scaler[1]                        # StandardScaler.transform
        with torch.no_grad():
            z = kwargs.get('z')
            if z is None:
                z = torch.randn(1, 1, c.shape[0] * config['hop_size'])       # drawn on the host like the reference (:96)
            c = np.pad(c, ((ctx, ctx), (0, 0)), 'edge')
            c = torch.as_tensor(c, dtype=torch.float32).unsqueeze(0).transpose(2, 1).to(self.device)
            p = kwargs.get('f0')
            if p is not None:
                p = f0_to_coarse(torch.as_tensor(np.asarray(p), dtype=torch.float32)).numpy()
                p = np.pad(p, (ctx, ctx), 'edge')
                p = torch.as_tensor(p, dtype=torch.long)[None, :].to(self.device)
            y = self.model(z.to(self.device), c, p).view(-1)
        return y.cpu().numpy()
