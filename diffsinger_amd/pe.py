"""PitchExtractor - mel -> f0 for the NSF vocoder (the reference runs it between the diffusion
sampler and HifiGAN.spec2wav when hparams['pe_enable'], inference/svs/base_svs_infer.py:61-70, tasks/tts/fs2.py:440-445) - as
nn.Modules whose eval forward runs on the HIP operators of libdsdenoise.so (include/dsf.h).

Mirrors the reference module tree (paths relative to the reference root) name for name, so `utils.load_ckpt(pe, hparams['pe_ckpt'],
'model', strict=True)` works on it:

    PitchExtractor, Prenet, ConvStacks, ConvBlock        modules/fastspeech/pe.py:8-148
    ConvNorm                                              modules/commons/common_layers.py:41-59
    PitchPredictor                                        modules/fastspeech/tts_modules.py:192-235 (diffsinger_amd.fs2.PitchPredictor)

What runs where: the seven k = 5 convolutions (+ ReLU), the four Linear layers and the predictor stack are k_fs_conv / k_fs_ln
launches; BatchNorm1d (eval) + the padding mask is k_fs_affine, GroupNorm + ReLU + residual is k_fs_group_norm.  The padding
mask (`mel.abs().sum(-1) == 0`), the positional-embedding lookup and denorm_f0 are torch index ops on the device.  Eval only
(BatchNorm uses its running statistics); no CPU path."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from .fs2 import Linear, PackedWeight, PitchPredictor, _need_hip, _stream, conv1d_cm, denorm_f0, from_cm, to_cm
from .hparams import hparams


def channel_affine_cm(x: torch.Tensor, T: int, a: torch.Tensor, b: torch.Tensor, keep=None) -> torch.Tensor:
    _need_hip(x, 'channel_affine')
    lib = _lib.load()
    B, C, TS = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.dsf_channel_affine(x.data_ptr(), a.data_ptr(), b.data_ptr(), keep.data_ptr() if keep is not None else None,
                                          out.data_ptr(), B, C, T, _stream(x.device)), 'dsf_channel_affine')
    return out


def group_norm_cm(x: torch.Tensor, T: int, groups: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float, *, relu=False, residual=None):
    _need_hip(x, 'group_norm')
    lib = _lib.load()
    B, C, TS = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.dsf_group_norm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), residual.data_ptr() if residual is not None else None,
                                      out.data_ptr(), B, C, groups, T, float(eps), int(relu), _stream(x.device)), 'dsf_group_norm')
    return out


class Prenet(nn.Module):
    """pe.py:8-41 (strides 1).  forward_cm: channel-major mel -> channel-major hidden (the module's second return value)."""

    def __init__(self, in_dim=80, out_dim=256, kernel=5, n_layers=3, strides=None):
        super().__init__()
        if strides is not None and any(s != 1 for s in strides):
            raise NotImplementedError('Prenet strides other than 1')
        self.kernel = kernel
        layers = []
        for _ in range(n_layers):
            layers.append(nn.Sequential(nn.Conv1d(in_dim, out_dim, kernel_size=kernel, padding=kernel // 2), nn.ReLU(), nn.BatchNorm1d(out_dim)))
            in_dim = out_dim
        self.layers = nn.ModuleList(layers)
        self.out_proj = nn.Linear(out_dim, out_dim)
        self._packs = [PackedWeight() for _ in range(n_layers)]
        self._pout = PackedWeight()

    def forward(self, x):
        """x [B,T,80] -> (hiddens [1,B,T,H], out [B,T,H]) like the reference module (pe.py:23-41)."""
        keep = (~x.abs().sum(-1).eq(0)).float().contiguous()
        T = x.shape[1]
        h, out = self.forward_cm(to_cm(x), T, keep, return_hidden=True)
        return from_cm(h, T)[None], from_cm(out, T)

    def forward_cm(self, x, T, keep, return_hidden=False):
        if self.training:
            raise RuntimeError('Prenet: eval mode only (BatchNorm1d runs on its running statistics)')
        for seq, pk in zip(self.layers, self._packs):
            conv, bn = seq[0], seq[2]
            y = conv1d_cm(x, T, conv.weight, pk, conv.bias, act='relu')
            inv = 1.0 / torch.sqrt(bn.running_var + bn.eps)          # aten batch_norm_cpu_transform_input: alpha = invstd * weight,
            a = (inv * bn.weight).contiguous()                       # beta = bias - mean * alpha, out = x * alpha + beta
            b = (bn.bias - bn.running_mean * a).contiguous()
            x = channel_affine_cm(y, T, a, b, keep)
        out = conv1d_cm(x, T, self.out_proj.weight, self._pout, self.out_proj.bias, keep=keep)
        return (x, out) if return_hidden else out


class ConvNorm(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1):
        super().__init__()
        assert kernel_size % 2 == 1
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, padding=(kernel_size - 1) // 2)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain('linear'))


class ConvBlock(nn.Module):
    """pe.py:44-79 with norm 'gn' (the only one PitchExtractor builds)."""

    def __init__(self, idim=80, n_chans=256, kernel_size=3, norm='gn'):
        super().__init__()
        if norm != 'gn':
            raise NotImplementedError(f"ConvBlock norm {norm!r}")
        self.conv = ConvNorm(idim, n_chans, kernel_size)
        self.norm = nn.GroupNorm(n_chans // 16, n_chans)
        self._pack = PackedWeight()


class ConvStacks(nn.Module):
    """pe.py:82-116, res=True, strides 1."""

    def __init__(self, idim=80, n_layers=5, n_chans=256, odim=32, kernel_size=5, norm='gn'):
        super().__init__()
        self.kernel_size = kernel_size
        self.in_proj = Linear(idim, n_chans)
        self.conv = nn.ModuleList([ConvBlock(n_chans, n_chans, kernel_size, norm=norm) for _ in range(n_layers)])
        self.out_proj = Linear(n_chans, odim)
        self._pin, self._pout = PackedWeight(), PackedWeight()

    def forward_cm(self, x, T):
        x = conv1d_cm(x, T, self.in_proj.weight, self._pin, self.in_proj.bias)
        for blk in self.conv:
            c = blk.conv.conv
            y = conv1d_cm(x, T, c.weight, blk._pack, c.bias)
            x = group_norm_cm(y, T, blk.norm.num_groups, blk.norm.weight, blk.norm.bias, blk.norm.eps, relu=True, residual=x)
        return conv1d_cm(x, T, self.out_proj.weight, self._pout, self.out_proj.bias)


class PitchExtractor(nn.Module):
    """pe.py:119-148.  forward(mel [B,T,80]) -> {'pitch_pred' [B,T,2], 'f0_denorm_pred' [B,T]}."""

    def __init__(self, n_mel_bins=80, conv_layers=2):
        super().__init__()
        self.hidden_size = hparams['hidden_size']
        self.predictor_hidden = hparams['predictor_hidden'] if hparams['predictor_hidden'] > 0 else self.hidden_size
        self.conv_layers = conv_layers
        self.mel_prenet = Prenet(n_mel_bins, self.hidden_size, strides=[1, 1, 1])
        if conv_layers > 0:
            self.mel_encoder = ConvStacks(idim=self.hidden_size, n_chans=self.hidden_size, odim=self.hidden_size, n_layers=conv_layers)
        self.pitch_predictor = PitchPredictor(self.hidden_size, n_chans=self.predictor_hidden, n_layers=5, dropout_rate=0.1, odim=2,
                                              padding=hparams['ffn_padding'], kernel_size=hparams['predictor_kernel'])

    @torch.no_grad()
    def forward(self, mel_input=None):
        _need_hip(mel_input, 'PitchExtractor')
        mel_input = mel_input.to(torch.float32)
        B, T, _ = mel_input.shape
        pitch_padding = mel_input.abs().sum(-1) == 0
        keep = (~pitch_padding).float().contiguous()
        x = self.mel_prenet.forward_cm(to_cm(mel_input), T, keep)
        if self.conv_layers > 0:
            x = self.mel_encoder.forward_cm(x, T)
        ret = {}
        ret['pitch_pred'] = pitch_pred = self.pitch_predictor(from_cm(x, T))
        use_uv = hparams['pitch_type'] == 'frame' and hparams['use_uv']
        ret['f0_denorm_pred'] = denorm_f0(pitch_pred[:, :, 0].clone(), (pitch_pred[:, :, 1] > 0) if use_uv else None, hparams,
                                          pitch_padding=pitch_padding)
        return ret
