"""`FFT`: the reference's transformer candidate denoiser (usr/diff/candidate_decoder.py:35-96, `diff_decoder_type: 'fft'`,
registered in usr/diffsinger_task.py:23-27) on the HIP FastSpeech2 operators (include/dsf.h) - SURVEY.md section 8 row f4.

Same constructor, parameter names and shapes as the reference class (a reference state_dict loads with strict=True), same
call `forward(spec [B,1,M,T], diffusion_step [B], cond [B,H,T]) -> [B,1,M,T]` (inference: no graph) and `forward_train(...)`, the same
function under autograd - the reference trains whatever `DIFF_DECODERS` returns (usr/diffsinger_task.py:23-27 with
usr/diff/shallow_diffusion_tts.py:213-231); every operator here has its HIP backward (diffsinger_amd/fs2.py, train.py), so
`GaussianDiffusion.p_losses` back-propagates through this denoiser like through the FastSpeech2 blocks it is made of.  Every contraction (input
projection, step MLP, get_decode_inp, the FFT blocks, get_mel_out), LayerNorm and the attention core run as HIP kernels; the
step's sinusoid, the channel concatenation and the position lookup are torch data movement on the device."""
from __future__ import annotations

import math

import torch
from torch import nn

from .fs2 import FastspeechDecoder, PackedWeight, _HipLinear, conv1d_cm, from_cm, to_cm
from .hparams import hparams
from .net import Mish


class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half_dim = self.dim // 2                                    # candidate_decoder.py:19-26
        emb = math.log(10000) / (half_dim - 1)
        emb = torch.exp(torch.arange(half_dim, device=x.device) * -emb)
        emb = x[:, None] * emb[None, :]
        return torch.cat((emb.sin(), emb.cos()), dim=-1)


class FFT(FastspeechDecoder):
    def __init__(self, hidden_size=None, num_layers=None, kernel_size=None, num_heads=None):
        super().__init__(hidden_size, num_layers, kernel_size, num_heads=num_heads)
        dim = hparams['residual_channels']
        self.input_projection = nn.Conv1d(hparams['audio_num_mel_bins'], dim, 1)
        nn.init.kaiming_normal_(self.input_projection.weight)
        self.diffusion_embedding = SinusoidalPosEmb(dim)
        self.mlp = nn.Sequential(_HipLinear(dim, dim * 4), Mish(), _HipLinear(dim * 4, dim))
        self.get_mel_out = nn.Linear(hparams['hidden_size'], 80, bias=True)
        self.get_decode_inp = nn.Linear(hparams['hidden_size'] + dim + dim, hparams['hidden_size'])
        self._pin, self._pdec, self._pmel = PackedWeight(), PackedWeight(), PackedWeight()

    @torch.no_grad()
    def forward(self, spec, diffusion_step, cond, padding_mask=None, attn_mask=None, return_hiddens=False):
        return self._forward(spec, diffusion_step, cond, padding_mask, attn_mask, return_hiddens)

    def forward_train(self, spec, diffusion_step, cond, padding_mask=None):
        """The same function with autograd on: what `p_losses` differentiates (candidate_decoder.py:66-96 under the reference's training step)."""
        return self._forward(spec, diffusion_step, cond, padding_mask)

    def _forward(self, spec, diffusion_step, cond, padding_mask=None, attn_mask=None, return_hiddens=False):
        if attn_mask is not None or return_hiddens:
            raise NotImplementedError('attn_mask / return_hiddens')
        B, _, M, T = spec.shape
        xc = conv1d_cm(to_cm(spec[:, 0].transpose(1, 2)), T, self.input_projection.weight, self._pin, self.input_projection.bias)   # [B][dim][TS]
        d = self.diffusion_embedding(diffusion_step.reshape(-1))
        d = self.mlp[2](self.mlp[0](d, act='mish'))                                          # [B, dim]
        TS = xc.shape[2]
        te = torch.zeros(B, d.shape[1], TS, device=spec.device, dtype=torch.float32)
        te[:, :, :T] = d[:, :, None]
        cat = torch.cat([xc, to_cm(cond.transpose(1, 2)), te], dim=1)                        # channels: x | cond | time embedding
        x = from_cm(conv1d_cm(cat, T, self.get_decode_inp.weight, self._pdec, self.get_decode_inp.bias), T)   # [B,T,H]
        yc, T, keep = self.forward_cm(x, padding_mask)                                       # FFTBlocks (tts_modules.py:288-314)
        out = conv1d_cm(yc, T, self.get_mel_out.weight, self._pmel, self.get_mel_out.bias)   # [B][80][TS]
        return out[:, :, :T].contiguous()[:, None, :, :]
