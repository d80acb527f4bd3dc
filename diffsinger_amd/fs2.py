"""FastSpeech2 / FastSpeech2MIDI - the conditioner + aux decoder in front of the diffusion hot path (SURVEY.md section 8
row f1) - as nn.Modules whose forward (and, under autograd, backward) runs on the HIP kernels of libdsdenoise.so (include/dsf.h).

Mirrors the reference module tree (paths relative to the reference root) name for name, so a reference checkpoint's
`model.fs2.*` state_dict loads with strict=True and vice versa:

    FastSpeech2 / FastSpeech2MIDI            modules/fastspeech/fs2.py:23-255, modules/diffsinger_midi/fs2.py:47-118
    FastspeechEncoder / MIDIEncoder / Decoder, FFTBlocks, DurationPredictor, PitchPredictor, LengthRegulator
                                              modules/fastspeech/tts_modules.py:58-356, modules/diffsinger_midi/fs2.py:10-37
    EncSALayer, TransformerFFNLayer, MultiheadAttention, SinusoidalPositionalEmbedding
                                              modules/commons/common_layers.py:88-147, 166-263, 486-588

What runs where: every contraction (attention projections, the k=9 conv-FFN, predictor convolutions, mel_out), every
LayerNorm and the softmax-attention core are HIP kernels on a channel-major [B][C][T] layout (>99.9 % of the FLOPs).  The
index plumbing between them - embedding lookups, the length-regulator gather, padding masks, f0 quantisation - is data
movement on [B,T,C] tensors and uses torch indexing ops on the device.  forward(infer=False) under autograd builds a graph whose
backward runs on HIP kernels as well (joint training of the e2e configurations); no CPU path: the ops raise when the tensors are
not on the MI355X.

Covered besides the shipped configurations: speaker d-vectors / speaker ids (use_spk_embed / use_spk_id), the energy embedding and
phone-level pitch (pitch_type 'ph').  Not covered (raise NotImplementedError): pitch_ar (raises in the reference itself: fs2.py:215 passes
two arguments to PitchPredictor.forward), dur_loss other than 'mse', ffn_padding 'LEFT', norm 'bn' - none is used by a shipped
DiffSpeech / DiffSinger config."""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .hparams import hparams

ACT = {'none': 0, 'relu': 1, 'gelu': 2, 'mish': 3}
f0_bin = 256
f0_mel_min = 1127 * np.log(1 + 50.0 / 700)
f0_mel_max = 1127 * np.log(1 + 1100.0 / 700)


# --------------------------------------------------------------------------------------------------------------
# thin wrappers over the C ABI (include/dsf.h).  Tensors in "cm" = channel-major [B][C][TS] fp32, TS = T up to 32.
# --------------------------------------------------------------------------------------------------------------
def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _need_hip(t: torch.Tensor, what: str):
    if t.device.type != 'cuda':
        raise RuntimeError(f'{what}: the FastSpeech2 HIP ops have no CPU path - move the module and its inputs to the MI355X')


def padded_frames(T: int) -> int:
    return (T + 31) // 32 * 32


class PackedWeight:
    """A Conv1d / Linear weight in MFMA-fragment order (dsf_pack_weight), re-packed when the parameter changes."""

    def __init__(self):
        self.tag = None
        self.buf = None

    def get(self, w: torch.Tensor) -> torch.Tensor:
        from .train_dist import param_generation                   # raw-pointer optimiser steps do not bump torch's version counters
        tag = (w.data_ptr(), w._version, tuple(w.shape), w.device, param_generation())
        if tag != self.tag:
            _need_hip(w, 'weight')
            lib = _lib.load()
            w3 = w.detach().to(torch.float32).contiguous()
            if w3.dim() == 2:
                w3 = w3[:, :, None]
            co, ci, k = w3.shape
            n = lib.dsf_packed_floats(co, ci, k)
            if n < 0:
                raise ValueError(f'unsupported weight shape {tuple(w.shape)} (input channels must be a multiple of 8)')
            buf = torch.empty(n, device=w.device, dtype=torch.float32)
            with torch.cuda.device(w.device):
                _lib.check(lib.dsf_pack_weight(w3.data_ptr(), co, ci, k, buf.data_ptr(), _stream(w.device)), 'dsf_pack_weight')
            w3.record_stream(torch.cuda.current_stream(w.device))        # w3 may be a temporary: keep its memory until the pack ran
            self.buf, self.tag = buf, tag
        return self.buf


def set_conv_split(mode: int):
    """Kernel choice of conv1d_cm: -1 by grid size (default: small grids run k_fs_conv_ks, 64-row workgroups whose waves split the contraction),
    0 never, 1 wherever the shape allows it.  Process-wide (dsf_set_conv_split); for tests and A/B measurements."""
    _lib.check(_lib.load().dsf_set_conv_split(int(mode)), 'dsf_set_conv_split')


def conv1d_cm(x: torch.Tensor, T: int, weight: torch.Tensor, packed: PackedWeight, bias: Optional[torch.Tensor] = None, *,
              scale: float = 1.0, act: str = 'none', residual: Optional[torch.Tensor] = None, keep: Optional[torch.Tensor] = None):
    """y = act(scale * (W * x + bias)) (+ residual) (* keep): nn.Conv1d 'SAME' / nn.Linear on a cm tensor."""
    _need_hip(x, 'conv1d')
    if _needs_grad(x, weight, bias, residual):
        return _conv1d_cm_train(x, T, weight, packed, bias, scale, act, residual, keep)
    lib = _lib.load()
    B, Ci, TS = x.shape
    Co = weight.shape[0]
    K = weight.shape[2] if weight.dim() == 3 else 1
    assert TS == padded_frames(T) and x.is_contiguous() and weight.shape[1] == Ci
    out = torch.empty(B, Co, TS, device=x.device, dtype=torch.float32)
    wp = packed.get(weight)
    with torch.cuda.device(x.device):
        _lib.check(lib.dsf_conv1d(x.data_ptr(), wp.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), B, Ci, Co, K, T,
                                  float(scale), ACT[act], residual.data_ptr() if residual is not None else None,
                                  keep.data_ptr() if keep is not None else None, _stream(x.device)), 'dsf_conv1d')
    return out


def layer_norm_cm(x: torch.Tensor, T: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float, *, relu_in: bool = False,
                  keep: Optional[torch.Tensor] = None):
    _need_hip(x, 'layer_norm')
    if _needs_grad(x, gamma, beta):
        return _LayerNormCM.apply(x, gamma, beta, T, float(eps), bool(relu_in), keep)
    return _layer_norm_raw(x, T, gamma, beta, eps, relu_in, keep)


def _layer_norm_raw(x, T, gamma, beta, eps, relu_in, keep):
    lib = _lib.load()
    B, Cc, TS = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.dsf_layer_norm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), B, Cc, T, float(eps), int(relu_in),
                                      keep.data_ptr() if keep is not None else None, _stream(x.device)), 'dsf_layer_norm')
    return out


def attention_cm(qkv: torch.Tensor, T: int, key_pad_u8: Optional[torch.Tensor], heads: int):
    _need_hip(qkv, 'attention')
    if _needs_grad(qkv):
        return _AttentionCM.apply(qkv, T, key_pad_u8, heads)
    return _attention_raw(qkv, T, key_pad_u8, heads)


def _attention_raw(qkv, T, key_pad_u8, heads):
    lib = _lib.load()
    B, C3, TS = qkv.shape
    out = torch.empty(B, C3 // 3, TS, device=qkv.device, dtype=torch.float32)
    with torch.cuda.device(qkv.device):
        _lib.check(lib.dsf_attention(qkv.data_ptr(), key_pad_u8.data_ptr() if key_pad_u8 is not None else None, out.data_ptr(), B, C3 // 3,
                                     heads, T, _stream(qkv.device)), 'dsf_attention')
    return out


def to_cm(x_btc: torch.Tensor) -> torch.Tensor:
    """[B,T,C] (any strides) -> channel-major [B][C][TS], zero in [T,TS)."""
    _need_hip(x_btc, 'to_cm')
    if _needs_grad(x_btc):
        return _ToCM.apply(x_btc)
    return _to_cm_raw(x_btc)


def _to_cm_raw(x_btc: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x_btc = x_btc.to(torch.float32)
    B, T, Cc = x_btc.shape
    out = torch.empty(B, Cc, padded_frames(T), device=x_btc.device, dtype=torch.float32)
    sb, st, sc = x_btc.stride()
    with torch.cuda.device(x_btc.device):
        _lib.check(lib.dsf_to_channel_major(x_btc.data_ptr(), sb, sc, st, out.data_ptr(), B, Cc, T, _stream(x_btc.device)), 'dsf_to_channel_major')
    return out


def from_cm(x_cm: torch.Tensor, T: int) -> torch.Tensor:
    _need_hip(x_cm, 'from_cm')
    if _needs_grad(x_cm):
        return _FromCM.apply(x_cm, T)
    return _from_cm_raw(x_cm, T)


def _from_cm_raw(x_cm: torch.Tensor, T: int) -> torch.Tensor:
    lib = _lib.load()
    B, Cc, TS = x_cm.shape
    out = torch.empty(B, T, Cc, device=x_cm.device, dtype=torch.float32)
    with torch.cuda.device(x_cm.device):
        _lib.check(lib.dsf_from_channel_major(x_cm.data_ptr(), out.data_ptr(), B, Cc, T, _stream(x_cm.device)), 'dsf_from_channel_major')
    return out


# --------------------------------------------------------------------------------------------------------------
# the same operators under autograd (training of FastSpeech2 / FastSpeech2MIDI: the Opencpop e2e configuration trains it jointly with the
# denoiser - usr/diffsinger_task.py:60-64, :273-300, usr/configs/midi/e2e/opencpop/ds1000.yaml:18).  Forward = the inference kernels;
# backward = HIP kernels too: convolutions through train._Conv1dCM (data gradient = the conv kernel with the flipped, transposed weight; weight
# and bias gradient = dsf_conv1d_wgrad), LayerNorm and the attention core through dsf_layer_norm_bwd / dsf_attention_bwd (csrc/fs2_train.hpp),
# the layout changes are each other's adjoints.  The fused epilogues of the inference convolution (scale, activation, residual, padding mask)
# are torch element-wise ops here - glue, not contractions.
# --------------------------------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------------------------------
# the index / mask glue of the forward as four operators (include/dsf.h; inference only - under autograd the torch ops below them run)
# --------------------------------------------------------------------------------------------------------------
_GLUE = True


def set_glue(on: bool):
    """A/B switch of the tests and of the measurement: False = the index / mask glue as the reference's torch op sequence (rounds 1-5)."""
    global _GLUE
    _GLUE = bool(on)


def _glue_ok(*tensors) -> bool:
    """The fused glue operators apply: no autograd graph is being recorded and every tensor lives on a HIP device."""
    return _GLUE and (not torch.is_grad_enabled()) and all(t is None or (torch.is_tensor(t) and t.is_cuda) for t in tensors)


def positions_op(tokens: Optional[torch.Tensor] = None, x: Optional[torch.Tensor] = None, padding_idx: int = 0) -> torch.Tensor:
    """make_positions (utils/__init__.py:145-157) of an int64 token tensor [B,T] or of channel 0 of a float tensor [B,T,C] -> int32 [B,T]."""
    src = tokens if tokens is not None else x
    B, T = src.shape[:2]
    pos = torch.empty(B, T, device=src.device, dtype=torch.int32)
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().dsf_positions(tokens.data_ptr() if tokens is not None else None, x.data_ptr() if x is not None else None, pos.data_ptr(),
                                             B, T, x.shape[2] if x is not None else 0, int(padding_idx), _stream(src.device)), 'dsf_positions')
    return pos


def input_cm_op(*, tokens=None, emb=None, emb_scale=1.0, adds=(), x=None, pos=None, pos_table=None, alpha=None, padding_mask=None, padding_idx=0, mask=True):
    """dsf_input_cm: (xc [B][C][TS], keep [B][T], key-padding mask u8 [B][T]) of an FFT stack's input."""
    src = tokens if tokens is not None else x
    B, T = src.shape[:2]
    C = emb.shape[1] if tokens is not None else x.shape[2]
    dev = src.device
    xc = torch.empty(B, C, padded_frames(T), device=dev, dtype=torch.float32)
    keep = torch.empty(B, T, device=dev, dtype=torch.float32)
    pad = torch.empty(B, T, device=dev, dtype=torch.uint8)
    adds = [a for a in adds if torch.is_tensor(a)]
    assert len(adds) <= 3 and all(a.shape == (B, T, C) and a.is_contiguous() and a.dtype == torch.float32 for a in adds)
    ap = [a.data_ptr() for a in adds] + [None] * (3 - len(adds))
    pm = None
    if padding_mask is not None:
        pm = padding_mask.contiguous()
        pm = pm.view(torch.uint8) if pm.dtype == torch.bool else pm.to(torch.uint8)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().dsf_input_cm(tokens.data_ptr() if tokens is not None else None, emb.data_ptr() if emb is not None else None, float(emb_scale),
                                            ap[0], ap[1], ap[2], x.data_ptr() if x is not None else None, pos.data_ptr() if pos is not None else None,
                                            pos_table.data_ptr() if pos_table is not None else None, alpha.data_ptr() if alpha is not None else None,
                                            pm.data_ptr() if pm is not None else None, xc.data_ptr(), keep.data_ptr(), pad.data_ptr(), B, T, C,
                                            int(padding_idx), int(bool(mask)), _stream(dev)), 'dsf_input_cm')
    return xc, keep, pad


def gather_frames_op(enc: torch.Tensor, mel2ph: torch.Tensor, spk):
    """fs2.py:128-134: (decoder_inp, (decoder_inp + spk) * (mel2ph > 0)); spk: [B,1,C] tensor or the integer 0."""
    B, Tp, C = enc.shape
    T = mel2ph.shape[1]
    out = torch.empty(B, T, C, device=enc.device, dtype=torch.float32)
    masked = torch.empty_like(out)
    spk2 = spk.reshape(B, C).contiguous() if torch.is_tensor(spk) else None
    with torch.cuda.device(enc.device):
        _lib.check(_lib.load().dsf_gather_frames(enc.data_ptr(), mel2ph.data_ptr(), spk2.data_ptr() if spk2 is not None else None, out.data_ptr(),
                                                 masked.data_ptr(), B, T, Tp, C, _stream(enc.device)), 'dsf_gather_frames')
    return out, masked


def sum_embed_op(dec: torch.Tensor, mel2ph: torch.Tensor, *, idx1=None, tab1=None, add1=None, idx2=None, tab2=None, spk=None):
    """fs2.py:136-141: (((decoder_inp + pitch embedding) + energy embedding) + spk) * (mel2ph > 0)."""
    B, T, C = dec.shape
    out = torch.empty_like(dec)
    spk2 = spk.reshape(B, C).contiguous() if torch.is_tensor(spk) else None
    p = lambda t: t.data_ptr() if t is not None else None
    with torch.cuda.device(dec.device):
        _lib.check(_lib.load().dsf_sum_embed(dec.data_ptr(), p(idx1), p(tab1), p(add1), p(idx2), p(tab2), p(spk2), mel2ph.data_ptr(), out.data_ptr(), B, T, C,
                                             _stream(dec.device)), 'dsf_sum_embed')
    return out


def token_masks_op(v: torch.Tensor, *, gt0=False, eq0=False, ne0=False):
    """dsf_token_masks on an int64 index tensor (txt_tokens, mel2ph): ((v > 0).float(), v == 0 [bool], (~(v == 0)).float()), None where not asked for."""
    v = v.contiguous()
    o_gt = torch.empty(v.shape, device=v.device, dtype=torch.float32) if gt0 else None
    o_eq = torch.empty(v.shape, device=v.device, dtype=torch.bool) if eq0 else None
    o_ne = torch.empty(v.shape, device=v.device, dtype=torch.float32) if ne0 else None
    p = lambda t: t.data_ptr() if t is not None else None
    with torch.cuda.device(v.device):
        _lib.check(_lib.load().dsf_token_masks(v.data_ptr(), p(o_gt), p(o_eq), p(o_ne), v.numel(), _stream(v.device)), 'dsf_token_masks')
    return o_gt, o_eq, o_ne


# How much of the pitch quantisation runs inside dsf_pitch_coarse: the device library's pow / log (the functions ATen's kernels call; compared over
# every float of the working range in tests/test_gpu_fs2.py) or, switched off, torch's own kernels around the split stages of the operator.
_PITCH_NATIVE = {'pow': True, 'log': True}


def set_pitch_native(pow: bool = True, log: bool = True):
    _PITCH_NATIVE['pow'], _PITCH_NATIVE['log'] = bool(pow), bool(log)


def _pitch_fusable(f0, uv, hp) -> bool:
    return (_glue_ok(f0, uv) and torch.is_tensor(f0) and f0.dtype == torch.float32 and f0.dim() == 2 and hp['pitch_norm'] in ('standard', 'log')
            and (uv is None or not hp['use_uv'] or (uv.shape == f0.shape and uv.dtype in (torch.float32, torch.bool, torch.uint8))))


def pitch_coarse_op(f0: torch.Tensor, uv, mel2ph, hp):
    """denorm_f0 + f0_to_coarse (utils/pitch_utils.py:64-77, :21-30) of f0 [B,T] (any strides) as ONE launch: (f0_denorm float [B,T], coarse int64 [B,T]).
    uv: float / bool [B,T] or None; mel2ph: int64 [B,T] whose zeros are padding frames, or None."""
    B, T = f0.shape
    dev = f0.device
    norm = 1 if hp['pitch_norm'] == 'standard' else 2
    native_pow = _PITCH_NATIVE['pow'] or norm == 1
    src = f0 if native_pow else (2 ** f0).contiguous()
    uv = uv if (uv is not None and hp['use_uv']) else None
    uv_f = uv.contiguous() if (uv is not None and uv.dtype == torch.float32) else None
    uv_b = uv.contiguous() if (uv is not None and uv.dtype != torch.float32) else None
    m2p = mel2ph.contiguous() if mel2ph is not None else None
    den = torch.empty(B, T, device=dev, dtype=torch.float32)
    coarse = torch.empty(B, T, device=dev, dtype=torch.int64)
    tmp = None if _PITCH_NATIVE['log'] else torch.empty(B, T, device=dev, dtype=torch.float32)
    p = lambda t: t.data_ptr() if t is not None else None
    mean, std = (float(hp['f0_mean']), float(hp['f0_std'])) if norm == 1 else (0.0, 1.0)
    lib = _lib.load()

    def run(stage):
        _lib.check(lib.dsf_pitch_coarse(src.data_ptr(), src.stride(0), src.stride(1), p(uv_f), p(uv_b), p(m2p), den.data_ptr(), p(tmp), coarse.data_ptr(), B, T,
                                        norm, mean, std, float(f0_mel_min), float(f0_mel_max), f0_bin, stage, int(not native_pow), _stream(dev)), 'dsf_pitch_coarse')
    with torch.cuda.device(dev):
        if tmp is None:
            run(0)
        else:
            run(1)
            tmp.log_()
            run(2)
    return den, coarse


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _keep_cm(keep: torch.Tensor, TS: int) -> torch.Tensor:
    """[B,T] mask -> [B,1,TS] (zero tail) for broadcasting over a cm tensor."""
    return F.pad(keep, (0, TS - keep.shape[1]))[:, None, :]


def _conv1d_cm_train(x, T, weight, packed, bias, scale, act, residual, keep):
    from .train import ConvCache
    cache = packed.__dict__.get('train')
    if cache is None:
        cache = packed.__dict__['train'] = ConvCache()
    y = cache(x.contiguous(), weight, bias, T)
    if scale != 1.0:
        y = y * scale
    if act == 'relu':
        y = F.relu(y)
    elif act == 'gelu':
        y = F.gelu(y)
    elif act == 'mish':
        y = y * torch.tanh(F.softplus(y))
    if residual is not None:
        y = y + residual
    if keep is not None:
        y = y * _keep_cm(keep, y.shape[2])
    return y


class _LayerNormCM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, T, eps, relu_in, keep):
        x = x.contiguous()
        out = _layer_norm_raw(x, T, gamma.detach(), beta.detach(), eps, relu_in, keep)
        ctx.save_for_backward(x, gamma.detach())
        ctx.keep, ctx.T, ctx.eps, ctx.relu_in = keep, T, eps, relu_in
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma = ctx.saved_tensors
        B, Cc, TS = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(Cc, device=x.device, dtype=torch.float32)
        db = torch.empty(Cc, device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.dsf_ln_bwd_workspace_floats(B, ctx.T), device=x.device, dtype=torch.float32)
        keep = ctx.keep
        with torch.cuda.device(x.device):
            _lib.check(lib.dsf_layer_norm_bwd(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), keep.data_ptr() if keep is not None else None,
                                              dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), B, Cc, ctx.T, float(ctx.eps), int(ctx.relu_in),
                                              _stream(x.device)), 'dsf_layer_norm_bwd')
        return dx, dg, db, None, None, None, None


class _AttentionCM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, T, key_pad_u8, heads):
        qkv = qkv.contiguous()
        out = _attention_raw(qkv, T, key_pad_u8, heads)
        ctx.save_for_backward(qkv)
        ctx.key_pad, ctx.T, ctx.heads = key_pad_u8, T, heads
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        qkv, = ctx.saved_tensors
        B, C3, TS = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        ws = torch.empty(lib.dsf_attention_bwd_workspace_floats(B, ctx.heads, ctx.T), device=qkv.device, dtype=torch.float32)
        kp = ctx.key_pad
        with torch.cuda.device(qkv.device):
            _lib.check(lib.dsf_attention_bwd(qkv.data_ptr(), kp.data_ptr() if kp is not None else None, dout.data_ptr(), dqkv.data_ptr(), ws.data_ptr(),
                                             B, C3 // 3, ctx.heads, ctx.T, _stream(qkv.device)), 'dsf_attention_bwd')
        return dqkv, None, None, None


class _ToCM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_btc):
        ctx.T = x_btc.shape[1]
        return _to_cm_raw(x_btc)

    @staticmethod
    def backward(ctx, g):
        return _from_cm_raw(g.contiguous(), ctx.T)


class _FromCM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cm, T):
        return _from_cm_raw(x_cm.contiguous(), T)

    @staticmethod
    def backward(ctx, g):
        return _to_cm_raw(g), None


def _drop(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    return F.dropout(x, p, True) if (training and p > 0) else x


# --------------------------------------------------------------------------------------------------------------
# parameter containers with the reference's names (init follows the reference's initialisers)
# --------------------------------------------------------------------------------------------------------------
def Embedding(num_embeddings, embedding_dim, padding_idx=None):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)       # common_layers.py:62-67
    nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
    if padding_idx is not None:
        nn.init.constant_(m.weight[padding_idx], 0)
    return m


def Linear(in_features, out_features, bias=True):
    m = nn.Linear(in_features, out_features, bias)                                  # common_layers.py:80-85
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.)
    return m


def make_positions(tensor, padding_idx):
    mask = tensor.ne(padding_idx).int()                                             # utils/__init__.py:145-157
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


class SinusoidalPositionalEmbedding(nn.Module):
    """common_layers.py:88-147.  The table is a constant built on the host exactly like the reference builds it (torch CPU
    ops at construction) and cached on the device; the lookup is an index_select."""

    def __init__(self, embedding_dim, padding_idx, init_size=1024):
        super().__init__()
        self.embedding_dim, self.padding_idx = embedding_dim, padding_idx
        self.weights = self.get_embedding(init_size, embedding_dim, padding_idx)
        self.register_buffer('_float_tensor', torch.FloatTensor(1))

    @staticmethod
    def get_embedding(num_embeddings, embedding_dim, padding_idx=None):
        half_dim = embedding_dim // 2
        emb = math.log(10000) / (half_dim - 1)
        emb = torch.exp(torch.arange(half_dim, dtype=torch.float) * -emb)
        emb = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * emb.unsqueeze(0)
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1).view(num_embeddings, -1)
        if embedding_dim % 2 == 1:
            emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
        if padding_idx is not None:
            emb[padding_idx, :] = 0
        return emb

    def table(self, seq_len: int) -> torch.Tensor:
        """The sinusoidal table, grown / moved exactly as forward() does it (common_layers.py:135-141)."""
        max_pos = self.padding_idx + 1 + seq_len
        if max_pos > self.weights.size(0):
            self.weights = self.get_embedding(max_pos, self.embedding_dim, self.padding_idx)
        self.weights = self.weights.to(self._float_tensor)
        return self.weights

    def forward(self, input, **kwargs):
        bsz, seq_len = input.shape[:2]
        self.table(seq_len)
        positions = make_positions(input, self.padding_idx)
        return self.weights.index_select(0, positions.view(-1)).view(bsz, seq_len, -1).detach()


class RelPositionalEncoding(nn.Module):
    """espnet_positional_embedding.py:86-112 as FastspeechMIDIEncoder uses it (x * sqrt(d) + pe[:T], reversed positions)."""

    def __init__(self, d_model, dropout_rate=0.0, max_len=5000):
        super().__init__()
        self.d_model, self.xscale, self.pe = d_model, math.sqrt(d_model), None
        self.extend_pe(max_len)

    def extend_pe(self, n):
        if self.pe is not None and self.pe.size(1) >= n:
            return
        pe = torch.zeros(n, self.d_model)
        position = torch.arange(n - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, self.d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / self.d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.pe = pe.unsqueeze(0)

    def forward(self, x):
        self.extend_pe(x.size(1))
        self.pe = self.pe.to(device=x.device, dtype=x.dtype)
        return x * self.xscale + self.pe[:, :x.size(1)]


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.Tensor(3 * embed_dim, embed_dim))
        self.register_parameter('in_proj_bias', None)                               # bias=False (common_layers.py:557-559)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj.weight)
        self._pin, self._pout = PackedWeight(), PackedWeight()


class TransformerFFNLayer(nn.Module):
    def __init__(self, hidden_size, filter_size, padding='SAME', kernel_size=1, act='gelu'):
        super().__init__()
        if padding != 'SAME' or act not in ('gelu', 'relu'):
            raise NotImplementedError(f'ffn_padding {padding} / ffn_act {act}')
        self.kernel_size, self.act = kernel_size, act
        self.ffn_1 = nn.Conv1d(hidden_size, filter_size, kernel_size, padding=kernel_size // 2)
        self.ffn_2 = Linear(filter_size, hidden_size)
        self._p1, self._p2 = PackedWeight(), PackedWeight()


class EncSALayer(nn.Module):
    def __init__(self, c, num_heads, kernel_size=9, padding='SAME', act='gelu', dropout=None):
        super().__init__()
        self.c, self.num_heads = c, num_heads
        self.dropout = hparams['dropout'] if dropout is None else dropout      # tts_modules.py:17-26: dropout = relu_dropout, attention_dropout 0
        self.layer_norm1 = nn.LayerNorm(c)
        self.self_attn = MultiheadAttention(c, num_heads)
        self.layer_norm2 = nn.LayerNorm(c)
        self.ffn = TransformerFFNLayer(c, 4 * c, kernel_size=kernel_size, padding=padding, act=act)

    def forward_cm(self, x, T, keep, pad_u8):
        """EncSALayer.forward (common_layers.py:565-588) on a cm tensor; with self.training and dropout > 0 the three dropouts of the reference
        (:576, TransformerFFNLayer :520, :584) sit between the convolutions and their residual adds, so those run unfused."""
        a, f = self.self_attn, self.ffn
        p = self.dropout if self.training else 0.0
        y = layer_norm_cm(x, T, self.layer_norm1.weight, self.layer_norm1.bias, 1e-5)
        qkv = conv1d_cm(y, T, a.in_proj_weight, a._pin)
        o = attention_cm(qkv, T, pad_u8, self.num_heads)
        if p > 0:
            x = (x + F.dropout(conv1d_cm(o, T, a.out_proj.weight, a._pout), p, True)) * _keep_cm(keep, x.shape[2])
        else:
            x = conv1d_cm(o, T, a.out_proj.weight, a._pout, residual=x, keep=keep)
        y = layer_norm_cm(x, T, self.layer_norm2.weight, self.layer_norm2.bias, 1e-5)
        hdn = conv1d_cm(y, T, f.ffn_1.weight, f._p1, f.ffn_1.bias, scale=f.kernel_size ** -0.5, act=f.act)
        if p > 0:
            out = conv1d_cm(F.dropout(hdn, p, True), T, f.ffn_2.weight, f._p2, f.ffn_2.bias)
            return (x + F.dropout(out, p, True)) * _keep_cm(keep, x.shape[2])
        return conv1d_cm(hdn, T, f.ffn_2.weight, f._p2, f.ffn_2.bias, residual=x, keep=keep)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, hidden_size, kernel_size, num_heads):
        super().__init__()
        self.op = EncSALayer(hidden_size, num_heads, kernel_size=kernel_size, padding=hparams['ffn_padding'], act=hparams['ffn_act'])


class FFTBlocks(nn.Module):
    def __init__(self, hidden_size, num_layers, ffn_kernel_size=9, num_heads=2, use_pos_embed=True, use_last_norm=True):
        super().__init__()
        self.num_layers, self.hidden_size, self.use_pos_embed = num_layers, hidden_size, use_pos_embed
        if use_pos_embed:
            self.padding_idx = 0
            self.pos_embed_alpha = nn.Parameter(torch.Tensor([1]))
            self.embed_positions = SinusoidalPositionalEmbedding(hidden_size, 0, init_size=2000)
        self.layers = nn.ModuleList([TransformerEncoderLayer(hidden_size, ffn_kernel_size, num_heads) for _ in range(num_layers)])
        self.layer_norm = nn.LayerNorm(hidden_size) if use_last_norm else None

    def forward_cm(self, x, padding_mask=None):
        """FFTBlocks.forward (tts_modules.py:288-314), eval mode.  x [B,T,C] -> channel-major [B][C][TS] (and T, keep)."""
        _need_hip(x, 'FFTBlocks')
        T = x.shape[1]
        if _glue_ok(x, padding_mask) and x.dtype == torch.float32 and not (self.use_pos_embed and self.training and hparams['dropout'] > 0):
            # padding mask, positions, positional embedding, `* nonpadding` and the transposition: two launches (dsf_positions, dsf_input_cm)
            x = x.contiguous()
            pos = tab = None
            if self.use_pos_embed:
                pos, tab = positions_op(x=x, padding_idx=self.embed_positions.padding_idx), self.embed_positions.table(T)
            xc, keep, pad_u8 = input_cm_op(x=x, pos=pos, pos_table=tab, alpha=self.pos_embed_alpha if self.use_pos_embed else None,
                                           padding_mask=padding_mask)
            return self.layers_cm(xc, T, keep, pad_u8)
        padding_mask = x.abs().sum(-1).eq(0) if padding_mask is None else padding_mask
        keep = (~padding_mask).float().contiguous()
        pad_u8 = padding_mask.to(torch.uint8).contiguous()
        if self.use_pos_embed:
            x = x + self.pos_embed_alpha * self.embed_positions(x[..., 0])
            x = _drop(x, hparams['dropout'], self.training)                          # tts_modules.py:293
        xc = to_cm(x * keep[:, :, None])
        return self.layers_cm(xc, T, keep, pad_u8)

    def layers_cm(self, xc, T, keep, pad_u8):
        for layer in self.layers:
            xc = layer.op.forward_cm(xc, T, keep, pad_u8)
        if self.layer_norm is not None:
            xc = layer_norm_cm(xc, T, self.layer_norm.weight, self.layer_norm.bias, 1e-5, keep=keep)
        return xc, T, keep

    def forward(self, x, padding_mask=None, **kwargs):
        xc, T, _ = self.forward_cm(x, padding_mask)
        return from_cm(xc, T)


class FastspeechEncoder(FFTBlocks):
    def __init__(self, embed_tokens, hidden_size, num_layers, kernel_size, num_heads=2):
        super().__init__(hidden_size, num_layers, kernel_size, num_heads=num_heads, use_pos_embed=False)
        self.embed_tokens = embed_tokens
        self.embed_scale = math.sqrt(hidden_size)
        self.padding_idx = 0
        if hparams.get('rel_pos'):
            self.embed_positions = RelPositionalEncoding(hidden_size, dropout_rate=0.0)
        else:
            self.embed_positions = SinusoidalPositionalEmbedding(hidden_size, 0, init_size=2000)

    def forward_embedding(self, txt_tokens):
        x = self.embed_scale * self.embed_tokens(txt_tokens)
        if hparams['use_pos_embed']:
            if hparams.get('rel_pos'):
                raise NotImplementedError('rel_pos without use_midi (the reference would scale the integer tokens)')
            x = x + self.embed_positions(txt_tokens)
        return _drop(x, hparams['dropout'], self.training)                           # tts_modules.py:346

    def _embed_cm(self, txt_tokens, adds=()):
        """forward_embedding + the front end of FFTBlocks.forward as two launches (inference; None: not applicable - rel_pos, dropout, CPU)."""
        if not _glue_ok(txt_tokens, *adds) or hparams.get('rel_pos') or (self.training and hparams['dropout'] > 0) or txt_tokens.dtype != torch.int64:
            return None
        T = txt_tokens.shape[1]
        tok = txt_tokens.contiguous()
        pos = tab = None
        if hparams['use_pos_embed']:
            pos, tab = positions_op(tokens=tok, padding_idx=self.padding_idx), self.embed_positions.table(T)
        xc, keep, pad_u8 = input_cm_op(tokens=tok, emb=self.embed_tokens.weight, emb_scale=self.embed_scale, adds=adds, pos=pos, pos_table=tab,
                                       padding_idx=self.padding_idx)
        return self.layers_cm(xc, T, keep, pad_u8)

    def forward(self, txt_tokens):
        fast = self._embed_cm(txt_tokens)
        if fast is not None:
            return from_cm(fast[0], fast[1])
        return FFTBlocks.forward(self, self.forward_embedding(txt_tokens), txt_tokens.eq(self.padding_idx))


class FastspeechMIDIEncoder(FastspeechEncoder):
    def forward_embedding(self, txt_tokens, midi_embedding, midi_dur_embedding, slur_embedding):
        x = self.embed_scale * self.embed_tokens(txt_tokens)
        x = x + midi_embedding + midi_dur_embedding + slur_embedding
        if hparams['use_pos_embed']:
            if hparams.get('rel_pos'):
                x = self.embed_positions(x)
            else:
                x = x + self.embed_positions(txt_tokens)
        return _drop(x, hparams['dropout'], self.training)                           # diffsinger_midi/fs2.py:28

    def forward(self, txt_tokens, midi_embedding, midi_dur_embedding, slur_embedding):
        adds = (midi_embedding, midi_dur_embedding, slur_embedding)
        if all(torch.is_tensor(a) and a.dim() == 3 and a.is_contiguous() and a.dtype == torch.float32 for a in adds if torch.is_tensor(a)):
            fast = self._embed_cm(txt_tokens, tuple(a for a in adds if torch.is_tensor(a)))       # (scale * E + midi) + dur) + slur, in this order
            if fast is not None:
                return from_cm(fast[0], fast[1])
        x = self.forward_embedding(txt_tokens, midi_embedding, midi_dur_embedding, slur_embedding)
        return FFTBlocks.forward(self, x, txt_tokens.eq(self.padding_idx))


class FastspeechDecoder(FFTBlocks):
    def __init__(self, hidden_size=None, num_layers=None, kernel_size=None, num_heads=None):
        super().__init__(hparams['hidden_size'] if hidden_size is None else hidden_size,
                         hparams['dec_layers'] if num_layers is None else num_layers,
                         hparams['dec_ffn_kernel_size'] if kernel_size is None else kernel_size,
                         num_heads=hparams['num_heads'] if num_heads is None else num_heads)


class _PredLayerNorm(nn.LayerNorm):
    """tts_modules.py:39-56: LayerNorm(nout, dim=1), eps 1e-12 (container; the arithmetic is k_fs_ln)."""

    def __init__(self, nout, dim=-1):
        super().__init__(nout, eps=1e-12)
        self.dim = dim


def _pred_convs(idim, n_layers, n_chans, kernel_size, dropout_rate, padding):
    if padding != 'SAME':
        raise NotImplementedError('ffn_padding LEFT')
    return nn.ModuleList([nn.Sequential(
        nn.ConstantPad1d(((kernel_size - 1) // 2, (kernel_size - 1) // 2), 0),
        nn.Conv1d(idim if i == 0 else n_chans, n_chans, kernel_size, stride=1, padding=0),
        nn.ReLU(), _PredLayerNorm(n_chans, dim=1), nn.Dropout(dropout_rate)) for i in range(n_layers)])


def _run_pred_convs(convs, packs, xc, T, keep):
    for seq, pk in zip(convs, packs):
        conv, ln = seq[1], seq[3]
        y = conv1d_cm(xc, T, conv.weight, pk, conv.bias)            # ReLU is fused into the LayerNorm kernel's load
        xc = layer_norm_cm(y, T, ln.weight, ln.bias, 1e-12, relu_in=True, keep=keep)
        xc = seq[4](xc)                                             # nn.Dropout(predictor_dropout): identity in eval mode (tts_modules.py:94, :216)
    return xc


class DurationPredictor(nn.Module):
    def __init__(self, idim, n_layers=2, n_chans=384, kernel_size=3, dropout_rate=0.1, offset=1.0, padding='SAME'):
        super().__init__()
        if hparams['dur_loss'] != 'mse':
            raise NotImplementedError(f"dur_loss {hparams['dur_loss']}")
        self.offset, self.kernel_size = offset, kernel_size
        self.conv = _pred_convs(idim, n_layers, n_chans, kernel_size, dropout_rate, padding)
        self.linear = nn.Linear(n_chans, 1)
        self._packs = [PackedWeight() for _ in range(n_layers)]
        self._plin = PackedWeight()

    def _forward(self, xs, x_masks, is_inference, keep=None):
        """tts_modules.py:107-120.  xs [B,T,idim]; x_masks [B,T] bool (True = pad); keep: (~x_masks).float() if the caller has it (dsf_token_masks)."""
        T = xs.shape[1]
        keep = (~x_masks).float().contiguous() if keep is None else keep
        xc = _run_pred_convs(self.conv, self._packs, to_cm(xs), T, keep)
        y = from_cm(conv1d_cm(xc, T, self.linear.weight, self._plin, self.linear.bias, keep=keep), T)        # [B,T,1]
        if is_inference:
            dur = torch.clamp(torch.round(y.squeeze(-1).exp() - self.offset), min=0).long()                 # out2dur :122-131
            return dur, y
        return y.squeeze(-1)

    def forward(self, xs, x_masks=None, keep=None):
        return self._forward(xs, x_masks, False, keep)

    def inference(self, xs, x_masks=None, keep=None):
        return self._forward(xs, x_masks, True, keep)


class LengthRegulator(nn.Module):
    def forward(self, dur, dur_padding=None, alpha=1.0):
        """tts_modules.py:158-186 (index arithmetic only)."""
        dur = torch.round(dur.float() * alpha).long()
        if dur_padding is not None:
            dur = dur * (1 - dur_padding.long())
        token_idx = torch.arange(1, dur.shape[1] + 1)[None, :, None].to(dur.device)
        dur_cumsum = torch.cumsum(dur, 1)
        dur_cumsum_prev = F.pad(dur_cumsum, [1, -1], mode='constant', value=0)
        pos_idx = torch.arange(int(dur.sum(-1).max()))[None, None].to(dur.device)
        token_mask = (pos_idx >= dur_cumsum_prev[:, :, None]) & (pos_idx < dur_cumsum[:, :, None])
        return (token_idx * token_mask.long()).sum(1)


class PitchPredictor(nn.Module):
    def __init__(self, idim, n_layers=5, n_chans=384, odim=2, kernel_size=5, dropout_rate=0.1, padding='SAME'):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv = _pred_convs(idim, n_layers, n_chans, kernel_size, dropout_rate, padding)
        self.linear = nn.Linear(n_chans, odim)
        self.embed_positions = SinusoidalPositionalEmbedding(idim, 0, init_size=4096)
        self.pos_embed_alpha = nn.Parameter(torch.Tensor([1]))
        self._packs = [PackedWeight() for _ in range(n_layers)]
        self._plin = PackedWeight()

    def forward(self, xs):
        """tts_modules.py:215-229.  xs [B,T,idim] -> [B,T,odim]."""
        T = xs.shape[1]
        if _glue_ok(xs) and xs.dtype == torch.float32 and xs.shape[2] % 4 == 0:
            xs = xs.contiguous()                                   # positions + positional embedding + transposition: two launches
            xc0, _, _ = input_cm_op(x=xs, pos=positions_op(x=xs, padding_idx=self.embed_positions.padding_idx), pos_table=self.embed_positions.table(T),
                                    alpha=self.pos_embed_alpha, mask=False)
        else:
            xs = xs + self.pos_embed_alpha * self.embed_positions(xs[..., 0])
            xc0 = to_cm(xs)
        xc = _run_pred_convs(self.conv, self._packs, xc0, T, None)
        return from_cm(conv1d_cm(xc, T, self.linear.weight, self._plin, self.linear.bias), T)


def _hip_mlp(x, layers):
    """A chain of _HipLinear layers on [B,C] / [B,T,C] input, channel-major in between: one layout change in and one out instead of a pair per
    layer (the values of layer(layer(...)): the layout changes are each other's inverses)."""
    x3 = x if x.dim() == 3 else x[:, None, :]
    T = x3.shape[1]
    xc = to_cm(x3)
    for lin, act in layers:
        xc = conv1d_cm(xc, T, lin.weight, lin._pack, lin.bias, act=act)
    y = from_cm(xc, T)
    return y if x.dim() == 3 else y[:, 0, :]


class _HipLinear(nn.Linear):
    """nn.Linear whose forward on [B,T,C] / [B,C] inputs runs through the conv kernel (K = 1)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._pack = PackedWeight()

    def forward(self, x, act='none'):
        x3 = x if x.dim() == 3 else x[:, None, :]
        T = x3.shape[1]
        y = from_cm(conv1d_cm(to_cm(x3), T, self.weight, self._pack, self.bias, act=act), T)
        return y if x.dim() == 3 else y[:, 0, :]


def denorm_f0(f0, uv, hp, pitch_padding=None):
    if hp['pitch_norm'] == 'standard':                                      # utils/pitch_utils.py:64-77
        f0 = f0 * hp['f0_std'] + hp['f0_mean']
    if hp['pitch_norm'] == 'log':
        f0 = 2 ** f0
    if uv is not None and hp['use_uv']:
        f0[uv > 0] = 0
    if pitch_padding is not None:
        f0[pitch_padding] = 0
    return f0


def norm_f0(f0, uv, hp):
    if hp['pitch_norm'] == 'standard':                                      # utils/pitch_utils.py:32-40
        f0 = (f0 - hp['f0_mean']) / hp['f0_std']
    if hp['pitch_norm'] == 'log':
        f0 = torch.log2(f0)
    if uv is not None and hp['use_uv']:
        f0[uv > 0] = 0
    return f0


def f0_to_coarse(f0):
    f0_mel = 1127 * (1 + f0 / 700).log()                                    # utils/pitch_utils.py:21-30
    # the reference's boolean-mask gather / scatter (`f0_mel[f0_mel > 0] = ...`) sizes a temporary from the mask - a device -> host
    # synchronisation; torch.where evaluates the same expression element-wise (bit-identical) without one, so the forward is capturable
    f0_mel = torch.where(f0_mel > 0, (f0_mel - f0_mel_min) * (f0_bin - 2) / (f0_mel_max - f0_mel_min) + 1, f0_mel)
    f0_mel = torch.where(f0_mel <= 1, torch.ones_like(f0_mel), f0_mel)
    f0_mel = torch.where(f0_mel > f0_bin - 1, torch.full_like(f0_mel, f0_bin - 1), f0_mel)
    return (f0_mel + 0.5).long()


class FastSpeech2(nn.Module):
    def __init__(self, dictionary, out_dims=None):
        super().__init__()
        if hparams.get('pitch_ar'):
            # fs2.py:215 passes two arguments to PitchPredictor.forward (tts_modules.py:222 takes one): the option raises in the reference itself
            raise NotImplementedError("hparams['pitch_ar'] is a dead option of the reference (TypeError at modules/fastspeech/fs2.py:215)")
        if hparams['encoder_type'] != 'fft' or hparams['decoder_type'] != 'fft':
            raise NotImplementedError('encoder_type / decoder_type other than fft')
        self.dictionary = dictionary
        n_vocab = dictionary if isinstance(dictionary, int) else len(dictionary)
        self.padding_idx = 0 if isinstance(dictionary, int) else dictionary.pad()
        self.enc_layers, self.dec_layers, self.hidden_size = hparams['enc_layers'], hparams['dec_layers'], hparams['hidden_size']
        H = self.hidden_size
        self.encoder_embed_tokens = Embedding(n_vocab, H, self.padding_idx)
        self.encoder = self._build_encoder()
        self.decoder = FastspeechDecoder(H, hparams['dec_layers'], hparams['dec_ffn_kernel_size'], hparams['num_heads'])
        self.out_dims = hparams['audio_num_mel_bins'] if out_dims is None else out_dims
        self.mel_out = Linear(H, self.out_dims, bias=True)
        self._pmel = PackedWeight()
        if hparams.get('use_spk_id'):                                   # fs2.py:37-44
            self.spk_embed_proj = Embedding(hparams['num_spk'] + 1, H)
            if hparams.get('use_split_spk_id'):
                self.spk_embed_f0 = Embedding(hparams['num_spk'] + 1, H)
                self.spk_embed_dur = Embedding(hparams['num_spk'] + 1, H)
        elif hparams.get('use_spk_embed'):
            self.spk_embed_proj = _HipLinear(256, H, bias=True)
        ph = hparams['predictor_hidden'] if hparams['predictor_hidden'] > 0 else H
        self.dur_predictor = DurationPredictor(H, n_chans=ph, n_layers=hparams['dur_predictor_layers'],
                                               dropout_rate=hparams['predictor_dropout'], padding=hparams['ffn_padding'],
                                               kernel_size=hparams['dur_predictor_kernel'])
        self.length_regulator = LengthRegulator()
        if hparams['use_pitch_embed']:
            self.pitch_embed = Embedding(300, H, self.padding_idx)
            if hparams['pitch_type'] == 'cwt':
                h = hparams['cwt_hidden_size']
                odim = 10 + (1 if hparams['use_uv'] else 0)
                self.cwt_predictor = nn.Sequential(
                    _HipLinear(H, h),
                    PitchPredictor(h, n_chans=ph, n_layers=hparams['predictor_layers'], dropout_rate=hparams['predictor_dropout'],
                                   odim=odim, padding=hparams['ffn_padding'], kernel_size=hparams['predictor_kernel']))
                self.cwt_stats_layers = nn.Sequential(_HipLinear(H, h), nn.ReLU(), _HipLinear(h, h), nn.ReLU(), _HipLinear(h, 2))
            else:                                                       # 'frame': (f0, uv) per frame; 'ph': f0 per phone (fs2.py:67-74)
                self.pitch_predictor = PitchPredictor(H, n_chans=ph, n_layers=hparams['predictor_layers'],
                                                      dropout_rate=hparams['predictor_dropout'],
                                                      odim=2 if hparams['pitch_type'] == 'frame' else 1,
                                                      padding=hparams['ffn_padding'], kernel_size=hparams['predictor_kernel'])
        if hparams.get('use_energy_embed'):                             # fs2.py:75-82; EnergyPredictor is PitchPredictor (tts_modules.py:253-254)
            self.energy_embed = Embedding(256, H, self.padding_idx)
            self.energy_predictor = PitchPredictor(H, n_chans=ph, n_layers=hparams['predictor_layers'],
                                                   dropout_rate=hparams['predictor_dropout'], odim=1,
                                                   padding=hparams['ffn_padding'], kernel_size=hparams['predictor_kernel'])

    def _build_encoder(self):
        return FastspeechEncoder(self.encoder_embed_tokens, self.hidden_size, hparams['enc_layers'], hparams['enc_ffn_kernel_size'],
                                 num_heads=hparams['num_heads'])

    # -- forward (fs2.py:93-149) -------------------------------------------------------------------------------------
    def _encode(self, txt_tokens, **kwargs):
        return self.encoder(txt_tokens)

    def forward(self, txt_tokens, mel2ph=None, spk_embed=None, ref_mels=None, f0=None, uv=None, energy=None, skip_decoder=False,
                spk_embed_dur_id=None, spk_embed_f0_id=None, infer=False, **kwargs):
        """fs2.py:93-149.  infer=True (or torch.no_grad()): the inference kernels with their fused epilogues.  infer=False with autograd on:
        the same forward as an autograd graph whose backward runs on HIP kernels too (the operator layer above) - what DiffSingerTask /
        DiffSingerMIDITask train when `fs2_ckpt` is empty (usr/diffsinger_task.py:60-64, :273-300); the gradient scaling of the predictor
        inputs (`predictor_grad`, fs2.py:153, :194) is part of the graph."""
        if infer or not torch.is_grad_enabled():
            with torch.no_grad():
                return self._forward(txt_tokens, mel2ph, f0, uv, skip_decoder, infer, spk_embed=spk_embed, energy=energy,
                                     spk_embed_dur_id=spk_embed_dur_id, spk_embed_f0_id=spk_embed_f0_id, **kwargs)
        return self._forward(txt_tokens, mel2ph, f0, uv, skip_decoder, infer, spk_embed=spk_embed, energy=energy,
                             spk_embed_dur_id=spk_embed_dur_id, spk_embed_f0_id=spk_embed_f0_id, **kwargs)

    def _speaker(self, spk_embed, spk_embed_dur_id, spk_embed_f0_id):
        """fs2.py:107-121 -> (spk_embed_dur, spk_embed_f0, spk_embed), each [B,1,H] or 0."""
        if hparams.get('use_spk_embed'):
            e = self.spk_embed_proj(spk_embed)[:, None, :]
            return e, e, e
        if hparams.get('use_spk_id'):
            sid = spk_embed
            e = self.spk_embed_proj(sid)[:, None, :]
            if not hparams.get('use_split_spk_id'):
                return e, e, e
            return (self.spk_embed_dur(sid if spk_embed_dur_id is None else spk_embed_dur_id)[:, None, :],
                    self.spk_embed_f0(sid if spk_embed_f0_id is None else spk_embed_f0_id)[:, None, :], e)
        return 0, 0, 0

    def _forward(self, txt_tokens, mel2ph, f0, uv, skip_decoder, infer, spk_embed=None, energy=None, spk_embed_dur_id=None,
                 spk_embed_f0_id=None, **kwargs):
        ret = {}
        encoder_out = self._encode(txt_tokens, **kwargs)                                        # [B,T_txt,H]
        spk_dur, spk_f0, spk = self._speaker(spk_embed, spk_embed_dur_id, spk_embed_f0_id)
        glue = _glue_ok(encoder_out, txt_tokens) and txt_tokens.dtype == torch.int64 and encoder_out.dtype == torch.float32 and encoder_out.shape[-1] % 4 == 0
        if glue:
            # fs2.py:98-101 / :157 / tts_modules.py:109: the three masks of the phone axis in one launch, `(encoder_out + spk) * nonpadding` in one
            # (dsf_sum_embed without tables computes exactly that) - seven torch launches before
            enc_c, tok_c = encoder_out.contiguous(), txt_tokens.contiguous()
            src_np, src_padding, src_keep = token_masks_op(tok_c, gt0=True, eq0=True, ne0=True)
            src_nonpadding = src_np[:, :, None]
            masked_enc = lambda s_: sum_embed_op(enc_c, tok_c, spk=s_ if torch.is_tensor(s_) else None)
            dur_inp = masked_enc(spk_dur)
            mel2ph = self.add_dur(dur_inp, mel2ph, txt_tokens, ret, src_padding=src_padding, keep=src_keep)
        else:
            src_nonpadding = (txt_tokens > 0).float()[:, :, None]
            dur_inp = (encoder_out + spk_dur) * src_nonpadding
            mel2ph = self.add_dur(dur_inp, mel2ph, txt_tokens, ret)
        if glue and _glue_ok(mel2ph) and mel2ph.dtype == torch.int64:
            # fs2.py:128-141 as two launches: the length regulator's gather (+ the predictors' masked input), then every embedding, the speaker
            # embedding and the mask in one pass over [B,T,H] (the pad / repeat / gather / add / mul sequence of the reference moved ~100 MB)
            m2p = mel2ph.contiguous()
            tgt_np, tgt_padding, _ = token_masks_op(m2p, gt0=True, eq0=True)
            tgt_nonpadding = tgt_np[:, :, None]
            decoder_inp, pitch_inp = gather_frames_op(enc_c, m2p, spk_f0)
            idx1 = tab1 = idx2 = tab2 = None
            if hparams['use_pitch_embed']:
                # the phone-rate input of the 'ph' / 'cwt' pitch paths (fs2.py:146); with no speaker embedding it IS the duration predictor's input
                enc_f0 = None
                if hparams['pitch_type'] in ('ph', 'cwt'):
                    enc_f0 = dur_inp if (not torch.is_tensor(spk_f0) and not torch.is_tensor(spk_dur)) else masked_enc(spk_f0)
                idx1 = self.add_pitch(pitch_inp, f0, uv, m2p, ret, encoder_out=enc_f0, want_index=True, pitch_padding=tgt_padding).contiguous()
                tab1 = self.pitch_embed.weight
            if hparams.get('use_energy_embed'):
                idx2, tab2 = self.add_energy(pitch_inp, energy, ret, want_index=True).contiguous(), self.energy_embed.weight
            ret['decoder_inp'] = decoder_inp = sum_embed_op(decoder_inp, m2p, idx1=idx1, tab1=tab1, idx2=idx2, tab2=tab2, spk=spk)
            if skip_decoder:
                return ret
            ret['mel_out'] = self.run_decoder(decoder_inp, tgt_nonpadding, ret, infer=infer, **kwargs)
            return ret
        tgt_nonpadding = (mel2ph > 0).float()[:, :, None]
        decoder_inp = F.pad(encoder_out, [0, 0, 1, 0])
        decoder_inp = torch.gather(decoder_inp, 1, mel2ph[..., None].repeat([1, 1, encoder_out.shape[-1]]))
        pitch_inp = (decoder_inp + spk_f0) * tgt_nonpadding
        if hparams['use_pitch_embed']:
            decoder_inp = decoder_inp + self.add_pitch(pitch_inp, f0, uv, mel2ph, ret, encoder_out=(encoder_out + spk_f0) * src_nonpadding)
        if hparams.get('use_energy_embed'):
            decoder_inp = decoder_inp + self.add_energy(pitch_inp, energy, ret)
        ret['decoder_inp'] = decoder_inp = (decoder_inp + spk) * tgt_nonpadding
        if skip_decoder:
            return ret
        ret['mel_out'] = self.run_decoder(decoder_inp, tgt_nonpadding, ret, infer=infer, **kwargs)
        return ret

    @staticmethod
    def _scale_grad(x):
        """x.detach() + predictor_grad * (x - x.detach()): the value of x, predictor_grad times its gradient (fs2.py:153, :194)."""
        if not (torch.is_grad_enabled() and x.requires_grad):
            return x
        return x.detach() + hparams['predictor_grad'] * (x - x.detach())

    def add_dur(self, dur_input, mel2ph, txt_tokens, ret, src_padding=None, keep=None):
        """fs2.py:151-172.  src_padding / keep: `txt_tokens == 0` and its float complement if the caller has them (dsf_token_masks)."""
        src_padding = (txt_tokens == 0) if src_padding is None else src_padding
        dur_input = self._scale_grad(dur_input)
        if mel2ph is None:
            dur, xs = self.dur_predictor.inference(dur_input, src_padding, keep=keep)
            ret['dur'], ret['dur_choice'] = xs, dur
            mel2ph = self.length_regulator(dur, src_padding).detach()
        else:
            ret['dur'] = self.dur_predictor(dur_input, src_padding, keep=keep)
        ret['mel2ph'] = mel2ph
        return mel2ph

    def add_energy(self, decoder_inp, energy, ret, want_index=False):
        """fs2.py:174-181.  want_index: the rows of energy_embed instead of the embedding (the fused glue of _forward looks them up itself)."""
        decoder_inp = self._scale_grad(decoder_inp)
        ret['energy_pred'] = energy_pred = self.energy_predictor(decoder_inp)[:, :, 0]
        if energy is None:
            energy = energy_pred
        energy = torch.clamp(energy * 256 // 4, max=255).long()
        return energy if want_index else self.energy_embed(energy)

    def add_pitch(self, decoder_inp, f0, uv, mel2ph, ret, encoder_out=None, want_index=False, pitch_padding=None):
        """fs2.py:183-231.  want_index: the rows of pitch_embed instead of the embedding; pitch_padding: `mel2ph == 0` if the caller has it."""
        if hparams['pitch_type'] == 'ph':                                # :184-196: predicted and quantised per phone, gathered to the frames
            pitch_pred_inp = self._scale_grad(encoder_out)
            pitch_padding = encoder_out.sum().abs() == 0
            ret['pitch_pred'] = pitch_pred = self.pitch_predictor(pitch_pred_inp)
            if f0 is None:
                f0 = pitch_pred[:, :, 0]
            ret['f0_denorm'] = f0_denorm = denorm_f0(f0, None, hparams, pitch_padding=pitch_padding)
            pitch = F.pad(f0_to_coarse(f0_denorm), [1, 0])
            idx = torch.gather(pitch, 1, mel2ph)
            return idx if want_index else self.pitch_embed(idx)
        decoder_inp = self._scale_grad(decoder_inp)
        pitch_padding = (mel2ph == 0) if pitch_padding is None else pitch_padding
        have_padding = True
        given = f0 is not None
        if hparams['pitch_type'] == 'cwt':
            pitch_padding, have_padding = None, False
            ret['cwt'] = cwt_out = self.cwt_predictor[1](self.cwt_predictor[0](decoder_inp))
            s = encoder_out[:, 0, :]
            st = self.cwt_stats_layers
            stats_out = _hip_mlp(s, ((st[0], 'relu'), (st[2], 'relu'), (st[4], 'none')))
            mean = ret['f0_mean'] = stats_out[:, 0]
            std = ret['f0_std'] = stats_out[:, 1]
            if f0 is None:
                std = std * hparams['cwt_std_scale']
                f0 = self.cwt2f0_norm(cwt_out[:, :, :10], mean, std, mel2ph)
                if hparams['use_uv']:
                    uv = cwt_out[:, :, -1] > 0
        else:
            ret['pitch_pred'] = pitch_pred = self.pitch_predictor(decoder_inp)
            if f0 is None:
                f0 = pitch_pred[:, :, 0]
            if hparams['use_uv'] and uv is None:
                uv = pitch_pred[:, :, 1] > 0
        if _pitch_fusable(f0, uv, hparams) and _glue_ok(mel2ph):
            # denorm_f0 + f0_to_coarse, 22 elementwise launches of the reference's op sequence, as one (dsf_pitch_coarse): the same values
            ret['f0_denorm'], pitch = pitch_coarse_op(f0, uv, mel2ph if have_padding else None, hparams)
        else:
            ret['f0_denorm'] = f0_denorm = denorm_f0(f0, uv, hparams, pitch_padding=pitch_padding)
            pitch = f0_to_coarse(f0_denorm)
        if pitch_padding is not None and not given:
            f0[pitch_padding] = 0           # the reference's in-place edit of the pitch_pred view (:225-226)
        return pitch if want_index else self.pitch_embed(pitch)

    def run_decoder(self, decoder_inp, tgt_nonpadding, ret, infer, **kwargs):
        xc, T, keep = self.decoder.forward_cm(decoder_inp)                                      # fs2.py:233-237
        return from_cm(conv1d_cm(xc, T, self.mel_out.weight, self._pmel, self.mel_out.bias, keep=tgt_nonpadding[:, :, 0].contiguous()), T)

    def cwt2f0_norm(self, cwt_spec, mean, std, mel2ph):
        b = (torch.arange(0, 10, device=cwt_spec.device).float()[None, None, :] + 1 + 2.5) ** (-2.5)     # utils/cwt.py:118-125
        rec = (cwt_spec * b).sum(-1)
        rec = (rec - rec.mean(-1, keepdim=True)) / rec.std(-1, keepdim=True)
        f0 = (rec * std[:, None] + mean[:, None]).exp()
        f0 = torch.cat([f0] + [f0[:, -1:]] * (mel2ph.shape[1] - f0.shape[1]), 1)
        return norm_f0(f0, None, hparams)

    def out2mel(self, out):
        return out

    @staticmethod
    def mel_norm(x):
        return (x + 5.5) / (6.3 / 2) - 1

    @staticmethod
    def mel_denorm(x):
        return (x + 1) * (6.3 / 2) - 5.5


class FastSpeech2MIDI(FastSpeech2):
    def __init__(self, dictionary, out_dims=None):
        super().__init__(dictionary, out_dims)
        self.midi_embed = Embedding(300, self.hidden_size, self.padding_idx)
        self.midi_dur_layer = Linear(1, self.hidden_size)
        self.is_slur_embed = Embedding(2, self.hidden_size)

    def _build_encoder(self):
        return FastspeechMIDIEncoder(self.encoder_embed_tokens, self.hidden_size, hparams['enc_layers'], hparams['enc_ffn_kernel_size'],
                                     num_heads=hparams['num_heads'])

    def _encode(self, txt_tokens, **kwargs):
        """diffsinger_midi/fs2.py:61-67."""
        midi_embedding = self.midi_embed(kwargs['pitch_midi'])
        midi_dur_embedding, slur_embedding = 0, 0
        if kwargs.get('midi_dur') is not None:
            # Linear(1, H) on [B,T,1]: an outer product (input width 1 is no contraction)
            midi_dur_embedding = kwargs['midi_dur'][:, :, None] * self.midi_dur_layer.weight[:, 0] + self.midi_dur_layer.bias
        if kwargs.get('is_slur') is not None:
            slur_embedding = self.is_slur_embed(kwargs['is_slur'])
        return self.encoder(txt_tokens, midi_embedding, midi_dur_embedding, slur_embedding)
