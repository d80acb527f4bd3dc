"""The FUSED training step of the DiffNet residual stack: the 20 ResidualBlocks of DiffNet.forward
(usr/diff/net.py:66-78, :121-126) as ONE autograd node whose forward and backward are the kernels of csrc/train_kernels.hpp behind
`dsf_stack_forward` / `dsf_stack_backward` (include/dsf.h) - what torch autograd runs for `p_losses`
(usr/diff/shallow_diffusion_tts.py:213-231) between the input projection and the skip projection.

Forward: one conditioner-projection launch for all layers + the inference layer kernel per layer (it additionally saves y = x + step and
the gate pre-activation).  Backward per layer: output-projection data gradient + gate derivative, transposed dilated convolution +
residual path, and one launch for the layer's three weight gradients (split-K over frames, fixed-order reduction: deterministic).
The operator-by-operator path of train.py stays as the fallback for shapes this build does not cover (channels != 256) and as the
comparison (`DSD_TRAIN_FUSED=0`)."""
from __future__ import annotations

import ctypes as C
import os
from typing import List

import torch

from . import _lib
from .fs2 import PackedWeight, padded_frames


class DsfStackWeights(C.Structure):
    _fields_ = [('dilated_conv_w', C.POINTER(C.c_void_p)), ('dilated_conv_b', C.POINTER(C.c_void_p)),
                ('cond_w', C.POINTER(C.c_void_p)), ('cond_b', C.POINTER(C.c_void_p)),
                ('out_w', C.POINTER(C.c_void_p)), ('out_b', C.POINTER(C.c_void_p)),
                ('dilations', C.POINTER(C.c_int32))]


class DsfStackGrads(C.Structure):
    _fields_ = [('dilated_conv_w', C.POINTER(C.c_void_p)), ('dilated_conv_b', C.POINTER(C.c_void_p)),
                ('cond_w', C.POINTER(C.c_void_p)), ('cond_b', C.POINTER(C.c_void_p)),
                ('out_w', C.POINTER(C.c_void_p)), ('out_b', C.POINTER(C.c_void_p)),
                ('dx0', C.c_void_p), ('dstep', C.c_void_p)]


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _table(tensors: List[torch.Tensor]):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def enabled() -> bool:
    return os.environ.get('DSD_TRAIN_FUSED', '1') != '0'


def supported(net) -> bool:
    """The fused stack covers the shapes every shipped DiffSpeech / DiffSinger config uses: 256 residual channels, 256 conditioner channels,
    at most 32 layers, dilations up to 8 (dilation_cycle_length <= 4)."""
    layers = net.residual_layers
    if len(layers) < 1 or len(layers) > 32:
        return False
    l0 = layers[0]
    return (tuple(l0.dilated_conv.weight.shape) == (512, 256, 3) and tuple(l0.conditioner_projection.weight.shape) == (512, 256, 1)
            and all(1 <= l.dilation <= 8 for l in layers))


def _bind(lib):
    if getattr(lib, '_stack_bound', False):
        return
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.dsf_stack_workspace_floats.argtypes = [i32, i32, i32, i32]
    lib.dsf_stack_workspace_floats.restype = i64
    lib.dsf_set_stack_mode.argtypes = [i32]
    lib.dsf_set_stack_conv.argtypes = [i32]
    lib.dsf_get_stack_conv.argtypes = []
    lib.dsf_set_wgrad_dual.argtypes = [i32]
    lib.dsf_stack_offsets.argtypes = [i32, i32, i32, i32, C.POINTER(i64), i32]
    lib.dsf_stack_forward.argtypes = [vp, vp, vp, C.POINTER(DsfStackWeights), i32, i32, i32, vp, vp, vp]
    lib.dsf_stack_backward.argtypes = [vp, vp, C.POINTER(DsfStackWeights), i32, i32, i32, vp, vp, C.POINTER(DsfStackGrads), vp, vp]
    lib.dsf_wgrad2_workspace_floats.argtypes = [i32, i32, i32]
    lib.dsf_wgrad2_workspace_floats.restype = i64
    lib.dsf_conv1d_wgrad2.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.dsf_wgrad_probe.argtypes = [i32]
    lib.dsf_wgrad_probe_read.argtypes = [C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    lib._stack_bound = True


def set_stack_mode(mode: int):
    """How the forward of the fused stack is launched (include/dsf.h dsf_set_stack_mode): 1 automatic, 0 one launch per layer, 2 the persistent
    kernel wherever an utterance fits the grid."""
    lib = _lib.load()
    _bind(lib)
    _lib.check(lib.dsf_set_stack_mode(int(mode)), 'dsf_set_stack_mode')


def set_stack_conv(mode):
    """The dilated convolution of the persistent forward (include/dsf.h dsf_set_stack_conv): 'wino' / 1 Winograd F(2,3) (default),
    'direct' / 0 the direct form (bit-identical to the per-layer launches)."""
    lib = _lib.load()
    _bind(lib)
    m = {'wino': 1, 'direct': 0}.get(mode, mode)
    _lib.check(lib.dsf_set_stack_conv(int(m)), 'dsf_set_stack_conv')


def set_wgrad_dual(on: bool):
    """The dilated convolution's weight gradient as the Winograd F(2,3) dual over frame pairs (include/dsf.h dsf_set_wgrad_dual; default on where
    the stack's convolution runs as Winograd) or as the three tap products of rounds 2-5 (the A/B switch of tests and tools)."""
    lib = _lib.load()
    _bind(lib)
    _lib.check(lib.dsf_set_wgrad_dual(int(bool(on))), 'dsf_set_wgrad_dual')


def stack_conv() -> str:
    lib = _lib.load()
    _bind(lib)
    return 'wino' if lib.dsf_get_stack_conv() == 1 else 'direct'


def _weights_struct(ws: List[torch.Tensor], L: int, dils: List[int]):
    """ws = [dilated_conv.weight] * L + [dilated_conv.bias] * L + [cond w] * L + [cond b] * L + [out w] * L + [out b] * L"""
    tabs = [_table(ws[k * L:(k + 1) * L]) for k in range(6)]
    dil = (C.c_int32 * L)(*dils)
    st = DsfStackWeights(*[C.cast(t, C.POINTER(C.c_void_p)) for t in tabs], C.cast(dil, C.POINTER(C.c_int32)))
    return st, (tabs, dil)          # keep the ctypes arrays alive as long as the struct


class _ResidualStack(torch.autograd.Function):
    """skip_sum [B][256][TS] = sum_l skip_l of the residual stack run on x0 [B][256][TS] with cond [B][256][TS], step [B][L][256]."""

    @staticmethod
    def forward(ctx, x0, cond, step, T, dils, cond_pack, *weights):
        lib = _lib.load()
        _bind(lib)
        L = len(dils)
        B, _, TS = x0.shape
        dev = x0.device
        x0, cond, step = x0.contiguous(), cond.contiguous(), step.contiguous()
        ws = [w.detach().contiguous() for w in weights]
        n = lib.dsf_stack_workspace_floats(B, T, L, 0)
        save = torch.empty(n, device=dev, dtype=torch.float32)
        skip = torch.empty(B, 256, TS, device=dev, dtype=torch.float32)
        st, keep = _weights_struct(ws, L, dils)
        with torch.cuda.device(dev):
            _lib.check(lib.dsf_stack_forward(x0.data_ptr(), cond.data_ptr(), step.data_ptr(), C.byref(st), B, T, L, save.data_ptr(), skip.data_ptr(),
                                             _stream(dev)), 'dsf_stack_forward')
        ctx.save_for_backward(cond, save, *ws)
        ctx.T, ctx.dils, ctx.cond_pack = T, dils, cond_pack
        return skip

    @staticmethod
    def backward(ctx, dskip):
        lib = _lib.load()
        cond, save, *ws = ctx.saved_tensors
        T, dils = ctx.T, ctx.dils
        L = len(dils)
        B, _, TS = cond.shape
        dev = cond.device
        dskip = dskip.contiguous()
        bws = torch.empty(lib.dsf_stack_workspace_floats(B, T, L, 1), device=dev, dtype=torch.float32)
        dx0 = torch.empty(B, 256, TS, device=dev, dtype=torch.float32)
        dstep = torch.empty(B, L, 256, device=dev, dtype=torch.float32)
        grads = [torch.empty_like(w) for w in ws]
        want_dcond = ctx.needs_input_grad[1]
        da_all = torch.empty(B, L * 512, TS, device=dev, dtype=torch.float32) if want_dcond else None
        st, keep = _weights_struct(ws, L, dils)
        gt = [_table(grads[k * L:(k + 1) * L]) for k in range(6)]
        gs = DsfStackGrads(*[C.cast(t, C.POINTER(C.c_void_p)) for t in gt], dx0.data_ptr(), dstep.data_ptr())
        with torch.cuda.device(dev):
            _lib.check(lib.dsf_stack_backward(dskip.data_ptr(), cond.data_ptr(), C.byref(st), B, T, L, save.data_ptr(), bws.data_ptr(), C.byref(gs),
                                              da_all.data_ptr() if want_dcond else None, _stream(dev)), 'dsf_stack_backward')
            dcond = None
            if want_dcond:
                # dcond = sum_l Wc_l^T da_l: ONE 1x1 convolution over the 512 L stacked channels (usr/diff/net.py:68 backward, all layers)
                # the packed Wc^T is keyed on the SOURCE parameters (address + version of every conditioner weight + the raw-pointer
                # optimiser's generation), never on the temporary: a fresh cat() has version 0 and usually the same address every step
                from .train_dist import param_generation
                cw = ws[2 * L:3 * L]
                tag = (tuple((w.data_ptr(), w._version) for w in cw), param_generation())
                pack = ctx.cond_pack
                if pack.get('tag') != tag:
                    pack['wcat'] = torch.cat([w.reshape(512, 256) for w in cw], 0).t().contiguous()          # [256][512 L], kept alive
                    pack['packed'] = PackedWeight()
                    pack['tag'] = tag
                wp = pack['packed'].get(pack['wcat'])
                dcond = torch.empty(B, 256, TS, device=dev, dtype=torch.float32)
                _lib.check(lib.dsf_conv1d(da_all.data_ptr(), wp.data_ptr(), None, dcond.data_ptr(), B, L * 512, 256, 1, T, 1.0, 0, None, None,
                                          _stream(dev)), 'dsf_conv1d (dcond)')
        return (dx0, dcond, dstep, None, None, None, *grads)


def residual_stack(net, x0: torch.Tensor, cond_cm: torch.Tensor, step_all: torch.Tensor, T: int) -> torch.Tensor:
    """x0 [B][256][TS], cond_cm [B][256][TS] (zero tails), step_all [B][L][256] -> the sum of the layers' skip outputs [B][256][TS]."""
    layers = list(net.residual_layers)
    ws = ([l.dilated_conv.weight for l in layers] + [l.dilated_conv.bias for l in layers] +
          [l.conditioner_projection.weight for l in layers] + [l.conditioner_projection.bias for l in layers] +
          [l.output_projection.weight for l in layers] + [l.output_projection.bias for l in layers])
    pack = net.__dict__.setdefault('_train_cond_pack', {})
    return _ResidualStack.apply(x0, cond_cm, step_all, T, [int(l.dilation) for l in layers], pack, *ws)


def conv1d_wgrad2(dy: torch.Tensor, x: torch.Tensor, K: int, dil: int, T: int, want_bias: bool = True):
    """The stack's weight-gradient kernel as a stand-alone operator (tests; dsf_conv1d_wgrad2): dy [B][Co][TS], x [B][Ci][TS] -> dw [Co][Ci][K], db."""
    lib = _lib.load()
    _bind(lib)
    B, Co, TS = dy.shape
    Ci = x.shape[1]
    n = lib.dsf_wgrad2_workspace_floats(Co, Ci, K)
    if n < 0:
        raise ValueError(f'dsf_conv1d_wgrad2: unsupported shape Co={Co} Ci={Ci} K={K}')
    ws = torch.empty(n, device=dy.device, dtype=torch.float32)
    dw = torch.empty(Co, Ci, K, device=dy.device, dtype=torch.float32)
    db = torch.empty(Co, device=dy.device, dtype=torch.float32) if want_bias else None
    with torch.cuda.device(dy.device):
        _lib.check(lib.dsf_conv1d_wgrad2(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if want_bias else None, ws.data_ptr(), B, Ci, Co, K, dil,
                                         T, _stream(dy.device)), 'dsf_conv1d_wgrad2')
    return dw, db


def wgrad_probe(on: bool):
    """Bracket every weight-gradient launch of the fused stack with events (bench.py --row train)."""
    lib = _lib.load()
    _bind(lib)
    _lib.check(lib.dsf_wgrad_probe(1 if on else 0), 'dsf_wgrad_probe')


def wgrad_probe_read():
    """-> (summed kernel time in ms, launches, algorithmic FLOP) of the launches since the probe was switched on / last read."""
    lib = _lib.load()
    _bind(lib)
    ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
    _lib.check(lib.dsf_wgrad_probe_read(C.byref(ms), C.byref(n), C.byref(fl)), 'dsf_wgrad_probe_read')
    return ms.value, n.value, fl.value
