"""`DiffNet`: the reference's WaveNet-style denoiser (usr/diff/net.py:81-130) as an nn.Module whose forward
runs on the hand-written HIP kernels.

Same constructor (`DiffNet(in_dims)` reading `hparams`), same sub-module / parameter names and shapes (a
reference state_dict loads with strict=True and vice versa), same RNG consumption at construction (so
`torch.manual_seed(s); DiffNet(80)` gives the reference's initial weights), same call signature
`forward(spec [B,1,M,T], diffusion_step [B], cond [B,H,T]) -> [B,1,M,T]`.

Under torch.no_grad() the forward is the fused inference path (engine.py).  With autograd enabled on parameters that require
grad it is the training path of diffsinger_amd/train.py (first slice): the same contractions as
stand-alone HIP operators with hand-written data / weight gradients, element-wise glue through torch autograd."""
from __future__ import annotations

import torch
from torch import nn

from .engine import DenoiserEngine
from .hparams import hparams


class Mish(nn.Module):
    """usr/diff/diffusion.py:68-70 - a parameter-free slot in `mlp` (keeps the state_dict keys mlp.0 / mlp.2).
    Its arithmetic runs inside the step-embedding table kernel, not here."""

    def forward(self, x):
        raise RuntimeError('Mish is evaluated inside the HIP step-table kernel; this module is a placeholder')


class SinusoidalPosEmb(nn.Module):
    """usr/diff/net.py:32-44 - parameter-free; evaluated by k_step_embed on the device."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim


def _conv1d(cin, cout, k):
    layer = nn.Conv1d(cin, cout, k)         # weights only: the convolution itself runs in k_layer
    nn.init.kaiming_normal_(layer.weight)   # net.py:47-50
    return layer


class ResidualBlock(nn.Module):
    """Parameter container for usr/diff/net.py:58-64 (dilated_conv keeps padding/dilation for introspection)."""

    def __init__(self, encoder_hidden, residual_channels, dilation):
        super().__init__()
        self.dilation = dilation
        self.dilated_conv = _conv1d(residual_channels, 2 * residual_channels, 3)
        self.dilated_conv.padding, self.dilated_conv.dilation = (dilation,), (dilation,)
        self.diffusion_projection = nn.Linear(residual_channels, residual_channels)
        self.conditioner_projection = _conv1d(encoder_hidden, 2 * residual_channels, 1)
        self.output_projection = _conv1d(residual_channels, 2 * residual_channels, 1)


class DiffNet(nn.Module):
    def __init__(self, in_dims=80):
        super().__init__()
        self.in_dims = in_dims
        self.encoder_hidden = hparams['hidden_size']
        self.n_layers = hparams['residual_layers']
        self.residual_channels = hparams['residual_channels']
        self.dilation_cycle_length = hparams['dilation_cycle_length']
        C = self.residual_channels
        self.input_projection = _conv1d(in_dims, C, 1)
        self.diffusion_embedding = SinusoidalPosEmb(C)
        self.mlp = nn.Sequential(nn.Linear(C, C * 4), Mish(), nn.Linear(C * 4, C))
        self.residual_layers = nn.ModuleList([
            ResidualBlock(self.encoder_hidden, C, 2 ** (i % self.dilation_cycle_length)) for i in range(self.n_layers)])
        self.skip_projection = _conv1d(C, C, 1)
        self.output_projection = _conv1d(C, in_dims, 1)
        nn.init.zeros_(self.output_projection.weight)       # net.py:105
        self._engine = None
        self._weights_tag = None
        self._cond_tag = None
        self._cond_ref = None

    # -- engine management -----------------------------------------------------------------------------------
    def _tag(self):
        from .train_dist import param_generation                 # raw-pointer optimiser steps do not bump torch's version counters
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + (param_generation(),)

    def engine(self) -> DenoiserEngine:
        """The per-device HIP engine, (re)packing the weights whenever a parameter tensor changed."""
        dev = self.input_projection.weight.device
        if dev.type != 'cuda':
            raise RuntimeError('DiffNet (HIP) has no CPU path: move the module to the MI355X first (.cuda())')
        if self._engine is None or self._engine.device != dev:
            self._engine = DenoiserEngine(self.in_dims, self.residual_channels, self.encoder_hidden, self.n_layers,
                                          self.dilation_cycle_length, dev)
            self._weights_tag = None
        tag = self._tag()
        if tag != self._weights_tag:
            self._engine.load_weights(self.state_dict())
            self._weights_tag = tag
            self._cond_tag = self._cond_ref = None
        return self._engine

    def bind_cond(self, cond: torch.Tensor) -> DenoiserEngine:
        """Hoist the conditioner projections for `cond` (once per utterance batch, not once per step)."""
        eng = self.engine()
        # The projections are cached per cond TENSOR: same storage address, same version counter, same view.  The
        # tensor is kept referenced while cached - otherwise the caching allocator may hand its address to a
        # different conditioner (same pointer, version 0, same shape) and the stale projections would be reused.
        tag = (cond.data_ptr(), cond._version, tuple(cond.shape), cond.stride())
        if (self._cond_ref is None or tag != self._cond_tag or eng.prepared_shape != (cond.shape[0], cond.shape[2])
                or eng.prepare_serial != getattr(self, '_cond_serial', -1)):      # (someone prepared the engine directly since: bench / tools)
            eng.prepare(cond)
            self._cond_tag = tag
            self._cond_ref = cond
            self._cond_serial = eng.prepare_serial
        return eng

    def fused(self) -> bool:
        """The fused engine (persistent loop, per-layer / latency kernels, include/dsd.h) is built for the width every shipped DiffSpeech /
        DiffSinger config uses: residual_channels == hidden_size == 256, at most 96 mel bins, dilations up to 8.  The reference reads the
        widths from hparams (usr/diff/net.py:85-90): any other width - multiples of 8 - runs on the generic HIP operators instead
        (`train.diffnet_forward_train` without autograd: one launch per Conv1d / element-wise piece) under the generic sampler of
        `GaussianDiffusion`: the same results and API, operator-path speed."""
        return (self.residual_channels == 256 and self.encoder_hidden == 256 and self.in_dims <= 96 and self.dilation_cycle_length <= 4)

    def forward(self, spec, diffusion_step, cond):
        """:param spec: [B, 1, M, T]  :param diffusion_step: [B]  :param cond: [B, H, T]  :return: [B, 1, M, T]"""
        if not self.fused():
            from .train import diffnet_forward_train
            if self.residual_channels % 8 or self.encoder_hidden % 8 or self.in_dims % 8:
                raise NotImplementedError('DiffNet on the HIP operators needs channel counts that are multiples of 8 '
                                          f'(residual_channels={self.residual_channels}, hidden_size={self.encoder_hidden}, mel bins={self.in_dims})')
            return diffnet_forward_train(self, spec, diffusion_step.reshape(-1), cond)
        if torch.is_grad_enabled() and (cond.requires_grad or any(p.requires_grad for p in self.parameters())):
            # training (p_losses, usr/diff/shallow_diffusion_tts.py:213-231): the autograd path on the HIP conv / wgrad operators
            from .train import diffnet_forward_train
            return diffnet_forward_train(self, spec, diffusion_step.reshape(-1), cond)
        eng = self.bind_cond(cond)
        t = diffusion_step.reshape(-1)
        eps = eng.denoise(spec, t)
        return eps[:, None, :, :]
