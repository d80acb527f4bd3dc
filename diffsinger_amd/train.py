"""Training path of the DiffNet denoiser (first slice): `p_losses` (usr/diff/shallow_diffusion_tts.py:
213-231) with the forward AND backward contractions on the HIP operators of include/dsf.h.

What is native: every Conv1d / Linear of DiffNet (usr/diff/net.py:91-105) in the forward pass, its data gradient (the same
MFMA convolution kernel with the flipped, transposed weight) and its weight / bias gradients (`dsf_conv1d_wgrad`: a split-K
contraction over frames on fp32 MFMA, reduced in a fixed order -> deterministic) - > 99 % of the FLOPs of a training step.
The element-wise pieces of the residual block (x + step, sigmoid * tanh gate, residual / skip update) are fused HIP kernels
with hand-written backward too (`dsf_train_*`).  What is left to torch autograd: the sum of the two convolution outputs, the
two ReLUs and the 1/sqrt(L) scale of the head, the Mish of the step-embedding MLP (its two Linear layers and the layers' step
projections are `dsf_linear_rows`), the L1 loss, and the optimiser.  DDP works unchanged on top
(gradients are ordinary `.grad` tensors; the all-reduce is torch.distributed's, RCCL on ROCm).

This is the functional slice of row f3, not yet the fused one: activations live channel-major [B][C][TS] (TS = T up to 32, zero
tail) and every operator is its own launch.  Parity: gradients of all 15 M parameters against torch autograd on the CPU oracle
(tests/test_gpu_train.py)."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib
from .fs2 import PackedWeight, padded_frames


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


class _LinearRows(torch.autograd.Function):
    """torch.nn.Linear on [rows, in] with rows = the batch size: the step-embedding MLP and the layers' step projections (usr/diff/net.py:
    94-98, :119-120, :67).  dsf_linear_rows / dsf_linear_rows_bwd: plain row-major products on the vector ALUs - no vendor BLAS on the path."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x, weight = x.contiguous(), weight.contiguous()
        rows, n_in = x.shape
        n_out = weight.shape[0]
        y = torch.empty(rows, n_out, device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.dsf_linear_rows_workspace_floats(rows, n_in, n_out), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(lib.dsf_linear_rows(x.data_ptr(), weight.data_ptr(), bias.contiguous().data_ptr() if bias is not None else None, y.data_ptr(),
                                           ws.data_ptr(), rows, n_in, n_out, _stream(x.device)), 'dsf_linear_rows')
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        rows, n_in = x.shape
        n_out = weight.shape[0]
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        db = torch.empty(n_out, device=x.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        ws = torch.empty(lib.dsf_linear_rows_workspace_floats(rows, n_in, n_out), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(lib.dsf_linear_rows_bwd(x.data_ptr(), weight.data_ptr(), dy.data_ptr(), ptr(dx), ptr(dw), ptr(db), ws.data_ptr(), rows, n_in,
                                               n_out, _stream(x.device)), 'dsf_linear_rows_bwd')
        return dx, dw, db


def linear_rows(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    return _LinearRows.apply(x, weight, bias)


class _Conv1dCM(torch.autograd.Function):
    """y = W * x + b on channel-major tensors [B][C][TS] with the zero-tail invariant; K in {1, 3}, dilation d."""

    @staticmethod
    def forward(ctx, x, weight, bias, T, dil, cache):
        lib = _lib.load()
        w3 = weight if weight.dim() == 3 else weight[:, :, None]
        Co, Ci, K = w3.shape
        B, _, TS = x.shape
        x = x.contiguous()
        out = torch.empty(B, Co, TS, device=x.device, dtype=torch.float32)
        wp = cache['fwd'].get(weight)
        with torch.cuda.device(x.device):
            _lib.check(lib.dsf_conv1d_dilated(x.data_ptr(), wp.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                              B, Ci, Co, K, dil, T, _stream(x.device)), 'dsf_conv1d_dilated')
        ctx.save_for_backward(x, weight, bias if bias is not None else torch.empty(0, device=x.device))
        ctx.T, ctx.dil, ctx.cache, ctx.has_bias = T, dil, cache, bias is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight, bias = ctx.saved_tensors
        T, dil, cache = ctx.T, ctx.dil, ctx.cache
        w3 = weight if weight.dim() == 3 else weight[:, :, None]
        Co, Ci, K = w3.shape
        B, _, TS = x.shape
        dy = dy.contiguous()
        if TS > T and ctx.needs_input_grad[0]:
            dy = dy.clone()
            dy[:, :, T:] = 0                                    # the data-gradient convolution relies on the zero tail (wgrad masks it itself)
        dev = x.device
        dx = dw = db = None
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0]:
                # data gradient = convolution of dy with W'[ci][co][k] = W[co][ci][K-1-k] (same dilation, same padding)
                wt = cache.get('bwd_w')
                from .train_dist import param_generation
                tag = (weight.data_ptr(), weight._version, param_generation())
                Cop = (Co + 7) // 8 * 8                         # the convolution kernel contracts over multiples of 8 channels: a narrow output
                if wt is None or cache.get('bwd_tag') != tag:   # (the duration predictor's Linear(C, 1)) is padded with zero channels
                    wt = w3.detach().flip(2).transpose(0, 1)
                    if Cop != Co:
                        wt = F.pad(wt, (0, 0, 0, Cop - Co))
                    wt = wt.contiguous()
                    cache['bwd_w'], cache['bwd_tag'] = wt, tag
                wtp = cache['bwd'].get(wt)
                dyd = dy if Cop == Co else F.pad(dy, (0, 0, 0, Cop - Co))
                dx = torch.empty(B, Ci, TS, device=dev, dtype=torch.float32)
                _lib.check(lib.dsf_conv1d_dilated(dyd.data_ptr(), wtp.data_ptr(), None, dx.data_ptr(), B, Cop, Ci, K, dil, T, _stream(dev)),
                           'dsf_conv1d_dilated (dgrad)')
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                dw = torch.empty(Co, Ci, K, device=dev, dtype=torch.float32)
                if want_db:
                    db = torch.empty(Co, device=dev, dtype=torch.float32)       # fused into the weight-gradient kernel (row sums of dy)
                ws = cache.get('ws')
                n = lib.dsf_wgrad_workspace_floats(Co, Ci, K)
                if ws is None or ws.numel() < n:
                    ws = cache['ws'] = torch.empty(n, device=dev, dtype=torch.float32)
                _lib.check(lib.dsf_conv1d_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if want_db else None, ws.data_ptr(),
                                                B, Ci, Co, K, dil, T, 0, _stream(dev)), 'dsf_conv1d_wgrad')
                dw = dw.reshape(weight.shape)
            elif want_db:
                db = torch.empty(Co, device=dev, dtype=torch.float32)
                _lib.check(lib.dsf_bias_grad(dy.data_ptr(), db.data_ptr(), B, Co, T, 0, _stream(dev)), 'dsf_bias_grad')
        return dx, dw, db, None, None, None


def _ew(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args), name)


class _AddStep(torch.autograd.Function):
    """y = x + step[b][c] with the zero tail (usr/diff/net.py:69)."""

    @staticmethod
    def forward(ctx, x, step, T):
        x, step = x.contiguous(), step.contiguous()
        B, C, TS = x.shape
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _ew('dsf_train_add_step', x.data_ptr(), step.data_ptr(), y.data_ptr(), B, C, T, _stream(x.device))
        ctx.T = T
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, C, TS = dy.shape
        dstep = torch.empty(B, C, device=dy.device, dtype=torch.float32)
        with torch.cuda.device(dy.device):
            _ew('dsf_train_rowsum', dy.data_ptr(), dstep.data_ptr(), B * C, ctx.T, _stream(dy.device))
        return dy, dstep, None


class _Gate(torch.autograd.Function):
    """g = sigmoid(a[:, :C]) * tanh(a[:, C:]) (net.py:73-74)."""

    @staticmethod
    def forward(ctx, a, T):
        a = a.contiguous()
        B, C2, TS = a.shape
        g = torch.empty(B, C2 // 2, TS, device=a.device, dtype=torch.float32)
        with torch.cuda.device(a.device):
            _ew('dsf_train_gate', a.data_ptr(), g.data_ptr(), B, C2 // 2, T, _stream(a.device))
        ctx.save_for_backward(a)
        ctx.T = T
        return g

    @staticmethod
    def backward(ctx, dg):
        a, = ctx.saved_tensors
        dg = dg.contiguous()
        B, C2, TS = a.shape
        da = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _ew('dsf_train_gate_bwd', a.data_ptr(), dg.data_ptr(), da.data_ptr(), B, C2 // 2, ctx.T, _stream(a.device))
        return da, None


class _ResSkip(torch.autograd.Function):
    """x' = (x + y[:, :C]) / sqrt(2), skip' = skip + y[:, C:] (net.py:76-78; the skip list of :121-126 summed on the fly)."""

    @staticmethod
    def forward(ctx, x, y, skip, T):
        x, y = x.contiguous(), y.contiguous()
        B, C, TS = x.shape
        xo, so = torch.empty_like(x), torch.empty_like(x)
        sk = skip.contiguous() if skip is not None else None
        with torch.cuda.device(x.device):
            _ew('dsf_train_res_skip', x.data_ptr(), y.data_ptr(), sk.data_ptr() if sk is not None else None, xo.data_ptr(), so.data_ptr(), B, C, T,
                _stream(x.device))
        ctx.T, ctx.has_skip = T, skip is not None
        return xo, so

    @staticmethod
    def backward(ctx, dxo, dso):
        dxo, dso = dxo.contiguous(), dso.contiguous()
        B, C, TS = dxo.shape
        dx = torch.empty_like(dxo)
        dy = torch.empty(B, 2 * C, TS, device=dxo.device, dtype=torch.float32)
        with torch.cuda.device(dxo.device):
            _ew('dsf_train_res_skip_bwd', dxo.data_ptr(), dso.data_ptr(), dx.data_ptr(), dy.data_ptr(), B, C, ctx.T, _stream(dxo.device))
        return dx, dy, (dso if ctx.has_skip else None), None


class ConvCache:
    """Per-layer cache of the packed forward / transposed weights and the split-K workspace."""

    def __init__(self):
        self.d = {'fwd': PackedWeight(), 'bwd': PackedWeight()}

    def __call__(self, x, weight, bias, T, dil=1):
        if weight.shape[1] % 8:
            raise ValueError('input channels must be a multiple of 8')
        return _Conv1dCM.apply(x, weight, bias, T, dil, self.d)


def step_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    half = dim // 2                                                 # usr/diff/net.py:37-44
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, device=t.device) * -e)
    e = t[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def _step_embedding_rows(net, t: torch.Tensor, n_steps: int) -> torch.Tensor:
    """step_embedding(t) as rows of a table over t = 0 .. n_steps - 1, made once per (device, size) by the same element-wise arithmetic (the
    same bits): six tiny launches per step become one gather.  Only when the caller knows the schedule length (p_losses does)."""
    tabs = net.__dict__.setdefault('_train_step_tables', {})
    key = (str(t.device), int(n_steps))
    tab = tabs.get(key)
    if tab is None:
        tab = tabs[key] = step_embedding(torch.arange(int(n_steps), device=t.device, dtype=t.dtype), net.residual_channels)
    return tab.index_select(0, t.long())


def diffnet_forward_train(net, spec: torch.Tensor, diffusion_step: torch.Tensor, cond: torch.Tensor, n_steps: Optional[int] = None) -> torch.Tensor:
    """DiffNet.forward (usr/diff/net.py:107-130) under autograd.  spec [B,1,M,T], diffusion_step [B], cond [B,H,T] -> [B,1,M,T].
    n_steps: the schedule length when the caller knows it (every diffusion_step < n_steps): the step embedding then comes from a table."""
    if spec.device.type != 'cuda':
        raise RuntimeError('the HIP training path has no CPU path')
    B, _, M, T = spec.shape
    TS = padded_frames(T)
    caches = net.__dict__.setdefault('_train_caches', {})

    def conv(name, x, mod, dil=1):
        c = caches.get(name)
        if c is None:
            c = caches[name] = ConvCache()
        return c(x, mod.weight, mod.bias, T, dil)

    pad = (0, TS - T)
    if TS == T:                                                     # (F.pad with nothing to pad still copies)
        xm, cm = spec[:, 0].contiguous(), cond.contiguous()
    else:
        xm = F.pad(spec[:, 0], pad)                                 # [B][M][TS], zero tail
        cm = F.pad(cond, pad).contiguous()                          # [B][H][TS]
    x = F.relu(conv('in', xm, net.input_projection))                # :116-118 (every conv output has the zero tail; relu keeps it)
    # the step-embedding MLP and the layers' step projections (net.py:94-98, :119-120, :67): [B, C] x [C, C'] products on B rows through
    # dsf_linear_rows (own vector-ALU kernels; round 2 left them on rocBLAS, a first attempt through the convolution operators cost +0.37 ms
    # per step in layout changes and re-packs and was reverted)
    d = step_embedding(diffusion_step, net.residual_channels) if n_steps is None else _step_embedding_rows(net, diffusion_step, n_steps)   # :119
    h = linear_rows(d, net.mlp[0].weight, net.mlp[0].bias)
    d = linear_rows(F.mish(h), net.mlp[2].weight, net.mlp[2].bias)  # :120 (Mish = x tanh(softplus(x)): ATen's fused forward / backward kernels)
    from . import train_fused
    if train_fused.enabled() and train_fused.supported(net):
        # the whole residual stack as ONE autograd node on the fused kernels (csrc/train_kernels.hpp); the step projections of all layers
        # (net.py:67) are one batched linear
        layers = list(net.residual_layers)
        wd = torch.cat([l.diffusion_projection.weight for l in layers], 0)
        bd = torch.cat([l.diffusion_projection.bias for l in layers], 0)
        step_all = linear_rows(d, wd, bd).view(B, len(layers), net.residual_channels)
        skip = train_fused.residual_stack(net, x, cm, step_all, T)
    else:
        skip = None
        for l, layer in enumerate(net.residual_layers):             # ResidualBlock.forward :66-78
            ds = linear_rows(d, layer.diffusion_projection.weight, layer.diffusion_projection.bias)
            y = _AddStep.apply(x, ds, T)
            a = conv(f'l{l}.dc', y, layer.dilated_conv, layer.dilation) + conv(f'l{l}.cp', cm, layer.conditioner_projection)
            g = _Gate.apply(a, T)
            y = conv(f'l{l}.op', g, layer.output_projection)
            x, skip = _ResSkip.apply(x, y, skip, T)
    x = skip / math.sqrt(len(net.residual_layers))                  # :126
    x = F.relu(conv('sp', x, net.skip_projection))                  # :127-128
    x = conv('out', x, net.output_projection)                       # :129
    return x[:, None, :, :T]


_FUSED_ENDS = True


def set_fused_ends(on: bool):
    """A/B switch of the tests and of the measurement: False = q_sample and the L1 loss of p_losses as the reference's tensor expressions
    (five + nine launches per step, rounds 2-5)."""
    global _FUSED_ENDS
    _FUSED_ENDS = bool(on)


class _L1Mean(torch.autograd.Function):
    """(noise - x_recon).abs().mean() (shallow_diffusion_tts.py:224-228) as two launches forward (sums in a fixed order) and one backward."""

    @staticmethod
    def forward(ctx, noise, x_recon):
        lib = _lib.load()
        n = noise.numel()
        ws = torch.empty(int(lib.dsf_l1_workspace_floats()), device=noise.device, dtype=torch.float32)
        out = torch.empty((), device=noise.device, dtype=torch.float32)
        with torch.cuda.device(noise.device):
            _lib.check(lib.dsf_l1_mean(noise.data_ptr(), x_recon.data_ptr(), ws.data_ptr(), out.data_ptr(), n, _stream(noise.device)), 'dsf_l1_mean')
        ctx.save_for_backward(noise, x_recon)
        return out

    @staticmethod
    def backward(ctx, g):
        noise, x_recon = ctx.saved_tensors
        g = g.contiguous()
        db = torch.empty_like(x_recon)
        with torch.cuda.device(noise.device):
            _lib.check(_lib.load().dsf_l1_mean_bwd(noise.data_ptr(), x_recon.data_ptr(), g.data_ptr(), db.data_ptr(), noise.numel(), _stream(noise.device)),
                       'dsf_l1_mean_bwd')
        return None, db


def _plain_f32(*ts) -> bool:
    return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts)


def q_sample_rows(gd, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """q_sample (:206-211) with a step per utterance; one launch when nothing of it needs a gradient, else the tensor expression."""
    B = x_start.shape[0]
    per_row = x_start[0].numel()
    if (_FUSED_ENDS and _plain_f32(x_start, noise, gd.sqrt_alphas_cumprod, gd.sqrt_one_minus_alphas_cumprod) and t.is_cuda and t.dtype == torch.int64
            and t.shape == (B,) and noise.shape == x_start.shape and per_row % 4 == 0 and B <= 65535
            and not (torch.is_grad_enabled() and (x_start.requires_grad or noise.requires_grad))):
        out = torch.empty_like(x_start)
        with torch.cuda.device(x_start.device):
            _lib.check(_lib.load().dsf_q_sample_rows(x_start.data_ptr(), noise.data_ptr(), t.contiguous().data_ptr(), gd.sqrt_alphas_cumprod.data_ptr(),
                                                     gd.sqrt_one_minus_alphas_cumprod.data_ptr(), int(gd.sqrt_alphas_cumprod.numel()), out.data_ptr(), B, per_row,
                                                     _stream(x_start.device)),
                       'dsf_q_sample_rows')
        return out
    shape = (B, 1, 1, 1)
    return gd.sqrt_alphas_cumprod.gather(-1, t).reshape(shape) * x_start + gd.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(shape) * noise


def p_losses(gd, x_start: torch.Tensor, t: torch.Tensor, cond: torch.Tensor, noise: Optional[torch.Tensor] = None,
             nonpadding: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GaussianDiffusion.p_losses (usr/diff/shallow_diffusion_tts.py:213-231)."""
    from .net import DiffNet
    fused = isinstance(gd.denoise_fn, DiffNet)
    if not fused and not hasattr(gd.denoise_fn, 'forward_train'):
        raise NotImplementedError(f'p_losses: {type(gd.denoise_fn).__name__} has no training forward (DiffNet: the fused stack; FFT: forward_train)')
    if noise is None:
        noise = torch.randn_like(x_start)
    x_noisy = q_sample_rows(gd, x_start, t, noise)                                             # q_sample :206-211
    # the denoiser the registry returned (usr/diffsinger_task.py:23-27): DiffNet on the fused training stack, the FFT candidate on the
    # FastSpeech2 operators under autograd
    x_recon = diffnet_forward_train(gd.denoise_fn, x_noisy, t, cond, n_steps=gd.num_timesteps) if fused else gd.denoise_fn.forward_train(x_noisy, t, cond)
    if gd.loss_type == 'l1':
        if nonpadding is not None:
            return ((noise - x_recon).abs() * nonpadding.unsqueeze(1)).mean()
        if _FUSED_ENDS and _plain_f32(noise, x_recon) and noise.shape == x_recon.shape and not (torch.is_grad_enabled() and noise.requires_grad):
            return _L1Mean.apply(noise, x_recon)
        return (noise - x_recon).abs().mean()
    if gd.loss_type == 'l2':
        return F.mse_loss(noise, x_recon)
    raise NotImplementedError(gd.loss_type)

