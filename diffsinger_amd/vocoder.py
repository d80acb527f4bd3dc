"""HiFi-GAN / NSF-HiFi-GAN generator - the vocoder behind the diffusion hot path - as an
nn.Module whose forward runs on the HIP kernels of libdsdenoise.so (include/dsv.h).

Mirrors the reference module tree (paths relative to the reference root) name for name, so the `model_gen` / `generator`
state_dict of a reference checkpoint loads with strict=True, before or after remove_weight_norm():

    HifiGanGenerator, ResBlock1, ResBlock2        modules/hifigan/hifigan.py:30-92, :104-179
    SourceModuleHnNSF, SineGen                    modules/parallel_wavegan/models/source.py:7-137, :484-531
    HifiGAN.spec2wav (the registered vocoder)     vocoders/hifigan.py:40-69

What runs where: every Conv1d / ConvTranspose1d with the leaky_relu in front of it and the residual / resblock-sum /
`/ num_kernels` / source add / tanh behind it is ONE launch of k_voc_conv (fp32 MFMA); the sine source (two cumulative
sums over the sample axis, sin, uv / noise mix, Linear + tanh) and the strided noise convolutions are their own kernels.
torch is used for buffers, the one-time weight preparation (weight-norm fold, polyphase form of the transposed
convolutions) and the random draws of the source module.  Inference only; no CPU path: forward raises when the tensors
are not on the MI355X."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _lib

LRELU_SLOPE = 0.1
MAX_REACH = 48        # kVocHaloWide of csrc/voc_kernels.hpp: taps may reach +-48 samples (beyond +-28 a second instantiation of the conv kernel runs)


def padded_samples(L: int) -> int:
    return (L + 31) // 32 * 32


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    return int((kernel_size * dilation - dilation) / 2)


def polyphase_weight(w: torch.Tensor, stride: int, padding: int):
    """ConvTranspose1d weight [Ci][Co][k] (stride u, padding p, k - 2 p == u) -> (W' [Co * u][Ci][K'], pad') such that
        out[co][u q + r] = sum_ci sum_t W'[co * u + r][ci][t] * in[ci][q + t - pad']
    i.e. the stride-1 convolution at the input rate that dsv_conv1d evaluates with up = u.  Output phase r is reached by
    the taps j = (r + p) % u + u m, each reading input sample q + (r + p) // u - m."""
    ci, co, k = w.shape
    u, p = int(stride), int(padding)
    if k - 2 * p != u:
        raise NotImplementedError(f'ConvTranspose1d with kernel {k}, stride {u}, padding {p}: only kernel - 2 * padding == stride (the HiFi-GAN '
                                  f'upsamplers) is supported')
    taps = []                                                        # (r, j, delta)
    for r in range(u):
        s = r + p
        m = 0
        while (s % u) + u * m < k:
            taps.append((r, (s % u) + u * m, s // u - m))
            m += 1
    dmin = min(t[2] for t in taps)
    dmax = max(t[2] for t in taps)
    kk = dmax - dmin + 1
    wp = torch.zeros(co, u, ci, kk, dtype=w.dtype, device=w.device)
    for r, j, d in taps:
        wp[:, r, :, d - dmin] = w[:, :, j].t()
    return wp.reshape(co * u, ci, kk).contiguous(), -dmin


def fold_weight(w: torch.Tensor, F: int) -> torch.Tensor:
    """Conv1d weight [Co][Ci][K] -> the F shifted copies dsv_conv1d_folded multiplies: W'[co * F + e][ci][s] = w[co][ci][s - e]
    (K + F - 1 taps; row co * F + e computes the output sample e dilation steps behind the column's first one)."""
    co, ci, k = w.shape
    wf = torch.zeros(co, F, ci, k + F - 1, dtype=w.dtype, device=w.device)
    for e in range(F):
        wf[:, e, :, e:e + k] = w
    return wf.reshape(co * F, ci, k + F - 1).contiguous()


import ctypes as _C


class DsvChainConv(_C.Structure):
    """include/dsv.h dsv_chain_conv"""
    _fields_ = [('w_offset', _C.c_int64), ('bias_offset', _C.c_int32), ('K', _C.c_int32), ('dil', _C.c_int32), ('reserved', _C.c_int32)]


class DsvConvDesc(_C.Structure):
    """include/dsv.h dsv_conv_desc"""
    _fields_ = [('in_', _C.c_void_p), ('wpacked', _C.c_void_p), ('bias', _C.c_void_p), ('out', _C.c_void_p), ('residual', _C.c_void_p),
                ('sum_in', _C.c_void_p), ('K', _C.c_int32), ('pad', _C.c_int32), ('dil', _C.c_int32), ('act', _C.c_int32), ('divide', _C.c_float),
                ('reserved', _C.c_int32)]


# How the ResBlock1 chains of a stage are launched (csrc/voc_chain.hpp): None = one launch per RESBLOCK (three conv pairs; the three parallel
# resblocks of a stage run one after the other, the running sum handed on through `sum_in`) wherever the library supports it (8 / 16 / 32
# channels; wider stages: one launch per convolution).  A launch's halo is the receptive field of ITS chain: 12 / 36 / 60 samples for the
# kernel-3 / 7 / 11 resblocks instead of 60 for all three in a whole-stage launch - the fastest grouping since round 6 (32 channels: 1.51 ms
# against 1.59 for the stage, 16: 0.91 / 0.95, 8: 0.595 / 0.60, profiles/r6_02_voc_chain_variants_modes.jsonl; rounds 3-5: the stage, r07).
# 'stage' / 'resblock' / 'pair' force that grouping; 'off' = one launch per convolution everywhere (the A/B switch of the measurement and of
# the bit-identity tests).
_CHAIN_MODE = None
# Round 6: None additionally MERGES two of the stage's resblocks into one launch (dsv_resblock_chain_multi) and lets the third form the sum
# (dsv_resblock_chain_sum) wherever `_merge_plan` models fewer rounds than for one launch per resblock; 'resblock' is the plain form.
# 'merged' forces the merge, 'merged0' / 'merged1' / 'merged2' with that resblock as the summing launch (tests: every split gives the same bits).


def set_chain_mode(mode):
    global _CHAIN_MODE
    if mode not in (None, 'stage', 'resblock', 'pair', 'off', 'merged', 'merged0', 'merged1', 'merged2'):
        raise ValueError(mode)
    _CHAIN_MODE = mode


def _merge_plan(W, d, slots, min_gain=0.03):
    """Which resblock of a stage to leave out of the merged launch.  A chain launch of W workgroups on `slots` co-resident places takes
    ceil(W / slots) rounds of its workgroup time d (profiles/r6_27_voc_tail_probe.jsonl); a merged launch - longest chain first - is list
    scheduling of its workgroups on the slots.  W[r], d[r]: workgroups and relative workgroup time of resblock r.  Returns (solo, order of
    the merged groups), or None when the modelled time does not beat three launches by `min_gain`."""
    import math
    n = len(W)
    if n != 3:
        return None
    rounds = lambda w: math.ceil(w / slots)
    separate = sum(rounds(W[r]) * d[r] for r in range(n))

    def merged_time(groups):                                      # list scheduling: every workgroup takes the slot that is free first
        import heapq
        free = [(0.0, slots)]                                     # (time, slots that become free then)
        end = 0.0
        for r in groups:
            left = W[r]
            while left > 0:
                t, c = heapq.heappop(free)
                k = min(c, left)
                left -= k
                heapq.heappush(free, (t + d[r], k))
                end = max(end, t + d[r])
                if k < c:
                    heapq.heappush(free, (t, c - k))
        return end

    # the summing launch: the resblock of MEDIAN workgroup time - the longest and the shortest chain share the merged grid, so that the short
    # workgroups fill the long ones' last round.  (Measured on the bench shape, every choice per stage, profiles/r6_30 / r6_32_voc_chain_variants.jsonl:
    # the median is the fastest at 32, 16 and 8 channels; the model alone - it ignores that a partial round runs faster - preferred the
    # shortest chain as the summing launch at 16 channels, 1.3 % slower.)  The model decides only WHETHER to merge.
    solo = sorted(range(n), key=lambda r: d[r])[n // 2]
    groups = sorted((r for r in range(n) if r != solo), key=lambda r: -d[r])
    t = merged_time(groups) + rounds(W[solo]) * d[solo]
    if t > (1.0 - min_gain) * separate:
        return None
    return solo, groups


class _HipOps:
    """The C ABI of include/dsv.h on torch device tensors (buffers in, buffers out).  There is no other implementation in the
    package: tests swap in a torch restatement of the header's formulas to check the orchestration on CPU."""

    def __init__(self):
        self.lib = _lib.load()

    @staticmethod
    def _s(dev) -> int:
        return torch.cuda.current_stream(dev).cuda_stream

    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return t.data_ptr() if t is not None else None

    def pack(self, w: torch.Tensor) -> torch.Tensor:
        rows, ci, k = w.shape
        n = self.lib.dsv_packed_floats(rows, ci, k)
        buf = torch.empty(n, device=w.device, dtype=torch.float32)
        w = w.contiguous()
        with torch.cuda.device(w.device):
            _lib.check(self.lib.dsv_pack_weight(w.data_ptr(), rows, ci, k, buf.data_ptr(), self._s(w.device)), 'dsv_pack_weight')
        w.record_stream(torch.cuda.current_stream(w.device))
        return buf

    def pad_rows(self, x: torch.Tensor) -> torch.Tensor:
        B, C, L = x.shape
        out = torch.empty(B, C, padded_samples(L), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(self.lib.dsv_pad_rows(x.data_ptr(), out.data_ptr(), B * C, L, self._s(x.device)), 'dsv_pad_rows')
        return out

    def conv(self, x, L_in, wp, bias, rows, ci, k, pad, dil, up=1, pre_slope=1.0, residual=None, sum_in=None, divide=1.0, act=0):
        B = x.shape[0]
        assert x.shape[1] == ci and x.shape[2] == padded_samples(L_in) and x.is_contiguous()
        out = torch.empty(B, rows // up, padded_samples(L_in * up), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(self.lib.dsv_conv1d(x.data_ptr(), wp.data_ptr(), self._p(bias), out.data_ptr(), B, ci, rows, k, pad, dil, L_in, up,
                                           float(pre_slope), self._p(residual), self._p(sum_in), float(divide), int(act), self._s(x.device)),
                       'dsv_conv1d')
        return out

    def conv_multi(self, xs, L_in, items, up=1, pre_slope=1.0):
        """Independent convolutions of one shape in ONE launch (dsv_conv1d_multi): xs[g] through items[g] = dict(wp, bias, rows, ci, k, pad, dil
        [, residual, sum_in, divide, act]); returns the outputs.  The first item's workgroups are dispatched first."""
        B, rows, ci = xs[0].shape[0], items[0]['rows'], items[0]['ci']
        outs, descs = [], (DsvConvDesc * len(xs))()
        for g, (x, it) in enumerate(zip(xs, items)):
            assert x.shape == xs[0].shape and x.shape[1] == ci and x.shape[2] == padded_samples(L_in) and x.is_contiguous() and it['rows'] == rows and it['ci'] == ci
            out = torch.empty(B, rows // up, padded_samples(L_in * up), device=x.device, dtype=torch.float32)
            outs.append(out)
            descs[g] = DsvConvDesc(x.data_ptr(), it['wp'].data_ptr(), self._p(it.get('bias')), out.data_ptr(), self._p(it.get('residual')),
                                   self._p(it.get('sum_in')), it['k'], it['pad'], it['dil'], int(it.get('act', 0)), float(it.get('divide', 1.0)), 0)
        with torch.cuda.device(xs[0].device):
            _lib.check(self.lib.dsv_conv1d_multi(len(xs), descs, B, ci, rows, L_in, up, float(pre_slope), self._s(xs[0].device)), 'dsv_conv1d_multi')
        return outs

    def fold_factor(self, co, ci, k, dil) -> int:
        return int(self.lib.dsv_fold_factor(co, ci, k, dil))

    def conv_folded(self, x, L, wp, bias, co, ci, k, F, dil, pre_slope=1.0, residual=None, sum_in=None, divide=1.0, act=0):
        B = x.shape[0]
        assert x.shape[1] == ci and x.shape[2] == padded_samples(L) and x.is_contiguous()
        out = torch.empty(B, co, padded_samples(L), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(self.lib.dsv_conv1d_folded(x.data_ptr(), wp.data_ptr(), self._p(bias), out.data_ptr(), B, ci, co, k, F, dil, L,
                                                  float(pre_slope), self._p(residual), self._p(sum_in), float(divide), int(act),
                                                  self._s(x.device)), 'dsv_conv1d_folded')
        return out

    def chain_fold(self, C: int) -> int:
        return int(self.lib.dsv_chain_fold(C))

    def chain_supported(self, C, nres, npairs, descs) -> int:
        return int(self.lib.dsv_chain_supported(C, nres, npairs, descs))

    def pack_into(self, w: torch.Tensor, buf: torch.Tensor, offset: int):
        """dsv_pack_weight of w [rows][Ci][K] into buf at float offset `offset` (the call also zeroes the slack behind the piece)."""
        rows, ci, k = w.shape
        w = w.contiguous()
        with torch.cuda.device(w.device):
            _lib.check(self.lib.dsv_pack_weight(w.data_ptr(), rows, ci, k, buf.data_ptr() + 4 * offset, self._s(w.device)), 'dsv_pack_weight')
        w.record_stream(torch.cuda.current_stream(w.device))

    def resblock_chain(self, x, L, wp, bias, C, nres, npairs, descs, sum_in=None, divide=1.0, pre_slope=LRELU_SLOPE):
        B = x.shape[0]
        assert x.shape[1] == C and x.shape[2] == padded_samples(L) and x.is_contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(self.lib.dsv_resblock_chain(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr(), self._p(sum_in), B, C, L, nres, npairs,
                                                   descs, float(pre_slope), float(divide), self._s(x.device)), 'dsv_resblock_chain')
        return out

    def resblock_chain_multi(self, x, L, wp, bias, C, npairs, descs, ngroups, pre_slope=LRELU_SLOPE):
        """`ngroups` independent resblocks (descs: [ngroups][npairs][2]) in one launch; returns their raw outputs, one buffer each."""
        B = x.shape[0]
        assert x.shape[1] == C and x.shape[2] == padded_samples(L) and x.is_contiguous()
        outs = [torch.empty_like(x) for _ in range(ngroups)]
        ptrs = (_C.c_void_p * ngroups)(*[o.data_ptr() for o in outs])
        with torch.cuda.device(x.device):
            _lib.check(self.lib.dsv_resblock_chain_multi(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), ptrs, B, C, L, ngroups, npairs, descs,
                                                         float(pre_slope), self._s(x.device)), 'dsv_resblock_chain_multi')
        return outs

    def resblock_chain_sum(self, x, L, wp, bias, C, npairs, descs, sum_in, sum_in2, own_last, divide, pre_slope=LRELU_SLOPE):
        B = x.shape[0]
        assert x.shape[1] == C and x.shape[2] == padded_samples(L) and x.is_contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(self.lib.dsv_resblock_chain_sum(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr(), sum_in.data_ptr(), sum_in2.data_ptr(),
                                                       int(bool(own_last)), B, C, L, npairs, descs, float(pre_slope), float(divide), self._s(x.device)),
                       'dsv_resblock_chain_sum')
        return out

    def noise_conv(self, har, L_har, w, bias, stride, pad, L_out):
        B = har.shape[0]
        C, K = w.shape
        out = torch.empty(B, C, padded_samples(L_out), device=har.device, dtype=torch.float32)
        with torch.cuda.device(har.device):
            _lib.check(self.lib.dsv_noise_conv(har.data_ptr(), w.data_ptr(), self._p(bias), out.data_ptr(), B, C, K, stride, pad, L_har, L_out,
                                               self._s(har.device)), 'dsv_noise_conv')
        return out

    def sine_source(self, f0, rand_ini, noise, lin_w, lin_b, up, sr, sine_amp, noise_std, thr):
        B, T = f0.shape
        H = rand_ini.shape[1]
        L = T * up
        ws = torch.empty(B, H, L, device=f0.device, dtype=torch.float32)
        har = torch.empty(B, padded_samples(L), device=f0.device, dtype=torch.float32)
        with torch.cuda.device(f0.device):
            _lib.check(self.lib.dsv_sine_source(f0.data_ptr(), rand_ini.data_ptr(), noise.data_ptr(), lin_w.data_ptr(), lin_b.data_ptr(),
                                                ws.data_ptr(), har.data_ptr(), B, T, up, H, float(sr), float(sine_amp), float(noise_std),
                                                float(thr), self._s(f0.device)), 'dsv_sine_source')
        return har


class _WNConv(nn.Module):
    """Parameter holder of weight_norm(Conv1d / ConvTranspose1d): weight_g / weight_v / bias as a checkpoint stores them
    (torch.nn.utils.weight_norm, dim 0), or weight / bias after remove_weight_norm()."""

    def __init__(self, wshape, nbias):
        super().__init__()
        self.weight_g = nn.Parameter(torch.ones(wshape[0], 1, 1))
        self.weight_v = nn.Parameter(torch.randn(wshape) * 0.01)
        self.bias = nn.Parameter(torch.zeros(nbias))

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

    def wshape(self):
        return tuple(self.weight_v.shape if hasattr(self, 'weight_g') else self.weight.shape)

    def tag(self):
        ps = ([self.weight_g, self.weight_v] if hasattr(self, 'weight_g') else [self.weight]) + [self.bias]
        return tuple((p.data_ptr(), p._version, p.device) for p in ps)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # a state saved AFTER remove_weight_norm() holds `weight`: switch this holder to the plain form before the strict load
        if prefix + 'weight' in state_dict and hasattr(self, 'weight_g'):
            self.remove_weight_norm()
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class _PlainConv(nn.Module):
    def __init__(self, wshape):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(wshape) * 0.01)
        self.bias = nn.Parameter(torch.zeros(wshape[0]))


class _ResBlock(nn.Module):
    """ResBlock1 (convs1 / convs2, hifigan.py:30-66) or ResBlock2 (convs, :69-92): parameters only."""

    def __init__(self, kind: str, ch: int, k: int, dils):
        super().__init__()
        dils = tuple(dils)[:3 if kind == '1' else 2]                 # ResBlock1 builds exactly three conv pairs, ResBlock2 exactly two convs
        self.kind, self.k, self.dils = kind, k, dils
        if kind == '1':
            self.convs1 = nn.ModuleList([_WNConv((ch, ch, k), ch) for _ in dils])
            self.convs2 = nn.ModuleList([_WNConv((ch, ch, k), ch) for _ in dils])
        else:
            self.convs = nn.ModuleList([_WNConv((ch, ch, k), ch) for _ in dils])

    def remove_weight_norm(self):
        for m in self.modules():
            if isinstance(m, _WNConv):
                m.remove_weight_norm()


class _Linear(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(o, i) * 0.1)
        self.bias = nn.Parameter(torch.zeros(o))


class _Source(nn.Module):
    """SourceModuleHnNSF (source.py:484-531): l_linear is its only parameter."""

    def __init__(self, harmonic_num):
        super().__init__()
        self.l_linear = _Linear(harmonic_num + 1, 1)


class HifiGanGenerator(nn.Module):
    """modules/hifigan/hifigan.py:104-179.  forward(x [B,80,T], f0 [B,T] or None) -> [B,1,T*prod(upsample_rates)]."""

    def __init__(self, h, c_out=1):
        super().__init__()
        self.h = h
        self.num_kernels = len(h['resblock_kernel_sizes'])
        self.num_upsamples = len(h['upsample_rates'])
        c0 = h['upsample_initial_channel']
        self.use_pitch_embed = bool(h.get('use_pitch_embed', False))
        if self.use_pitch_embed:
            self.harmonic_num = 8
            self.m_source = _Source(self.harmonic_num)
            self.noise_convs = nn.ModuleList()
        self.conv_pre = _WNConv((c0, 80, 7), c0)
        self.ups = nn.ModuleList()
        rates, ksz = list(h['upsample_rates']), list(h['upsample_kernel_sizes'])
        ch = c0
        for i, (u, k) in enumerate(zip(rates, ksz)):
            ch = c0 // (2 ** (i + 1))
            self.ups.append(_WNConv((ch * 2, ch, k), ch))
            if self.use_pitch_embed:
                if i + 1 < len(rates):
                    s = int(np.prod(rates[i + 1:]))
                    self.noise_convs.append(_PlainConv((ch, 1, s * 2)))
                else:
                    self.noise_convs.append(_PlainConv((ch, 1, 1)))
        self.resblocks = nn.ModuleList()
        for i in range(len(rates)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes']):
                self.resblocks.append(_ResBlock(str(h['resblock']), ch, k, d))
        self.conv_post = _WNConv((c_out, ch, 7), c_out)
        self.c_out = c_out
        self._ops = None
        self._packed = {}

    def remove_weight_norm(self):
        for m in self.modules():
            if isinstance(m, _WNConv):
                m.remove_weight_norm()

    # ---- weight preparation (cached per parameter version) ---------------------------------------------------------------
    def _prep(self, key: str, conv: _WNConv, transposed_stride: int = 0, transposed_pad: int = 0, fold: int = 1):
        tag = conv.tag() + (fold,)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1]
        w = conv.plain_weight().to(torch.float32)
        pad = None
        if transposed_stride:
            w, pad = polyphase_weight(w, transposed_stride, transposed_pad)
        elif fold > 1:
            w = fold_weight(w, fold)
        rows, ci, k = w.shape
        entry = dict(wp=self._ops.pack(w), rows=rows, ci=ci, k=k, pad=pad, bias=conv.bias.detach().to(torch.float32).contiguous())
        self._packed[key] = (tag, entry)
        return entry

    def _conv(self, key, conv, x, L, dil=1, **kw):
        """One 'same'-padded Conv1d with its fused neighbours; the narrow layers take the folded kernel when the library offers it."""
        co, ci, k = conv.wshape()
        if get_padding(k, dil) > MAX_REACH:
            raise NotImplementedError(f'Conv1d kernel {k} at dilation {dil} reaches {get_padding(k, dil)} samples; the vocoder kernels stage +-{MAX_REACH} '
                                      f'(kernel 11 at dilation 5 = 25 on the shipped generators, kernel 7 at dilation 12 = 36 on the official v3)')
        F = self._ops.fold_factor(co, ci, k, dil)
        e = self._prep(key, conv, fold=F)
        if F > 1:
            return self._ops.conv_folded(x, L, e['wp'], e['bias'], co, ci, k, F, dil, **kw)
        return self._ops.conv(x, L, e['wp'], e['bias'], e['rows'], e['ci'], e['k'], get_padding(e['k'], dil), dil, **kw)

    def _chain_prep(self, i: int):
        """Stage i's ResBlock1 convolutions as the chain kernel wants them: ONE packed weight buffer (per convolution the F shifted copies of
        the filter, dsv_pack_weight(rows = 32, C, K + F - 1)), one bias buffer, and a descriptor per convolution.  None: not a ResBlock1 stage
        of 8 / 16 / 32 channels.  Cached per parameter version."""
        rbs = [self.resblocks[i * self.num_kernels + j] for j in range(self.num_kernels)]
        if any(rb.kind != '1' for rb in rbs) or len({len(rb.dils) for rb in rbs}) != 1:
            return None
        C = rbs[0].convs1[0].wshape()[0]
        F = self._ops.chain_fold(C)
        if not F:
            return None
        convs = [(c, d if which == 0 else 1) for rb in rbs for q, d in enumerate(rb.dils) for which, c in ((0, rb.convs1[q]), (1, rb.convs2[q]))]
        tag = tuple(c.tag() for c, _ in convs)
        hit = self._packed.get(f'chain{i}')
        if hit is not None and hit[0] == tag:
            return hit[1]
        sizes = [(C // 8) * (c.wshape()[2] + F - 1) * 256 for c, _ in convs]
        dev = convs[0][0].bias.device
        slack = self._ops.lib.dsv_packed_floats(32, C, 1) - (C // 8) * 256          # the A prefetch overruns the last piece by up to five chunks
        wp = torch.zeros(sum(sizes) + slack, device=dev, dtype=torch.float32)
        bias = torch.empty(len(convs), C, device=dev, dtype=torch.float32)
        descs = (DsvChainConv * len(convs))()
        off = 0
        for n, ((c, d), sz) in enumerate(zip(convs, sizes)):
            w = c.plain_weight().to(torch.float32)
            self._ops.pack_into(fold_weight(w, F) if F > 1 else w.contiguous(), wp, off)
            bias[n] = c.bias.detach().to(torch.float32)
            descs[n] = DsvChainConv(off, n * C, int(w.shape[2]), int(d), 0)
            off += sz
        entry = dict(wp=wp, bias=bias, descs=descs, C=C, npairs=len(rbs[0].dils), nres=len(rbs))
        self._packed[f'chain{i}'] = (tag, entry)
        return entry

    def _merge_plan_for(self, i: int, e, B: int, L: int, force: bool = False):
        """_merge_plan for stage i at batch B, length L (cached per shape): workgroups per resblock from dsv_chain_supported's tile size, relative
        workgroup time = its chunks (C / 8 x folded taps per convolution) + 5.5 chunk times per convolution of epilogue / barrier / restart
        (the fit of profiles/r6_27_voc_tail_probe.jsonl: 6.0 / 5.6 / 4.5 at 32 / 16 / 8 channels), slots = 256 CUs x the co-resident workgroups
        of the default instantiation."""
        key = (i, B, L, force)
        hit = self._packed.get('merge_plans', {})
        if key in hit:
            return hit[key]
        C, nres, npairs, ops = e['C'], e['nres'], e['npairs'], self._ops
        F = ops.chain_fold(C)
        plan = None
        Ns = [ops.chain_supported(C, 1, npairs, (DsvChainConv * (npairs * 2))(*[e['descs'][(r * npairs + q) * 2 + k] for q in range(npairs) for k in range(2)]))
              for r in range(nres)]
        if all(Ns):
            LS = padded_samples(L)
            W = [B * ((LS + n - 1) // n) for n in Ns]
            d = [sum((C // 8) * (e['descs'][(r * npairs + q) * 2 + k].K + F - 1) + 5.5 for q in range(npairs) for k in range(2)) for r in range(nres)]
            slots = 256 * (2 if C == 32 else 3)
            plan = _merge_plan(W, d, slots, min_gain=-1e9 if force else 0.03)
        hit[key] = plan
        self._packed['merge_plans'] = hit
        return plan

    def _stage_resblocks(self, i: int, x, L):
        """x = (sum_j resblock_{i, j}(x)) / num_kernels (hifigan.py:161-166): fused chains where the library offers them, else one launch per
        convolution.  Every grouping gives the same bits."""
        mode = _CHAIN_MODE
        e = self._chain_prep(i) if mode != 'off' else None
        if e is not None:
            C, nres, npairs = e['C'], e['nres'], e['npairs']
            if mode is None:
                mode = 'resblock'
            sub = lambda r0, q0, nr, nq: (DsvChainConv * (nr * nq * 2))(*[e['descs'][(r * npairs + q) * 2 + k] for r in range(r0, r0 + nr)
                                                                       for q in range(q0, q0 + nq) for k in range(2)])
            ops, nk = self._ops, float(self.num_kernels)
            per_resblock = all(ops.chain_supported(C, 1, npairs, sub(r, 0, 1, npairs)) for r in range(nres))
            plan = None
            if nres == 3 and per_resblock:
                if mode in ('merged0', 'merged1', 'merged2'):                     # tests: that resblock as the summing launch
                    solo = int(mode[6])
                    plan = (solo, sorted((r for r in range(3) if r != solo), reverse=True))
                elif mode == 'merged' or _CHAIN_MODE is None:
                    plan = self._merge_plan_for(i, e, x.shape[0], L, force=mode == 'merged')
            if mode.startswith('merged'):
                mode = 'resblock'
            if plan is not None and mode == 'resblock':
                # two resblocks in ONE launch, longest first, each to its own buffer; the third forms xs = y_0 + y_1 + y_2 in that order with
                # its own y first / second - (s + y) + s2 - or last - (s + s2) + y
                solo, groups = plan
                descs2 = (DsvChainConv * (2 * npairs * 2))(*[e['descs'][(r * npairs + q) * 2 + k] for r in groups for q in range(npairs) for k in range(2)])
                ys = dict(zip(groups, ops.resblock_chain_multi(x, L, e['wp'], e['bias'], C, npairs, descs2, 2)))
                a, b = (ys[r] for r in sorted(groups))
                return ops.resblock_chain_sum(x, L, e['wp'], e['bias'], C, npairs, sub(solo, 0, 1, npairs), a, b, solo == 2, nk)
            if mode == 'stage' and ops.chain_supported(C, nres, npairs, e['descs']):
                return ops.resblock_chain(x, L, e['wp'], e['bias'], C, nres, npairs, e['descs'], divide=nk)
            if mode in ('stage', 'resblock') and per_resblock:
                acc = None
                for r in range(nres):
                    acc = ops.resblock_chain(x, L, e['wp'], e['bias'], C, 1, npairs, sub(r, 0, 1, npairs), sum_in=acc,
                                             divide=nk if r == nres - 1 else 1.0)
                return acc
            if all(ops.chain_supported(C, 1, 1, sub(r, q, 1, 1)) for r in range(nres) for q in range(npairs)):
                acc = None
                for r in range(nres):
                    y = x
                    for q in range(npairs):
                        last = q == npairs - 1
                        y = ops.resblock_chain(y, L, e['wp'], e['bias'], C, 1, 1, sub(r, q, 1, 1), sum_in=acc if last else None,
                                               divide=nk if (last and r == nres - 1) else 1.0)
                    acc = y
                return acc
        if _CHAIN_MODE is None:
            y = self._stage_resblocks_by_level(i, x, L)
            if y is not None:
                return y
        acc = None
        for j in range(self.num_kernels):                            # xs = rb0(x); xs += rb1(x); ...; x = xs / num_kernels
            last = j == self.num_kernels - 1
            acc = self._resblock(i * self.num_kernels + j, x, L, acc, float(self.num_kernels) if last else 1.0)
        return acc

    def _stage_resblocks_by_level(self, i: int, x, L):
        """The stage's parallel ResBlock1 as chains advanced LEVEL BY LEVEL (round 6; stages without a chain kernel - 64 channels on the shipped
        generator): the j-th resblocks' convolutions of a level are independent, so they share ONE launch (dsv_conv1d_multi, the largest kernel
        first) - 3 + 2 merged launches and the three convolutions that carry the running sum `xs +=` instead of 18.  Every convolution is the
        one `_resblock` runs, operand for operand: the same bits.  None: the stage does not qualify (ResBlock2, folded layers, one resblock)."""
        nk, ops = self.num_kernels, self._ops
        rbs = [self.resblocks[i * nk + j] for j in range(nk)]
        if nk < 2 or nk > 3 or not hasattr(ops, 'conv_multi') or any(rb.kind != '1' for rb in rbs) or len({len(rb.dils) for rb in rbs}) != 1:
            return None
        convs = [c for rb in rbs for q in range(len(rb.dils)) for c in (rb.convs1[q], rb.convs2[q])]
        if len({c.wshape()[:2] for c in convs}) != 1:
            return None
        for rb in rbs:
            for q, d in enumerate(rb.dils):
                for c, dd in ((rb.convs1[q], d), (rb.convs2[q], 1)):
                    co, ci, k = c.wshape()
                    if get_padding(k, dd) > MAX_REACH or ops.fold_factor(co, ci, k, dd) != 1:
                        return None
        order = sorted(range(nk), key=lambda j: -rbs[j].convs1[0].wshape()[2])          # the largest kernel's workgroups first
        n = len(rbs[0].dils)

        def item(j, which, q, dil, **kw):
            conv = (rbs[j].convs1 if which == 1 else rbs[j].convs2)[q]
            e = self._prep(f'rb{i * nk + j}.c{which}.{q}', conv, fold=1)
            return dict(wp=e['wp'], bias=e['bias'], rows=e['rows'], ci=e['ci'], k=e['k'], pad=get_padding(e['k'], dil), dil=dil, **kw)

        xr = {j: x for j in range(nk)}
        for q in range(n):
            xt = dict(zip(order, ops.conv_multi([xr[j] for j in order], L, [item(j, 1, q, rbs[j].dils[q]) for j in order], pre_slope=LRELU_SLOPE)))
            if q < n - 1:
                xr = dict(zip(order, ops.conv_multi([xt[j] for j in order], L, [item(j, 2, q, 1, residual=xr[j]) for j in order], pre_slope=LRELU_SLOPE)))
            else:
                acc = None
                for j in range(nk):                                  # xs = rb0(x); xs += rb1(x); ...; x = xs / num_kernels
                    acc = self._conv(f'rb{i * nk + j}.c2.{q}', rbs[j].convs2[q], xt[j], L, dil=1, pre_slope=LRELU_SLOPE, residual=xr[j], sum_in=acc,
                                     divide=float(nk) if j == nk - 1 else 1.0)
                return acc

    def _resblock(self, idx: int, x, L, sum_in, divide):
        """One ResBlock on x; its last convolution also adds the block output into `sum_in` (xs += ...) and divides."""
        rb = self.resblocks[idx]
        n = len(rb.dils)
        for q, d in enumerate(rb.dils):
            last = q == n - 1
            if rb.kind == '1':                                       # hifigan.py:54-61
                xt = self._conv(f'rb{idx}.c1.{q}', rb.convs1[q], x, L, dil=d, pre_slope=LRELU_SLOPE)
                x = self._conv(f'rb{idx}.c2.{q}', rb.convs2[q], xt, L, dil=1, pre_slope=LRELU_SLOPE, residual=x,
                               sum_in=sum_in if last else None, divide=divide if last else 1.0)
            else:                                                    # hifigan.py:82-87
                x = self._conv(f'rb{idx}.c.{q}', rb.convs[q], x, L, dil=d, pre_slope=LRELU_SLOPE, residual=x,
                               sum_in=sum_in if last else None, divide=divide if last else 1.0)
        return x

    @torch.no_grad()
    def forward(self, x, f0=None, *, rand_ini: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """x [B,80,T] mel; f0 [B,T] Hz (NSF) or None.  rand_ini [B,9] / noise [B,T*hop,9]: the source module's draws
        (torch.rand / torch.randn_like in SineGen, source.py:57, :131); drawn from torch's generator on x's device when
        omitted, in the reference's order."""
        if self._ops is None:
            if x.device.type != 'cuda':
                raise RuntimeError('HifiGanGenerator: the HIP vocoder has no CPU path - move the module and its inputs to the MI355X')
            self._ops = _HipOps()
        ops = self._ops
        h = self.h
        rates, ksz = list(h['upsample_rates']), list(h['upsample_kernel_sizes'])
        B, M, T = x.shape
        x = x.to(torch.float32).contiguous()
        har = None
        hop = int(np.prod(rates))
        if f0 is not None:
            if not self.use_pitch_embed:
                raise ValueError('f0 given but the generator was built without use_pitch_embed')
            if hop & (hop - 1):
                raise NotImplementedError('nearest f0 upsampling is index // hop here: exact for power-of-two hop sizes only')
            f0 = f0.to(torch.float32).contiguous()
            H = self.harmonic_num + 1
            Lh = T * hop
            if rand_ini is None:
                rand_ini = torch.rand(B, H, device=x.device)
            if noise is None:
                noise = torch.randn(B, Lh, H, device=x.device)
                torch.randn(B, Lh, 1, device=x.device)               # SourceModuleHnNSF draws its (unused) noise branch too, source.py:529
            lin = self.m_source.l_linear
            har = ops.sine_source(f0, rand_ini.to(torch.float32).contiguous(), noise.to(torch.float32).contiguous(),
                                  lin.weight.detach().reshape(-1).contiguous(), lin.bias.detach().contiguous(), hop,
                                  h['audio_sample_rate'], 0.1, 0.003, 0.0)
        L = T
        x = ops.pad_rows(x)
        x = self._conv('pre', self.conv_pre, x, L)
        for i, (u, k) in enumerate(zip(rates, ksz)):
            xs = None
            if har is not None:                                      # hifigan.py:158-160
                nc = self.noise_convs[i]
                if i + 1 < len(rates):
                    s = int(np.prod(rates[i + 1:]))
                    xs = ops.noise_conv(har, T * hop, nc.weight.detach()[:, 0].contiguous(), nc.bias.detach().contiguous(), s, s // 2, L * u)
                else:
                    xs = ops.noise_conv(har, T * hop, nc.weight.detach()[:, 0].contiguous(), nc.bias.detach().contiguous(), 1, 0, L * u)
            e = self._prep(f'ups{i}', self.ups[i], transposed_stride=u, transposed_pad=(k - u) // 2)
            x = ops.conv(x, L, e['wp'], e['bias'], e['rows'], e['ci'], e['k'], e['pad'], 1, up=u, pre_slope=LRELU_SLOPE, residual=xs)
            L = L * u
            x = self._stage_resblocks(i, x, L)
        x = self._conv('post', self.conv_post, x, L, pre_slope=0.01, act=1)                                  # F.leaky_relu default slope, tanh
        return x[:, :, :L].contiguous()


VOCODERS = {}


def register_vocoder(cls):
    """vocoders/base_vocoder.py:5-8."""
    VOCODERS[cls.__name__.lower()] = cls
    VOCODERS[cls.__name__] = cls
    return cls


def get_vocoder_cls(hp):
    """vocoders/base_vocoder.py:11-19: a registered name, or a dotted `package.Class` path (the shipped configs say
    `vocoder: vocoders.hifigan.HifiGAN`; inside the reference tree that import then finds the class `register_vocoders` rebound)."""
    import importlib
    from . import pwg  # noqa: F401  (registers PWG: `vocoder: pwg` is the default of configs/tts/base.yaml:88)
    if hp['vocoder'] in VOCODERS:
        return VOCODERS[hp['vocoder']]
    pkg, cls_name = hp['vocoder'].rsplit('.', 1)
    if cls_name in VOCODERS and pkg in ('vocoders.hifigan', 'vocoders.pwg', 'diffsinger_amd.vocoder', 'diffsinger_amd.pwg'):
        return VOCODERS[cls_name]
    return getattr(importlib.import_module(pkg), cls_name)


def register_vocoders(*registries, modules=(), pwg_modules=()):
    """Rebind HifiGAN and PWG in the reference's registry (vocoders.base_vocoder.VOCODERS) and, for configs that name a class by its dotted
    path, in the modules that export it:
        register_vocoders(vocoders.base_vocoder.VOCODERS, modules=[vocoders.hifigan], pwg_modules=[vocoders.pwg])"""
    from .pwg import PWG
    for reg in registries:
        reg['hifigan'] = reg['HifiGAN'] = HifiGAN
        reg['pwg'] = reg['PWG'] = PWG
    for mod in modules:
        mod.HifiGAN = HifiGAN
    for mod in pwg_modules:
        mod.PWG = PWG
    return registries


def load_model(config_path, checkpoint_path, device=None):
    """vocoders/hifigan.py:17-32: (generator, config, device) from a checkpoint directory's config + weights.  `.yaml` configs hold the
    training run's flattened hparams and the state under ['state_dict']['model_gen']; `.json` configs are the official release
    (['generator'])."""
    import json

    import yaml
    if device is None:
        device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    from .ckpt import _torch_load
    ckpt_dict = _torch_load(checkpoint_path, trusted=True)         # the reference's own torch.load (vocoders/hifigan.py:19) unpickles everything
    if '.yaml' in config_path:
        with open(config_path) as f:
            config = yaml.safe_load(f)
        if config.get('base_config'):
            raise NotImplementedError('vocoder config.yaml with a base_config chain: flatten it (the reference saves the merged hparams)')
        state = ckpt_dict['state_dict']['model_gen']
    elif '.json' in config_path:
        with open(config_path) as f:
            config = json.load(f)
        state = ckpt_dict['generator']
    else:
        raise ValueError(config_path)
    config.setdefault('use_pitch_embed', False)
    model = HifiGanGenerator(config)
    model.load_state_dict(state, strict=True)
    model.remove_weight_norm()
    model = model.eval().to(device)
    print(f'| Loaded model parameters from {checkpoint_path}.')
    print(f'| HifiGAN device: {device}.')
    return model, config, device


def _hann_periodic(win_length: int, n_fft: int) -> np.ndarray:
    """scipy.signal.get_window('hann', win_length, fftbins=True), zero-padded on both sides to n_fft (librosa.util.pad_center)."""
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
    lpad = (n_fft - win_length) // 2
    return np.pad(w, (lpad, n_fft - win_length - lpad))


def denoise(wav: np.ndarray, v: float = 0.1, *, fft_size: int, hop_size: int, win_size: int) -> np.ndarray:
    """The reference's spectral-subtraction post-filter (vocoders/vocoder_utils.py:7-15), applied ON THE HOST to the waveform the
    generator returned, as there: magnitude of a centred Hann STFT minus `v`, clipped at 0, phases kept, inverse STFT.

    The reference calls librosa 0.8.0 (requirements.txt:2), which is not in this image: this is a numpy restatement of librosa's
    published stft / istft for the arguments used there - window 'hann' (periodic) padded to n_fft, center=True with
    pad_mode='constant', frames of n_fft at hop_length, rfft in double, spectrum kept as complex64; inverse: irfft x window overlap-added
    into a float32 signal, divided by the window sum-of-squares where it exceeds `tiny`, trimmed by n_fft // 2 on both sides.
    PARITY UNPINNED against librosa itself (absent); pinned against an independent implementation (torch.stft / torch.istft) in
    tests/test_vocoder_host.py."""
    wav = np.asarray(wav, dtype=np.float32).reshape(-1)
    n_fft, hop = int(fft_size), int(hop_size)
    win = _hann_periodic(int(win_size), n_fft)
    y = np.pad(wav, n_fft // 2, mode='constant')
    n_frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    spec = np.fft.rfft(win[:, None] * y[idx], axis=0).astype(np.complex64)             # [1 + n_fft / 2, n_frames]
    mag = np.clip(np.abs(spec) - v, a_min=0, a_max=None)
    spec = mag * np.exp(1j * np.angle(spec))
    frames = win[:, None] * np.fft.irfft(spec, n=n_fft, axis=0)
    out = np.zeros(n_fft + hop * (n_frames - 1), dtype=np.float32)
    wss = np.zeros_like(out)
    wsq = (win * win).astype(np.float32)
    for k in range(n_frames):
        out[k * hop:k * hop + n_fft] += frames[:, k].astype(np.float32)
        wss[k * hop:k * hop + n_fft] += wsq
    nz = wss > np.finfo(np.float32).tiny
    out[nz] /= wss[nz]
    return out[n_fft // 2:len(out) - n_fft // 2]


@register_vocoder
class HifiGAN:
    """vocoders/hifigan.py:40-69.  HifiGAN() discovers the checkpoint like the reference does (hparams['vocoder_ckpt']: config.yaml +
    newest model_ckpt_steps_*.ckpt, else config.json + generator_v1) and reads hparams['use_nsf']; HifiGAN(model, device, use_nsf)
    wraps a generator that is already loaded.  spec2wav(mel [T,80], f0=[T]) -> wav [T*hop] as a numpy array."""

    def __init__(self, model: Optional[HifiGanGenerator] = None, device=None, use_nsf: Optional[bool] = None):
        from .hparams import hparams
        if model is None:
            import glob
            import os
            import re
            base_dir = hparams['vocoder_ckpt']
            config_path = f'{base_dir}/config.yaml'
            if os.path.exists(config_path):
                ckpt = sorted(glob.glob(f'{base_dir}/model_ckpt_steps_*.ckpt'),
                              key=lambda x: int(re.findall(rf'{re.escape(base_dir)}/model_ckpt_steps_(\d+).ckpt', x)[0]))[-1]
                print('| load HifiGAN: ', ckpt)
            else:
                config_path, ckpt = f'{base_dir}/config.json', f'{base_dir}/generator_v1'
                if not os.path.exists(config_path):
                    raise FileNotFoundError(f'no config.yaml / config.json under {base_dir}')
            model, self.config, device = load_model(config_path, ckpt, device)
        if device is None:
            device = 'cuda'
        self.model = model.eval().to(device)
        self.device = torch.device(device)
        self.use_nsf = bool(hparams.get('use_nsf')) if use_nsf is None else use_nsf

    @classmethod
    def from_state_dict(cls, config: dict, state: dict, device='cuda', use_nsf: bool = False):
        """load_model (vocoders/hifigan.py:17-32): strict load of a weight-normed checkpoint state, then remove_weight_norm()."""
        model = HifiGanGenerator(config)
        model.load_state_dict(state, strict=True)
        model.remove_weight_norm()
        return cls(model, device, use_nsf)

    def spec2wav(self, mel, **kwargs):
        from .hparams import hparams
        with torch.no_grad():
            c = torch.as_tensor(mel, dtype=torch.float32).unsqueeze(0).transpose(2, 1).to(self.device)
            f0 = kwargs.get('f0')
            if f0 is not None and self.use_nsf:
                f0 = torch.as_tensor(f0, dtype=torch.float32)[None, :].to(self.device)
                y = self.model(c, f0).view(-1)
            else:
                y = self.model(c).view(-1)
        wav_out = y.cpu().numpy()
        if hparams.get('vocoder_denoise_c', 0.0) > 0:                       # vocoders/hifigan.py:63-64
            wav_out = denoise(wav_out, v=hparams['vocoder_denoise_c'], fft_size=hparams['fft_size'], hop_size=hparams['hop_size'],
                              win_size=hparams['win_size'])
        return wav_out
