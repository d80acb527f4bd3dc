"""Test infrastructure for the vocoder (row f2): a torch CPU restatement of the FORMULAS include/dsv.h documents for each
entry point (not of the kernels).  Swapped into diffsinger_amd.vocoder.HifiGanGenerator in place of the C-ABI ops so that the
host orchestration (weight-norm fold, polyphase form of the transposed convolutions, call order, residual / running-sum /
divide wiring, source-module plumbing) is checked against the oracle without a GPU.  Lives under tests/ only."""
import numpy as np
import torch
import torch.nn.functional as F

from diffsinger_amd.vocoder import padded_samples


class HeaderFormulaOps:
    """Same method surface as diffsinger_amd.vocoder._HipOps; `pack` is the identity (the emulation reads the plain weight)."""

    def pack(self, w):
        return w.clone()

    def pad_rows(self, x):
        B, C, L = x.shape
        out = torch.zeros(B, C, padded_samples(L))
        out[:, :, :L] = x
        return out

    def conv(self, x, L_in, wp, bias, rows, ci, k, pad, dil, up=1, pre_slope=1.0, residual=None, sum_in=None, divide=1.0, act=0):
        assert x.shape[2] == padded_samples(L_in) and float(x[:, :, L_in:].abs().sum()) == 0.0, 'input tail must be zero'
        assert pad <= 48 and (k - 1) * dil - pad <= 48
        xin = x[:, :, :L_in]
        if pre_slope != 1.0:
            xin = torch.where(xin > 0, xin, xin * pre_slope)
        right = (k - 1) * dil - pad
        y = F.conv1d(F.pad(xin, (pad, right)), wp, None, dilation=dil)                  # y[row][q] = sum W[row][ci][t] in[q + t dil - pad]
        B = x.shape[0]
        co = rows // up
        y = y.reshape(B, co, up, L_in).permute(0, 1, 3, 2).reshape(B, co, L_in * up)     # n = q * up + phase
        Lo = L_in * up
        v = y + (bias[None, :, None] if bias is not None else 0.0)
        if residual is not None:
            v = v + residual[:, :, :Lo]
        if sum_in is not None:
            v = sum_in[:, :, :Lo] + v
        if divide != 1.0:
            v = v / divide
        if act == 1:
            v = torch.tanh(v)
        out = torch.zeros(B, co, padded_samples(Lo))
        out[:, :, :Lo] = v
        return out

    def conv_multi(self, xs, L_in, items, up=1, pre_slope=1.0):
        """dsv_conv1d_multi: convolution g is dsv_conv1d on its own operands."""
        return [self.conv(x, L_in, it['wp'], it.get('bias'), it['rows'], it['ci'], it['k'], it['pad'], it['dil'], up=up, pre_slope=pre_slope,
                          residual=it.get('residual'), sum_in=it.get('sum_in'), divide=it.get('divide', 1.0), act=it.get('act', 0)) for x, it in zip(xs, items)]

    def chain_fold(self, C):
        return 0                                                      # the fused ResBlock1 chains (dsv_resblock_chain) exist on the device only

    fold = True                                                       # mirrors dsv_fold_factor / dsv_set_fold

    def fold_factor(self, co, ci, k, dil):
        if not self.fold or ci > 16 or (k - 1) * dil // 2 > 25:
            return 1
        return 4 if co <= 8 else 2 if co <= 16 else 1

    def conv_folded(self, x, L, wp, bias, co, ci, k, F, dil, pre_slope=1.0, residual=None, sum_in=None, divide=1.0, act=0):
        """The formula of dsv_conv1d_folded, literally: columns c, pos(c), the F rows per channel, gathered inputs."""
        assert wp.shape == (co * F, ci, k + F - 1)
        LS = padded_samples(L)
        assert x.shape[2] == LS and float(x[:, :, L:].abs().sum()) == 0.0
        pad, KT, fd = (k - 1) * dil // 2, k + F - 1, F * dil
        xin = x[:, :, :L]
        if pre_slope != 1.0:
            xin = torch.where(xin > 0, xin, xin * pre_slope)
        groups = (LS + fd - 1) // fd
        c = torch.arange(groups * dil)
        pos = (c // dil) * fd + c % dil
        idx = pos[None, :] + (torch.arange(KT) * dil)[:, None] - pad                   # [KT][cols]
        valid = ((idx >= 0) & (idx < L)).float()
        xg = xin[:, :, idx.clamp(0, L - 1)] * valid                                    # [B][ci][KT][cols]
        y = torch.einsum('rcs,bcsn->brn', wp, xg)                                      # [B][co * F][cols]
        B = x.shape[0]
        full = torch.zeros(B, co, groups * fd)
        hits = torch.zeros(groups * fd)
        for e in range(F):
            n = pos + e * dil
            full[:, :, n] = y[:, e::F, :]
            hits[n] += 1
        assert bool((hits == 1).all())                                                 # every sample produced exactly once
        v = full[:, :, :L] + (bias[None, :, None] if bias is not None else 0.0)
        if residual is not None:
            v = v + residual[:, :, :L]
        if sum_in is not None:
            v = sum_in[:, :, :L] + v
        if divide != 1.0:
            v = v / divide
        if act == 1:
            v = torch.tanh(v)
        out = torch.zeros(B, co, LS)
        out[:, :, :L] = v
        return out

    def noise_conv(self, har, L_har, w, bias, stride, pad, L_out):
        y = F.conv1d(har[:, None, :L_har], w[:, None, :], bias, stride=stride, padding=pad)
        assert y.shape[2] == L_out
        out = torch.zeros(y.shape[0], y.shape[1], padded_samples(L_out))
        out[:, :, :L_out] = y
        return out

    def sine_source(self, f0, rand_ini, noise, lin_w, lin_b, up, sr, sine_amp, noise_std, thr):
        sw = chunked_sine(f0.numpy(), rand_ini.numpy(), up, float(sr), float(sine_amp))           # [B][H][L]
        B, T = f0.shape
        L = T * up
        f0u = f0.repeat_interleave(up, dim=1)                                                    # index // up
        uv = (f0u > thr).float()[:, :, None]
        namp = uv * noise_std + ((1 - uv) * sine_amp) / 3
        s = torch.from_numpy(sw).permute(0, 2, 1) * uv + namp * noise
        har = torch.tanh(F.linear(s, lin_w[None, :], lin_b))[:, :, 0]
        out = torch.zeros(B, padded_samples(L))
        out[:, :L] = har
        return out


def _mod1(a):
    m = np.fmod(a, np.float32(1.0))
    return np.where((m != 0) & (m < 0), m + np.float32(1.0), m).astype(np.float32)


def chunked_sine(f0, rand_ini, up, sr, sine_amp, nthreads=256):
    """numpy model of k_voc_sine's algorithm: 256 contiguous pieces per (utterance, harmonic), two exclusive scans of fp64
    piece sums, every element rounded to fp32 from the fp64 running sum - including the piece-boundary handling."""
    B, T = f0.shape
    H = rand_ini.shape[1]
    L = T * up
    out = np.zeros((B, H, L), np.float32)
    per = (L + nthreads - 1) // nthreads
    for b in range(B):
        f0u = np.repeat(f0[b].astype(np.float32), up)
        for h in range(H):
            f = f0u * np.float32(h + 1) if h > 0 else f0u
            rad = _mod1((f / np.float32(sr)).astype(np.float32))
            rad[0] = np.float32(rad[0] + (np.float32(0) if h == 0 else rand_ini[b, h]))
            bounds = [(min(L, k * per), min(L, min(L, k * per) + per)) for k in range(nthreads)]
            s1 = np.array([rad[a:e].astype(np.float64).sum() if e > a else 0.0 for a, e in bounds])      # order inside a piece: sequential
            base1 = np.concatenate([[0.0], np.cumsum(s1)[:-1]])
            shift = np.zeros(L, np.float32)
            s2 = np.zeros(nthreads)
            for k, (a, e) in enumerate(bounds):
                if e <= a:
                    continue
                c = base1[k] + np.cumsum(rad[a:e].astype(np.float64))
                over = _mod1(c.astype(np.float32))
                prev = np.concatenate([_mod1(np.array([base1[k]], np.float64).astype(np.float32)), over[:-1]])
                sh = np.where((over - prev) < 0, np.float32(-1), np.float32(0)).astype(np.float32)
                if a == 0:
                    sh[0] = 0
                shift[a:e] = sh
                s2[k] = (rad[a:e] + sh).astype(np.float32).astype(np.float64).sum()
            base2 = np.concatenate([[0.0], np.cumsum(s2)[:-1]])
            for k, (a, e) in enumerate(bounds):
                if e <= a:
                    continue
                c2 = base2[k] + np.cumsum((rad[a:e] + shift[a:e]).astype(np.float32).astype(np.float64))
                ph = c2.astype(np.float32)
                arg = (ph * np.float32(2.0)) * np.float32(np.pi)
                out[b, h, a:e] = np.sin(arg.astype(np.float32)).astype(np.float32) * np.float32(sine_amp)
    return out


def draws_like_reference(seed, B, L, H=9):
    """The source module's three draws from torch's global CPU generator in the reference's order (source.py:57, :131, :529)."""
    torch.manual_seed(seed)
    rand_ini = torch.rand(B, H)
    noise = torch.randn(B, L, H)
    torch.randn(B, L, 1)
    return rand_ini, noise
