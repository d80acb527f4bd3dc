"""CPU study kept as a test ("beyond the fp32-MFMA ceiling"): would an fp32 GEMM evaluated as bf16 plane products on the
bf16 matrix pipe keep the sampler within the path's error budget?  The oracle's convolutions are replaced by their split form - operands
split exactly into three bf16 planes, the plane products (exact in fp32) accumulated in fp32 - and the WHOLE K=100 DDPM golden case
generated from the reference is re-run:
    six products (i + j <= 2)   fp32-class: the mel stays within a few 1e-6 of the reference fixture (the fp32 path itself: 3.3e-6 on the GPU)
    three products (i + j <= 1) ~16 mantissa bits per product: still inside the 1e-4 budget, ~10x the fp32 error
Nothing here touches the product; tools/mfma_split_probe.hip measures the same two forms on the hardware."""
import types
from unittest import mock

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import diffnet_oracle as O
from tests import helpers as H

SIX = [(0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)]          # smallest terms first
THREE = [(0, 1), (1, 0), (0, 0)]


def split3(t):
    a = t.bfloat16().float()
    r = t - a
    b = r.bfloat16().float()
    return a, b, (r - b).bfloat16().float()


def make_shim(terms):
    real = F.conv1d

    def conv1d(x, w, b=None, **kw):
        xs, ws = split3(x), split3(w)
        acc = None
        for i, j in terms:
            y = real(xs[j], ws[i], None, **kw)
            acc = y if acc is None else acc + y
        return acc if b is None else acc + b[None, :, None]

    shim = types.SimpleNamespace(**{k: getattr(F, k) for k in ('relu', 'linear', 'softplus', 'pad')})
    shim.conv1d = conv1d
    return shim


def test_the_three_planes_are_exact():
    x = torch.randn(10000) * torch.logspace(-6, 6, 10000)
    a, b, c = split3(x)
    assert torch.equal(a.double() + b.double() + c.double(), x.double())


@pytest.mark.parametrize('terms,budget', [(SIX, 1e-5), (THREE, 1e-4)], ids=['six_products', 'three_products'])
def test_split_gemms_keep_the_k100_sampler_within_budget(terms, budget):
    g = H.load_golden('ddpm_lj_k100')
    with mock.patch.object(O, 'F', make_shim(terms)):
        out = H.run_oracle_case('ddpm_lj_k100')
    err = float(np.abs(out - g['out']).max())
    print(f'{len(terms)} plane products: max-abs mel error vs the reference fixture after 100 DDPM steps = {err:.3e}')
    assert err < budget, err


# ---- the pair format (csrc/dsd_loop_split.hpp, SplitPipeF): two scaled fp16 planes, x = h0 + 2^-11 h1, product = h0 g0 + 2^-11 (h0 g1 + h1 g0) ----
def split2h(t):
    h0 = t.half().float()
    h1 = ((t - h0) * 2048.0).half().float()
    return h0, h1


def make_pair_shim():
    real = F.conv1d

    def conv1d(x, w, b=None, **kw):
        (g0, g1), (h0, h1) = split2h(x), split2h(w)
        cross = real(g0, h1, None, **kw) + real(g1, h0, None, **kw)          # the loop keeps these in their own accumulators
        acc = real(g0, h0, None, **kw) + cross * (1.0 / 2048.0)
        return acc if b is None else acc + b[None, :, None]

    shim = types.SimpleNamespace(**{k: getattr(F, k) for k in ('relu', 'linear', 'softplus', 'pad')})
    shim.conv1d = conv1d
    return shim


def test_the_pair_keeps_22_bits_where_the_first_plane_is_normal():
    x = torch.randn(100000) * torch.logspace(-3, 4, 100000)                  # 1e-3 .. 1e4: weights to activations
    h0, h1 = split2h(x)
    err = (h0.double() + h1.double() / 2048.0 - x.double()).abs()
    normal = x.abs() >= 2.0 ** -14                                           # fp16's smallest normal: below it h0 is a denormal
    assert float((err[normal] / x.double().abs()[normal]).max()) <= 2.0 ** -22
    assert float(err[~normal].max()) <= 2.0 ** -35                           # absolute, not relative: half a denormal step of the scaled second plane
    assert float(h1[normal & (h1 != 0)].abs().min()) >= 2.0 ** -24 and float(h1.abs().max()) < 65504.0


def test_pair_gemms_keep_the_k100_sampler_within_budget():
    """The whole K = 100 DDPM golden case with EVERY convolution of the oracle in the pair format (the loop only does the layers' two): the mel
    stays within a few 1e-6 of the reference fixture - the class of the six-product bf16 split and of the fp32 path itself."""
    g = H.load_golden('ddpm_lj_k100')
    with mock.patch.object(O, 'F', make_pair_shim()):
        out = H.run_oracle_case('ddpm_lj_k100')
    err = float(np.abs(out - g['out']).max())
    print(f'pair format (3 fp16 plane products): max-abs mel error vs the reference fixture after 100 DDPM steps = {err:.3e}')
    assert err < 1e-5, err
