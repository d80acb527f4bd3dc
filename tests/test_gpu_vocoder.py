"""GPU parity of the HIP vocoder through the C ABI of include/dsv.h:
  * every entry point against the formula its header documents (tests/voc_helpers.HeaderFormulaOps, torch CPU fp32) on shapes that
    exercise the three kernel tilings, ragged lengths, both store paths of the transposed convolution and every fused neighbour;
  * the whole generator against the fixtures recorded from the REAL reference generator (tests/golden/hifigan_*.npz) and against
    the oracle on a longer input.
Tolerance: the generator is fp32 end to end (fp32 MFMA, no reduced precision); the fixture differs from an fp64 evaluation of the
same network by 5.5e-7, the kernels sum in a different order: 2e-5 absolute on a waveform in [-1, 1] (observed values are printed)."""
import os

import numpy as np
import pytest
import torch

from diffsinger_amd import _lib
from diffsinger_amd.vocoder import HifiGAN, HifiGanGenerator, _HipOps, fold_weight, padded_samples, polyphase_weight
from oracle import hifigan_oracle as HO
from oracle.make_golden_hifigan import CASES, CONFIG, inputs
from tests.voc_helpers import HeaderFormulaOps, draws_like_reference

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda:0'
TOL = 2e-5


def _cm(x, L):
    out = torch.zeros(x.shape[0], x.shape[1], padded_samples(L))
    out[:, :, :L] = x
    return out


# (Ci, rows, K, dil, L, up, pre_slope, residual, sum_in, divide, act)  -  pad follows the module: 'same' for a conv, polyphase for up > 1
CONV_CASES = [
    dict(ci=8, co=8, k=11, dil=5, L=1000, slope=0.1, res=True, acc=True, div=3.0),          # stage-4 resblock tail: <4,4> tiling, ragged L
    dict(ci=8, co=8, k=3, dil=1, L=513, slope=0.1, res=True),                                # one sample into the second workgroup
    dict(ci=16, co=16, k=7, dil=3, L=700, slope=0.1),
    dict(ci=32, co=32, k=11, dil=1, L=96, slope=0.1, res=True, acc=True),
    dict(ci=64, co=64, k=7, dil=5, L=300, slope=0.1, res=True),                             # <2,2> tiling
    dict(ci=80, co=128, k=7, dil=1, L=37),                                                   # conv_pre: <1,1> tiling, Ci = 80
    dict(ci=512, co=256, k=3, dil=1, L=50),                                                  # more than one channel slab, two row tiles
    dict(ci=8, co=1, k=7, dil=1, L=2000, slope=0.01, act=1),                                 # conv_post -> tanh
    dict(ci=12, co=20, k=5, dil=2, L=77, slope=0.1),                                         # nothing a multiple of 8 / 32
    dict(ci=32, co=32, k=7, dil=12, L=900, slope=0.1, res=True),                            # reach 36 > 28: the wide-window instantiation (official v3)
    dict(ci=64, co=64, k=9, dil=12, L=300, slope=0.1),                                      # reach 48, <2,2> tiling
    dict(ci=128, co=128, k=7, dil=12, L=100, slope=0.1, res=True, acc=True),               # reach 36, <1,1> tiling
]


@pytest.mark.parametrize('c', CONV_CASES, ids=lambda c: f"ci{c['ci']}co{c['co']}k{c['k']}d{c['dil']}L{c['L']}")
def test_conv1d_matches_header_formula(c):
    g = torch.Generator().manual_seed(c['ci'] * 1000 + c['L'])
    B, L = 2, c['L']
    w = torch.randn(c['co'], c['ci'], c['k'], generator=g) / (c['ci'] * c['k']) ** 0.5
    bias = torch.randn(c['co'], generator=g)
    x = _cm(torch.randn(B, c['ci'], L, generator=g), L)
    res = _cm(torch.randn(B, c['co'], L, generator=g), L) if c.get('res') else None
    acc = _cm(torch.randn(B, c['co'], L, generator=g), L) if c.get('acc') else None
    pad = (c['k'] - 1) * c['dil'] // 2
    kw = dict(pre_slope=c.get('slope', 1.0), divide=c.get('div', 1.0), act=c.get('act', 0))
    want = HeaderFormulaOps().conv(x, L, w, bias, c['co'], c['ci'], c['k'], pad, c['dil'], residual=res, sum_in=acc, **kw)
    ops = _HipOps()
    d = lambda t: None if t is None else t.to(DEV)
    got = ops.conv(d(x), L, ops.pack(d(w)), d(bias), c['co'], c['ci'], c['k'], pad, c['dil'], residual=d(res), sum_in=d(acc), **kw).cpu()
    assert got.shape == want.shape
    assert float(got[:, :, L:].abs().max()) == 0.0 if got.shape[2] > L else True            # zero tail
    err = float((got - want).abs().max())
    print('conv err', err)
    assert err < 1e-5, err


FOLD_CASES = [
    dict(ci=8, co=8, k=11, dil=5, L=3000, res=True, acc=True, div=3.0),                     # stage-4 resblock tail, F = 4, scattered stores
    dict(ci=8, co=8, k=3, dil=1, L=1000, res=True),                                         # F = 4, 16-byte store path
    dict(ci=8, co=8, k=7, dil=3, L=513),
    dict(ci=8, co=8, k=11, dil=1, L=1025, res=True, acc=True),
    dict(ci=16, co=16, k=11, dil=5, L=2000, res=True),                                      # F = 2
    dict(ci=16, co=16, k=7, dil=1, L=700, res=True, acc=True, div=3.0),
    dict(ci=16, co=16, k=3, dil=3, L=100),
    dict(ci=8, co=1, k=7, dil=1, L=2000, slope=0.01, act=1),                                 # conv_post -> tanh
    dict(ci=12, co=5, k=5, dil=2, L=300),                                                    # ragged channel counts, even dilation
]


@pytest.fixture
def fold_on():
    lib = _lib.load()
    lib.dsv_set_fold(1)
    yield lib
    lib.dsv_set_fold(1)


@pytest.mark.parametrize('c', FOLD_CASES, ids=lambda c: f"ci{c['ci']}co{c['co']}k{c['k']}d{c['dil']}L{c['L']}")
def test_folded_conv_matches_the_plain_convolution(c, fold_on):
    """dsv_conv1d_folded against the PLAIN convolution formula (original weights), not against its own folded restatement."""
    g = torch.Generator().manual_seed(c['co'] * 1000 + c['L'])
    B, L = 2, c['L']
    w = torch.randn(c['co'], c['ci'], c['k'], generator=g) / (c['ci'] * c['k']) ** 0.5
    bias = torch.randn(c['co'], generator=g)
    x = _cm(torch.randn(B, c['ci'], L, generator=g), L)
    res = _cm(torch.randn(B, c['co'], L, generator=g), L) if c.get('res') else None
    acc = _cm(torch.randn(B, c['co'], L, generator=g), L) if c.get('acc') else None
    kw = dict(pre_slope=c.get('slope', 0.1), divide=c.get('div', 1.0), act=c.get('act', 0))
    want = HeaderFormulaOps().conv(x, L, w, bias, c['co'], c['ci'], c['k'], (c['k'] - 1) * c['dil'] // 2, c['dil'], residual=res, sum_in=acc, **kw)
    ops = _HipOps()
    F = ops.fold_factor(c['co'], c['ci'], c['k'], c['dil'])
    assert F == (4 if c['co'] <= 8 else 2)
    d = lambda t: None if t is None else t.to(DEV)
    got = ops.conv_folded(d(x), L, ops.pack(fold_weight(w, F).to(DEV)), d(bias), c['co'], c['ci'], c['k'], F, c['dil'], residual=d(res),
                          sum_in=d(acc), **kw).cpu()
    assert got.shape == want.shape
    if got.shape[2] > L:
        assert float(got[:, :, L:].abs().max()) == 0.0
    err = float((got - want).abs().max())
    print('folded conv err', err)
    assert err < 1e-5, err


def test_fold_switch():
    lib = _lib.load()
    lib.dsv_set_fold(0)
    assert lib.dsv_fold_factor(8, 8, 11, 5) == 1
    lib.dsv_set_fold(1)
    assert lib.dsv_fold_factor(8, 8, 11, 5) == 4 and lib.dsv_fold_factor(16, 16, 3, 1) == 2 and lib.dsv_fold_factor(32, 32, 3, 1) == 1
    assert lib.dsv_fold_factor(8, 32, 3, 1) == 1                                             # more input channels than the folded tile stages


@pytest.mark.parametrize('ci,co,u,k,L', [(128, 64, 8, 16, 37), (64, 32, 8, 16, 300), (32, 16, 2, 4, 1000), (16, 8, 2, 4, 3001), (24, 8, 4, 8, 100)])
def test_transposed_conv_matches_torch(ci, co, u, k, L):
    g = torch.Generator().manual_seed(ci + u + L)
    B = 2
    w = torch.randn(ci, co, k, generator=g) / (ci * k / u) ** 0.5
    bias = torch.randn(co, generator=g)
    x = torch.randn(B, ci, L, generator=g)
    src = torch.randn(B, co, L * u, generator=g)
    xa = torch.where(x > 0, x, x * 0.1)
    want = torch.nn.functional.conv_transpose1d(xa, w, bias, stride=u, padding=(k - u) // 2) + src
    wp, pad = polyphase_weight(w, u, (k - u) // 2)
    ops = _HipOps()
    got = ops.conv(_cm(x, L).to(DEV), L, ops.pack(wp.to(DEV)), bias.to(DEV), co * u, ci, wp.shape[2], pad, 1, up=u, pre_slope=0.1,
                   residual=_cm(src, L * u).to(DEV)).cpu()
    assert got.shape == (B, co, padded_samples(L * u))
    assert float(got[:, :, L * u:].abs().sum()) == 0.0
    err = float((got[:, :, :L * u] - want).abs().max())
    print('convT err', err)
    assert err < 1e-5, err


def test_pad_rows_and_noise_conv():
    g = torch.Generator().manual_seed(5)
    ops, emu = _HipOps(), HeaderFormulaOps()
    x = torch.randn(3, 80, 45, generator=g)
    assert torch.equal(ops.pad_rows(x.to(DEV)).cpu(), emu.pad_rows(x))
    Lh = 24 * 256
    har = _cm(torch.randn(2, 1, Lh, generator=g), Lh)[:, 0].contiguous()
    for C, s in [(64, 32), (32, 4), (16, 2)]:
        w = torch.randn(C, 2 * s, generator=g) * 0.1
        b = torch.randn(C, generator=g)
        want = emu.noise_conv(har, Lh, w, b, s, s // 2, Lh // s)
        got = ops.noise_conv(har.to(DEV), Lh, w.to(DEV), b.to(DEV), s, s // 2, Lh // s).cpu()
        assert float((got - want).abs().max()) < 2e-6
    w = torch.randn(8, 1, generator=g)
    b = torch.randn(8, generator=g)
    assert float((ops.noise_conv(har.to(DEV), Lh, w.to(DEV), b.to(DEV), 1, 0, Lh).cpu() - emu.noise_conv(har, Lh, w, b, 1, 0, Lh)).abs().max()) < 1e-6


def test_sine_source_matches_the_reference_module():
    """dsv_sine_source against SourceModuleHnNSF through the oracle (same f0, same draws)."""
    B, T, up = 2, 40, 256
    g = torch.Generator().manual_seed(3)
    f0 = torch.rand(B, T, generator=g) * 500 + 60
    f0[0, 5:9] = 0
    f0[1, 30:] = 0
    p = {'m_source.l_linear.weight': torch.randn(1, 9, generator=g) * 0.5, 'm_source.l_linear.bias': torch.randn(1, generator=g) * 0.1}
    torch.manual_seed(21)
    f0u = torch.nn.functional.interpolate(f0[:, None], scale_factor=float(up), mode='nearest').transpose(1, 2)
    want = HO.source_module(p, f0u, 24000)[:, :, 0]
    rand_ini, noise = draws_like_reference(21, B, T * up)
    got = _HipOps().sine_source(f0.to(DEV), rand_ini.to(DEV), noise.to(DEV), p['m_source.l_linear.weight'].reshape(-1).to(DEV),
                                p['m_source.l_linear.bias'].to(DEV), up, 24000, 0.1, 0.003, 0.0).cpu()
    assert float(got[:, T * up:].abs().sum()) == 0.0
    err = float((got[:, :T * up] - want).abs().max())
    print('source err', err)
    assert err < 5e-6, err


def _generator(case, weight_norm=False):
    h = dict(CONFIG, use_pitch_embed=case['nsf'])
    p = HO.synth_generator_params(h, case['seed'] + 1000)
    m = HifiGanGenerator(h)
    if weight_norm:
        sd = {}
        for k, v in p.items():
            if k.endswith('.weight') and not k.startswith(('noise_convs', 'm_source')):
                sd[k[:-7] + '.weight_v'] = v.clone()
                sd[k[:-7] + '.weight_g'] = v.flatten(1).norm(dim=1).reshape(-1, 1, 1).clone()
            else:
                sd[k] = v.clone()
        m.load_state_dict(sd, strict=True)
    else:
        m.load_state_dict(p, strict=True)
    return h, p, m.to(DEV)


@pytest.mark.parametrize('name', ['hifigan_plain', 'hifigan_nsf'])
@pytest.mark.parametrize('weight_norm,fold', [(False, 1), (True, 1), (False, 0)])
def test_generator_matches_reference_fixture(name, weight_norm, fold, fold_on):
    fold_on.dsv_set_fold(fold)                                        # narrow stages on the folded kernel (default) / on dsv_conv1d
    case = CASES[name]
    h, p, m = _generator(case, weight_norm)
    mel, f0 = inputs(case)
    kw = {}
    if f0 is not None:
        ri, nz = draws_like_reference(case['seed'], case['B'], case['T'] * 256)
        kw = dict(rand_ini=ri.to(DEV), noise=nz.to(DEV))
    wav = m(mel.to(DEV), None if f0 is None else f0.to(DEV), **kw).cpu().numpy()
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))['wav']
    assert wav.shape == g.shape
    err = float(np.abs(wav - g).max())
    print(name, 'max-abs err vs reference fixture', err)
    assert err < TOL, err
    assert any('libdsdenoise' in ln for ln in open('/proc/self/maps'))


def test_generator_longer_input_matches_oracle_and_is_deterministic():
    case = dict(nsf=True, B=3, T=150, seed=77)
    h, p, m = _generator(case)
    g = torch.Generator().manual_seed(5)
    mel = torch.randn(case['B'], 80, case['T'], generator=g)
    f0 = torch.rand(case['B'], case['T'], generator=g) * 400 + 70
    f0[0, 40:60] = 0
    torch.manual_seed(9)
    want = HO.generator(p, h, mel, f0)
    ri, nz = draws_like_reference(9, case['B'], case['T'] * 256)
    a = m(mel.to(DEV), f0.to(DEV), rand_ini=ri.to(DEV), noise=nz.to(DEV)).cpu()
    b = m(mel.to(DEV), f0.to(DEV), rand_ini=ri.to(DEV), noise=nz.to(DEV)).cpu()
    assert torch.equal(a, b)
    err = float((a - want).abs().max())
    print('T=150 NSF err vs oracle', err)
    assert err < TOL, err
    # utterances are independent: the first one alone gives the same samples
    solo = m(mel[:1].to(DEV), f0[:1].to(DEV), rand_ini=ri[:1].to(DEV), noise=nz[:1].to(DEV)).cpu()
    assert torch.equal(solo, a[:1])


def test_spec2wav_wrapper_and_own_draws():
    case = CASES['hifigan_nsf']
    h, p, m = _generator(case)
    voc = HifiGAN(m, DEV, use_nsf=True)
    mel, f0 = inputs(case)
    wav = voc.spec2wav(mel[0].t().numpy(), f0=f0[0].numpy())
    assert wav.shape == (case['T'] * 256,) and wav.dtype == np.float32 and np.isfinite(wav).all() and np.abs(wav).max() <= 1.0
    plain = HifiGAN(m, DEV, use_nsf=False).spec2wav(mel[0].t().numpy(), f0=f0[0].numpy())          # use_nsf off: f0 ignored (vocoders/hifigan.py:62)
    assert plain.shape == wav.shape and not np.array_equal(plain, wav)


# ---- fused ResBlock1 chains (csrc/voc_chain.hpp, dsv_resblock_chain) ----------------------------------------------------------------

DEFAULT_CHAIN_VARIANT = {8: (2, 1), 16: (2, 1), 32: (4, 1)}       # csrc/voc_abi.hpp g_chain_variant


@pytest.fixture
def chain_default():
    from diffsinger_amd import vocoder
    vocoder.set_chain_mode(None)
    yield vocoder
    vocoder.set_chain_mode(None)


@pytest.mark.parametrize('stage,L', [(3, 5000), (3, 896), (3, 33), (2, 3001), (2, 640), (1, 1500), (1, 449), (0, 300)])
@pytest.mark.parametrize('mode', ['stage', 'resblock', 'pair', 'merged0', 'merged1', 'merged2'])
def test_resblock_chain_is_bit_identical_to_the_single_convolutions(stage, L, mode, chain_default):
    """One stage's `(sum_j resblock_j(x)) / 3` (hifigan.py:161-166, :54-61) with the ResBlock1 chains fused in LDS - the whole stage, one
    resblock, or one conv pair per launch, or (round 6) two resblocks merged into one launch and the third (0, 1 or 2) forming the sum -
    against one launch per convolution: same chunk order, same epilogue arithmetic -> the same BITS.
    Ragged lengths: several tiles with a partial last one, exactly one tile, a tile shorter than the halo; stage 0 (64 channels) has no chain
    kernel and must fall through unchanged."""
    case = dict(nsf=False, B=2, T=8, seed=41 + stage)
    h, p, m = _generator(case)
    m(torch.zeros(1, 80, 4, device=DEV))                              # creates the op table
    C = 128 >> (stage + 1)
    g = torch.Generator().manual_seed(100 * stage + L)
    x = _cm(torch.randn(3, C, L, generator=g), L).to(DEV)
    chain_default.set_chain_mode('off')
    want = m._stage_resblocks(stage, x, L)
    chain_default.set_chain_mode(mode)
    got = m._stage_resblocks(stage, x, L)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and float(got[:, :, L:].abs().max() if got.shape[2] > L else 0) == 0
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize('C,L', [(64, 300), (64, 1000), (32, 515), (128, 77)])
def test_conv1d_multi_is_bit_identical_to_the_single_launches(C, L):
    """dsv_conv1d_multi (include/dsv.h, round 6): three independent convolutions of one shape - kernels 11 / 7 / 3 at dilations 5 / 3 / 1, one with a
    residual, one with residual + running sum + divisor - in ONE launch against three dsv_conv1d launches: the same BITS (same tiles, same chunk
    order); ragged lengths, every tiling (<2,2> at 64 rows, <4,4> at 32, <1,1> at 128); and the argument contract."""
    import ctypes
    from diffsinger_amd.vocoder import DsvConvDesc
    ops = _HipOps()
    g = torch.Generator().manual_seed(C + L)
    B = 2
    xs, items = [], []
    for k, dil in ((11, 5), (7, 3), (3, 1)):
        w = (torch.randn(C, C, k, generator=g) / (C * k) ** 0.5).to(DEV)
        xs.append(_cm(torch.randn(B, C, L, generator=g), L).to(DEV))
        items.append(dict(wp=ops.pack(w), bias=torch.randn(C, generator=g).to(DEV), rows=C, ci=C, k=k, pad=(k - 1) * dil // 2, dil=dil))
    items[1]['residual'] = _cm(torch.randn(B, C, L, generator=g), L).to(DEV)
    items[2]['residual'] = _cm(torch.randn(B, C, L, generator=g), L).to(DEV)
    items[2]['sum_in'] = _cm(torch.randn(B, C, L, generator=g), L).to(DEV)
    items[2]['divide'] = 3.0
    want = [ops.conv(x, L, it['wp'], it['bias'], C, C, it['k'], it['pad'], it['dil'], pre_slope=0.1, residual=it.get('residual'), sum_in=it.get('sum_in'),
                     divide=it.get('divide', 1.0)) for x, it in zip(xs, items)]
    got = ops.conv_multi(xs, L, items, pre_slope=0.1)
    two = ops.conv_multi(xs[:2], L, items[:2], pre_slope=0.1)
    torch.cuda.synchronize()
    for a, b in zip(got + two, want + want[:2]):
        assert torch.equal(a, b), float((a - b).abs().max())
    lib = _lib.load()
    mk = lambda i, out, **kw: DsvConvDesc(kw.get('inp', xs[i].data_ptr()), items[i]['wp'].data_ptr(), items[i]['bias'].data_ptr(), out, kw.get('res'), None,
                                         items[i]['k'], items[i]['pad'], items[i]['dil'], 0, 1.0, 0)
    o0, o1 = torch.empty_like(xs[0]), torch.empty_like(xs[0])
    call = lambda *d: lib.dsv_conv1d_multi(len(d), (DsvConvDesc * len(d))(*d), B, C, C, L, 1, 0.1, None)
    assert call(mk(0, o0.data_ptr()), mk(1, o1.data_ptr())) == 0
    assert call(mk(0, o0.data_ptr()), mk(1, o0.data_ptr())) != 0                                  # two convolutions, one output
    assert call(mk(0, o0.data_ptr()), mk(1, o1.data_ptr(), inp=o0.data_ptr())) != 0              # an output is another convolution's input
    assert call(mk(0, o0.data_ptr()), mk(1, o1.data_ptr(), res=o0.data_ptr())) != 0              # ... or its residual
    assert lib.dsv_conv1d_multi(0, (DsvConvDesc * 1)(mk(0, o0.data_ptr())), B, C, C, L, 1, 0.1, None) != 0
    assert lib.dsv_conv1d_multi(4, (DsvConvDesc * 4)(*[mk(0, o0.data_ptr())] * 4), B, C, C, L, 1, 0.1, None) != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize('ci,rows,L,B', [(32, 32, 65536, 4), (16, 16, 70001, 3), (8, 6, 4100, 9)])
def test_lean_build_of_the_stride2_transposed_convolution_is_bit_identical(ci, rows, L, B):
    """Round 6 (r6_52): the stride-2 transposed convolutions run on a lean build of k_voc_conv (half the tile, a 4-sample halo, LDS by the
    channel count, three workgroups per CU instead of two: 64-69 -> 58-60 us per launch at the bench shape).  Same chunk order: the same BITS as
    the standard build (dsv_set_lean(0)) - the shipped shapes, a ragged length, a row count that does not fill the block."""
    lib = _lib.load()
    ops = _HipOps()
    g = torch.Generator(device=DEV).manual_seed(ci + L)
    k, up = 2, 2
    w = torch.randn(rows, ci, k, generator=g, device=DEV) / (ci * k) ** 0.5
    bias = torch.randn(rows // up, generator=g, device=DEV)
    x = torch.zeros(B, ci, padded_samples(L), device=DEV)
    x[:, :, :L] = torch.randn(B, ci, L, generator=g, device=DEV)
    wp = ops.pack(w)
    try:
        assert lib.dsv_set_lean(0) == 0
        want = ops.conv(x, L, wp, bias, rows, ci, k, 1, 1, up=up, pre_slope=0.1)
        assert lib.dsv_set_lean(1) == 0
        got = ops.conv(x, L, wp, bias, rows, ci, k, 1, 1, up=up, pre_slope=0.1)
        torch.cuda.synchronize()
    finally:
        lib.dsv_set_lean(1)
    assert torch.isfinite(want).all() and float(want.abs().max()) > 0.1
    assert torch.equal(got, want), float((got - want).abs().max())
    assert got.shape[2] == L * up or float(got[:, :, L * up:].abs().max()) == 0


def test_merged_chain_entry_points_contract_and_plan():
    """dsv_resblock_chain_multi / dsv_resblock_chain_sum (include/dsv.h): argument checks, and the host model that picks the split
    (diffsinger_amd.vocoder._merge_plan) on the bench shape - 1 096 / 1 264 / 1 368 workgroups of kernel 3 / 7 / 11 on 512 slots: kernel 11 and
    kernel 3 share a launch (4.8 rounds' worth instead of 3 + 3), kernel 7 sums."""
    from diffsinger_amd.vocoder import DsvChainConv, _merge_plan
    import ctypes
    assert _merge_plan([1096, 1264, 1368], [6 * 18, 6 * 34, 6 * 50], 512) == (1, [2, 0])
    assert _merge_plan([512, 512, 512], [1.0, 2.0, 3.0], 512) is None                      # whole rounds: nothing to gain
    lib = _lib.load()
    x = torch.zeros(1, 32, 1024, device=DEV)
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    wp, b = torch.zeros(65536, device=DEV), torch.zeros(128, device=DEV)
    d = (DsvChainConv * 4)(DsvChainConv(0, 0, 3, 1, 0), DsvChainConv(3072, 32, 3, 1, 0), DsvChainConv(6144, 64, 3, 3, 0), DsvChainConv(9216, 96, 3, 1, 0))
    outs = (ctypes.c_void_p * 2)(o1.data_ptr(), o2.data_ptr())
    same = (ctypes.c_void_p * 2)(o1.data_ptr(), o1.data_ptr())
    isin = (ctypes.c_void_p * 2)(o1.data_ptr(), x.data_ptr())
    call = lambda ptrs, ng: lib.dsv_resblock_chain_multi(x.data_ptr(), wp.data_ptr(), b.data_ptr(), ptrs, 1, 32, 1000, ng, 1, d, 0.1, None)
    assert call(outs, 2) == 0
    assert call(same, 2) != 0 and call(isin, 2) != 0 and call(outs, 4) != 0 and call(outs, 0) != 0
    out = torch.empty_like(x)
    summ = lambda o, a, c: lib.dsv_resblock_chain_sum(x.data_ptr(), wp.data_ptr(), b.data_ptr(), o, a, c, 0, 1, 32, 1000, 1, d, 0.1, 3.0, None)
    assert summ(out.data_ptr(), o1.data_ptr(), o2.data_ptr()) == 0
    assert summ(o1.data_ptr(), o1.data_ptr(), o2.data_ptr()) != 0 and summ(out.data_ptr(), None, o2.data_ptr()) != 0 and summ(x.data_ptr(), o1.data_ptr(), o2.data_ptr()) != 0
    torch.cuda.synchronize()


CHAIN_VARIANTS = [(32, 4, 0), (32, 4, 1), (16, 2, 0), (16, 2, 1), (16, 4, 1), (8, 2, 0), (8, 2, 1), (8, 4, 1)]


@pytest.mark.parametrize('C,nb,ip', CHAIN_VARIANTS)
@pytest.mark.parametrize('mode', ['stage', 'pair', 'merged1'])
def test_every_chain_variant_is_bit_identical(C, nb, ip, mode, chain_default):
    """dsv_set_chain_variant (include/dsv.h): the window (nb column blocks per wave) and one tile rewritten in place / two tiles select
    between instantiations of the same sums in the same order - several tiles with a partial last one, and with a running sum coming in.
    'merged1': the merged launches run the in-place instantiation with the variant's window (the MG kernels of csrc/voc_chain.hpp)."""
    if mode == 'merged1' and not ip:
        pytest.skip('the merged launches exist for the in-place form only')
    lib = _lib.load()
    stage = {32: 1, 16: 2, 8: 3}[C]
    case = dict(nsf=False, B=2, T=8, seed=41 + stage)
    h, p, m = _generator(case)
    m(torch.zeros(1, 80, 4, device=DEV))
    L = {32: 1500, 16: 3001, 8: 5000}[C]
    g = torch.Generator().manual_seed(7 * C + nb + ip)
    x = _cm(torch.randn(3, C, L, generator=g), L).to(DEV)
    chain_default.set_chain_mode('off')
    want = m._stage_resblocks(stage, x, L)
    try:
        assert lib.dsv_set_chain_variant(C, nb, ip) == 0, lib.dsd_last_error()
        chain_default.set_chain_mode(mode)
        got = m._stage_resblocks(stage, x, L)
        again = m._stage_resblocks(stage, x, L)
        torch.cuda.synchronize()
    finally:
        for c in (8, 16, 32):
            lib.dsv_set_chain_variant(c, *DEFAULT_CHAIN_VARIANT[c])
    assert float(got[:, :, L:].abs().max()) == 0
    assert torch.equal(got, want), float((got - want).abs().max())
    assert torch.equal(again, got)
    assert lib.dsv_set_chain_variant(32, 2, 1) != 0 and lib.dsv_set_chain_variant(64, 4, 1) != 0


@pytest.mark.parametrize('mode', [None, 'stage', 'resblock', 'pair'])
def test_generator_with_fused_chains_equals_the_unfused_generator(mode, chain_default):
    case = dict(nsf=True, B=3, T=150, seed=77)
    h, p, m = _generator(case)
    g = torch.Generator().manual_seed(5)
    mel = torch.randn(case['B'], 80, case['T'], generator=g).to(DEV)
    f0 = (torch.rand(case['B'], case['T'], generator=g) * 400 + 70).to(DEV)
    ri, nz = draws_like_reference(9, case['B'], case['T'] * 256)
    chain_default.set_chain_mode('off')
    want = m(mel, f0, rand_ini=ri.to(DEV), noise=nz.to(DEV))
    chain_default.set_chain_mode(mode)
    got = m(mel, f0, rand_ini=ri.to(DEV), noise=nz.to(DEV))
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize('dils,B,T', [([[1, 3, 5], [1, 6, 12], [1, 3, 5]], 2, 40), ([[1, 3, 5], [1, 3, 5], [1, 3, 5]], 1, 7), ([[1, 2], [2, 6], [3, 4]], 3, 33)])
def test_default_launch_forms_on_other_configs_equal_the_unfused_generator(dils, B, T, chain_default):
    """The default launch forms of round 6 - merged chain launches, the level-by-level 64-channel stage - against one launch per convolution
    on generators beyond the shipped one (hifigan.py:104-179 accepts any config): a kernel-7 resblock at dilation 12 (reach 36: no chain
    kernel for ITS stage's resblocks -> the stages fall back, the wide-window instantiations of dsv_conv1d_multi run), one utterance of 7
    frames (every launch a partial tile), two conv pairs per resblock.  Same bits."""
    h = dict(CONFIG, use_pitch_embed=False, resblock_dilation_sizes=dils)
    m = HifiGanGenerator(h)
    m.remove_weight_norm()
    g = torch.Generator().manual_seed(B * 100 + T)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 / max(1, p[0].numel()) ** 0.5 if n.endswith('weight') else 0.1))
    m = m.to(DEV).eval()
    mel = torch.randn(B, 80, T, generator=g).to(DEV)
    chain_default.set_chain_mode('off')
    want = m(mel)
    chain_default.set_chain_mode(None)
    got = m(mel)
    again = m(mel)
    torch.cuda.synchronize()
    assert torch.isfinite(want).all() and float(want.abs().max()) > 1e-3
    assert torch.equal(got, want), float((got - want).abs().max())
    assert torch.equal(again, got)


@pytest.mark.parametrize('stage', [0, 1, 2, 3])
def test_default_launch_forms_at_the_bench_shape(stage, chain_default):
    """8 x 1024 mel frames (bench.py --row vocoder): the stage's resblocks in the default form - several ROUNDS of co-resident workgroups per
    launch, the merged grid's groups meeting in the middle of a round (stage 0: three convolutions per launch) - against one launch per
    convolution.  The small cases above never fill the chip once."""
    case = dict(nsf=False, B=2, T=8, seed=7)
    h, p, m = _generator(case)
    m(torch.zeros(1, 80, 4, device=DEV))
    C = 128 >> (stage + 1)
    L = 1024 * [8, 64, 128, 256][stage]
    g = torch.Generator(device=DEV).manual_seed(stage)
    x = torch.zeros(8, C, padded_samples(L), device=DEV)
    x[:, :, :L] = torch.randn(8, C, L, generator=g, device=DEV)
    chain_default.set_chain_mode('off')
    want = m._stage_resblocks(stage, x, L)
    chain_default.set_chain_mode(None)
    if stage:
        assert m._merge_plan_for(stage, m._chain_prep(stage), 8, L) is not None          # the bench shape IS merged
    got = m._stage_resblocks(stage, x, L)
    torch.cuda.synchronize()
    assert torch.equal(got, want), float((got - want).abs().max())


def test_chain_entry_point_refuses_what_it_cannot_tile():
    from diffsinger_amd.vocoder import DsvChainConv
    lib = _lib.load()
    ok = (DsvChainConv * 2)(DsvChainConv(0, 0, 11, 5, 0), DsvChainConv(3584, 8, 11, 1, 0))
    assert lib.dsv_chain_supported(8, 1, 1, ok) > 0 and lib.dsv_chain_fold(8) == 4 and lib.dsv_chain_fold(64) == 0
    wide = (DsvChainConv * 2)(DsvChainConv(0, 0, 7, 12, 0), DsvChainConv(2560, 8, 7, 1, 0))           # the official v3 generator: 36 samples of reach
    assert lib.dsv_chain_supported(8, 1, 1, wide) == 0
    second_dilated = (DsvChainConv * 2)(DsvChainConv(0, 0, 3, 1, 0), DsvChainConv(1536, 8, 3, 3, 0))
    assert lib.dsv_chain_supported(8, 1, 1, second_dilated) == 0
    assert lib.dsv_chain_supported(64, 1, 1, ok) == 0 and lib.dsv_chain_supported(8, 4, 3, ok) == 0
    x = torch.zeros(1, 8, 64, device=DEV)
    assert lib.dsv_resblock_chain(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, 1, 8, 64, 1, 1, ok, 0.1, 1.0, None) != 0
    assert b'different buffers' in lib.dsd_last_error()
    y = torch.zeros(1, 8, 64, device=DEV)
    assert lib.dsv_resblock_chain(x.data_ptr(), x.data_ptr(), x.data_ptr(), y.data_ptr(), y.data_ptr(), 1, 8, 64, 1, 1, ok, 0.1, 1.0, None) != 0
    assert b'sum_in and out' in lib.dsd_last_error()
