"""CPU: the host side of the HIP vocoder (row f2) - parameter tree, weight-norm fold, polyphase form of the transposed
convolutions, the call sequence into include/dsv.h - checked against the oracle / the reference fixtures with the header's
formulas standing in for the kernels (tests/voc_helpers.py).  The kernels themselves are checked on the GPU
(tests/test_gpu_vocoder.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from diffsinger_amd import _lib
from diffsinger_amd.vocoder import HifiGanGenerator, fold_weight, polyphase_weight
from oracle import hifigan_oracle as HO
from oracle.make_golden_hifigan import CASES, CONFIG, inputs
from tests.voc_helpers import HeaderFormulaOps, chunked_sine, draws_like_reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('u,k', [(8, 16), (2, 4), (4, 8), (5, 11), (3, 9), (1, 3)])
def test_polyphase_form_equals_conv_transpose(u, k):
    g = torch.Generator().manual_seed(u * 100 + k)
    ci, co, L, B = 6, 5, 19, 2
    w = torch.randn(ci, co, k, generator=g)
    x = torch.randn(B, ci, L, generator=g)
    p = (k - u) // 2
    ref = F.conv_transpose1d(x, w, None, stride=u, padding=p)
    wp, pad = polyphase_weight(w, u, p)
    assert wp.shape[0] == co * u and wp.shape[1] == ci
    y = F.conv1d(F.pad(x, (pad, wp.shape[2] - 1 - pad)), wp)
    y = y.reshape(B, co, u, L).permute(0, 1, 3, 2).reshape(B, co, L * u)
    assert ref.shape == y.shape
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('F,k', [(4, 11), (2, 7), (4, 3)])
def test_fold_weight_rows_are_shifted_filters(F, k):
    w = torch.randn(3, 5, k)
    wf = fold_weight(w, F)
    assert wf.shape == (3 * F, 5, k + F - 1)
    for e in range(F):
        assert torch.equal(wf[e::F, :, e:e + k], w)
        assert float(wf[e::F, :, :e].abs().sum()) == 0.0 and float(wf[e::F, :, e + k:].abs().sum()) == 0.0


def test_polyphase_rejects_other_paddings():
    with pytest.raises(NotImplementedError):
        polyphase_weight(torch.zeros(4, 4, 16), 8, 3)


def test_chunked_scan_reproduces_the_reference_sine_generator():
    """The two-pass fp64 block scan of k_voc_sine (numpy model) against SineGen through the oracle: same excitation."""
    B, T, up = 2, 24, 256
    g = torch.Generator().manual_seed(3)
    f0 = torch.rand(B, T, generator=g) * 300 + 80
    f0[0, 5:9] = 0
    f0[1, 20:] = 0
    torch.manual_seed(11)
    f0u = F.interpolate(f0[:, None], scale_factor=float(up), mode='nearest').transpose(1, 2)
    sines, uv = HO.sine_gen(f0u, 24000)
    rand_ini, noise = draws_like_reference(11, B, T * up)
    sw = torch.from_numpy(chunked_sine(f0.numpy(), rand_ini.numpy(), up, 24000.0, 0.1)).permute(0, 2, 1)
    namp = uv * 0.003 + (1 - uv) * 0.1 / 3
    mine = sw * uv + namp * noise
    assert float((mine - sines).abs().max()) < 2e-6


def _build(case, weight_norm):
    h = dict(CONFIG, use_pitch_embed=case['nsf'])
    p = HO.synth_generator_params(h, case['seed'] + 1000)
    m = HifiGanGenerator(h)
    if weight_norm:                                                   # a checkpoint as saved: weight_g / weight_v
        sd = {}
        for k, v in p.items():
            if k.endswith('.weight') and not k.startswith(('noise_convs', 'm_source')):
                sd[k[:-7] + '.weight_v'] = v.clone()
                sd[k[:-7] + '.weight_g'] = v.flatten(1).norm(dim=1).reshape(-1, 1, 1).clone()
            else:
                sd[k] = v.clone()
        m.load_state_dict(sd, strict=True)
        assert any(k.endswith('weight_g') for k in m.state_dict())
    else:
        m.load_state_dict(p, strict=True)                             # a state saved after remove_weight_norm()
        assert not any(k.endswith('weight_g') for k in m.state_dict())
        assert sorted(m.state_dict()) == sorted(p)
    return h, p, m


@pytest.mark.parametrize('name', ['hifigan_plain', 'hifigan_nsf'])
@pytest.mark.parametrize('weight_norm,fold', [(False, True), (True, True), (False, False)])
def test_host_orchestration_matches_reference_fixture(name, weight_norm, fold):
    case = CASES[name]
    h, p, m = _build(case, weight_norm)
    if weight_norm:
        m.remove_weight_norm()
    m._ops = HeaderFormulaOps()
    m._ops.fold = fold                                                # the narrow stages through the folded formula / the plain one
    mel, f0 = inputs(case)
    kw = {}
    if f0 is not None:
        kw['rand_ini'], kw['noise'] = draws_like_reference(case['seed'], case['B'], case['T'] * 256)
    wav = m(mel, f0, **kw)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))['wav']
    assert wav.shape == g.shape
    err = float(np.abs(wav.numpy() - g).max())
    assert err < 5e-5, err


def test_state_dict_names_match_the_reference_layout():
    h = dict(CONFIG, use_pitch_embed=True)
    m = HifiGanGenerator(h)
    names = set(m.state_dict())
    assert 'conv_pre.weight_g' in names and 'ups.3.weight_v' in names and 'resblocks.11.convs2.2.bias' in names
    assert 'noise_convs.0.weight' in names and 'm_source.l_linear.weight' in names
    m.remove_weight_norm()
    assert sorted(m.state_dict()) == sorted(HO.generator_shapes(h))
    for k, shp in HO.generator_shapes(h).items():
        assert tuple(m.state_dict()[k].shape) == tuple(shp), k


def test_no_cpu_path():
    m = HifiGanGenerator(dict(CONFIG, use_pitch_embed=False))
    with pytest.raises(RuntimeError, match='no CPU path'):
        m(torch.zeros(1, 80, 8))


def test_dsv_header_symbols_bound_and_exported():
    from tests.test_abi import _header_symbols
    assert _header_symbols('dsv.h', 'dsv_') == sorted(_lib.SYMBOLS_VOC)
    lib = _lib.load()
    for name in _lib.SYMBOLS_VOC:
        assert hasattr(lib, name), name
    assert lib.dsv_padded_samples(33) == 64
    assert lib.dsv_packed_floats(8, 8, 11) == (1 * 1 * 11 * 64 + 8192) * 4
    assert lib.dsv_packed_floats(512, 128, 3) == (16 * 16 * 3 * 64 + 8192) * 4
    assert lib.dsv_conv1d(None, None, None, None, 1, 8, 8, 1, 0, 1, 8, 1, 1.0, None, None, 1.0, 0, None) == -1     # rejected before any HIP call
    assert lib.dsv_fold_factor(8, 8, 11, 5) in (1, 4) and lib.dsv_fold_factor(64, 64, 11, 5) == 1
    assert b'dsv_conv1d' in lib.dsd_last_error()


def test_vocoder_registry_and_checkpoint_discovery(tmp_path, capsys):
    """HifiGAN() finds config.yaml + the newest model_ckpt_steps_*.ckpt under hparams['vocoder_ckpt'] like vocoders/hifigan.py:40-54,
    loads the weight-normed `model_gen` state strictly and folds the weight norm; the registry mirrors vocoders/base_vocoder.py."""
    import yaml
    from diffsinger_amd import hparams
    from diffsinger_amd import vocoder as V
    h = dict(CONFIG, use_pitch_embed=True)
    p = HO.synth_generator_params(h, 5)
    sd = {}
    for k, v in p.items():
        if k.endswith('.weight') and not k.startswith(('noise_convs', 'm_source')):
            sd[k[:-7] + '.weight_v'] = v.clone()
            sd[k[:-7] + '.weight_g'] = v.flatten(1).norm(dim=1).reshape(-1, 1, 1).clone()
        else:
            sd[k] = v.clone()
    with open(os.path.join(tmp_path, 'config.yaml'), 'w') as f:
        yaml.safe_dump(h, f)
    for step in (9000, 120000, 40000):
        torch.save({'state_dict': {'model_gen': sd if step == 120000 else {}}}, os.path.join(tmp_path, f'model_ckpt_steps_{step}.ckpt'))
    hparams.clear()
    hparams.update(vocoder_ckpt=str(tmp_path), use_nsf=True, vocoder='vocoders.hifigan.HifiGAN')
    voc = V.get_vocoder_cls(hparams)(device='cpu')
    assert isinstance(voc, V.HifiGAN) and voc.use_nsf and voc.config['upsample_rates'] == [8, 8, 2, 2]
    assert 'model_ckpt_steps_120000.ckpt' in capsys.readouterr().out
    got = voc.model.state_dict()
    assert sorted(got) == sorted(p)
    assert float((got['ups.2.weight'] - p['ups.2.weight']).abs().max()) < 1e-6
    assert V.VOCODERS['hifigan'] is V.HifiGAN and V.get_vocoder_cls({'vocoder': 'HifiGAN'}) is V.HifiGAN
    reg, mod = {}, type('m', (), {})()
    V.register_vocoders(reg, modules=[mod])
    assert reg['HifiGAN'] is V.HifiGAN and reg['hifigan'] is V.HifiGAN and mod.HifiGAN is V.HifiGAN
    with pytest.raises(RuntimeError, match='no CPU path'):
        voc.spec2wav(np.zeros((8, 80), np.float32))


@pytest.mark.parametrize('cfg', [
    dict(resblock='2', upsample_rates=[8, 8, 4], upsample_kernel_sizes=[16, 16, 8], upsample_initial_channel=64, resblock_kernel_sizes=[3, 5, 7],
         resblock_dilation_sizes=[[1, 2], [2, 6], [3, 9]], nsf=False),                        # a "v3"-style generator: ResBlock2, hop 256 in 3 stages
    dict(resblock='1', upsample_rates=[4, 4, 4, 2], upsample_kernel_sizes=[8, 8, 8, 4], upsample_initial_channel=256, resblock_kernel_sizes=[3, 7, 11],
         resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], nsf=True),                # wider, hop 128, NSF
], ids=['resblock2_hop256', 'wide_hop128_nsf'])
def test_other_generator_configs_through_the_header_formulas(cfg):
    """Configurations beyond the shipped configs/tts/hifigan.yaml: the orchestration (ResBlock2 wiring, other strides / widths, which
    layers fold) against the oracle, on CPU with the header formulas standing in for the kernels."""
    cfg = dict(cfg)
    nsf = cfg.pop('nsf')
    h = dict(cfg, use_pitch_embed=nsf, audio_sample_rate=24000)
    p = HO.synth_generator_params(h, 77)
    m = HifiGanGenerator(h)
    m.load_state_dict(p, strict=True)
    m._ops = HeaderFormulaOps()
    g = torch.Generator().manual_seed(4)
    B, T = 2, 11
    hop = int(np.prod(cfg['upsample_rates']))
    mel = torch.randn(B, 80, T, generator=g)
    f0, kw = None, {}
    if nsf:
        f0 = torch.rand(B, T, generator=g) * 300 + 80
        f0[1, 7:] = 0
        kw['rand_ini'], kw['noise'] = draws_like_reference(123, B, T * hop)
    torch.manual_seed(123)
    with torch.no_grad():
        want = HO.generator(p, h, mel, f0)
    got = m(mel, f0, **kw)
    assert got.shape == want.shape == (B, 1, T * hop)
    assert float((got - want).abs().max()) < 5e-5


def test_taps_beyond_the_staged_window_are_refused():
    """Taps may reach +-48 samples (the official v3 generator's kernel 7 at dilation 12 = 36 runs, tests/test_gpu_vocoder.py); beyond that the
    module says so instead of being wrong."""
    h = dict(CONFIG, resblock='2', resblock_kernel_sizes=[11], resblock_dilation_sizes=[[3, 12]], use_pitch_embed=False)
    m = HifiGanGenerator(h)
    m.remove_weight_norm()
    m._ops = HeaderFormulaOps()
    with pytest.raises(NotImplementedError, match='reaches 60'):
        m(torch.zeros(1, 80, 4))
    h = dict(CONFIG, resblock='2', resblock_kernel_sizes=[7], resblock_dilation_sizes=[[3, 12]], use_pitch_embed=False)
    m = HifiGanGenerator(h)
    m.remove_weight_norm()
    m._ops = HeaderFormulaOps()
    assert m(torch.zeros(1, 80, 4)).shape[0] == 1          # reach 36: accepted


def test_denoise_post_filter_matches_an_independent_stft_and_is_identity_at_zero():
    """vocoders/vocoder_utils.py:7-15 (librosa 0.8.0 stft -> |S| - v clipped at 0, phase kept -> istft) restated in numpy
    (diffsinger_amd.vocoder.denoise; librosa is absent here) against torch.stft / torch.istft with the same arguments."""
    import numpy as np
    import torch
    from diffsinger_amd.vocoder import denoise
    g = torch.Generator().manual_seed(4)
    for n_fft, hop, win, n in ((512, 128, 512, 24000), (1024, 256, 1024, 22050), (1024, 256, 800, 9000)):
        wav = (torch.randn(n, generator=g) * 0.1 + 0.3 * torch.sin(torch.arange(n) * 0.05)).numpy().astype(np.float32)
        window = torch.zeros(n_fft, dtype=torch.float64)
        lp = (n_fft - win) // 2
        window[lp:lp + win] = torch.hann_window(win, periodic=True, dtype=torch.float64)
        for v in (0.0, 0.1, 0.5):
            got = denoise(wav, v=v, fft_size=n_fft, hop_size=hop, win_size=win)
            S = torch.stft(torch.from_numpy(wav).double(), n_fft, hop_length=hop, win_length=n_fft, window=window, center=True, pad_mode='constant',
                           onesided=True, return_complex=True)
            S = torch.polar(torch.clamp(S.abs() - v, min=0), S.angle())
            want = torch.istft(S, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True).numpy()
            m = min(len(got), len(want))
            assert abs(len(got) - len(want)) <= hop and m > n - 2 * n_fft
            edge = n_fft                                           # (librosa keeps the ramp-up samples torch trims differently: compare the interior)
            err = float(np.abs(got[edge:m - edge] - want[edge:m - edge]).max())
            assert err < 2e-5, (n_fft, hop, win, v, err)
            if v == 0.0:
                assert float(np.abs(got[edge:m - edge] - wav[edge:m - edge]).max()) < 2e-5     # no subtraction: the filter is the identity
            else:
                assert float(np.abs(got).mean()) < float(np.abs(wav).mean())                  # energy went down


def test_merge_plan_counts_rounds():
    """diffsinger_amd.vocoder._merge_plan (round 6): a chain launch of W workgroups on S co-resident places costs ceil(W / S) rounds
    (profiles/r6_27_voc_tail_probe.jsonl).  The bench shape at 32 channels - 1 096 / 1 264 / 1 368 workgroups of the kernel-3 / 7 / 11 resblocks
    on 512 places - pays 3 + 3 + 3 rounds as three launches; with kernel 11 and kernel 3 in one grid the short workgroups fill the long ones'
    last round.  The summing launch is the resblock of median workgroup time; whole rounds gain nothing and are left alone."""
    from diffsinger_amd.vocoder import _merge_plan
    assert _merge_plan([1096, 1264, 1368], [6 * 18, 6 * 34, 6 * 50], 512) == (1, [2, 0])
    assert _merge_plan([2192, 2528, 2736], [81.0, 129.0, 177.0], 768) == (1, [2, 0])
    assert _merge_plan([512, 512, 512], [1.0, 2.0, 3.0], 512) is None                       # three whole rounds
    assert _merge_plan([1096, 1264, 1368], [3.0, 2.0, 1.0], 512)[0] == 1                    # median by TIME, longest group first
    assert _merge_plan([1096, 1264, 1368], [3.0, 2.0, 1.0], 512)[1] == [0, 2]
    assert _merge_plan([100, 100], [1.0, 2.0], 512) is None                                 # two resblocks: nothing to leave for the sum


def test_level_by_level_resblocks_equal_the_sequential_form():
    """HifiGanGenerator._stage_resblocks_by_level (round 6: the stages without a chain kernel advance their three parallel ResBlock1 level by
    level, the convolutions of a level in one dsv_conv1d_multi launch) against `_resblock` one after the other - on the header-formula
    emulation the two are the same torch calls on the same operands: equal bits; a ResBlock2 generator does not qualify."""
    import diffsinger_amd.vocoder as V
    cfg = dict(CONFIG)
    m = HifiGanGenerator(cfg)
    m.remove_weight_norm()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    m._ops = HeaderFormulaOps()
    L = 90
    C = cfg['upsample_initial_channel'] // 2
    x = torch.zeros(2, C, V.padded_samples(L))
    x[:, :, :L] = torch.randn(2, C, L, generator=g)
    got = m._stage_resblocks_by_level(0, x, L)
    assert got is not None
    acc = None
    for j in range(m.num_kernels):
        acc = m._resblock(j, x, L, acc, float(m.num_kernels) if j == m.num_kernels - 1 else 1.0)
    assert torch.equal(got, acc)
    V.set_chain_mode('off')
    try:
        assert torch.equal(m._stage_resblocks(0, x, L), acc)
    finally:
        V.set_chain_mode(None)
    assert torch.equal(m._stage_resblocks(0, x, L), acc)
    m2 = HifiGanGenerator(dict(cfg, resblock='2'))
    m2.remove_weight_norm()
    m2._ops = HeaderFormulaOps()
    assert m2._stage_resblocks_by_level(0, x, L) is None
