"""GPU parity of the HIP PitchExtractor (mel -> f0 for the NSF vocoder): the two new operators against torch's
own BatchNorm / GroupNorm on the CPU, the module against the fixture recorded from the REAL reference module and against the oracle.
Tolerances: fp32 end to end; pitch_pred (log2 f0 and the voicing logit, |x| < 10) within 2e-4 after 15 convolution / normalisation
layers; f0 in Hz compared where the voicing decision is not within 1e-3 of its threshold (f0 = 2 ** x amplifies by f0 ln 2)."""
import ast
import os

import numpy as np
import pytest
import torch

from diffsinger_amd import hparams
from oracle import pe_oracle as PO
from oracle.make_golden_pe import CASE

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda:0'


def _build(seed):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pe_opencpop.npz'))
    hp = ast.literal_eval(str(g['hp']))
    hparams.clear()
    hparams.update(hp, dur_loss='mse')
    from diffsinger_amd.pe import PitchExtractor
    m = PitchExtractor().eval()
    p = PO.synth_extractor_params(hp, seed)
    m.load_state_dict(p, strict=True)
    return g, hp, p, m.to(DEV)


def _compare(r, want_pp, want_f0):
    pp = r['pitch_pred'].cpu().numpy()
    f0 = r['f0_denorm_pred'].cpu().numpy()
    err = float(np.abs(pp - want_pp).max())
    sure = np.abs(want_pp[:, :, 1]) > 1e-3
    rel = float((np.abs(f0 - want_f0)[sure] / np.maximum(want_f0[sure], 1.0)).max())
    print('pitch_pred err', err, 'f0 rel err', rel)
    assert err < 2e-4, err
    assert rel < 2e-4, rel
    assert ((f0 == 0) == (want_f0 == 0))[sure].all()


def test_operators_match_torch_norms():
    from diffsinger_amd.fs2 import from_cm, to_cm
    from diffsinger_amd.pe import channel_affine_cm, group_norm_cm
    g = torch.Generator().manual_seed(1)
    B, C, T = 3, 256, 75
    x = torch.randn(B, T, C, generator=g) * 2 + 0.5
    keep = (torch.rand(B, T, generator=g) > 0.2).float()
    bn = torch.nn.BatchNorm1d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * torch.randn(C, generator=g)); bn.bias.copy_(0.1 * torch.randn(C, generator=g))
        bn.running_mean.copy_(0.3 * torch.randn(C, generator=g)); bn.running_var.copy_(0.5 + torch.rand(C, generator=g))
        want = (bn(x.transpose(1, 2)) * keep[:, None, :]).transpose(1, 2)
        inv = 1.0 / torch.sqrt(bn.running_var + bn.eps)
        a = inv * bn.weight
        b = bn.bias - bn.running_mean * a
    xc = to_cm(x.to(DEV))
    got = channel_affine_cm(xc, T, a.to(DEV), b.to(DEV), keep.to(DEV))
    assert float(got[:, :, T:].abs().sum()) == 0.0
    assert float((from_cm(got, T).cpu() - want).abs().max()) < 2e-6
    gn = torch.nn.GroupNorm(16, C)
    with torch.no_grad():
        gn.weight.copy_(1 + 0.1 * torch.randn(C, generator=g)); gn.bias.copy_(0.1 * torch.randn(C, generator=g))
        res = torch.randn(B, T, C, generator=g)
        want = (res.transpose(1, 2) + torch.relu(gn(x.transpose(1, 2)))).transpose(1, 2)
    got = group_norm_cm(xc, T, 16, gn.weight.detach().to(DEV), gn.bias.detach().to(DEV), gn.eps, relu=True, residual=to_cm(res.to(DEV)))
    assert float(got[:, :, T:].abs().sum()) == 0.0
    err = float((from_cm(got, T).cpu() - want).abs().max())
    print('group_norm err', err)
    assert err < 5e-6, err


def test_pitch_extractor_matches_reference_fixture():
    g, hp, p, m = _build(CASE['seed'] + 1000)
    mel = PO.synth_mel(CASE['B'], CASE['T'], CASE['seed'])
    r = m(mel.to(DEV))
    _compare(r, g['pitch_pred'], g['f0_denorm_pred'])
    f0 = r['f0_denorm_pred'].cpu()
    assert (f0[1, -5:] == 0).all() and (f0[2, -10:] == 0).all()
    assert any('libdsdenoise' in ln for ln in open('/proc/self/maps'))


def test_pitch_extractor_longer_batch_matches_oracle_and_prenet_surface():
    g, hp, p, m = _build(99)
    mel = PO.synth_mel(4, 333, 5)
    with torch.no_grad():
        want = PO.pitch_extractor(p, hp, mel)
    r = m(mel.to(DEV))
    _compare(r, want['pitch_pred'].numpy(), want['f0_denorm_pred'].numpy())
    hid, out = m.mel_prenet(mel.to(DEV))
    with torch.no_grad():
        w = PO.prenet(p, 'mel_prenet.', mel)
    assert hid.shape == (1, 4, 333, 256) and float((out.cpu() - w).abs().max()) < 5e-5
