"""GPU: the ParallelWaveGAN generator on the HIP operators (diffsinger_amd/pwg.py, csrc/pwg_kernels.hpp) against the fixtures of the live
reference (tests/golden/pwg_*.npz) and, operator by operator, against the oracle (oracle/pwg_oracle.py).  fp32 MFMA with k-ordered sums
through 30 gated layers: waveform within 2e-5 absolute (|y| <= 1)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pwg_oracle as PO
from oracle.pwg_cases import CASES
from tests import pwg_helpers as PH

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


@pytest.mark.parametrize('name', list(CASES))
def test_generator_matches_the_reference(name):
    case, cfg, m, state, inp = PH.case_setup(name)
    g = PH.load_golden(name)
    m.remove_weight_norm()
    m = m.to(DEV).eval()
    y = m(inp['x'].to(DEV), inp['c'].to(DEV), inp['pitch'].to(DEV) if 'pitch' in inp else None)
    from diffsinger_amd import _lib
    assert _lib._lib is not None, 'libdsdenoise.so was not loaded'
    assert y.shape == g['y'].shape
    err = float(np.abs(y.cpu().numpy() - g['y']).max())
    print(f'{name}: waveform max-abs err {err:.3e} (max|ref| {float(np.abs(g["y"]).max()):.3f})')
    assert err <= 2e-5
    # weight-normed module (not removed) gives the same waveform: the packed weights come from plain_weight()
    case, cfg, m2, state, inp = PH.case_setup(name)
    y2 = m2.to(DEV).eval()(inp['x'].to(DEV), inp['c'].to(DEV), inp['pitch'].to(DEV) if 'pitch' in inp else None)
    assert float((y2 - y).abs().max()) <= 1e-6


@pytest.mark.parametrize('L,scale', [(7, 4), (33, 2), (100, 8), (1, 4)])
def test_upsample_stage(L, scale):
    from diffsinger_amd import _lib
    from diffsinger_amd.vocoder import _HipOps, padded_samples
    ops = _HipOps()
    g = torch.Generator().manual_seed(L * 10 + scale)
    x = torch.randn(2, 80, L, generator=g)
    w = torch.randn(1, 1, 1, 2 * scale + 1, generator=g)
    ref = F.conv2d(F.interpolate(x.unsqueeze(1), scale_factor=(1, scale), mode='nearest'), w, padding=(0, scale)).squeeze(1)
    xd = ops.pad_rows(x.to(DEV))
    out = torch.full((2, 80, padded_samples(L * scale)), 7.0, device=DEV)
    _lib.check(ops.lib.dsv_pwg_upsample(xd.data_ptr(), w.reshape(-1).to(DEV).data_ptr(), out.data_ptr(), 160, L, scale, ops._s(DEV)))
    assert float((out[:, :, :L * scale].cpu() - ref).abs().max()) <= 2e-6
    if out.shape[2] > L * scale:
        assert float(out[:, :, L * scale:].abs().max()) == 0


@pytest.mark.parametrize('B,L,dil,aux,first', [(2, 100, 1, 80, True), (1, 777, 4, 80, False), (2, 1500, 512, 80, False), (3, 65, 64, 80, False),
                                               (1, 300, 32, 0, True), (2, 96, 256, 8, False)])
def test_residual_block(B, L, dil, aux, first):
    from diffsinger_amd import _lib
    from diffsinger_amd.vocoder import _HipOps, padded_samples
    ops = _HipOps()
    g = torch.Generator().manual_seed(B * 1000 + L + dil)
    x = torch.randn(B, 64, L, generator=g)
    c = torch.randn(B, aux, L, generator=g) if aux else None
    p = {'conv.weight': torch.randn(128, 64, 3, generator=g) * (192 ** -0.5) * 2, 'conv.bias': torch.randn(128, generator=g) * 0.1,
         'conv1x1_out.weight': torch.randn(64, 64, 1, generator=g) * 0.125, 'conv1x1_out.bias': torch.randn(64, generator=g) * 0.1,
         'conv1x1_skip.weight': torch.randn(64, 64, 1, generator=g) * 0.125, 'conv1x1_skip.bias': torch.randn(64, generator=g) * 0.1}
    if aux:
        p['conv1x1_aux.weight'] = torch.randn(128, aux, 1, generator=g) * (aux ** -0.5)
    prev = torch.randn(B, 64, L, generator=g)
    xr, sr = PO.residual_block(p, '', x, c, dil)
    if not first:
        sr = prev + sr
    cols = [p['conv.weight'].permute(0, 2, 1).reshape(128, -1)] + ([p['conv1x1_aux.weight'][:, :, 0]] if aux else [])
    w1 = ops.pack(torch.cat(cols, 1)[:, :, None].to(DEV))
    w2 = ops.pack(torch.cat([p['conv1x1_out.weight'], p['conv1x1_skip.weight']], 0).to(DEV))
    b1 = p['conv.bias'].to(DEV)
    b2 = torch.cat([p['conv1x1_out.bias'], p['conv1x1_skip.bias']]).to(DEV)
    xd = ops.pad_rows(x.to(DEV))
    cd = ops.pad_rows(c.to(DEV)) if aux else None
    LS = padded_samples(L)
    xo = torch.full((B, 64, LS), 7.0, device=DEV)
    sk = ops.pad_rows(prev.to(DEV)) if not first else torch.full((B, 64, LS), 7.0, device=DEV)
    _lib.check(ops.lib.dsv_pwg_layer(xd.data_ptr(), cd.data_ptr() if aux else None, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                     xo.data_ptr(), sk.data_ptr(), B, L, aux, dil, 1 if first else 0, ops._s(DEV)))
    ex = float((xo[:, :, :L].cpu() - xr).abs().max())
    es = float((sk[:, :, :L].cpu() - sr).abs().max())
    print(f'residual block B={B} L={L} dil={dil} aux={aux}: x err {ex:.2e}, skip err {es:.2e}')
    assert ex <= 5e-6 and es <= 5e-6
    if LS > L:
        assert float(xo[:, :, L:].abs().max()) == 0 and float(sk[:, :, L:].abs().max()) == 0


def test_spec2wav_wrapper():
    """vocoders/pwg.py:85-104: edge padding by the context window, host-drawn noise, f0 -> coarse pitch ids."""
    from diffsinger_amd.pwg import PWG
    case, cfg, m, state, inp = PH.case_setup('pwg_pitch')
    m.remove_weight_norm()
    hop = int(np.prod(cfg['upsample_scales']))
    config = {'generator_params': dict(case['gen']), 'hop_size': hop}
    config['generator_params'].setdefault('aux_context_window', 2)
    voc = PWG(m, config, None, DEV)
    g = torch.Generator().manual_seed(9)
    T = 17
    mel = torch.randn(T, 80, generator=g).numpy()
    f0 = (torch.rand(T, generator=g) * 300 + 80).numpy()
    f0[3:6] = 0
    z = torch.randn(1, 1, T * hop, generator=g)
    wav = voc.spec2wav(mel, f0=f0, z=z)
    assert wav.shape == (T * hop,) and np.isfinite(wav).all()
    # the same through the oracle
    from diffsinger_amd.fs2 import f0_to_coarse
    c = torch.as_tensor(np.pad(mel, ((2, 2), (0, 0)), 'edge')).float().unsqueeze(0).transpose(2, 1)
    p = torch.as_tensor(np.pad(f0_to_coarse(torch.as_tensor(f0)).numpy(), (2, 2), 'edge'))[None].long()
    with torch.no_grad():
        y, _ = PO.generator_forward(PO.plain_params(state), cfg, z, c, p)
    assert float(np.abs(wav - y.view(-1).numpy()).max()) <= 2e-5
    torch.manual_seed(4)
    assert voc.spec2wav(mel, f0=f0).shape == (T * hop,)                    # draws its own noise on the host
