"""GPU: the FUSED training stack (diffsinger_amd/train_fused.py, csrc/train_kernels.hpp) against torch autograd
in float64 on the CPU of the reference's ResidualBlock arithmetic (usr/diff/net.py:66-78, :121-126): the stand-alone weight-gradient
operator, the forward (skip sum and the tensors it saves), and every gradient of the backward pass - dx0, dcond, dstep and the six
parameter tensors of every layer.  Errors are judged relative to the max-abs of the reference tensor (gradients are sums over B * T frames)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double().cpu() - b).abs().max() / max(float(b.abs().max()), 1e-30))


@pytest.mark.parametrize('B,T,Ci,Co,K,dil', [(2, 50, 256, 512, 3, 1), (3, 77, 256, 512, 3, 8), (2, 96, 256, 512, 3, 2), (2, 64, 256, 512, 1, 1),
                                           (1, 33, 512, 256, 3, 4), (8, 1024, 256, 512, 3, 1), (48, 512, 256, 512, 1, 1)])
def test_wgrad2_operator(B, T, Ci, Co, K, dil):
    from diffsinger_amd import fs2, train_fused
    g = torch.Generator().manual_seed(T + Co + K)
    TS = fs2.padded_frames(T)
    x = torch.randn(B, Ci, T, generator=g)
    dy = torch.randn(B, Co, T, generator=g)
    big = B * T > 4096
    dev = torch.device('cuda', 0)
    if big:         # reference on the GPU (float64 autograd of conv1d)
        xr, dyr = x.to(dev).double(), dy.to(dev).double()
    else:
        xr, dyr = x.double(), dy.double()
    w = torch.zeros(Co, Ci, K, dtype=torch.float64, device=xr.device, requires_grad=True)
    bias = torch.zeros(Co, dtype=torch.float64, device=xr.device, requires_grad=True)
    F.conv1d(xr, w, bias, padding=dil * (K - 1) // 2, dilation=dil).backward(dyr)
    xd = F.pad(x, (0, TS - T)).to(dev).contiguous()
    dyd = F.pad(dy, (0, TS - T)).to(dev).contiguous()
    dyd[:, :, T:] = 7.0                                  # garbage in the tail of dy must not leak
    dw, db = train_fused.conv1d_wgrad2(dyd, xd, K, dil, T)
    dw2, _ = train_fused.conv1d_wgrad2(dyd, xd, K, dil, T)
    e_w, e_b = _rel(dw, w.grad.cpu()), _rel(db, bias.grad.cpu())
    print(f'wgrad2 B={B} T={T} Ci={Ci} Co={Co} K={K} dil={dil}: dw {e_w:.2e} db {e_b:.2e}')
    assert torch.equal(dw, dw2)                          # deterministic
    assert e_w <= 2e-5 and e_b <= 2e-5


def _make_stack(L, cycle, seed):
    g = torch.Generator().manual_seed(seed)
    ws = {}
    ws['dc_w'] = [torch.randn(512, 256, 3, generator=g) * (256 * 3) ** -0.5 for _ in range(L)]
    ws['dc_b'] = [torch.randn(512, generator=g) * 0.1 for _ in range(L)]
    ws['cp_w'] = [torch.randn(512, 256, 1, generator=g) * 256 ** -0.5 for _ in range(L)]
    ws['cp_b'] = [torch.randn(512, generator=g) * 0.1 for _ in range(L)]
    ws['op_w'] = [torch.randn(512, 256, 1, generator=g) * 256 ** -0.5 for _ in range(L)]
    ws['op_b'] = [torch.randn(512, generator=g) * 0.1 for _ in range(L)]
    dils = [2 ** (l % cycle) for l in range(L)]
    return ws, dils


def _ref_stack(x0, cond, step, ws, dils, keep=None):
    """float64 reference of the residual stack on [B][256][T] tensors (no padding); keep: dict that receives y / a of every layer."""
    x, skip = x0, 0
    L = len(dils)
    for l in range(L):
        y = x + step[:, l, :, None]
        a = F.conv1d(y, ws['dc_w'][l], ws['dc_b'][l], padding=dils[l], dilation=dils[l]) + F.conv1d(cond, ws['cp_w'][l], ws['cp_b'][l])
        if keep is not None:
            keep.setdefault('y', []).append(y.detach())
            keep.setdefault('a', []).append(a.detach())
        g = torch.sigmoid(a[:, :256]) * torch.tanh(a[:, 256:])
        o = F.conv1d(g, ws['op_w'][l], ws['op_b'][l])
        x = (x + o[:, :256]) / math.sqrt(2.0)
        skip = skip + o[:, 256:]
    return skip


def _frag_to_rows(frag, ntiles):
    """[ntiles][w4][mb4][q4][lane64][e4] (the layout of the saved pre-activation) -> [ntiles][512 rows][32 frames]"""
    f = frag.reshape(ntiles, 4, 4, 4, 2, 32, 4)                 # tile, w, mb, q, h, j, e
    out = torch.zeros(ntiles, 512, 32)
    for w in range(4):
        for mb in range(4):
            base = (64 * w + 32 * mb) if mb < 2 else (256 + 64 * w + 32 * (mb - 2))
            for q in range(4):
                for h in range(2):
                    for e in range(4):
                        out[:, base + 8 * q + 4 * h + e, :] = f[:, w, mb, q, h, :, e]
    return out


@pytest.fixture
def stack_conv(request):
    """the convolution of the persistent forward for one test (dsf_set_stack_conv), the default restored behind it"""
    from diffsinger_amd import train_fused
    train_fused.set_stack_conv('wino' if request.param == 'wino-taps' else request.param)
    train_fused.set_wgrad_dual(request.param != 'wino-taps')      # 'wino-taps': the Winograd convolutions with the weight gradient as three tap products (rounds 2-5)
    yield request.param
    train_fused.set_stack_conv('wino')
    train_fused.set_wgrad_dual(True)


@pytest.mark.parametrize('stack_conv', ['wino', 'wino-taps', 'direct'], indirect=True)
@pytest.mark.parametrize('B,T,L,cycle', [(2, 50, 3, 4), (3, 96, 5, 1), (2, 70, 20, 4), (1, 5, 1, 1), (1, 32, 2, 4), (4, 33, 2, 2), (9, 129, 4, 3), (1, 8300, 2, 4)])
def test_stack_forward_and_backward(B, T, L, cycle, stack_conv):
    """(the short shapes take the persistent forward: with the Winograd convolution, csrc/train_loop_wino.hpp, and with the direct one; the
    utterance of 8300 frames = 260 tiles does not fit the co-resident grid of a 256-CU part: per-layer forward launches, and the Winograd /
    direct data-gradient kernels behind them)"""
    from diffsinger_amd import _lib, fs2, train_fused
    lib = _lib.load()
    train_fused._bind(lib)
    assert train_fused.stack_conv() == ('wino' if stack_conv == 'wino-taps' else stack_conv)
    ws, dils = _make_stack(L, cycle, seed=L + T)
    g = torch.Generator().manual_seed(5 + T)
    x0 = torch.relu(torch.randn(B, 256, T, generator=g))
    cond = torch.randn(B, 256, T, generator=g)
    step = torch.randn(B, L, 256, generator=g) * 0.5
    dskip = torch.randn(B, 256, T, generator=g)
    # float64 reference with autograd
    r = {k: [t.double().requires_grad_(True) for t in v] for k, v in ws.items()}
    x0r, condr, stepr = x0.double().requires_grad_(True), cond.double().requires_grad_(True), step.double().requires_grad_(True)
    keep = {}
    skip_ref = _ref_stack(x0r, condr, stepr, r, dils, keep)
    skip_ref.backward(dskip.double())

    dev = torch.device('cuda', 0)
    TS = fs2.padded_frames(T)
    ntiles = B * TS // 32
    pad = lambda t: F.pad(t, (0, TS - T)).to(dev).contiguous()
    x0d, condd = pad(x0).requires_grad_(True), pad(cond).requires_grad_(True)
    stepd = step.to(dev).requires_grad_(True)
    order = ['dc_w', 'dc_b', 'cp_w', 'cp_b', 'op_w', 'op_b']
    wd = [t.to(dev).requires_grad_(True) for k in order for t in ws[k]]
    skip = train_fused._ResidualStack.apply(x0d, condd, stepd, T, dils, {}, *wd)
    e_skip = _rel(skip[:, :, :T], skip_ref.detach())
    assert float(skip[:, :, T:].abs().max() if TS > T else 0) == 0
    # the tensors the forward saved for the backward pass
    save = skip.grad_fn.saved_tensors[1]
    off = (C.c_int64 * 16)()
    _lib.check(lib.dsf_stack_offsets(B, T, L, 0, off, 16))
    oY, oA, Yl, Al = off[6], off[7], off[12], off[13]
    e_y = e_a = 0.0
    for l in range(L):
        yp = save[oY + l * Yl: oY + (l + 1) * Yl].reshape(B, 256, TS + 16).cpu()     # rows carry 8 zero floats on both sides (kTrYPad)
        y = yp[:, :, 8:8 + TS]
        e_y = max(e_y, _rel(y[:, :, :T], keep['y'][l]))
        assert float(y[:, :, T:].abs().max() if TS > T else 0) == 0 and float(yp[:, :, :8].abs().max()) == 0 and float(yp[:, :, 8 + TS:].abs().max()) == 0
        a = _frag_to_rows(save[oA + l * Al: oA + (l + 1) * Al].cpu(), ntiles).reshape(B, TS // 32, 512, 32).permute(0, 2, 1, 3).reshape(B, 512, TS)
        e_a = max(e_a, _rel(a[:, :, :T], keep['a'][l]))
    print(f'stack B={B} T={T} L={L}: skip {e_skip:.2e}, saved y {e_y:.2e}, saved a {e_a:.2e}')
    assert e_skip <= 2e-5 and e_y <= 1e-5 and e_a <= 2e-5

    dsk = pad(dskip)
    dsk[:, :, T:] = 3.0                                   # garbage in the tail of the incoming gradient must not leak
    skip.backward(dsk)
    errs = {'dx0': _rel(x0d.grad[:, :, :T], x0r.grad), 'dcond': _rel(condd.grad[:, :, :T], condr.grad), 'dstep': _rel(stepd.grad, stepr.grad)}
    assert float(x0d.grad[:, :, T:].abs().max() if TS > T else 0) == 0
    per_layer = []
    for ki, k in enumerate(order):
        worst = 0.0
        for l in range(L):
            e = _rel(wd[ki * L + l].grad, r[k][l].grad)
            per_layer.append((k, l, e))
            worst = max(worst, e)
        errs[k] = worst
    print('  backward: ' + ', '.join(f'{k} {v:.2e}' for k, v in errs.items()))
    bad = [(k, l, e) for k, l, e in per_layer if e > 2e-5]
    if bad:
        print('  layers over tolerance: ' + ', '.join(f'{k}[{l}] {e:.1e}' for k, l, e in bad[:40]))
    assert all(v <= 2e-5 for v in errs.values()), errs


@pytest.mark.parametrize('B,T,L,cycle,dcond', [(9, 1000, 3, 4, False), (2, 50, 3, 4, True), (3, 96, 20, 4, False), (1, 5, 1, 1, True), (17, 500, 2, 2, True),
                                               (1, 200, 4, 1, False)])
def test_persistent_kernels_equal_per_layer_launches(B, T, L, cycle, dcond):
    """csrc/train_loop.hpp (the forward as ONE launch per chunk of whole utterances, neighbour exchange through flags; the DIRECT convolution:
    dsf_set_stack_conv(0)) against the per-layer launches: the skip sum, everything saved for the backward and every gradient are the same
    BITS; (9, 1000) and (17, 500) take two chunks on a 256-CU part; dcond selects the caller-kept da_all layout of the backward.  The Winograd
    forward (the default) on the same inputs: every one of those tensors within 3e-5 of the direct form's (relative to the tensor's max)."""
    from diffsinger_amd import _lib, fs2, train_fused
    lib = _lib.load()
    train_fused._bind(lib)
    ws, dils = _make_stack(L, cycle, seed=3 * L + T)
    g = torch.Generator().manual_seed(T)
    dev = torch.device('cuda', 0)
    TS = fs2.padded_frames(T)
    pad = lambda t: F.pad(t, (0, TS - T)).to(dev).contiguous()
    x0, cond = pad(torch.randn(B, 256, T, generator=g)), pad(torch.randn(B, 256, T, generator=g))
    step = (torch.randn(B, L, 256, generator=g) * 0.5).to(dev)
    dskip = pad(torch.randn(B, 256, T, generator=g))
    dskip[:, :, T:] = -2.0
    wsrc = [t.to(dev) for k in ['dc_w', 'dc_b', 'cp_w', 'cp_b', 'op_w', 'op_b'] for t in ws[k]]
    off = (C.c_int64 * 16)()
    _lib.check(lib.dsf_stack_offsets(B, T, L, 0, off, 16))
    oY, oA, Yl, Al = off[6], off[7], off[12], off[13]
    got = {}
    for mode in ('0', '2', 'w', '0w'):           # '0w': per-layer (direct) forward launches + the Winograd data-gradient kernels - what a batch that does not fit the persistent grid runs
        train_fused.set_stack_mode({'0': 0, '2': 2, 'w': 2, '0w': 0}[mode])
        train_fused.set_stack_conv('wino' if 'w' in mode else 'direct')
        xin, cin, sin = x0.clone().requires_grad_(True), cond.clone().requires_grad_(dcond), step.clone().requires_grad_(True)
        wd = [t.clone().requires_grad_(True) for t in wsrc]
        skip = train_fused._ResidualStack.apply(xin, cin, sin, T, dils, {}, *wd)
        save = skip.grad_fn.saved_tensors[1]
        res = [skip.detach().clone(), save[oY: oY + L * Yl].clone(), save[oA: oA + L * Al].clone()]
        skip.backward(dskip)
        res += [xin.grad.clone(), sin.grad.clone()] + [t.grad.clone() for t in wd]
        if dcond:
            res.append(cin.grad.clone())
        got[mode] = res
        torch.cuda.synchronize()
    train_fused.set_stack_mode(1)
    train_fused.set_stack_conv('wino')
    assert all(bool(torch.isfinite(t).all()) for t in got['2']) and all(bool(torch.isfinite(t).all()) for t in got['w'])
    for i, (a, b) in enumerate(zip(got['0'], got['2'])):
        assert torch.equal(a, b), f'tensor {i} differs: {float((a - b).abs().max())}'
    worst = 0.0
    for tag in ('w', '0w'):
        for i, (a, b) in enumerate(zip(got['2'], got[tag])):
            e = float((a - b).abs().max() / max(float(a.abs().max()), 1e-30))
            worst = max(worst, e)
            assert e <= 3e-5, f'Winograd kernels ({tag}): tensor {i} off by {e:.2e}'
    assert worst > 0.0                                    # (they ARE other kernels)
    for i in range(3):                                    # '0w': the forward is the per-layer one - skip sum and saved tensors are its bits
        assert torch.equal(got['0'][i], got['0w'][i])
    print(f'Winograd forward vs direct persistent forward, B={B} T={T} L={L}: worst tensor {worst:.2e}')


def test_p_losses_gradients_fused_equals_operator_path(monkeypatch):
    """The fused stack and the operator-by-operator path of train.py give the same loss and gradients on a real DiffNet."""
    import diffsinger_amd
    from diffsinger_amd import hparams
    from tests import helpers as H
    pre = H.presets()['opencpop_ds60_rel']
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('DSD_TRAIN_FUSED', mode)
        hparams.clear()
        diffsinger_amd.use_preset('opencpop_ds60_rel')
        torch.manual_seed(7)
        net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
        torch.nn.init.normal_(net.output_projection.weight, std=0.02)
        gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1',
                                              spec_min=pre['spec_min'], spec_max=pre['spec_max']).cuda().train()
        g = torch.Generator().manual_seed(3)
        x0 = torch.clamp(torch.randn(3, 1, 80, 83, generator=g) * 0.5, -1, 1).cuda()
        noise = torch.randn(3, 1, 80, 83, generator=g).cuda()
        cond = torch.randn(3, 83, 256, generator=g).transpose(1, 2).cuda().requires_grad_(True)
        t = torch.tensor([5, 40, 17]).cuda()
        loss = gd.p_losses(x0, t, cond, noise=noise)
        loss.backward()
        res[mode] = (float(loss), {k: p.grad.detach().cpu() for k, p in net.named_parameters()}, cond.grad.detach().cpu())
    assert abs(res['1'][0] - res['0'][0]) <= 2e-6 * abs(res['0'][0])
    worst = ('', 0.0)
    for k, gref in res['0'][1].items():
        e = float((res['1'][1][k] - gref).abs().max() / max(float(gref.abs().max()), 1e-30))
        if e > worst[1]:
            worst = (k, e)
    e_c = float((res['1'][2] - res['0'][2]).abs().max() / float(res['0'][2].abs().max()))
    print(f'fused vs operator path: loss {res["1"][0]:.6f} / {res["0"][0]:.6f}, worst parameter gradient {worst[1]:.2e} at {worst[0]}, dcond {e_c:.2e}')
    assert worst[1] <= 5e-5 and e_c <= 5e-5


def test_fused_ends_of_p_losses_equal_the_tensor_expressions():
    """Round 6: q_sample with a step per utterance (five torch launches) and the L1 loss with its backward (nine) run as dsf_q_sample_rows /
    dsf_l1_mean / dsf_l1_mean_bwd.  x_noisy carries the bits of the tensor expression; the loss sums in another (fixed) order - within 1e-6 -
    and its gradient -(sign(noise - x_recon) / N) is the same bits, so every parameter gradient is bit-identical to the torch ends'."""
    import diffsinger_amd
    from diffsinger_amd import hparams, train
    from tests import helpers as H
    pre = H.presets()['opencpop_ds60_rel']
    res = {}
    try:
        for fused in (True, False):
            train.set_fused_ends(fused)
            hparams.clear()
            diffsinger_amd.use_preset('opencpop_ds60_rel')
            torch.manual_seed(7)
            net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
            torch.nn.init.normal_(net.output_projection.weight, std=0.02)
            gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1',
                                                  spec_min=pre['spec_min'], spec_max=pre['spec_max']).cuda().train()
            g = torch.Generator().manual_seed(3)
            B, T = 3, 96                                                   # T a multiple of 32: x_recon is contiguous, the fused loss applies
            x0 = torch.clamp(torch.randn(B, 1, 80, T, generator=g) * 0.5, -1, 1).cuda()
            noise = torch.randn(B, 1, 80, T, generator=g).cuda()
            cond = torch.randn(B, T, 256, generator=g).transpose(1, 2).cuda().requires_grad_(True)
            t = torch.tensor([5, 40, 17]).cuda()
            xn = train.q_sample_rows(gd, x0, t, noise)
            if fused:                                                      # a step outside the schedule is loud: a row of NaN (torch.gather raises)
                bad = train.q_sample_rows(gd, x0, torch.tensor([5, 10 ** 6, 17]).cuda(), noise)
                assert bool(torch.isnan(bad[1]).all()) and bool(torch.isfinite(bad[0]).all()) and bool(torch.isfinite(bad[2]).all())
            loss = gd.p_losses(x0, t, cond, noise=noise)
            assert (type(loss.grad_fn).__name__ == '_L1MeanBackward') == fused
            (loss * 3.0).backward()                                        # a scaled loss: the upstream gradient is a device scalar, not 1
            res[fused] = (xn.cpu(), float(loss), {k: p.grad.detach().cpu() for k, p in net.named_parameters()}, cond.grad.detach().cpu())
    finally:
        train.set_fused_ends(True)
    assert torch.equal(res[True][0], res[False][0])
    assert abs(res[True][1] - res[False][1]) <= 1e-6 * abs(res[False][1]) and res[False][1] > 0
    for k, gref in res[False][2].items():
        assert torch.equal(res[True][2][k], gref), k
    assert torch.equal(res[True][3], res[False][3])


def test_dcond_follows_the_weights_over_optimizer_steps(monkeypatch):
    """the packed Wc^T of the dcond convolution must be rebuilt when a STOCK torch optimiser changes the conditioner weights
    (joint FastSpeech2 training, usr/diffsinger_task.py:60-64): three optimiser steps with cond.requires_grad on the fused path and on the
    operator-by-operator path (DSD_TRAIN_FUSED=0) - dcond of every step must agree, and must differ from step to step."""
    import diffsinger_amd
    from diffsinger_amd import hparams
    from tests import helpers as H
    pre = H.presets()['opencpop_ds60_rel']
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('DSD_TRAIN_FUSED', mode)
        hparams.clear()
        diffsinger_amd.use_preset('opencpop_ds60_rel')
        torch.manual_seed(11)
        net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
        torch.nn.init.normal_(net.output_projection.weight, std=0.02)
        gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1',
                                              spec_min=pre['spec_min'], spec_max=pre['spec_max']).cuda().train()
        # a stock torch optimiser that updates the parameters in place.  SGD: smooth in the gradient (Adam divides by |g|: the 1e-6 differences
        # between the two paths flip the sign of near-zero gradients and the NETS part within two steps - measured, profiles/r07_diag_dcond_*.txt)
        opt = torch.optim.SGD(net.parameters(), lr=10.0)
        g = torch.Generator().manual_seed(4)
        x0 = torch.clamp(torch.randn(2, 1, 80, 70, generator=g) * 0.5, -1, 1).cuda()
        noise = torch.randn(2, 1, 80, 70, generator=g).cuda()
        cond0 = torch.randn(2, 70, 256, generator=g).transpose(1, 2).cuda()
        t = torch.tensor([9, 33]).cuda()
        dconds = []
        for it in range(3):
            cond = cond0.clone().requires_grad_(True)
            opt.zero_grad(set_to_none=True)
            loss = gd.p_losses(x0, t, cond, noise=noise)
            loss.backward()
            dconds.append(cond.grad.detach().cpu().clone())
            opt.step()
        res[mode] = dconds
    for it in range(3):
        ref = res['0'][it]
        e = float((res['1'][it] - ref).abs().max() / float(ref.abs().max()))
        print(f'step {it}: dcond fused vs operator path {e:.2e}')
        assert e <= 1e-4, (it, e)
    moved = float((res['0'][2] - res['0'][0]).abs().max() / float(res['0'][0].abs().max()))
    assert moved > 3e-3, moved                      # the test is only meaningful when the weights changed dcond


@pytest.mark.parametrize('loss_type,masked', [('l1', True), ('l2', False)])
def test_p_losses_variants_on_the_fused_path(loss_type, masked):
    """The masked L1 loss (shallow_diffusion_tts.py:222-226: `* nonpadding.unsqueeze(1)`) and the L2 loss (:228) through the fused stack: loss and every
    parameter gradient against torch autograd on the CPU oracle; padded frames of a shorter utterance carry no loss."""
    import diffsinger_amd
    from diffsinger_amd import hparams
    from oracle import diffnet_oracle as O
    from tests import helpers as H
    pre = H.presets()['lj_ds_beta6']
    cfg = H.net_config(pre)
    params = {k: v.clone().requires_grad_(True) for k, v in H.oracle_params(cfg).items()}
    B, T = 3, 77
    # The seed is chosen for CONDITIONING: the loss is a mean over 18 480 elements (gradients ~1e-4), so one ReLU whose pre-activation sits
    # within the forward's 3e-6 of zero and falls on the other side changes skip_projection.weight's gradient by 5 % of its max (seed 23:
    # min |pre-activation| 8.4e-8, element (0, 41, 47) - found with tools/diag_linear_rows.py when the step MLP moved off rocBLAS and its
    # outputs moved by 6e-7).  The margin of this seed is asserted below on the oracle.
    g = torch.Generator().manual_seed(26)
    x0 = torch.clamp(torch.randn(B, 1, 80, T, generator=g) * 0.5, -1, 1)
    noise = torch.randn(B, 1, 80, T, generator=g)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    t = torch.tensor([3, 99, 41])
    keep = torch.ones(B, 1, T)                                          # [B,1,T]: nonpadding.unsqueeze(1) multiplies [B,1,M,T] like the reference's broadcast
    keep[1, :, 50:] = 0
    keep[2, :, 10:] = 0
    sch = O.make_schedule(H.betas_for(pre))
    margins, relu0 = [], F.relu

    def relu_probe(x, *a, **kw):
        margins.append(float(x.detach().abs().min()))
        return relu0(x, *a, **kw)
    F.relu = relu_probe                                                 # (oracle.F is this module)
    try:
        eps_ref = O.diffnet_forward(params, cfg, O.q_sample(sch, x0, t, noise), t, cond)
    finally:
        F.relu = relu0
    assert len(margins) == 2 and min(margins) >= 8e-6, f'ill-conditioned case: a ReLU pre-activation within {min(margins):.1e} of zero'
    if loss_type == 'l1':
        loss_ref = ((noise - eps_ref).abs() * keep.unsqueeze(1)).mean() if masked else (noise - eps_ref).abs().mean()
    else:
        loss_ref = F.mse_loss(noise, eps_ref)
    loss_ref.backward()
    hparams.clear()
    diffsinger_amd.use_preset('lj_ds_beta6')
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    net.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type=loss_type,
                                          spec_min=pre['spec_min'], spec_max=pre['spec_max']).cuda().train()
    loss = gd.p_losses(x0.cuda(), t.cuda(), cond.cuda(), noise=noise.cuda(), nonpadding=keep.cuda() if masked else None)
    loss.backward()
    worst = 0.0
    for k, p in net.named_parameters():
        worst = max(worst, float((p.grad.cpu() - params[k].grad).abs().max() / max(float(params[k].grad.abs().max()), 1e-30)))
    print(f'{loss_type} masked={masked}: loss {float(loss.detach()):.6f} (ref {float(loss_ref.detach()):.6f}), worst gradient rel err {worst:.2e}')
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 2e-6 * abs(float(loss_ref.detach())) and worst <= 2e-4
