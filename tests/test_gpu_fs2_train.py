"""GPU: FastSpeech2 / FastSpeech2MIDI UNDER AUTOGRAD (SURVEY section 8 rows f1 x f3): the training forward of the HIP modules
(infer=False, skip_decoder=True as GaussianDiffusion.forward calls it, usr/diff/shallow_diffusion_tts.py:236) and its backward - convolutions on
dsf_conv1d_dilated / dsf_conv1d_wgrad, LayerNorm and the attention core on dsf_layer_norm_bwd / dsf_attention_bwd (csrc/fs2_train.hpp) - against
torch autograd on oracle/fs2_oracle.py, whose gradients are bit-equal to the live reference's (oracle/check_fs2_grad.py, build container)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.check_fs2_grad import loss_of
from oracle.fs2_cases import CASES
from tests import fs2_helpers as FH

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TEACHER = [n for n, c in CASES.items() if c['mode'] == 'teacher']


def _rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / max(float(b.abs().max()), 1e-30))


@pytest.mark.parametrize('B,T,relu_in,masked', [(2, 50, False, True), (3, 96, True, True), (1, 5, False, False), (2, 33, True, False)])
def test_layer_norm_backward_operator(B, T, relu_in, masked):
    from diffsinger_amd import fs2
    g = torch.Generator().manual_seed(T + B)
    x = torch.randn(B, T, 256, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    keep = (torch.rand(B, T, generator=g) > 0.2).float() if masked else None
    dy = torch.randn(B, T, 256, generator=g)
    xr, gr, br = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = F.layer_norm(F.relu(xr) if relu_in else xr, (256,), gr, br, 1e-5)
    if keep is not None:
        y = y * keep.double()[:, :, None]
    y.backward(dy.double())
    xc = fs2.to_cm(x.to(DEV)).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    out = fs2.layer_norm_cm(xc, T, gd, bd, 1e-5, relu_in=relu_in, keep=keep.to(DEV).contiguous() if keep is not None else None)
    dyc = fs2.to_cm(dy.to(DEV))
    dyc[:, :, T:] = 5.0                                  # garbage in the tail of the incoming gradient must not leak
    out.backward(dyc)
    e = {'y': _rel(fs2.from_cm(out.detach(), T), y.detach()), 'dx': _rel(fs2.from_cm(xc.grad, T), xr.grad), 'dgamma': _rel(gd.grad, gr.grad),
         'dbeta': _rel(bd.grad, br.grad)}
    print(f'layer_norm bwd B={B} T={T} relu_in={relu_in} masked={masked}:', {k: f'{v:.1e}' for k, v in e.items()})
    assert float(xc.grad[:, :, T:].abs().max() if xc.shape[2] > T else 0) == 0
    assert all(v <= 1e-5 for v in e.values()), e


@pytest.mark.parametrize('B,T,heads', [(2, 50, 2), (3, 21, 2), (1, 130, 2), (2, 1, 2)])
def test_attention_backward_operator(B, T, heads):
    from diffsinger_amd import fs2
    C = 128 * heads
    g = torch.Generator().manual_seed(7 * T + B)
    qkv = torch.randn(B, T, 3 * C, generator=g)
    pad = torch.zeros(B, T, dtype=torch.bool)
    for b in range(1, B):
        pad[b, max(1, T - 3 * b):] = True
    do = torch.randn(B, T, C, generator=g)
    r = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(B, T, heads, 128).transpose(1, 2) for t in r.chunk(3, -1)]
    s = (q * 128 ** -0.5) @ k.transpose(-1, -2)
    s = s.masked_fill(pad[:, None, None, :], float('-inf'))
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, C)
    o.backward(do.double())
    qc = fs2.to_cm(qkv.to(DEV)).requires_grad_(True)
    out = fs2.attention_cm(qc, T, pad.to(torch.uint8).to(DEV).contiguous(), heads)
    out.backward(fs2.to_cm(do.to(DEV)))
    e = {'o': _rel(fs2.from_cm(out.detach(), T), o.detach()), 'dqkv': _rel(fs2.from_cm(qc.grad, T), r.grad)}
    print(f'attention bwd B={B} T={T}:', {k: f'{v:.1e}' for k, v in e.items()})
    assert float(qc.grad[:, :, T:].abs().max() if qc.shape[2] > T else 0) == 0
    assert all(v <= 1e-5 for v in e.values()), e


@pytest.mark.parametrize('Ci,Co,K', [(256, 1024, 9), (256, 256, 5), (256, 1, 1), (1024, 256, 1)])
def test_conv_weight_gradient_for_wide_kernels(Ci, Co, K):
    """dsf_conv1d_wgrad beyond 3 taps (the k = 9 conv-FFN, the k = 5 pitch predictor): groups of three taps."""
    from diffsinger_amd import fs2
    from diffsinger_amd.train import ConvCache
    B, T = 2, 70
    g = torch.Generator().manual_seed(Ci + Co + K)
    x = torch.randn(B, Ci, T, generator=g)
    w = torch.randn(Co, Ci, K, generator=g) * (Ci * K) ** -0.5
    bias = torch.randn(Co, generator=g) * 0.1
    dy = torch.randn(B, Co, T, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    F.conv1d(xr, wr, br, padding=K // 2).backward(dy.double())
    TS = fs2.padded_frames(T)
    xd = F.pad(x, (0, TS - T)).to(DEV).contiguous().requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    y = ConvCache()(xd, wd, bd, T)
    y.backward(F.pad(dy, (0, TS - T)).to(DEV).contiguous())
    e = {'dx': _rel(xd.grad[:, :, :T], xr.grad), 'dw': _rel(wd.grad, wr.grad), 'db': _rel(bd.grad, br.grad)}
    print(f'conv bwd Ci={Ci} Co={Co} K={K}:', {k: f'{v:.1e}' for k, v in e.items()})
    assert all(v <= 2e-5 for v in e.values()), e


@pytest.mark.parametrize('name', TEACHER)
def test_training_forward_and_every_parameter_gradient_match_the_oracle(name):
    from oracle import fs2_oracle as FO
    case, m, hp, params, inp = FH.case_setup(name)
    # oracle (CPU, autograd)
    p = FH.oracle_params(params)
    kw = {k: v.clone() for k, v in inp.items() if k != 'txt_tokens'}
    o = FO.fs2_forward(p, hp, inp['txt_tokens'], skip_decoder=True, **kw)
    lo = loss_of(o, case['seed'])
    lo.backward()
    alias = {'encoder.embed_tokens.weight', 'encoder_embed_tokens.weight'}
    # HIP module under autograd (eval(): dropout is the identity, like the pin against the reference)
    m = m.to(DEV).eval()
    kw = {k: v.clone().to(DEV) for k, v in inp.items() if k != 'txt_tokens'}
    r = m(inp['txt_tokens'].to(DEV), skip_decoder=True, infer=False, **kw)
    assert r['decoder_inp'].requires_grad
    lh = loss_of(r, case['seed'])
    lh.backward()
    e_fwd = {k: _rel(r[k].detach(), o[k].detach()) for k in ('decoder_inp', 'dur') if k in r}
    worst, n = ('', 0.0), 0
    for k, prm in m.named_parameters():
        if k in alias:
            parts = [p[a].grad for a in alias if a in p and p[a].grad is not None]
            og = sum(parts) if parts else None
        else:
            og = p[k].grad if k in p else None
        if og is None:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, f'{k}: gradient where the oracle has none'
            continue
        assert prm.grad is not None, f'{k}: no gradient'
        e = _rel(prm.grad, og)
        n += 1
        if e > worst[1]:
            worst = (k, e)
    print(f'{name}: loss {float(lh.detach()):.6f} (oracle {float(lo.detach()):.6f}), forward {e_fwd}, {n} parameter gradients, worst rel err {worst[1]:.2e} at {worst[0]}')
    assert abs(float(lh.detach()) - float(lo.detach())) <= 2e-5 * max(1.0, abs(float(lo.detach())))
    assert all(v <= 2e-5 for v in e_fwd.values())
    assert n >= 60 and worst[1] <= 5e-6, worst                  # every FastSpeech2 parameter gradient within 5e-6 (relative)


def test_e2e_training_step_flows_from_the_diffusion_loss_into_fastspeech2():
    """The Opencpop e2e step (usr/diffsinger_task.py:273-300 with fs2_ckpt '' -> FastSpeech2MIDI trainable): GaussianDiffusion.forward(infer=False)
    = fs2(skip_decoder=True) -> p_losses; diff_loss.backward() reaches the encoder through `cond` - against the oracle pair (FastSpeech2 oracle ->
    denoiser oracle) under torch autograd on the CPU with the same t and noise."""
    import diffsinger_amd
    from diffsinger_amd import hparams
    from oracle import diffnet_oracle as O
    from oracle import fs2_oracle as FO
    from tests import helpers as H
    name = 'fs2_midi_e2e_teacher'
    case, m, hp, params, inp = FH.case_setup(name)
    pre = H.presets()[case['preset']]
    cfg = H.net_config(pre)
    dparams = {k: v.clone().requires_grad_(True) for k, v in H.oracle_params(cfg).items()}
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    net.load_state_dict({k: v.detach() for k, v in dparams.items()}, strict=True)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max'], fs2=m).to(DEV).eval()
    B, T = inp['mel2ph'].shape
    g = torch.Generator().manual_seed(99)
    mel = torch.randn(B, T, 80, generator=g) * 0.5 - 3.0
    t = torch.randint(0, pre['K_step'], (B,), generator=g)
    noise = torch.randn(B, 1, 80, T, generator=g)
    # oracle pair on the CPU
    p = FH.oracle_params(params)
    kw = {k: v.clone() for k, v in inp.items() if k != 'txt_tokens'}
    o = FO.fs2_forward(p, hp, inp['txt_tokens'], skip_decoder=True, **kw)
    cond = o['decoder_inp'].transpose(1, 2)
    sch = O.make_schedule(H.betas_for(pre))
    smin, smax = torch.tensor(pre['spec_min'])[None, None, :], torch.tensor(pre['spec_max'])[None, None, :]
    x0 = ((mel - smin) / (smax - smin) * 2 - 1).transpose(1, 2)[:, None]
    eps = O.diffnet_forward(dparams, cfg, O.q_sample(sch, x0, t, noise), t, cond)
    loss_ref = (noise - eps).abs().mean()
    loss_ref.backward()
    # HIP: the same t / noise through the module's own training branch pieces
    kw = {k: v.clone().to(DEV) for k, v in inp.items() if k != 'txt_tokens'}
    ret = gd.fs2(inp['txt_tokens'].to(DEV), skip_decoder=True, infer=False, **kw)
    x = gd.norm_spec(mel.to(DEV)).transpose(1, 2)[:, None, :, :]
    loss = gd.p_losses(x, t.to(DEV), ret['decoder_inp'].transpose(1, 2), noise=noise.to(DEV))
    loss.backward()
    alias = {'encoder.embed_tokens.weight', 'encoder_embed_tokens.weight'}
    worst, n = ('', 0.0), 0
    for k, prm in m.named_parameters():
        parts = [p[a].grad for a in (alias if k in alias else {k}) if a in p and p[a].grad is not None]
        if not parts:
            continue
        e = _rel(prm.grad, sum(parts))
        n += 1
        if e > worst[1]:
            worst = (k, e)
    print(f'e2e step: diff_loss {float(loss.detach()):.6f} (oracle {float(loss_ref.detach()):.6f}); {n} FastSpeech2MIDI gradients through cond, worst rel err {worst[1]:.2e} at {worst[0]}')
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 5e-6 * abs(float(loss_ref.detach()))
    assert n >= 40 and worst[1] <= 1e-5, worst
    # and the module's forward(infer=False) returns the loss dict key the task reads
    out = gd(inp['txt_tokens'].to(DEV), ref_mels=mel.to(DEV), infer=False, **{k: v.clone().to(DEV) for k, v in inp.items() if k != 'txt_tokens'})
    assert out['diff_loss'].requires_grad and torch.isfinite(out['diff_loss'])
