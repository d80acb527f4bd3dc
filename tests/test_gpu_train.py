"""GPU: the training slice of row f3 (diffsinger_amd/train.py) - loss and the gradients of ALL DiffNet parameters from the HIP
forward / data-gradient / weight-gradient operators against torch autograd on the CPU oracle (oracle/diffnet_oracle.py, the
reference's own arithmetic), plus the single operators against torch's conv1d backward.

Tolerances: fp32 throughout; a gradient is a sum over B*T frames, so errors are judged relative to the tensor's max-abs gradient:
<= 2e-4 per parameter tensor, loss <= 1e-6 relative."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import diffnet_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,T,Ci,Co,K,dil', [(2, 50, 256, 512, 3, 1), (3, 77, 256, 512, 3, 8), (2, 64, 256, 256, 1, 1), (2, 45, 80, 256, 1, 1),
                                           (2, 70, 256, 80, 1, 1)])
def test_conv_backward_operators(B, T, Ci, Co, K, dil):
    from diffsinger_amd import fs2
    from diffsinger_amd.train import ConvCache
    g = torch.Generator().manual_seed(T + Co)
    x = torch.randn(B, Ci, T, generator=g)
    w = (torch.randn(Co, Ci, K, generator=g) * (Ci * K) ** -0.5).requires_grad_(True)
    bias = (torch.randn(Co, generator=g) * 0.1).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = F.conv1d(xr, w, bias, padding=dil * (K - 1) // 2, dilation=dil)
    dy = torch.randn(B, Co, T, generator=g)
    y.backward(dy)
    d = torch.device('cuda', 0)
    TS = fs2.padded_frames(T)
    xd = F.pad(x, (0, TS - T)).to(d).requires_grad_(True)
    wd, bd = w.detach().to(d).requires_grad_(True), bias.detach().to(d).requires_grad_(True)
    yd = ConvCache()(xd, wd, bd, T, dil)
    err_y = float((yd[:, :, :T].cpu() - y).abs().max())
    dyd = F.pad(dy, (0, TS - T)).to(d)
    dyd[:, :, T:] = 7.0                                  # garbage in the tail must not leak into any gradient
    yd.backward(dyd)
    e = {'y': err_y, 'dx': float((xd.grad[:, :, :T].cpu() - xr.grad).abs().max() / xr.grad.abs().max()),
         'dw': float((wd.grad.cpu() - w.grad).abs().max() / w.grad.abs().max()), 'db': float((bd.grad.cpu() - bias.grad).abs().max() / bias.grad.abs().max())}
    print(f'conv Ci={Ci} Co={Co} K={K} dil={dil}: ' + ', '.join(f'{k} {v:.2e}' for k, v in e.items()))
    assert float(xd.grad[:, :, T:].abs().max() if TS > T else 0) == 0
    assert e['y'] <= 2e-5 and e['dx'] <= 2e-5 and e['dw'] <= 2e-5 and e['db'] <= 2e-5


@pytest.mark.parametrize('rows,n_in,n_out', [(8, 256, 1024), (3, 1024, 256), (5, 256, 20 * 256), (1, 384, 384), (48, 100, 70)])
def test_linear_rows_forward_and_gradients(rows, n_in, n_out):
    """dsf_linear_rows / dsf_linear_rows_bwd (the step-embedding MLP and the layers' step projections under training, usr/diff/net.py:94-98,
    :119-120, :67) against torch's Linear on the CPU: fp32, sums over up to 5120 terms in k order - 5e-6 relative to the tensor's max-abs."""
    from diffsinger_amd.train import linear_rows
    g = torch.Generator().manual_seed(rows + n_out)
    x = torch.randn(rows, n_in, generator=g).requires_grad_(True)
    w = (torch.randn(n_out, n_in, generator=g) * n_in ** -0.5).requires_grad_(True)
    b = (torch.randn(n_out, generator=g) * 0.1).requires_grad_(True)
    y = F.linear(x, w, b)
    dy = torch.randn(rows, n_out, generator=g)
    y.backward(dy)
    d = torch.device('cuda', 0)
    xd, wd, bd = (t.detach().to(d).requires_grad_(True) for t in (x, w, b))
    yd = linear_rows(xd, wd, bd)
    yd.backward(dy.to(d))
    rel = lambda got, want: float((got.cpu() - want).abs().max() / want.abs().max())
    e = {'y': rel(yd.detach(), y.detach()), 'dx': rel(xd.grad, x.grad), 'dw': rel(wd.grad, w.grad), 'db': rel(bd.grad, b.grad)}
    print(f'linear_rows {rows}x{n_in}->{n_out}: {e}')
    assert max(e.values()) <= 5e-6, e
    # no bias, input without gradient (the first MLP layer reads the sinusoidal embedding)
    y2 = linear_rows(xd.detach(), wd, None)
    assert rel(y2.detach(), F.linear(x, w).detach()) <= 5e-6


@pytest.mark.parametrize('preset,B,T', [('opencpop_ds60_rel', 2, 50), ('lj_ds_beta6', 3, 96)])
def test_p_losses_and_all_parameter_gradients(preset, B, T):
    import diffsinger_amd
    from diffsinger_amd import hparams
    pre = H.presets()[preset]
    cfg = H.net_config(pre)
    params = {k: v.clone().requires_grad_(True) for k, v in H.oracle_params(cfg).items()}
    g = torch.Generator().manual_seed(17)
    x0 = torch.clamp(torch.randn(B, 1, 80, T, generator=g) * 0.5, -1, 1)
    noise = torch.randn(B, 1, 80, T, generator=g)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    t = torch.tensor([37, 0, 59][:B])
    sch = O.make_schedule(H.betas_for(pre))
    xn = O.q_sample(sch, x0, t, noise)
    loss_ref = (noise - O.diffnet_forward(params, cfg, xn, t, cond)).abs().mean()
    loss_ref.backward()

    hparams.clear()
    diffsinger_amd.use_preset(preset)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    net.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1',
                                          spec_min=pre['spec_min'], spec_max=pre['spec_max']).cuda().train()
    loss = gd.p_losses(x0.cuda(), t.cuda(), cond.cuda(), noise=noise.cuda())
    loss.backward()
    rel_loss = abs(float(loss) - float(loss_ref)) / abs(float(loss_ref))
    worst = ('', 0.0)
    for k, p in net.named_parameters():
        gr = params[k].grad
        assert p.grad is not None and gr is not None, k
        e = float((p.grad.cpu() - gr).abs().max() / max(float(gr.abs().max()), 1e-30))
        if e > worst[1]:
            worst = (k, e)
    print(f'{preset}: loss {float(loss):.6f} (ref {float(loss_ref):.6f}, rel err {rel_loss:.2e}); worst gradient rel err {worst[1]:.2e} at {worst[0]}')
    assert rel_loss <= 1e-6 and worst[1] <= 2e-4
    # the inference path still works on the same module afterwards (weights re-packed on demand)
    with torch.no_grad():
        eps = net(xn.cuda(), t.cuda(), cond.cuda())
    assert float((eps.cpu() - O.diffnet_forward({k: v.detach() for k, v in params.items()}, cfg, xn, t, cond)).abs().max()) <= 1e-5

