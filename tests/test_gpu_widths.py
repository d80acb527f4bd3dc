"""GPU: DiffNet widths other than the fused engine's 256 / 256 (the reference reads `residual_channels` / `hidden_size` from hparams,
usr/diff/net.py:85-90; configs/tts/base.yaml ships hidden_size 384 for plain FastSpeech2).  Such a DiffNet runs on the generic HIP
operators under the generic sampler of GaussianDiffusion - the same API and results against the oracle (the reference's arithmetic):
a single evaluation, a K-step DDPM loop with explicit noise, the PLMS loop, the shallow-diffusion start, and the training step."""
from collections import deque

import numpy as np
import pytest
import torch

from oracle import diffnet_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

WIDTHS = [(384, 256, 3, 2), (128, 384, 4, 4)]          # residual_channels, hidden_size, layers, dilation cycle


def _build(C, Hd, L, cyc):
    import diffsinger_amd
    from diffsinger_amd import hparams
    pre = H.presets()['lj_ds_beta6']
    hparams.clear()
    diffsinger_amd.use_preset('lj_ds_beta6')
    hparams.update(residual_channels=C, hidden_size=Hd, residual_layers=L, dilation_cycle_length=cyc)
    cfg = O.NetConfig(mel_bins=80, residual_channels=C, encoder_hidden=Hd, residual_layers=L, dilation_cycle_length=cyc)
    params = O.init_diffnet_params(cfg, 4321, 0.02)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    net.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
    assert not net.fused()
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1',
                                          spec_min=pre['spec_min'], spec_max=pre['spec_max']).cuda()
    sch = O.make_schedule(H.betas_for(pre))
    smin = torch.tensor(pre['spec_min'])[None, None, :]
    smax = torch.tensor(pre['spec_max'])[None, None, :]
    return gd, net, cfg, params, sch, smin, smax, pre


@pytest.mark.parametrize('C,Hd,L,cyc', WIDTHS)
def test_other_widths_match_the_oracle(C, Hd, L, cyc):
    gd, net, cfg, params, sch, smin, smax, pre = _build(C, Hd, L, cyc)
    gd.eval()
    B, T, K = 2, 70, 6
    g = torch.Generator().manual_seed(C + Hd)
    cond = torch.randn(B, T, Hd, generator=g).transpose(1, 2)
    x = torch.randn(B, 1, 80, T, generator=g)
    t = torch.tensor([17, 3])
    with torch.no_grad():
        eps_ref = O.diffnet_forward(params, cfg, x, t, cond)
        eps = net(x.cuda(), t.cuda(), cond.cuda())
    e_eps = float((eps.cpu() - eps_ref).abs().max())
    # DDPM, K steps from a Gaussian start with explicit noise
    noises = [torch.randn(B, 1, 80, T, generator=g) for _ in range(K)]
    x_ref = O.sample_ddpm(params, cfg, sch, x.clone(), cond, noises, K)
    mel_ref = O.denorm_spec(x_ref[:, 0].transpose(1, 2), smin, smax)
    mel = gd.inference(cond.cuda(), x_T=x.cuda(), noise=torch.stack(noises).cuda(), K_step=K)
    e_ddpm = float((mel.cpu() - mel_ref).abs().max())
    # shallow start (q_sample of the aux mel) + mask
    fs2_mel = O.denorm_spec(torch.clamp(torch.randn(B, T, 80, generator=g) * 0.5, -1, 1), smin, smax)
    qn = torch.randn(B, 1, 80, T, generator=g)
    mask = (torch.rand(B, T, generator=g) > 0.2).float()
    mel_ref2 = O.infer_mel(params, cfg, sch, cond, smin, smax, k_step=K, noises=noises, fs2_mel=fs2_mel, q_noise=qn, mel_mask=mask)
    mel2 = gd.inference(cond.cuda(), fs2_mels=fs2_mel.cuda(), q_noise=qn.cuda(), noise=torch.stack(noises).cuda(), K_step=K, gaussian_start=False,
                        mel_mask=mask.cuda())
    e_sh = float((mel2.cpu() - mel_ref2).abs().max())
    # PLMS: 100-step schedule, interval 20 (5 iterations, 6 evaluations), graded relative to max |x_0| (no clamp in PLMS)
    xp_ref = O.sample_plms(params, cfg, sch, x.clone(), cond, 100, 20)
    _, xp = gd.inference(cond.cuda(), x_T=x.cuda(), K_step=100, pndm_speedup=20, return_x=True)
    e_plms = float((xp.cpu() - xp_ref).abs().max() / xp_ref.abs().max())
    print(f'C={C} H={Hd} L={L}: eps {e_eps:.2e}, DDPM K={K} mel {e_ddpm:.2e}, shallow mel {e_sh:.2e}, PLMS rel {e_plms:.2e}')
    assert e_eps <= 1e-5 and e_ddpm <= 1e-4 and e_sh <= 1e-4 and e_plms <= 1e-4
    # single steps of the reference API
    tt = torch.tensor([5, 0])
    z = torch.randn(B, 1, 80, T, generator=g)
    ps_ref = O.p_sample(params, cfg, sch, x.clone(), tt, cond, z)
    ps = gd.p_sample(x.cuda(), tt.cuda(), cond.cuda(), noise=z.cuda())
    assert float((ps.cpu() - ps_ref).abs().max()) <= 1e-5
    qs = gd.q_sample(x.cuda(), torch.tensor([9]).cuda(), noise=z.cuda())
    assert float((qs.cpu() - O.q_sample(sch, x, torch.tensor([9]), z)).abs().max()) <= 1e-6


def test_other_width_trains():
    C, Hd, L, cyc = WIDTHS[0]
    gd, net, cfg, params, sch, smin, smax, pre = _build(C, Hd, L, cyc)
    gd.train()
    B, T = 2, 45
    g = torch.Generator().manual_seed(9)
    x0 = torch.clamp(torch.randn(B, 1, 80, T, generator=g) * 0.5, -1, 1)
    noise = torch.randn(B, 1, 80, T, generator=g)
    cond = torch.randn(B, T, Hd, generator=g).transpose(1, 2)
    t = torch.tensor([37, 2])
    pr = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    loss_ref = (noise - O.diffnet_forward(pr, cfg, O.q_sample(sch, x0, t, noise), t, cond)).abs().mean()
    loss_ref.backward()
    loss = gd.p_losses(x0.cuda(), t.cuda(), cond.cuda(), noise=noise.cuda())
    loss.backward()
    worst = 0.0
    for k, p in net.named_parameters():
        worst = max(worst, float((p.grad.cpu() - pr[k].grad).abs().max() / max(float(pr[k].grad.abs().max()), 1e-30)))
    print(f'C={C} H={Hd}: loss {float(loss):.6f} (ref {float(loss_ref):.6f}), worst gradient rel err {worst:.2e}')
    assert abs(float(loss) - float(loss_ref)) <= 1e-6 * abs(float(loss_ref)) and worst <= 2e-4
