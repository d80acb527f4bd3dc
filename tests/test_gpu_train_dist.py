"""GPU: the optimiser step of row f3 - the fused AdamW kernel (dsf_adamw_step) against torch.optim.AdamW on the device, and
ShardedAdamW (single process: flat buffers + fused step + clip, no communication) driving two training steps of the DiffNet slice against
the same steps with clip_grad_norm_ + torch.optim.AdamW.  The multi-rank exchange around the kernel is covered on CPU over gloo
(tests/test_train_dist_gloo.py).

First run on the MI355X in round 2 (profiles/r02a_pytest_gpu_zz_train_dist.txt: parameters within 2.4e-7 of torch.optim.AdamW)."""
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
HP = dict(lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)


@pytest.mark.parametrize('n', [1 << 20, 1000003, 64])
def test_fused_adamw_matches_torch_optim(n):
    from diffsinger_amd.train_dist import _hip_adamw
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (10.0 ** (i - 1)) for i in range(3)]
    ref = torch.nn.Parameter(p0.clone().to(DEV))
    opt = torch.optim.AdamW([ref], **HP)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    scale = torch.tensor([0.5], device=DEV)
    for i, gr in enumerate(grads):
        ref.grad = (gr.to(DEV) * 0.5)
        opt.step()
        _hip_adamw(p, gr.to(DEV), m, v, HP['lr'], HP['betas'][0], HP['betas'][1], HP['eps'], HP['weight_decay'], i + 1, scale)
    st = opt.state[ref]
    err = float((p - ref.detach()).abs().max())
    print('adamw n', n, 'param err', err, 'm err', float((m - st['exp_avg']).abs().max()), 'v rel err',
          float(((v - st['exp_avg_sq']).abs() / (st['exp_avg_sq'].abs() + 1e-12)).max()))
    assert err < 1e-6
    assert float((m - st['exp_avg']).abs().max()) < 1e-5 * float(st['exp_avg'].abs().max())
    assert float(((v - st['exp_avg_sq']).abs() / (st['exp_avg_sq'].abs() + 1e-12)).max()) < 1e-5


def test_sharded_adamw_drives_the_training_slice():
    import copy

    import diffsinger_amd
    from diffsinger_amd import hparams
    from diffsinger_amd.train_dist import ShardedAdamW
    preset, B, T = 'lj_ds_beta6', 2, 64
    pre = H.presets()[preset]
    hparams.clear()
    diffsinger_amd.use_preset(preset)
    torch.manual_seed(3)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    mk = lambda n: diffsinger_amd.GaussianDiffusion(None, 80, n, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1',
                                                    spec_min=pre['spec_min'], spec_max=pre['spec_max']).to(DEV).train()
    gd_a, gd_b = mk(net), mk(copy.deepcopy(net))
    g = torch.Generator().manual_seed(17)
    x0 = torch.clamp(torch.randn(B, 1, 80, T, generator=g) * 0.5, -1, 1).to(DEV)
    noise = torch.randn(B, 1, 80, T, generator=g).to(DEV)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2).to(DEV)
    t = torch.tensor([37, 5], device=DEV)
    pa = [p for p in gd_a.denoise_fn.parameters()]
    pb = [p for p in gd_b.denoise_fn.parameters()]
    opt_a = ShardedAdamW(pa, clip_grad_norm=1.0, **HP)
    opt_b = torch.optim.AdamW(pb, **HP)
    losses = []
    for step in range(3):
        la = gd_a.p_losses(x0, t, cond, noise=noise) * 50
        la.backward()
        opt_a.step()
        opt_a.zero_grad()
        lb = gd_b.p_losses(x0, t, cond, noise=noise) * 50
        lb.backward()
        torch.nn.utils.clip_grad_norm_(pb, 1.0)
        opt_b.step()
        opt_b.zero_grad()
        losses.append((float(la), float(lb)))
    print('losses (sharded fused, torch):', losses)
    assert losses[0][0] == losses[0][1]                               # same weights, same kernels before the first update
    assert losses[2][0] < losses[0][0]                                # the updates were seen by the next forward (weights re-packed)
    for (la, lb) in losses:
        assert abs(la - lb) <= 1e-4 * abs(lb)
    worst = max(float((a - b).abs().max()) for a, b in zip(pa, pb))
    print('worst parameter difference after 3 steps', worst)
    assert worst < 5e-5
