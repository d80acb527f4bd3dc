"""Developer diagnostic (GPU): p_losses gradients with dsf_linear_rows vs torch's F.linear for the step MLP / step projections, per parameter."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import diffsinger_amd
from diffsinger_amd import hparams, train
from tests import helpers as H
from oracle import diffnet_oracle as O


def main():
    pre = H.presets()['lj_ds_beta6']
    cfg = H.net_config(pre)
    params = {k: v.clone().requires_grad_(True) for k, v in H.oracle_params(cfg).items()}
    B, T = 3, 77
    g = torch.Generator().manual_seed(23)
    x0 = torch.clamp(torch.randn(B, 1, 80, T, generator=g) * 0.5, -1, 1)
    noise = torch.randn(B, 1, 80, T, generator=g)
    cond = torch.randn(B, T, 256, generator=g).transpose(1, 2)
    t = torch.tensor([3, 99, 41])
    sch = O.make_schedule(H.betas_for(pre))
    eps_ref = O.diffnet_forward(params, cfg, O.q_sample(sch, x0, t, noise), t, cond)
    F.mse_loss(noise, eps_ref).backward()
    res = {}
    relu_dump = {}
    orig = train.linear_rows
    def poison(value):
        # fill the caching allocator's free lists with `value`: every torch.empty() below gets memory that holds it
        blocks = []
        for rep in range(3):
            for k in range(8, 27):
                blocks.append(torch.full((2 ** k + 64 * rep,), value, device='cuda'))
        del blocks

    class FwdOnly(torch.autograd.Function):          # my forward kernel, torch's backward arithmetic
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return orig(x.detach(), w.detach(), b.detach())

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            return dy @ w, dy.t() @ x, dy.sum(0)

    class BwdOnly(torch.autograd.Function):          # torch's forward, my backward kernels
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return F.linear(x, w, b)

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            xr, wr, br = x.detach().requires_grad_(True), w.detach().requires_grad_(True), torch.zeros(w.shape[0], device=w.device, requires_grad=True)
            with torch.enable_grad():
                y = orig(xr, wr, br)
            return torch.autograd.grad(y, (xr, wr, br), dy)

    def small_only(x, w, b):
        return orig(x, w, b) if w.shape[0] <= 1024 else F.linear(x, w, b)

    def big_only(x, w, b):
        return orig(x, w, b) if w.shape[0] > 1024 else F.linear(x, w, b)

    def perturbed(x, w, b):
        return F.linear(x, w, b) * (1 + 3e-7)

    def torch_plus_launch(x, w, b):
        y = F.linear(x, w, b)
        orig(x.detach(), w.detach(), b.detach())           # the kernels run, their result is dropped
        return y

    def torch_plus_alloc(x, w, b):
        y = F.linear(x, w, b)
        junk = torch.empty(x.shape[0], w.shape[0], device=x.device).fill_(1.0)
        del junk
        return y

    dvals = {}

    def rec(tag, f):
        def g(x, w, b):
            y = f(x, w, b)
            dvals.setdefault(tag, []).append(y.detach().cpu().clone())
            return y
        return g

    variants = {'rows': orig, 'torch': lambda x, w, b: F.linear(x, w, b), 'perturbed': perturbed, 'torch+launch': torch_plus_launch,
                'torch+alloc': torch_plus_alloc, 'mlp_only': small_only}
    variants = {k: rec(k, variants[k]) for k in ('rows', 'torch')}
    from diffsinger_amd import train_fused
    log = {}
    cur = ['']

    relus = []

    class Shim:
        def __getattr__(self, k):
            return getattr(F, k)

        @staticmethod
        def relu(x):
            y = F.relu(x)
            torch.cuda.synchronize()
            relus.append((y, y.detach().clone(), x, x.detach().clone()))
            return y
    train.F = Shim()

    def wrap(cls, name):
        inner = cls.backward

        def bw(ctx, *gs):
            torch.cuda.synchronize()
            pre = [(float((y.detach() - c).abs().max()), float((x.detach() - cx).abs().max())) for y, c, x, cx in relus[-2:]]
            gs_pre = [g.detach().cpu().clone() for g in gs if torch.is_tensor(g)]
            outs = inner(ctx, *gs)
            torch.cuda.synchronize()
            log.setdefault(cur[0] + ':pre', []).append(gs_pre)
            changed = [float((g.detach().cpu() - a).abs().max()) for g, a in zip([g for g in gs if torch.is_tensor(g)], gs_pre)]
            print(f'[{cur[0]}] backward {name}: grad-in changed during the node by {changed}')
            post = [(float((y.detach() - c).abs().max()), float((x.detach() - cx).abs().max())) for y, c, x, cx in relus[-2:]]
            print(f'[{cur[0]}] backward {name}: relu outputs / inputs changed since forward: before this node {pre}, after {post}; data_ptrs y {[hex(r[0].data_ptr()) for r in relus[-2:]]} '
                  f'grads in {[hex(g.data_ptr()) for g in gs if torch.is_tensor(g)]} out {[hex(o.data_ptr()) for o in outs if torch.is_tensor(o)]}')
            log.setdefault(cur[0], []).append((name, [g.detach().cpu().clone() for g in gs if torch.is_tensor(g)],
                                               [o.detach().cpu().clone() if torch.is_tensor(o) else None for o in outs]))
            return outs
        cls.backward = staticmethod(bw)

    wrap(train._Conv1dCM, 'conv')
    wrap(train_fused._ResidualStack, 'stack')
    for mode in variants:
        train.linear_rows = variants[mode]
        cur[0] = mode
        if mode == 'torch+nan':
            poison(float('nan'))
        if mode == 'torch+7':
            poison(7.0)
        hparams.clear()
        diffsinger_amd.use_preset('lj_ds_beta6')
        net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
        net.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
        gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l2',
                                              spec_min=pre['spec_min'], spec_max=pre['spec_max']).cuda().train()
        with torch.no_grad():
            eps_ng = train.diffnet_forward_train(net, O.q_sample(sch, x0, t, noise).cuda(), t.cuda(), cond.cuda())
        eps_g = train.diffnet_forward_train(net, O.q_sample(sch, x0, t, noise).cuda(), t.cuda(), cond.cuda())
        print(mode, 'eps vs oracle: no_grad', float((eps_ng.cpu() - eps_ref.detach()).abs().max()), 'grad mode', float((eps_g.detach().cpu() - eps_ref.detach()).abs().max()),
              'max|eps|', float(eps_ref.abs().max()))
        loss = gd.p_losses(x0.cuda(), t.cuda(), cond.cuda(), noise=noise.cuda())
        relu_dump[mode] = [(r[1].cpu(), r[3].cpu()) for r in relus[-2:]]
        loss.backward()
        res[mode] = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
        print(mode, 'loss', float(loss))
    rel = lambda a, b: float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))
    for mode in dvals:
        print(mode, 'linear outputs vs torch:', [f'{rel(a, b):.2e}' for a, b in zip(dvals[mode], dvals['torch'])])
    os.makedirs('gpurun_out/diag', exist_ok=True)
    torch.save({'log': {k: v[:2] for k, v in log.items()}, 'relus': relu_dump}, 'gpurun_out/diag/diag.pt')
    relz = lambda a, b: float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))
    for i, (a, b) in enumerate(zip(log['rows'], log['torch'])):
        gi = [f'{relz(x, y):.1e}' for x, y in zip(a[1], b[1])]
        go = [('-' if x is None else f'{relz(x, y):.1e}') for x, y in zip(a[2], b[2])][:8]
        tails = [('-' if x is None or x.dim() != 3 or x.shape[2] <= T else f'{float(x[:, :, T:].abs().max()):.1e}') for x in a[2]][:3]
        gpre = [f'{relz(x, y):.1e}' for x, y in zip(log['rows:pre'][i], log['torch:pre'][i])]
        print(f'backward call {i} {a[0]}: grad-in BEFORE the node rows-vs-torch {gpre}')
        print(f'backward call {i} {a[0]}: grad-in rows-vs-torch {gi}  outputs {go}  |tail| of outputs (rows mode) {tails}  shapes {[tuple(x.shape) for x in a[1]]}')
    for mode in res:
        rows = sorted(((rel(res[mode][k], params[k].grad), k) for k in res[mode]), key=lambda r: (not (r[0] == r[0]), -r[0] if r[0] == r[0] else 0))
        bad = [r for r in rows if not (r[0] <= 2e-4)]
        print(f'{mode}: {len(bad)} of {len(rows)} parameter gradients off by more than 2e-4 (or NaN); worst:')
        for r in (bad[:4] + rows[:2]):
            print(f'    {r[0]:.3e}  {r[1]}')
        if mode == 'rows':
            ok = [r[1] for r in rows if r[0] <= 2e-4]
            print('    within 2e-4:', ok)
            for k in ('output_projection.weight', 'output_projection.bias', 'skip_projection.weight', 'skip_projection.bias', 'input_projection.weight',
                      'mlp.0.weight', 'mlp.2.weight', 'residual_layers.0.diffusion_projection.weight', 'residual_layers.19.output_projection.bias',
                      'residual_layers.19.output_projection.weight'):
                print(f'    {k}: {rel(res[mode][k], params[k].grad):.3e}')


if __name__ == '__main__':
    main()
