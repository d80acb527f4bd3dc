"""Developer diagnostic (GPU): where do the fused and the operator-by-operator training paths part over optimizer steps?  For every step:
dcond of both paths (+ variants with their weight caches dropped before every backward) and the largest weight difference between the nets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffsinger_amd
from diffsinger_amd import hparams
from diffsinger_amd.synth import presets

OPT = sys.argv[1] if len(sys.argv) > 1 else 'adamw'
pre = presets()['opencpop_ds60_rel']


def run(fused, drop_caches):
    os.environ['DSD_TRAIN_FUSED'] = '1' if fused else '0'
    hparams.clear()
    diffsinger_amd.use_preset('opencpop_ds60_rel')
    torch.manual_seed(11)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1',
                                          spec_min=pre['spec_min'], spec_max=pre['spec_max']).cuda().train()
    opt = torch.optim.AdamW(net.parameters(), lr=2e-3, weight_decay=0.0) if OPT == 'adamw' else torch.optim.SGD(net.parameters(), lr=20.0)
    g = torch.Generator().manual_seed(4)
    x0 = torch.clamp(torch.randn(2, 1, 80, 70, generator=g) * 0.5, -1, 1).cuda()
    noise = torch.randn(2, 1, 80, 70, generator=g).cuda()
    cond0 = torch.randn(2, 70, 256, generator=g).transpose(1, 2).cuda()
    t = torch.tensor([9, 33]).cuda()
    out = []
    for it in range(4):
        if drop_caches:
            net.__dict__.pop('_train_cond_pack', None)
            net.__dict__.pop('_train_caches', None)
        cond = cond0.clone().requires_grad_(True)
        opt.zero_grad(set_to_none=True)
        loss = gd.p_losses(x0, t, cond, noise=noise)
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        out.append((float(loss), cond.grad.detach().clone(), {k: p.detach().clone() for k, p in net.named_parameters()}, grads))
        opt.step()
    return out


res = {(f, d): run(f, d) for f in (True, False) for d in (False, True)}
rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for it in range(4):
    ref = res[(False, True)][it]
    print(f'--- step {it} ({OPT}); reference = operator path with caches dropped; loss {ref[0]:.6f}')
    for key, name in (((True, False), 'fused'), ((True, True), 'fused, caches dropped'), ((False, False), 'operator path')):
        r = res[key][it]
        wd = max(((rel(r[2][k], ref[2][k]), k) for k in ref[2]))
        gd_ = max(((rel(r[3][k], ref[3][k]), k) for k in ref[3]))
        print(f'  {name:24s} loss {r[0]:.6f}  dcond rel diff {rel(r[1], ref[1]):.2e}  worst weight diff {wd[0]:.2e} ({wd[1]})  worst grad diff {gd_[0]:.2e} ({gd_[1]})')
