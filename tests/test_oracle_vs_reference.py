"""Build container only (skipped where /root/reference is absent): the oracle restatement against the
LIVE reference on a configuration the goldens do not cover (narrow model, dilation cycle 2, cosine
schedule, per-step comparison of p_sample / q_sample / p_sample_plms; p_losses and the gradients of all parameters, with and
without the non-padding mask - the oracle of the training row)."""
import os
import subprocess
import sys

import pytest

from oracle.ref_driver import reference_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from collections import deque
from oracle.ref_driver import Reference
from oracle import diffnet_oracle as O
ov = dict(residual_channels=32, residual_layers=5, hidden_size=48, dilation_cycle_length=2,
          audio_num_mel_bins=20, keep_bins=20, schedule_type=%(sched)r, use_midi=False)
ref = Reference('usr/configs/lj_ds_beta6.yaml', ov)
smin, smax = [-5.0 - 0.01 * i for i in range(20)], [0.5 - 0.02 * i for i in range(20)]
net, gd = ref.build(7, 0.05, 50, 50, smin, smax)
cfg = O.NetConfig(20, 32, 48, 5, 2)
p = O.init_diffnet_params(cfg, 7, 0.05)
for k, v in net.state_dict().items():
    assert torch.equal(v, p[k]), k
betas = O.linear_beta_schedule(50, 0.06) if %(sched)r == 'linear' else O.cosine_beta_schedule(50)
sch = O.make_schedule(betas)
for k, v in sch.items():
    assert torch.equal(v, getattr(gd, k)), k
g = torch.Generator().manual_seed(3)
B, T = 3, 41
cond = torch.randn(B, T, 48, generator=g).transpose(1, 2)
x = torch.randn(B, 1, 20, T, generator=g)
z = torch.randn(B, 1, 20, T, generator=g)
with torch.no_grad():
    t = torch.tensor([49, 7, 0])
    assert torch.equal(net(x, t, cond=cond), O.diffnet_forward(p, cfg, x, t, cond))
    ref.sdt.noise_like = lambda shape, device, repeat=False: z
    assert torch.equal(gd.p_sample(x, t, cond), O.p_sample(p, cfg, sch, x, t, cond, z))
    assert torch.equal(gd.q_sample(x, torch.tensor([30]), noise=z), O.q_sample(sch, x, torch.tensor([30]), z))
    mel = torch.randn(B, T, 20, generator=g)
    sm, sx = gd.spec_min, gd.spec_max
    assert torch.equal(gd.norm_spec(mel), O.norm_spec(mel, sm, sx))
    assert torch.equal(gd.denorm_spec(mel), O.denorm_spec(mel, sm, sx))
    # PLMS, B = 1 (the only batch size the reference supports), every warm-up order + t < interval tail
    xb, cb = x[:1], cond[:1]
    gd.noise_list = deque(maxlen=4); hist = deque(maxlen=4)
    xr = xo = xb
    for i in reversed(range(0, 50, 8)):
        tt = torch.full((1,), i, dtype=torch.long)
        xr = gd.p_sample_plms(xr, tt, 8, cb)
        xo = O.p_sample_plms(p, cfg, sch, xo, tt, 8, cb, hist)
        assert torch.equal(xr, xo), i
# training (row f3): p_losses and the gradient of EVERY parameter - the reference module under autograd vs torch autograd on the oracle
for nonpad in (None, (torch.rand(B, T, generator=g) > 0.2).float()):
    tt = torch.tensor([49, 7, 0])
    net.zero_grad()
    loss_r = gd.p_losses(x, tt, cond, noise=z, nonpadding=nonpad)
    loss_r.backward()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    diff = z - O.diffnet_forward(po, cfg, O.q_sample(sch, x, tt, z), tt, cond)
    loss_o = (diff.abs() * nonpad.unsqueeze(1)).mean() if nonpad is not None else diff.abs().mean()
    loss_o.backward()
    assert torch.equal(loss_r.detach(), loss_o.detach()), (float(loss_r), float(loss_o))
    for k, v in net.named_parameters():
        assert v.grad is not None and po[k].grad is not None, k
        assert torch.equal(v.grad, po[k].grad), (k, float((v.grad - po[k].grad).abs().max()))
print('REFERENCE_EQUAL_OK')
'''


@pytest.mark.skipif(not reference_available(), reason='/root/reference not mounted')
@pytest.mark.parametrize('sched', ['linear', 'cosine'])
def test_oracle_bit_equal_to_live_reference(sched):
    res = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, sched=sched)], capture_output=True, text=True)
    assert 'REFERENCE_EQUAL_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
