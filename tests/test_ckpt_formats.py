"""CPU: the on-disk formats of row f4 (diffsinger_amd/ckpt.py) - the reference's checkpoint layout and loader semantics, the optimiser
state exchange between ShardedAdamW and torch.optim.AdamW, the offline aux-decoder mels - including a round trip through the
reference's own utils.load_ckpt in the build container."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from diffsinger_amd import ckpt as CK
from diffsinger_amd.train_dist import ShardedAdamW
from oracle.ref_driver import reference_available
from tests.test_train_dist_gloo import HP, data, make_model, torch_adamw_update

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_save_and_load_follow_the_reference_layout(tmp_path, capsys):
    m = make_model()
    for step in (9, 100, 10):                                         # numeric, not lexicographic, order decides which is newest
        with torch.no_grad():
            m[0].bias.fill_(float(step))
        p = CK.save_ckpt(str(tmp_path), m, step, epoch=3, optimizer_states=[{'k': 1}])
        assert os.path.basename(p) == f'model_ckpt_steps_{step}.ckpt' and not os.path.exists(p + '.part')
    raw = torch.load(os.path.join(tmp_path, 'model_ckpt_steps_100.ckpt'), map_location='cpu')
    assert set(raw) >= {'epoch', 'global_step', 'optimizer_states', 'lr_schedulers', 'state_dict', 'checkpoint_callback_best'}
    assert raw['global_step'] == 100 and all(k.startswith('model.') for k in raw['state_dict'])
    m2 = make_model()
    used = CK.load_ckpt(m2, str(tmp_path), 'model')
    assert used.endswith('model_ckpt_steps_100.ckpt') and float(m2[0].bias.detach()[0]) == 100.0
    assert "| load 'model' from" in capsys.readouterr().out
    CK.load_ckpt(m2, os.path.join(tmp_path, 'model_ckpt_steps_9.ckpt'), 'model')       # a file instead of a directory
    assert float(m2[0].bias.detach()[0]) == 9.0


def test_missing_and_mismatched(tmp_path):
    m = make_model()
    with pytest.raises(AssertionError, match='ckpt not found'):
        CK.load_ckpt(m, str(tmp_path), 'model')
    assert CK.load_ckpt(m, str(tmp_path), 'model', force=False) is None
    CK.save_ckpt(str(tmp_path), m, 5)
    other = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Tanh(), torch.nn.Linear(13, 6), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    with pytest.raises(RuntimeError):
        CK.load_ckpt(other, str(tmp_path), 'model', strict=True)
    before = other[2].weight.clone()
    CK.load_ckpt(other, str(tmp_path), 'model', strict=False)          # shape-mismatched tensors are dropped, the rest is loaded
    assert torch.equal(other[2].weight, before) and torch.equal(other[0].weight, m[0].weight)
    CK.load_ckpt(m, str(tmp_path), 'nothing_under_this_prefix', strict=False)


def _train(m, opt, steps, clip=1.0):
    x, y = data(12)
    for _ in range(steps):
        (((m(x) - y).abs()).mean() * 40).backward()
        if isinstance(opt, torch.optim.Optimizer):
            torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
        opt.step()
        opt.zero_grad()


def test_optimizer_state_moves_between_sharded_and_torch_adamw():
    ma, mb = make_model(), make_model()
    oa = ShardedAdamW(ma.parameters(), clip_grad_norm=1.0, _update=torch_adamw_update, **HP)
    ob = torch.optim.AdamW(mb.parameters(), **HP)
    _train(ma, oa, 3)
    _train(mb, ob, 3)
    sd = CK.adamw_state_from_sharded(oa)
    for i, p in enumerate(mb.parameters()):                           # same moments as the reference optimiser holds
        assert float((sd['state'][i]['exp_avg'] - ob.state[p]['exp_avg']).abs().max()) < 1e-6
        assert sd['state'][i]['exp_avg'].shape == p.shape and float(sd['state'][i]['step']) == 3.0
    # sharded -> torch: a fresh torch optimiser resumes from the converted state
    mc = make_model()
    mc.load_state_dict(ma.state_dict())
    oc = torch.optim.AdamW(mc.parameters(), **HP)
    oc.load_state_dict(sd)
    # torch -> sharded: a fresh sharded optimiser resumes from the torch state
    md = make_model()
    md.load_state_dict(mb.state_dict())
    od = ShardedAdamW(md.parameters(), clip_grad_norm=1.0, _update=torch_adamw_update, **HP)
    CK.adamw_state_to_sharded(od, ob.state_dict())
    assert od.step_count == 3
    for m, o in ((ma, oa), (mb, ob), (mc, oc), (md, od)):
        _train(m, o, 2)
    for pa, pb, pc, pd in zip(*[[q.detach() for q in m.parameters()] for m in (ma, mb, mc, md)]):
        assert float((pa - pb).abs().max()) < 2e-6 and float((pc - pb).abs().max()) < 2e-6 and float((pd - pb).abs().max()) < 2e-6


def test_offline_mels_round_trip(tmp_path):
    g = torch.Generator().manual_seed(1)
    mels = {'utt_a': torch.randn(37, 80, generator=g).numpy(), 'utt_b': torch.randn(52, 80, generator=g).numpy()}
    for k, v in mels.items():
        CK.save_offline_mel(str(tmp_path), k, v)
    batch = CK.load_offline_mels(os.path.join(tmp_path, 'model_ckpt_steps_160000.ckpt'), ['utt_b', 'utt_a'])
    assert batch.shape == (2, 52, 80) and batch.dtype == torch.float32
    assert np.array_equal(batch[1, :37].numpy(), mels['utt_a']) and float(batch[1, 37:].abs().sum()) == 0.0
    assert np.array_equal(batch[0].numpy(), mels['utt_b'])


CHILD = r'''
import os, sys, types
sys.path.insert(0, %(root)r)
for n in ('librosa', 'pycwt'):
    sys.modules.setdefault(n, types.ModuleType(n))
import torch
from diffsinger_amd import ckpt as CK
from tests.test_train_dist_gloo import make_model
ref = %(ref)r
sys.path.insert(0, ref)
os.chdir(ref)
import utils                                                          # the reference's utils/__init__.py
m = make_model()
CK.save_ckpt(%(tmp)r, m, 160000)
m2 = make_model()
with torch.no_grad():
    for p in m2.parameters():
        p.zero_()
utils.load_ckpt(m2, %(tmp)r, 'model', strict=True)                    # the reference reads what we wrote
assert all(torch.equal(a, b) for a, b in zip(m.parameters(), m2.parameters()))
torch.save({'state_dict': {'model.' + k: v for k, v in m.state_dict().items()}, 'global_step': 7}, os.path.join(%(tmp)r, 'model_ckpt_steps_170000.ckpt'))
m3 = make_model()
with torch.no_grad():
    for p in m3.parameters():
        p.zero_()
assert CK.load_ckpt(m3, %(tmp)r, 'model').endswith('170000.ckpt')      # and we read what the reference layout holds
assert all(torch.equal(a, b) for a, b in zip(m.parameters(), m3.parameters()))
print('CKPT_INTEROP_OK')
'''


@pytest.mark.skipif(not reference_available(), reason='/root/reference not mounted')
def test_interop_with_the_reference_loader(tmp_path):
    from oracle.ref_driver import REFERENCE_ROOT
    res = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, ref=REFERENCE_ROOT, tmp=str(tmp_path))], capture_output=True, text=True)
    assert 'CKPT_INTEROP_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
