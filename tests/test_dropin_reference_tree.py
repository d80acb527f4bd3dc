"""Build container only (skipped where /root/reference is absent): the drop-in surface of the package INSIDE the
reference tree - what INTEGRATION.md section 2a relies on.  No compute (no GPU here): construction, registry rebinding,
strict state_dict exchange in both directions, buffer equality, shared hparams dict, loud failure without a device."""
import os
import subprocess
import sys

import pytest

from oracle.ref_driver import reference_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import torch
from oracle.ref_driver import Reference
ref = Reference(%(config)r)
hp = ref.hparams
import diffsinger_amd
from diffsinger_amd import hparams as our_hp
assert our_hp is hp, 'inside the reference tree the package must share utils.hparams.hparams'

# registry: same lookup expression as usr/diffspeech_task.py:27 / usr/diffsinger_task.py:54
reg = {'wavenet': lambda h: ref.DiffNet(h['audio_num_mel_bins'])}
diffsinger_amd.register(reg)
assert set(reg) == {'wavenet', 'wavenet_hip'}
torch.manual_seed(1234)
ours = reg[hp['diff_decoder_type']](hp)
assert isinstance(ours, diffsinger_amd.DiffNet)
torch.manual_seed(1234)
theirs = ref.DiffNet(hp['audio_num_mel_bins'])
sd_o, sd_t = ours.state_dict(), theirs.state_dict()
assert list(sd_o.keys()) == list(sd_t.keys())
for k in sd_t:
    assert sd_o[k].shape == sd_t[k].shape and torch.equal(sd_o[k], sd_t[k]), k     # same init RNG consumption
torch.nn.init.normal_(theirs.output_projection.weight, std=0.02)
ours.load_state_dict(theirs.state_dict(), strict=True)
theirs.load_state_dict(ours.state_dict(), strict=True)

# sampler: constructor signature of shallow_diffusion_tts.py:72-73, fs2 built like :76-79, 14 buffers
enc = ref.TokenTextEncoder(None, vocab_list=['a', 'b', 'c'], replace_oov=',')
kw = dict(timesteps=hp['timesteps'], K_step=hp['K_step'], loss_type=hp['diff_loss_type'],
          spec_min=hp['spec_min'], spec_max=hp['spec_max'])
torch.manual_seed(7)
gd_t = ref.sdt.GaussianDiffusion(phone_encoder=enc, out_dims=hp['audio_num_mel_bins'], denoise_fn=theirs, **kw)
torch.manual_seed(7)
gd_o = diffsinger_amd.GaussianDiffusion(phone_encoder=enc, out_dims=hp['audio_num_mel_bins'], denoise_fn=ours, **kw)
assert type(gd_o.fs2) is type(gd_t.fs2), (type(gd_o.fs2), type(gd_t.fs2))
so, st = gd_o.state_dict(), gd_t.state_dict()
assert set(so.keys()) == set(st.keys()), set(so.keys()) ^ set(st.keys())
bufs_o, bufs_t = dict(gd_o.named_buffers(recurse=False)), dict(gd_t.named_buffers(recurse=False))
assert list(bufs_o) == list(bufs_t) and len(bufs_o) == 14
for k in bufs_t:
    assert bufs_o[k].dtype == bufs_t[k].dtype and torch.equal(bufs_o[k], bufs_t[k]), k
gd_o.load_state_dict(gd_t.state_dict(), strict=True)                                  # a reference checkpoint loads
gd_t.load_state_dict(gd_o.state_dict(), strict=True)
for attr in ('K_step', 'num_timesteps', 'mel_bins', 'loss_type'):
    assert getattr(gd_o, attr) == getattr(gd_t, attr), attr
assert gd_o.noise_list.maxlen == 4
for name in ('q_sample', 'p_sample', 'p_sample_plms', 'norm_spec', 'denorm_spec', 'cwt2f0_norm', 'out2mel', 'forward', 'inference'):
    assert callable(getattr(gd_o, name)), name
mel = torch.randn(2, 9, hp['audio_num_mel_bins'])
assert torch.equal(gd_o.norm_spec(mel), gd_t.norm_spec(mel)) and torch.equal(gd_o.denorm_spec(mel), gd_t.denorm_spec(mel))

# no device -> loud failure, never another implementation
x = torch.randn(1, 1, hp['audio_num_mel_bins'], 8); cond = torch.randn(1, hp['hidden_size'], 8)
for call in (lambda: ours(x, torch.tensor([3]), cond), lambda: gd_o.inference(cond, x_T=x, pndm_speedup=0, K_step=2)):
    try:
        with torch.no_grad():
            call()
    except RuntimeError as e:
        assert 'no CPU path' in str(e) or 'HIP' in str(e), e
    else:
        raise AssertionError('CPU call must raise')
# the training branch exists (HIP operators with hand-written gradients, diffsinger_amd/train.py) but, like inference, only on the device
try:
    ours(x.requires_grad_(False), torch.tensor([3]), cond)            # autograd on, parameters require grad -> training path
except RuntimeError as e:
    assert 'no CPU path' in str(e), e
else:
    raise AssertionError('CPU training call must raise')
print('DROPIN_OK')
'''


@pytest.mark.skipif(not reference_available(), reason='/root/reference not mounted')
@pytest.mark.parametrize('config', ['usr/configs/lj_ds_beta6.yaml', 'usr/configs/midi/e2e/opencpop/ds1000.yaml'])
def test_dropin_surface_inside_reference_tree(config):
    res = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, config=config)], capture_output=True, text=True)
    assert 'DROPIN_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
