"""GPU: the parts of the reference's sampler surface beyond the plain inference loop (b):

  * `p_sample(x, t, cond, clip_denoised, repeat_noise)` with a step index PER UTTERANCE, without the clamp, with one repeated noise draw
    (usr/diff/shallow_diffusion_tts.py:159-166, noise_like :38-41) - dsd_p_sample_ex - against the oracle;
  * `OfflineGaussianDiffusion.forward` (:291-323) and the legacy `usr/diff/diffusion.py::GaussianDiffusion.forward` (:296-320), both
    branches, with the HIP FastSpeech2 attached: infer=True against FastSpeech2-oracle -> sampler-oracle (<= 1e-4 on the mel), infer=False
    (p_losses on the HIP training operators) against the oracle's loss on the same t / noise;
  * two handles sampling on two streams of one device (the persistent loop's one-at-a-time rule is enforced by the library);
  * the REAL sampler through dist.sharded_inference / gather_mels on a world-1 RCCL group: bit-identical to the unsharded batch.
Everything goes through ctypes -> the C ABI (include/dsd.h, dsf.h)."""
import os
import socket

import numpy as np
import pytest
import torch

import diffsinger_amd
from diffsinger_amd import hparams
from diffsinger_amd.synth import make_inputs, presets
from oracle import diffnet_oracle as O
from tests import fs2_helpers as FH
from tests import helpers as H
from tests.gpu_helpers import build_hip

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _lj(k_step=100):
    gd, cfg, pre = build_hip('lj_ds_beta6', k_step)
    return gd, cfg, pre, H.oracle_params(cfg), O.make_schedule(H.betas_for(pre))


@pytest.mark.parametrize('clip,repeat,ts', [(True, False, [99, 0, 37]), (False, False, [5, 5, 5]), (True, True, [60, 60, 60]),
                                            (False, True, [0, 1, 98])])
def test_p_sample_full_signature_matches_oracle(clip, repeat, ts):
    gd, cfg, pre, p, sch = _lj()
    B, T = 3, 77
    inp = make_inputs(31, B, T, n_noise=1)
    x, cond = inp['x_T'] * 0.7, inp['cond']
    z = inp['noise'][0]
    if repeat:
        z = z[:1]                                           # noise_like(repeat=True): one [1,1,M,T] draw for the whole batch
    t = torch.tensor(ts)
    with torch.no_grad():
        want = O.p_sample(p, cfg, sch, x, t, cond, z.expand(B, -1, -1, -1), clip_denoised=clip)
    got = gd.p_sample(x.to(DEV), t.to(DEV), cond.to(DEV), clip_denoised=clip, repeat_noise=repeat, noise=z.to(DEV))
    assert got.shape == x.shape
    scale = max(1.0, float(want.abs().max()))               # without the clamp x_0 is O(1/sqrt(alpha_bar)) at large t
    err = float((got.cpu() - want).abs().max()) / scale
    print(f'p_sample clip={clip} repeat={repeat} t={ts}: max-abs err {err:.3e} (/ {scale:.3g})')
    assert err <= 1e-5
    # no explicit noise: the draw has the reference's shape and the call still works
    out = gd.p_sample(x.to(DEV), t.to(DEV), cond.to(DEV), clip_denoised=clip, repeat_noise=repeat)
    assert bool(torch.isfinite(out).all())


def _fs2_and_gd(case_name, cls, **ctor):
    case, fs2m, hp, params, inp = FH.case_setup(case_name)
    pre = presets()[case['preset']]
    cfg = H.net_config(pre)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    net.load_state_dict(H.oracle_params(cfg), strict=True)
    gd = cls(None, 80, net, loss_type='l1', spec_min=pre['spec_min'], spec_max=pre['spec_max'], fs2=fs2m, **ctor).to(DEV)
    return case, gd, hp, params, inp, pre, cfg


def _oracle_cond(hp, params, inp):
    from oracle import fs2_oracle as FO
    kw = {k: v.clone() for k, v in inp.items() if k != 'txt_tokens'}
    with torch.no_grad():
        r = FO.fs2_forward(FH.oracle_params(params), hp, inp['txt_tokens'], skip_decoder=True, **kw)
    return r['decoder_inp'].detach().transpose(1, 2)


def _inject(gd, **fixed):
    """forward() draws its noise itself; the test makes the draws explicit by wrapping inference()."""
    orig = gd.inference
    gd.inference = lambda cond, **kw: orig(cond, **{**kw, **fixed})


def test_offline_gaussian_diffusion_forward_infer_matches_oracles():
    from diffsinger_amd.diffusion import OfflineGaussianDiffusion
    K = 12
    case, gd, hp, params, inp, pre, cfg = _fs2_and_gd('fs2_popcs_teacher', OfflineGaussianDiffusion, timesteps=pre_timesteps('popcs_ds_beta6'), K_step=K)
    gd.eval()
    B, T = inp['mel2ph'].shape
    g = torch.Generator().manual_seed(77)
    smin = torch.tensor(pre['spec_min'])[None, None, :]
    smax = torch.tensor(pre['spec_max'])[None, None, :]
    target = (torch.clamp(torch.randn(B, T, 80, generator=g) * 0.5, -1, 1) + 1) / 2 * (smax - smin) + smin
    offline = (torch.clamp(torch.randn(B, T, 80, generator=g) * 0.5, -1, 1) + 1) / 2 * (smax - smin) + smin      # P_mels_npy aux mel
    qn = torch.randn(B, 1, 80, T, generator=g)
    nz = torch.randn(K, B, 1, 80, T, generator=g)
    _inject(gd, q_noise=qn.to(DEV), noise=nz.to(DEV))
    kw = {k: v.to(DEV) for k, v in inp.items() if k != 'txt_tokens'}
    with torch.no_grad():
        ret = gd(inp['txt_tokens'].to(DEV), ref_mels=[target.to(DEV), offline.to(DEV)], infer=True, **kw)
    sch = O.make_schedule(H.betas_for(pre))
    want = O.infer_mel(H.oracle_params(cfg), cfg, sch, _oracle_cond(hp, params, inp), smin, smax, k_step=K, noises=list(nz), fs2_mel=offline, q_noise=qn)
    err = float((ret['mel_out'].cpu() - want).abs().max())
    print(f'OfflineGaussianDiffusion.forward(infer=True): max-abs mel err {err:.3e}')
    assert ret['mel_out'].shape == (B, T, 80) and err <= 1e-4
    assert 'fs2_mel' not in ret                               # skip_decoder=True: no aux-decoder mel (:295-296)


def pre_timesteps(name):
    return presets()[name]['timesteps']


def test_offline_gaussian_diffusion_forward_train_matches_oracle_loss():
    from diffsinger_amd.diffusion import OfflineGaussianDiffusion
    K = 51
    case, gd, hp, params, inp, pre, cfg = _fs2_and_gd('fs2_popcs_teacher', OfflineGaussianDiffusion, timesteps=pre_timesteps('popcs_ds_beta6'), K_step=K)
    gd.train()
    gd.fs2.eval()                                             # dropout off in the conditioner: the oracle it is compared with has none
    B, T = inp['mel2ph'].shape
    g = torch.Generator().manual_seed(78)
    smin = torch.tensor(pre['spec_min'])[None, None, :]
    smax = torch.tensor(pre['spec_max'])[None, None, :]
    target = (torch.clamp(torch.randn(B, T, 80, generator=g) * 0.5, -1, 1) + 1) / 2 * (smax - smin) + smin
    kw = {k: v.to(DEV) for k, v in inp.items() if k != 'txt_tokens'}
    torch.manual_seed(5)
    ret = gd(inp['txt_tokens'].to(DEV), ref_mels=[target.to(DEV), target.to(DEV)], infer=False, **kw)
    loss = ret['diff_loss']
    loss.backward()
    assert gd.denoise_fn.input_projection.weight.grad is not None
    # the draws forward() made: t = randint(0, K_step, (B,)), then noise = randn_like(x) inside p_losses - x is the TRANSPOSED VIEW
    # norm_spec(mel).transpose(1, 2)[:, None] (:302-304) and randn_like fills a tensor of those strides
    torch.manual_seed(5)
    t = torch.randint(0, K, (B,), device=DEV).long()
    noise = torch.randn_like(gd.norm_spec(target.to(DEV)).transpose(1, 2)[:, None, :, :])
    sch = O.make_schedule(H.betas_for(pre))
    x0 = O.norm_spec(target, smin, smax).transpose(1, 2)[:, None]
    with torch.no_grad():
        p = H.oracle_params(cfg)
        want = (noise.cpu() - O.diffnet_forward(p, cfg, O.q_sample(sch, x0, t.cpu(), noise.cpu()), t.cpu(), _oracle_cond(hp, params, inp))).abs().mean()
    print(f'OfflineGaussianDiffusion.forward(infer=False): loss {float(loss):.7f}, oracle {float(want):.7f}')
    assert abs(float(loss) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))


def test_legacy_gaussian_diffusion_forward_both_branches():
    from diffsinger_amd.legacy import GaussianDiffusion as LegacyGD
    K = 16
    case, gd, hp, params, inp, pre, cfg = _fs2_and_gd('fs2_lj_teacher', LegacyGD, timesteps=K)
    assert gd.fs2.decoder is None and gd.num_timesteps == K
    B, T = inp['mel2ph'].shape
    g = torch.Generator().manual_seed(79)
    smin = torch.tensor(pre['spec_min'])[None, None, :]
    smax = torch.tensor(pre['spec_max'])[None, None, :]
    sch = O.make_schedule(O.cosine_beta_schedule(K))             # usr/diff/diffusion.py:192-195: always cosine
    p = H.oracle_params(cfg)
    cond = _oracle_cond(hp, params, inp)
    kw = {k: v.to(DEV) for k, v in inp.items() if k != 'txt_tokens'}
    # infer=True: Gaussian start, K ancestral steps, no mask on the output (:313-320)
    gd.eval()
    x_T = torch.randn(B, 1, 80, T, generator=g)
    nz = torch.randn(K, B, 1, 80, T, generator=g)
    _inject(gd, x_T=x_T.to(DEV), noise=nz.to(DEV))
    with torch.no_grad():
        ret = gd(inp['txt_tokens'].to(DEV), infer=True, **kw)
    want = O.infer_mel(p, cfg, sch, cond, smin, smax, k_step=K, noises=list(nz), x_T=x_T)
    err = float((ret['mel_out'].cpu() - want).abs().max())
    print(f'legacy GaussianDiffusion.forward(infer=True): max-abs mel err {err:.3e}')
    assert err <= 1e-4
    # infer=False: t ~ U[0, num_timesteps), L1 with the (mel2ph != 0) factor broadcast exactly like the reference writes it (:284-286, :304-305)
    gd.train()
    gd.fs2.eval()                                             # dropout off in the conditioner: the oracle it is compared with has none
    target = (torch.clamp(torch.randn(B, T, 80, generator=g) * 0.5, -1, 1) + 1) / 2 * (smax - smin) + smin
    torch.manual_seed(6)
    loss = gd(inp['txt_tokens'].to(DEV), ref_mels=target.to(DEV), infer=False, **kw)['diff_loss']
    torch.manual_seed(6)
    t = torch.randint(0, K, (B,), device=DEV).long().cpu()
    noise = torch.randn_like(gd.norm_spec(target.to(DEV)).transpose(1, 2)[:, None, :, :]).cpu()     # strides of the view forward() builds
    x0 = O.norm_spec(target, smin, smax).transpose(1, 2)[:, None]
    with torch.no_grad():
        rec = O.diffnet_forward(p, cfg, O.q_sample(sch, x0, t, noise), t, cond)
        want_loss = ((noise - rec).abs() * (inp['mel2ph'] != 0).float().unsqueeze(1)).mean()
    print(f'legacy GaussianDiffusion.forward(infer=False): loss {float(loss):.7f}, oracle {float(want_loss):.7f}')
    assert abs(float(loss) - float(want_loss)) <= 2e-6 * max(1.0, abs(float(want_loss)))
    # the loss back-propagates through `cond` into a trainable FastSpeech2 (usr/diff/diffusion.py:296-311 under DiffFsTask, usr/task.py:56-84):
    # round 5 still froze the HIP FastSpeech2 here
    gd.zero_grad(set_to_none=True)
    loss.backward()
    enc = [(n, q.grad) for n, q in gd.fs2.named_parameters() if n.startswith('encoder.') and q.requires_grad]
    assert enc and all(g_ is not None and bool(torch.isfinite(g_).all()) for _, g_ in enc)
    assert sum(float(g_.abs().sum()) for _, g_ in enc) > 0, 'no gradient reached the FastSpeech2 encoder through the conditioner'
    assert all(q.grad is not None for q in gd.denoise_fn.parameters())


def test_two_handles_on_two_streams_are_serialised_not_starved():
    K, B, T = 12, 8, 1024                                      # 256 tiles each: either loop alone needs every CU
    gds = [_lj(K)[0] for _ in range(2)]
    inps = [make_inputs(40 + i, B, T, n_noise=K) for i in range(2)]
    dev = [{k: v.to(DEV) for k, v in inp.items()} for inp in inps]
    want = [gds[i].inference(dev[i]['cond'], x_T=dev[i]['x_T'], noise=dev[i]['noise'], K_step=K, pndm_speedup=0).clone() for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [None, None]
    for rep in range(3):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                got[i] = gds[i].inference(dev[i]['cond'], x_T=dev[i]['x_T'], noise=dev[i]['noise'], K_step=K, pndm_speedup=0)
    torch.cuda.synchronize()
    for i in (0, 1):
        eng = gds[i].denoise_fn.engine()
        assert eng.loop_mode() == 1 and eng.loop_launches() == 1
        with torch.cuda.stream(streams[i]):
            assert eng.loop_timeouts() == 0
        assert torch.equal(got[i], want[i])


def test_real_sampler_through_sharded_inference_on_rccl_world1():
    import torch.distributed as dist
    from diffsinger_amd.dist import gather_mels, sharded_inference
    K, n, T = 10, 5, 70
    gd = _lj(K)[0]
    inp = make_inputs(55, n, T, n_noise=K)
    cond, x_T, noise = inp['cond'].to(DEV), inp['x_T'].to(DEV), inp['noise'].to(DEV)
    want = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).clone()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        conds = [cond[i] for i in range(n)]
        got = sharded_inference(gd, conds, micro_batch=2, dst=0, x_T=lambda idx: x_T[idx], noise=lambda idx: noise[:, idx].contiguous(),
                                K_step=K, pndm_speedup=0)
        assert torch.equal(got, want)                         # micro-batches of (2, 2, 1): utterances are independent, bit for bit
        gathered = gather_mels(got, n, dst=0)                 # the RCCL collective itself (world 1: rank 0 receives its own shard)
        assert torch.equal(gathered, want)
        view = gather_mels(got, n, dst=0, order='view')
        assert view.shape == (n, 1, T, 80) and torch.equal(view[:, 0], want)
    finally:
        dist.destroy_process_group()


def test_utterances_of_different_lengths_through_sharded_inference_vs_the_oracle():
    """Different lengths (diffsinger_amd/dist.py plan_ragged): length-sorted, 32-frame-bucketed micro-batches, the shorter utterances of a
    micro-batch zero-padded to its longest - the reference's batched inference (collated, padded batches: tasks/tts/fs2.py:340-369,
    utils/__init__.py:89-142).  Every utterance's mel against the ORACLE run on the same padded micro-batch, original order kept."""
    from diffsinger_amd.dist import pad_stack, plan_ragged, sharded_inference
    K = 8
    gd, cfg, pre, p, sch = _lj(K)
    lengths = [70, 91, 40, 96, 85, 33, 64, 90]
    g = torch.Generator().manual_seed(77)
    conds = [torch.randn(t, 256, generator=g).t() for t in lengths]                       # [H, T_i] views of [T_i, H]
    xs = [torch.randn(1, 80, t, generator=g) for t in lengths]
    nz = [torch.randn(K, 1, 80, t, generator=g) for t in lengths]
    T_of = lambda idx: max(lengths[i] for i in idx)
    got = sharded_inference(gd, [c.to(DEV) for c in conds], micro_batch=3, max_frames=8192, dst=0, K_step=K, pndm_speedup=0,
                            x_T=lambda idx: pad_stack([xs[i] for i in idx], T_of(idx)).to(DEV),
                            noise=lambda idx: torch.stack([pad_stack([nz[i][k] for i in idx], T_of(idx)) for k in range(K)]).to(DEV))
    assert [tuple(m.shape) for m in got] == [(t, 80) for t in lengths]
    smin = torch.tensor(pre['spec_min'], dtype=torch.float32)[None, None, :]
    smax = torch.tensor(pre['spec_max'], dtype=torch.float32)[None, None, :]
    batches, owner = plan_ragged(lengths, 1, micro_batch=3, max_frames=8192)
    assert sorted(i for b in batches for i in b) == list(range(len(lengths))) and set(owner) == {0}
    worst = 0.0
    for idx in batches:
        Tm = T_of(idx)
        assert len({(lengths[i] + 31) // 32 for i in idx}) == 1                            # one 32-frame bucket per micro-batch
        want = O.infer_mel(p, cfg, sch, pad_stack([conds[i] for i in idx], Tm), smin, smax, k_step=K,
                           noises=[pad_stack([nz[i][k] for i in idx], Tm) for k in range(K)], x_T=pad_stack([xs[i] for i in idx], Tm))
        for b, i in enumerate(idx):
            worst = max(worst, float((got[i].cpu() - want[b, :lengths[i]]).abs().max()))
    print(f'ragged micro-batches {batches}: max-abs mel err vs the oracle on the same padded batches {worst:.3e}')
    assert worst <= 1e-4
