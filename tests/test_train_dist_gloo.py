"""CPU, world_size 1 and 2 over gloo: the data-parallel optimiser step of diffsinger_amd/train_dist.py -
flat parameter / gradient buffers, gradient reduce-scatter, clip coefficient from one all-reduced float, AdamW on the rank's shard,
parameter all-gather - against what the reference computes: torch.optim.AdamW + clip_grad_norm_ on the gradient of the FULL batch
(= DDP's averaged gradient).  The fused HIP kernel is replaced by a torch restatement of torch.optim.AdamW's single-tensor update
(injected through `_update`, tests only); the kernel itself is checked on the GPU (tests/test_gpu_train_dist.py)."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffsinger_amd import train_dist
from diffsinger_amd.train_dist import ShardedAdamW, StepLR

HP = dict(lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)


def torch_adamw_update(p, g, m, v, lr, b1, b2, eps, wd, step, gscale):
    """torch/optim/adamw.py _single_tensor_adamw, verbatim arithmetic, on a flat range."""
    g = g * gscale
    p.mul_(1 - lr * wd)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def make_model():
    torch.manual_seed(5)
    return torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Tanh(), torch.nn.Linear(13, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def data(n):
    g = torch.Generator().manual_seed(9)
    return torch.randn(n, 7, generator=g) * 3, torch.randn(n, 3, generator=g)


def reference_run(n, steps, clip, return_opt=False):
    m = make_model()
    opt = torch.optim.AdamW(m.parameters(), **HP)
    sched = torch.optim.lr_scheduler.StepLR(opt, 2, gamma=0.5)
    x, y = data(n)
    for _ in range(steps):
        loss = ((m(x) - y).abs()).mean() * 40                        # large gradients: the clip is active
        loss.backward()
        if clip:
            torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
        opt.step()
        opt.zero_grad()
        sched.step()
    if return_opt:
        return [p.detach().clone() for p in m.parameters()], [opt.state[p]['exp_avg'].clone() for p in m.parameters()]
    return [p.detach().clone() for p in m.parameters()]


def sharded_run(rank, world, n, steps, clip):
    m = make_model()
    opt = ShardedAdamW(m.parameters(), clip_grad_norm=clip, _update=torch_adamw_update, **HP)
    sched = StepLR(opt, 2, gamma=0.5)
    assert opt.padded % (64 * world) == 0 and opt.shard * world == opt.padded and opt.total == sum(p.numel() for p in m.parameters())
    assert all(p.data_ptr() >= opt.flat_p.data_ptr() for p in m.parameters())
    x, y = data(n)
    x, y = x[rank::world], y[rank::world]                             # equal shares: mean of rank means = global mean
    for s in range(steps):
        loss = ((m(x) - y).abs()).mean() * 40
        loss.backward()
        gen = train_dist.param_generation()
        opt.step()
        assert train_dist.param_generation() == gen + 1
        opt.zero_grad()
        sched.step(s + 1)
    assert float(opt.flat_g.abs().sum()) == 0.0
    return [p.detach().clone() for p in m.parameters()], opt


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, steps, clip, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        params, opt = sharded_run(rank, world, n, steps, clip)
        from diffsinger_amd.ckpt import adamw_state_from_sharded
        sd = adamw_state_from_sharded(opt)                            # collective: the moments of all shards gathered on rank 0
        assert (sd is None) == (rank != 0)
        moments = [sd['state'][i]['exp_avg'].numpy() for i in range(len(params))] if rank == 0 else None
        q.put((rank, [p.numpy() for p in params], float(opt.last_grad_norm) if clip else None, moments))     # by value: the worker may exit first
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('clip', [None, 1.0])
def test_world2_matches_full_batch_adamw(clip):
    world, n, steps = 2, 12, 5
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, steps, clip, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get() for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want, want_m = reference_run(n, steps, clip, return_opt=True)
    got.sort(key=lambda t: t[0])
    for a, w in zip(got[0][3], want_m):                               # the gathered first moments are torch.optim.AdamW's
        assert a.shape == tuple(w.shape) and float(abs(torch.from_numpy(a) - w).max()) < 1e-6
    for a, b in zip(got[0][1], got[1][1]):
        assert (a == b).all()                                         # every rank ends with the same parameters
    for a, w in zip(got[0][1], want):
        assert float(abs(torch.from_numpy(a) - w).max()) < 2e-6
    if clip:
        assert got[0][2] > clip                                       # the clip was active in the last step


@pytest.mark.parametrize('clip', [None, 1.0, 0, 0.0])         # 0 = the reference's 'no clipping' (configs/config_base.yaml clip_grad_norm: 0;
def test_single_process_matches_adamw(clip):                   # utils/pl_utils.py:1165-1168 clips only when > 0): the parameters must MOVE
    params, opt = sharded_run(0, 1, 12, 5, clip)
    want = reference_run(12, 5, clip)
    for a, w in zip(params, want):
        assert float((a - w).abs().max()) < 2e-6
    start = [p.detach().clone() for p in make_model().parameters()]
    assert all(float((a - b).abs().max()) > 1e-4 for a, b in zip(params, start))     # training really happened (clip 0 must not zero the gradient)
    sd = opt.state_dict()
    assert sd['step'] == 5 and sd['exp_avg_shard'].numel() == opt.shard
    assert sd['exp_avg_shard'].data_ptr() != opt.exp_avg.data_ptr()                   # a snapshot, not a live reference
    opt.lr = 123.0
    opt.load_state_dict(sd)
    assert opt.lr == sd['lr'] != 123.0


def test_steplr_base_after_resuming_a_decayed_torch_state():
    """adamw_state_to_sharded takes the UN-decayed `initial_lr` a torch scheduler left in the param group as the StepLR base."""
    from diffsinger_amd.ckpt import adamw_state_from_sharded, adamw_state_to_sharded
    _, opt = sharded_run(0, 1, 12, 2, None)
    sd = adamw_state_from_sharded(opt)
    sd['param_groups'][0]['initial_lr'] = HP['lr']
    sd['param_groups'][0]['lr'] = HP['lr'] * 0.25                                     # what StepLR(2, 0.5) leaves after 4 steps
    m2 = make_model()
    opt2 = ShardedAdamW(m2.parameters(), _update=torch_adamw_update, **HP)
    adamw_state_to_sharded(opt2, sd)
    assert opt2.lr == HP['lr'] * 0.25
    sched = StepLR(opt2, 2, gamma=0.5)
    sched.step(4)
    assert abs(opt2.lr - HP['lr'] * 0.25) < 1e-12                                     # not decayed twice


def test_no_cpu_path_for_the_fused_step():
    m = make_model()
    opt = ShardedAdamW(m.parameters(), **HP)
    (m(torch.zeros(2, 7)).sum()).backward()
    with pytest.raises(RuntimeError, match='no CPU path'):
        opt.step()
