"""CPU, world_size 2 over gloo: bench.py's BASELINE configs[4] branch ITSELF (`bench.cfg5_workload`: shard r::W -> micro-batches -> one gather ->
shape assert on rank 0) with a stand-in sampler - the exact host code the 8-GPU run executes has then run at world > 1 somewhere (VERDICT r2
item 4).  Also: the same-workload N = 1 leg (`local_only`) covers exactly this rank's shard."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_UTTS, T, MICRO, K, M, H = 13, 40, 3, 2, 5, 8


class _Stub:
    """mel[b, t, m] = mean_h(cond[b, :, t]) + 0.01 m: a function of the utterance's conditioner only (which depends on the utterance index
    only), so the collated result must not depend on the world size.  Counts its calls and checks the per-batch keyword tensors."""
    mel_bins = M

    def __init__(self):
        self.calls = []

    def inference(self, cond, x_T=None, noise=None, K_step=None, pndm_speedup=None):
        B = cond.shape[0]
        assert cond.shape[1:] == (H, T) and x_T.shape == (B, 1, M, T) and noise.shape == (K, B, 1, M, T) and noise.is_contiguous()
        assert K_step == K and pndm_speedup == 0
        self.calls.append(B)
        return cond.mean(1)[:, :, None] + 0.01 * torch.arange(M)[None, None, :]


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import bench
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        gd = _Stub()
        step, local_only, mine, first_cond, x_T, noise = bench.cfg5_workload(gd, rank, world, torch.device('cpu'), n_utts=N_UTTS, T=T, micro=MICRO,
                                                                              K=K, M=M, H=H)
        assert mine == list(range(rank, N_UTTS, world)) and first_cond.shape == (MICRO, H, T)
        out = step()
        assert gd.calls == [MICRO] * (len(mine) // MICRO) + ([len(mine) % MICRO] if len(mine) % MICRO else [])
        out2 = step()                                            # the timed region calls it repeatedly
        if rank == 0:
            assert torch.equal(out, out2)
            q.put(out.clone())
        loc = local_only()
        assert sum(o.shape[0] for o in loc) == len(mine)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_cfg5_branch_at_world_2_equals_world_1():
    import bench
    gd = _Stub()
    step, local_only, mine, _, _, _ = bench.cfg5_workload(gd, 0, 1, torch.device('cpu'), n_utts=N_UTTS, T=T, micro=MICRO, K=K, M=M, H=H)
    want = step()
    assert want.shape == (N_UTTS, T, M) and mine == list(range(N_UTTS))
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got.shape == (N_UTTS, T, M)
    assert torch.equal(got, want)                                # utterance i is the same mel wherever it was sampled, in the original order


def test_metric_string_is_baselines():
    import json
    import bench
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))['metric']
    assert base.startswith(bench.BASELINE_METRIC)
