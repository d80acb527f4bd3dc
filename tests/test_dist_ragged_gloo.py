"""CPU, world_size 2 and 3 over gloo: utterances of DIFFERENT lengths through diffsinger_amd/dist.py ("variable-length
utterances: padded buffer + lengths").  LJSpeech-like lengths (200-1550 frames, configs/tts/base.yaml:35-39 max_frames
1550): length-sorted, 32-frame-bucketed, frame-budgeted micro-batches (the reference's batch_by_size, utils/__init__.py:89-142) dealt to the
ranks by a longest-processing-time greedy on frames, ONE padded gather.  Checked: original order, exact equality with the single-process run
(the stand-in sampler's output depends on the micro-batch an utterance is computed in - its size and padded length - so equality proves that
the micro-batches do not depend on the world size), rank balance max / mean <= 1.05."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffsinger_amd.dist import pad_stack, plan_ragged, sharded_inference

N, H, M, MICRO, BUDGET = 96, 4, 5, 8, 8192


class _Sampler:
    """mel[b, t, m] = cond[b, :, t].mean() + 0.001 m + x_T[b, 0, m, t] + 1e-3 * T_run + 1e-2 * B: a function of the utterance AND of the
    micro-batch it is computed in (like the real sampler, whose padded tail sees the micro-batch's length)."""
    mel_bins = M

    def __init__(self):
        self.frames = 0

    def inference(self, cond, x_T=None):
        B, _, T = cond.shape
        assert x_T.shape == (B, 1, M, T)
        self.frames += B * ((T + 31) // 32 * 32)
        return cond.mean(1)[:, :, None] + 0.001 * torch.arange(M)[None, None, :] + x_T[:, 0].transpose(1, 2) + 1e-3 * T + 1e-2 * B


def _inputs():
    rnd = random.Random(7)
    lengths = [rnd.randint(200, 1550) for _ in range(N)]
    g = torch.Generator().manual_seed(3)
    conds = [torch.randn(H, t, generator=g) for t in lengths]
    xs = [torch.randn(1, M, t, generator=g) for t in lengths]
    return lengths, conds, xs


def _run(world_rank=None):
    lengths, conds, xs = _inputs()
    m = _Sampler()
    out = sharded_inference(m, conds, micro_batch=MICRO, max_frames=BUDGET, dst=0,
                            x_T=lambda idx: pad_stack([xs[i] for i in idx], max(lengths[i] for i in idx)))
    return out, m.frames, lengths


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        out, frames, lengths = _run()
        fr = torch.tensor([float(frames)])
        allf = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(allf, fr)
        if rank == 0:
            q.put((torch.cat(out).numpy().copy(), [float(v) for v in allf]))      # (by value: the worker exits before the parent reads)
        else:
            assert out is None
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_ragged_lengths_equal_the_single_process_run_and_are_balanced(world):
    want, frames1, lengths = _run()
    assert [w.shape for w in want] == [(t, M) for t in lengths]
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat, per_rank = q.get()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    got = list(torch.from_numpy(flat).split(lengths))
    assert len(got) == N
    for i in range(N):
        assert got[i].shape == (lengths[i], M)
        assert torch.equal(got[i], want[i]), i                          # original order, bit-identical to world 1
    assert sum(per_rank) == frames1                                      # the same micro-batches, only dealt out
    balance = max(per_rank) / (sum(per_rank) / world)
    print(f'world {world}: padded frames per rank {per_rank}, max / mean {balance:.4f}; padding overhead {frames1 / sum(lengths):.4f}')
    assert balance <= 1.05


def test_plan_is_a_partition_bucketed_and_world_independent():
    rnd = random.Random(1)
    lengths = [rnd.randint(200, 1550) for _ in range(512)]
    b1, _ = plan_ragged(lengths, 1, micro_batch=16, max_frames=8192)
    for world in (2, 4, 8):
        b, owner = plan_ragged(lengths, world, micro_batch=16, max_frames=8192)
        assert b == b1                                                   # micro-batches never depend on the world size
        assert sorted(i for bt in b for i in bt) == list(range(512))
        load = [0] * world
        for bt, r in zip(b, owner):
            ts = {(lengths[i] + 31) // 32 for i in bt}
            assert len(ts) == 1 and len(bt) <= 16 and len(bt) * 32 * ts.pop() <= 8192       # one 32-frame bucket, within both budgets
            load[r] += len(bt) * ((max(lengths[i] for i in bt) + 31) // 32 * 32)
        assert max(load) / (sum(load) / world) <= 1.05, (world, load)
    # equal lengths stay on the reference's r::W split (bench configs[4]); a single utterance is one batch on rank 0
    assert plan_ragged([100], 4)[1] == [0]


def test_shard_retry_pins_the_engine_off_the_persistent_path_for_more_micro_batches_than_the_parking_lasts():
    """a handle parks itself for 16 sampling calls after a reported timeout; a shard of MORE micro-batches than that used
    to re-arm the persistent loop in the middle of the retry, against the foreign kernel that starved it - three failed retries, the rank
    raises before the gather and the other ranks hang.  The retry now pins loop mode 0 (per-layer hipGraph: no co-residency requirement)
    and restores the caller's mode afterwards.  Model of the engine's parking rule, no GPU."""
    from diffsinger_amd.dist import _run_checked

    class Eng:
        def __init__(self):
            self.req, self.parked, self.calls_log = 2, 0, []

        def set_loop_mode(self, m):
            self.req, self.parked = m, 0                 # an explicit choice re-arms (dsd_set_loop_mode)

        def requested_loop_mode(self):
            return self.req

        def sample(self):                                # one micro-batch; the foreign kernel is resident all the time
            persistent = self.req in (1, 2) and self.parked == 0
            self.calls_log.append(persistent)
            if persistent:
                self.parked = 16                         # reported: parked for the next 16 calls (kParkedCalls)
                raise RuntimeError('dsd_sample_ddpm: 1 persistent K-step loop launch(es) ... hit the inter-workgroup spin bound')
            if self.parked:
                self.parked -= 1

    class Net:
        pass

    class Model:
        def __init__(self):
            self.denoise_fn = Net()
            self.denoise_fn._engine = Eng()

        def check_loops(self):
            pass

    m = Model()
    eng = m.denoise_fn._engine

    def run_shard():
        for _ in range(24):                              # 24 micro-batches > 16
            eng.sample()
        return 'mels'

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert _run_checked(m, run_shard, 0) == 'mels'
    assert eng.calls_log[0] is True and not any(eng.calls_log[1:])      # one starved loop, then 24 micro-batches on the per-layer path
    # the caller's mode is NOT restored at once (dsd_set_loop_mode would clear the handle's parking and every following shard would
    # run into the seconds-long spin bound again while the foreign kernel is still there): the engine stays pinned for the next
    # PIN_SHARDS_AFTER_RETRY shards, then the mode comes back
    from diffsinger_amd.dist import PIN_SHARDS_AFTER_RETRY
    assert eng.requested_loop_mode() == 0
    for k in range(PIN_SHARDS_AFTER_RETRY - 1):
        n0 = len(eng.calls_log)
        assert _run_checked(m, run_shard, 0) == 'mels' and eng.requested_loop_mode() == 0 and not any(eng.calls_log[n0:])
    eng.sample = lambda: eng.calls_log.append('free')   # the foreign kernel has gone
    assert _run_checked(m, run_shard, 0) == 'mels'
    assert eng.requested_loop_mode() == 2                # restored, the persistent path re-armed
