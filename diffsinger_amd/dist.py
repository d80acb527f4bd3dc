"""Multi-GPU inference: utterances shard across ranks, one RCCL gather collates the finished mels (equal or different lengths).

The path shards naturally: every op of the denoiser and sampler is per-utterance, so rank
r takes utterances r, r+W, r+2W, ... (what the reference's DDP inference does with batches, tasks/tts/tts.py:
85-88), runs the whole K-step loop with ZERO communication, and only the finished [n_local, T, M] mels are
gathered to rank 0 - a flat gather over xGMI's direct peer links (each rank has its own link to the root; no
ring).  The reference collates through the filesystem (tasks/tts/fs2.py:414-431); the gather is new.

One process per GPU (`torch.distributed`, backend 'nccl' == RCCL on ROCm; 'gloo' on CPU for the tests)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Utterance i -> rank i mod W."""
    return list(range(rank, n_items, world))


def unshard_order(n_items: int, world: int) -> List[int]:
    """Position in the rank-major concatenation [rank0 items..., rank1 items..., ...] of each original item."""
    order = []
    for r in range(world):
        order.extend(shard_indices(n_items, r, world))
    inv = [0] * n_items
    for pos, i in enumerate(order):
        inv[i] = pos
    return inv


def gather_mels(local: torch.Tensor, n_items: int, dst: int = 0, group=None, order: str = 'original') -> Optional[torch.Tensor]:
    """local: this rank's [n_local, T, M] mels (all ranks the same T, M; n_local may differ by one).
    Returns on `dst` the [n_items, T, M] tensor in ORIGINAL utterance order, None elsewhere.
    One collective into ONE receive buffer [W, n_max, T, M] (the gather list is its W slices; ranks with fewer items pad to the maximum
    with zeros).  Utterance i = k W + r sits at [r][k], so the original order is the transposed view [n_max, W] flattened: one strided copy
    (`order='original'`), or no copy at all for a consumer that indexes the view itself (`order='view'` returns that [n_max, W, T, M] view;
    entry [k][r] is utterance k W + r, entries with k W + r >= n_items are padding)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_max = (n_items + world - 1) // world
    T, M = local.shape[1], local.shape[2]
    buf = local
    if local.shape[0] < n_max:
        buf = torch.zeros(n_max, T, M, dtype=local.dtype, device=local.device)
        buf[:local.shape[0]] = local
    buf = buf.contiguous()
    if rank == dst:
        recv = torch.empty(world, n_max, T, M, dtype=local.dtype, device=local.device)
        dist.gather(buf, gather_list=list(recv.unbind(0)), dst=dst, group=group)
        view = recv.transpose(0, 1)
        if order == 'view':
            return view
        return view.reshape(n_max * world, T, M)[:n_items]
    dist.gather(buf, gather_list=None, dst=dst, group=group)
    return None


def plan_ragged(lengths: Sequence[int], world: int, *, micro_batch: int = 16, max_frames: Optional[int] = None):
    """Length-aware work split for utterances of DIFFERENT lengths (frames).  Returns (batches, owner):

      batches  micro-batches of utterance indices, formed from the lengths ALONE (never from the world size): utterances sorted by length,
               consecutive ones that share a 32-frame bucket (ceil32(T): the denoiser's tile, so padding inside a micro-batch costs no extra
               tile) are grouped up to `micro_batch` utterances and `max_frames` padded frames (B x ceil32(T_max), the reference's
               max_tokens: utils/__init__.py:89-142 batch_by_size over length-sorted indices, tasks/tts/tts.py:64-69).  A micro-batch runs
               at T = its longest utterance; the shorter ones are zero-padded inside it only.
      owner    owner[k] = rank of micro-batch k: longest-processing-time greedy on padded frames (heaviest micro-batch first, always to the
               least loaded rank; ties to the lower rank) - not r::W, which is blind to length (the reference's DDP split
               batches[rank::world], tasks/tts/tts.py:85-88, shards token-budgeted batches and is balanced for the same reason).

    Because the micro-batches do not depend on `world`, utterance i is computed inside the same micro-batch at every world size: the collated
    result is bit-identical to the single-process run."""
    n = len(lengths)
    order = sorted(range(n), key=lambda i: (int(lengths[i]), i))
    ceil32 = lambda t: (int(t) + 31) // 32 * 32
    batches: List[List[int]] = []
    cur: List[int] = []
    for i in order:
        ts = ceil32(lengths[i])
        if cur and (ceil32(lengths[cur[0]]) != ts or len(cur) >= micro_batch or (max_frames is not None and (len(cur) + 1) * ts > max_frames)):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    cost = [len(bt) * ceil32(max(lengths[i] for i in bt)) for bt in batches]
    load = [0] * world
    owner = [0] * len(batches)
    for k in sorted(range(len(batches)), key=lambda k: (-cost[k], k)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[k] = r
        load[r] += cost[k]
    return batches, owner


def pad_stack(tensors: Sequence[torch.Tensor], T: int) -> torch.Tensor:
    """Stack tensors whose LAST axis is the frame axis (lengths <= T) into one batch, zero-padded to T."""
    out = tensors[0].new_zeros((len(tensors),) + tuple(tensors[0].shape[:-1]) + (T,))
    for b, t in enumerate(tensors):
        out[b, ..., :t.shape[-1]] = t
    return out


class ShardFailed(RuntimeError):
    """Raised on EVERY rank of a sharded call when the shard of at least one rank failed: `.rank` is the lowest failed rank (on that rank the
    original exception is the __cause__)."""

    def __init__(self, rank: int, here: bool, detail: str = ''):
        self.rank = rank
        super().__init__(f'sharded inference: the shard of rank {rank} failed' + (f' ({detail})' if here and detail else '') +
                         ('' if here else ' - this rank\'s shard was fine; nothing was gathered'))


def _agree_or_raise(err: Optional[BaseException], rank: int, world: int, group, device):
    """A failed rank must not leave the others blocked in the gather (rounds 2-5: it raised before `dist.gather`, the other ranks waited in the
    collective for ever).  Before every gather the ranks exchange ONE int - all_reduce(MIN) over `rank if failed else world` - and on any
    failure every rank raises the same ShardFailed naming the lowest failed rank; nobody enters the gather."""
    if world == 1:
        if err is not None:
            raise err
        return
    dev = _collective_device(group) if device is None else device
    st = torch.tensor([rank if err is not None else world], dtype=torch.int32, device=dev)
    dist.all_reduce(st, op=dist.ReduceOp.MIN, group=group)
    bad = int(st.item())
    if bad < world:
        if err is not None:
            raise ShardFailed(bad, bad == rank, f'{type(err).__name__}: {err}') from err
        raise ShardFailed(bad, False)


# How many following shards a handle stays pinned on the hipGraph path after a shard had to be repeated there (restoring the
# caller's mode at once re-arms the persistent loop - dsd_set_loop_mode clears the handle's own 16-call parking - against a foreign kernel
# that may still hold CUs: every later shard would run into the seconds-long spin bound again)
PIN_SHARDS_AFTER_RETRY = 4


def _engine_of(model):
    return getattr(getattr(model, 'denoise_fn', None), '_engine', None)


def _unpin_if_due(model):
    eng = _engine_of(model)
    pin = getattr(eng, '_dist_pin', None) if eng is not None else None
    if pin is None:
        return
    pin['left'] -= 1
    if pin['left'] <= 0:
        eng.set_loop_mode(pin['prev'])                   # (an explicit choice re-arms the persistent path)
        eng._dist_pin = None


def _run_checked(model, run_shard, rank):
    """run_shard() with the loud-failure contract of the persistent loops (include/dsd.h dsd_check): a loop starved by a foreign kernel
    raises 'spin bound'; every rank must still arrive at the collective, so the shard is repeated - with the engine PINNED on the per-layer
    hipGraph path (loop mode 0: no co-residency requirement) for the whole retry AND the next PIN_SHARDS_AFTER_RETRY shards, then its mode restored.  (The handle parks itself
    after a report, but only for 16 sampling calls: a shard of more micro-batches than that - 96 ragged utterances in micro-batches of 8 -
    would re-arm the persistent loop in the middle of the retry, against the same foreign kernel: ADVICE r4.)  Loops of LATER micro-batches
    may have been enqueued against the same foreign kernel and latch their timeout after the report: drain the stream and swallow those
    late reports before every retry."""
    _unpin_if_due(model)
    try:
        return run_shard()
    except RuntimeError as e:
        if 'spin bound' not in str(e):
            raise
        first = e
    import warnings
    warnings.warn(f'rank {rank}: {first}  -- repeating the shard on the hipGraph path')
    eng = _engine_of(model)
    pin = getattr(eng, '_dist_pin', None) if eng is not None else None
    prev = pin['prev'] if pin else (eng.requested_loop_mode() if eng is not None and hasattr(eng, 'requested_loop_mode') else None)
    if prev is not None:
        eng.set_loop_mode(0)
        # stays pinned for the next shards too; _unpin_if_due restores the caller's mode
        eng._dist_pin = {'prev': prev, 'left': PIN_SHARDS_AFTER_RETRY}
    return _retry_shard(model, run_shard, first)


def _retry_shard(model, run_shard, first):
    for attempt in range(3):
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        if hasattr(model, 'check_loops'):
            try:
                model.check_loops()
            except RuntimeError as late:
                if 'spin bound' not in str(late):
                    raise
        try:
            return run_shard()
        except RuntimeError as e:
            if 'spin bound' not in str(e) or attempt == 2:
                raise
    raise first


def sharded_inference(model, conds: Sequence[torch.Tensor], *, micro_batch: int = 16, group=None, dst: int = 0, lengths: Optional[Sequence[int]] = None,
                      max_frames: Optional[int] = None, **infer_kw):
    """Run `model.inference` over a list of utterance conditioners [H,T_i] (already on this rank's device; a rank only needs ITS utterances'
    conditioners, the others may be None when `lengths` is given), sharded over the ranks, in micro-batches, and collate the mels on `dst`.

    Equal lengths (the bench's BASELINE configs[4]): utterances r::W (what the reference's DDP inference does with batches,
    tasks/tts/tts.py:85-88), one gather; returns [n, T, M] on `dst` in original order, None elsewhere.
    Different lengths (real test sets: LJSpeech 200-1550 frames): `plan_ragged` - length-sorted, 32-frame-bucketed, frame-budgeted
    micro-batches dealt to the ranks by a longest-processing-time greedy on frames; ONE padded gather of every rank's flat [frames, M]
    buffer, no lengths exchange (every rank computes the same plan); returns on `dst` the list of n mels [T_i, M] in original order (views
    of the receive buffer), None elsewhere.
    Per-batch keyword tensors (x_T, noise, fs2_mels, ...) are passed as callables `idx_list -> batched tensor` (ragged: padded to the
    longest utterance of idx_list along the frame axis, see pad_stack)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if lengths is None:             # (a rank may hold None for utterances it does not own: those then have the common length)
        some_ = next(c for c in conds if c is not None)
        lengths = [(c.shape[-1] if c is not None else some_.shape[-1]) for c in conds]
    lengths = [int(v) for v in lengths]
    if len(lengths) != len(conds):
        raise ValueError('lengths must have one entry per utterance')
    if len(set(lengths)) > 1:
        return _sharded_inference_ragged(model, conds, lengths, world, rank, micro_batch, max_frames, group, dst, infer_kw)
    mine = shard_indices(len(conds), rank, world)

    def run_shard():
        outs = []
        for s in range(0, len(mine), micro_batch):
            idx = mine[s:s + micro_batch]
            cond = torch.stack([conds[i] for i in idx])
            kw = {k: (v(idx) if callable(v) else v) for k, v in infer_kw.items()}
            outs.append(model.inference(cond, **kw))
        if hasattr(model, 'check_loops'):
            model.check_loops()             # ONE wait per shard: a persistent loop starved by a foreign kernel must not reach the gather as NaN
        return outs

    some = next(c for c in conds if c is not None)          # a rank only needs ITS utterances' conditioners; the others may be None
    outs, err = None, None
    try:
        outs = _run_checked(model, run_shard, rank)
    except Exception as e:                                  # noqa: BLE001 - whatever it was, the other ranks must hear of it
        err = e
    _agree_or_raise(err, rank, world, group, some.device if dist.is_initialized() and dist.get_backend(group) == 'nccl' else None)
    T = some.shape[-1]
    local = torch.cat(outs) if outs else torch.zeros(0, T, model.mel_bins, device=some.device)
    if world == 1:
        return local
    return gather_mels(local, len(conds), dst=dst, group=group)


def _sharded_inference_ragged(model, conds, lengths, world, rank, micro_batch, max_frames, group, dst, infer_kw):
    batches, owner = plan_ragged(lengths, world, micro_batch=micro_batch, max_frames=max_frames)
    my_batches = [bt for bt, r in zip(batches, owner) if r == rank]
    some = next(c for c in conds if c is not None)
    M = model.mel_bins

    def run_shard():
        outs = []
        for idx in my_batches:
            T_run = max(lengths[i] for i in idx)
            cond = pad_stack([conds[i] for i in idx], T_run)
            kw = {k: (v(idx) if callable(v) else v) for k, v in infer_kw.items()}
            mel = model.inference(cond, **kw)                                   # [B, T_run, M]
            outs.extend(mel[b, :lengths[i]] for b, i in enumerate(idx))
        if hasattr(model, 'check_loops'):
            model.check_loops()
        return outs

    outs, err = None, None
    try:
        outs = _run_checked(model, run_shard, rank)
    except Exception as e:                                  # noqa: BLE001
        err = e
    _agree_or_raise(err, rank, world, group, some.device if dist.is_initialized() and dist.get_backend(group) == 'nccl' else None)
    # every rank knows every rank's utterances and lengths (the plan is a function of `lengths` and `world`): one gather of flat
    # [frames, M] buffers padded to the heaviest rank, no lengths collective
    seq = [[i for bt, r in zip(batches, owner) if r == rr for i in bt] for rr in range(world)]
    frames = [sum(lengths[i] for i in s_) for s_ in seq]
    if world == 1:
        out = [None] * len(conds)
        for i, m in zip(seq[0], outs):
            out[i] = m
        return out
    F = max(frames)
    flat = torch.zeros(F, M, dtype=torch.float32, device=some.device)
    if outs:
        flat[:frames[rank]] = torch.cat(outs)
    if rank == dst:
        recv = torch.empty(world, F, M, dtype=flat.dtype, device=flat.device)
        dist.gather(flat, gather_list=list(recv.unbind(0)), dst=dst, group=group)
        out = [None] * len(conds)
        for rr in range(world):
            pos = 0
            for i in seq[rr]:
                out[i] = recv[rr, pos:pos + lengths[i]]
                pos += lengths[i]
        return out
    dist.gather(flat, gather_list=None, dst=dst, group=group)
    return None


def _collective_device(group=None) -> torch.device:
    """Where a tensor has to live to take part in a collective of this group: the current HIP device under RCCL, the host under gloo."""
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')


def gather_ragged(local: Sequence[torch.Tensor], n_items: int, dst: int = 0, group=None, device=None, dtype=None):
    """Collate per-utterance 1-D results of DIFFERENT lengths (waveforms of the vocoder stage, f0 tracks) sharded r::W: this rank's
    tensors in shard order -> on `dst` the list of n_items tensors in ORIGINAL utterance order, None elsewhere.  Three collectives: the
    lengths (one int64 gather), the longest length (one all-reduce), then one gather of the payloads padded to the longest - the reference
    writes one file per utterance from every rank instead (tasks/tts/fs2.py:414-431).  A rank WITHOUT items (n_items < world) still takes
    part: its buffers live on `device` (default: the current HIP device under RCCL, the host under gloo) with `dtype` (default float32;
    pass it when the payloads are not float32 - every rank must use the same dtype)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_max = (n_items + world - 1) // world
    dev = torch.device(device) if device is not None else (local[0].device if len(local) else _collective_device(group))
    dtype = dtype if dtype is not None else (local[0].dtype if len(local) else torch.float32)
    lens = torch.zeros(n_max, dtype=torch.int64, device=dev)
    for i, t in enumerate(local):
        lens[i] = t.numel()
    all_lens = [torch.empty_like(lens) for _ in range(world)] if rank == dst else None
    dist.gather(lens, gather_list=all_lens, dst=dst, group=group)
    longest = lens.max().reshape(1)
    dist.all_reduce(longest, op=dist.ReduceOp.MAX, group=group)
    L = int(longest.item())
    buf = torch.zeros(n_max, L, dtype=dtype, device=dev)
    for i, t in enumerate(local):
        buf[i, :t.numel()] = t.reshape(-1)
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gather_list=parts, dst=dst, group=group)
    if rank != dst:
        return None
    out = [None] * n_items
    for r in range(world):
        for k, i in enumerate(shard_indices(n_items, r, world)):
            out[i] = parts[r][k, :int(all_lens[r][k])].clone()
    return out
