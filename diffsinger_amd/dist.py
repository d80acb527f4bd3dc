"""Multi-GPU inference: utterances shard across ranks, one RCCL gather collates the finished mels.

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


def sharded_inference(model, conds: Sequence[torch.Tensor], *, micro_batch: int = 16, group=None, dst: int = 0, **infer_kw):
    """Run `model.inference` over a list of equal-length utterance conditioners [H,T] (already on this rank's
    device), sharded r::W, in micro-batches, and gather the mels on `dst` in original order.
    Per-batch keyword tensors (x_T, noise, fs2_mels, ...) are passed as callables `idx_list -> batched tensor`."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
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

    try:
        outs = run_shard()
    except RuntimeError as e:
        if 'spin bound' not in str(e):
            raise
        # every rank must still arrive at the collective: repeat this rank's shard - the engine is parked on the hipGraph path (per-layer
        # kernels, no co-residency requirement) since the report - and only raise if that fails as well
        import warnings
        warnings.warn(f'rank {rank}: {e}  -- repeating the shard on the hipGraph path')
        outs = run_shard()
    some = next(c for c in conds if c is not None)          # a rank only needs ITS utterances' conditioners; the others may be None
    T = some.shape[-1]
    local = torch.cat(outs) if outs else torch.zeros(0, T, model.mel_bins, device=some.device)
    if world == 1:
        return local
    return gather_mels(local, len(conds), dst=dst, group=group)


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
