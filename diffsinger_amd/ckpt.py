"""On-disk formats around the hot path: the reference's `.ckpt` files and the offline aux-decoder mels.

    load_ckpt                 utils/__init__.py:178-209 - newest `model_ckpt_steps_*.ckpt` of a work dir (or one file), the sub-dict
                              under `state_dict` whose keys start with `<prefix>.`, strict or shape-filtered load
    save_ckpt                 the checkpoint dict of Trainer.dump_checkpoint / _atomic_save (utils/pl_utils.py:813-870): epoch,
                              global_step, optimizer_states, lr_schedulers, state_dict - so a model trained here resumes in the
                              reference and vice versa
    adamw_state_from_sharded / adamw_state_to_sharded
                              ShardedAdamW's 1/W moments <-> a torch.optim.AdamW state_dict (what `optimizer_states` holds)
    load_offline_mels / save_offline_mel
                              `P_mels_npy/<item_name>.npy` written by the FFT-Singer test run (tasks/tts/fs2.py:414-431) and read by
                              ShallowDiffusionOfflineDataset (usr/diffsinger_task.py:100-118) as `fs2_mels` of OfflineGaussianDiffusion

Plain torch / numpy host code: file formats are not on the device path."""
from __future__ import annotations

import glob
import os
import re
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def newest_ckpt(base_dir: str) -> Optional[str]:
    paths = sorted(glob.glob(f'{base_dir}/model_ckpt_steps_*.ckpt'),
                   key=lambda x: int(re.findall(rf'{re.escape(base_dir)}/model_ckpt_steps_(\d+).ckpt', x)[0]))
    return paths[-1] if paths else None


def _torch_load(path, trusted: bool = True):
    """A reference trainer checkpoint carries non-tensor objects (numpy scalars in `checkpoint_callback_best`, optimizer / scheduler
    dicts): torch >= 2.6 refuses to unpickle those under its default `weights_only=True`.  The reference's own `torch.load(path,
    map_location='cpu')` (utils/__init__.py:191) predates that switch and always unpickled everything - `trusted=True` (default) keeps
    that behaviour; pass `trusted=False` for files of unknown origin: tensors-only loading, with a clear error if the file needs more."""
    if trusted:
        return torch.load(path, map_location='cpu', weights_only=False)
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except Exception as e:
        raise RuntimeError(f'{path}: not loadable with weights_only=True ({type(e).__name__}: {str(e)[:200]}); a reference trainer checkpoint '
                           f'holds pickled non-tensor objects - pass trusted=True only if you trust the file') from e


def load_ckpt(cur_model, ckpt_base_dir, prefix_in_ckpt='model', force=True, strict=True, trusted=True):
    """Same contract as the reference's utils.load_ckpt (including its messages and the assert when nothing is found and force)."""
    if os.path.isfile(ckpt_base_dir):
        base_dir, checkpoint_path = os.path.dirname(ckpt_base_dir), ckpt_base_dir
    else:
        base_dir, checkpoint_path = ckpt_base_dir, newest_ckpt(ckpt_base_dir)
    if checkpoint_path is None:
        e_msg = f"| ckpt not found in {base_dir}."
        if force:
            assert False, e_msg
        print(e_msg)
        return None
    state_dict = _torch_load(checkpoint_path, trusted)['state_dict']
    state_dict = {k[len(prefix_in_ckpt) + 1:]: v for k, v in state_dict.items() if k.startswith(f'{prefix_in_ckpt}.')}
    if not strict:
        cur = cur_model.state_dict()
        for key in [k for k, v in state_dict.items() if k in cur and cur[k].shape != v.shape]:
            print('| Unmatched keys: ', key, cur[key].shape, state_dict[key].shape)
            del state_dict[key]
    cur_model.load_state_dict(state_dict, strict=strict)
    print(f"| load '{prefix_in_ckpt}' from '{checkpoint_path}'.")
    return checkpoint_path


def save_ckpt(work_dir: str, model, global_step: int, *, epoch: int = 0, prefix_in_ckpt: str = 'model', optimizer_states: Sequence[dict] = (),
              lr_schedulers: Sequence[dict] = (), best=None, extra: Optional[dict] = None) -> str:
    """Write `work_dir/model_ckpt_steps_<global_step>.ckpt` in the reference's layout, atomically (.part + replace)."""
    os.makedirs(work_dir, exist_ok=True)
    ckpt = {'epoch': epoch, 'global_step': global_step, 'checkpoint_callback_best': best, 'optimizer_states': list(optimizer_states),
            'lr_schedulers': list(lr_schedulers),
            'state_dict': {f'{prefix_in_ckpt}.{k}': v.detach().cpu() for k, v in model.state_dict().items()}}
    if extra:
        ckpt.update(extra)
    path = os.path.join(work_dir, f'model_ckpt_steps_{global_step}.ckpt')
    torch.save(ckpt, path + '.part')
    os.replace(path + '.part', path)
    return path


# ---- optimiser state: ShardedAdamW <-> torch.optim.AdamW ---------------------------------------------------------------------
def _gather_flat(shard: torch.Tensor, opt, dst: int = 0) -> Optional[torch.Tensor]:
    if opt.world == 1:
        return shard.detach().cpu()
    parts = [torch.empty_like(shard) for _ in range(opt.world)] if opt.rank == dst else None
    dist.gather(shard.contiguous(), gather_list=parts, dst=dst, group=opt.group)
    return torch.cat(parts).cpu() if opt.rank == dst else None


def adamw_state_from_sharded(opt, dst: int = 0) -> Optional[dict]:
    """A torch.optim.AdamW state_dict (one param group, parameters in the optimiser's order) from the sharded moments; on `dst` only
    (None elsewhere).  Collective when world > 1."""
    m, v = _gather_flat(opt.exp_avg, opt, dst), _gather_flat(opt.exp_avg_sq, opt, dst)
    if m is None:
        return None
    state, off = {}, 0
    for i, p in enumerate(opt.params):
        n = p.numel()
        state[i] = {'step': torch.tensor(float(opt.step_count)), 'exp_avg': m[off:off + n].view(p.shape).clone(),
                    'exp_avg_sq': v[off:off + n].view(p.shape).clone()}
        off += n
    group = {'lr': opt.lr, 'betas': tuple(opt.betas), 'eps': opt.eps, 'weight_decay': opt.weight_decay, 'amsgrad': False, 'maximize': False,
             'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(opt.params)))}
    return {'state': state, 'param_groups': [group]}


def adamw_state_to_sharded(opt, sd: dict):
    """Load a torch.optim.AdamW state_dict (every rank passes the same dict, e.g. read from the checkpoint file) into the rank's shard."""
    flat_m = torch.zeros(opt.padded)
    flat_v = torch.zeros(opt.padded)
    off, steps = 0, set()
    for i, p in enumerate(opt.params):
        n = p.numel()
        st = sd['state'].get(i)
        if st is not None:
            flat_m[off:off + n] = st['exp_avg'].reshape(-1).float()
            flat_v[off:off + n] = st['exp_avg_sq'].reshape(-1).float()
            steps.add(int(float(st['step'])))
        off += n
    if len(steps) > 1:
        raise ValueError(f'parameters with different step counts {sorted(steps)}: not a state ShardedAdamW can continue')
    lo = opt.rank * opt.shard
    opt.exp_avg.copy_(flat_m[lo:lo + opt.shard])
    opt.exp_avg_sq.copy_(flat_v[lo:lo + opt.shard])
    opt.step_count = steps.pop() if steps else 0
    g = sd['param_groups'][0]
    opt.lr, opt.betas, opt.eps, opt.weight_decay = g['lr'], tuple(g['betas']), g['eps'], g['weight_decay']
    # torch's lr schedulers store the un-decayed rate as `initial_lr` in the group; `lr` is already decayed.  A StepLR built on this
    # optimiser afterwards must start from the base rate (else it decays twice): diffsinger_amd.train_dist.StepLR reads `base_lr`.
    opt.base_lr = g.get('initial_lr', g['lr'])


# ---- offline aux-decoder mels -----------------------------------------------------------------------------------------------
def save_offline_mel(work_dir: str, item_name: str, mel: np.ndarray, kind: str = 'P') -> str:
    """tasks/tts/fs2.py:447-450 save_result: `np.save(f'{work_dir}/{kind}_mels_npy/{item_name}', mel)` with mel [T, 80]."""
    d = os.path.join(work_dir, f'{kind}_mels_npy')
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, item_name + '.npy')
    np.save(path, np.asarray(mel, dtype=np.float32))
    return path


def load_offline_mels(fs2_ckpt: str, item_names: List[str], pad_value: float = 0.0) -> torch.Tensor:
    """usr/diffsinger_task.py:107-117: the aux mels of a batch from `dirname(fs2_ckpt)/P_mels_npy/<item>.npy`, right-padded to the
    longest (utils.collate_2d) -> [B, T_max, 80] - the `fs2_mels` half of OfflineGaussianDiffusion's `ref_mels`."""
    base = os.path.dirname(fs2_ckpt)
    mels = [torch.Tensor(np.load(f'{base}/P_mels_npy/{n}.npy')) for n in item_names]
    T = max(m.shape[0] for m in mels)
    out = mels[0].new_full((len(mels), T, mels[0].shape[1]), pad_value)
    for i, m in enumerate(mels):
        out[i, :m.shape[0]] = m
    return out
