"""Data-parallel training step of the denoiser: gradient exchange + optimiser, one process per GPU.

What the reference does (paths relative to the reference root): wraps the task in torch DDP (tasks/base_task.py:261-275,
utils/pl_utils.py:180-312) - a bucketed RING all-reduce of all 15 M gradients (60 MB) - then every rank runs the same
clip_grad_norm_ (utils/pl_utils.py:1165-1168, clip_grad_norm 1) and the same torch.optim.AdamW step
(usr/diffspeech_task.py:40-46) on all parameters, followed by StepLR (usr/task.py:75-84).

What this does instead, MI355X-first.  xGMI is point-to-point (7 links per GPU, every peer one hop away), so the exchange is
the two DIRECT collectives an all-reduce consists of, with the optimiser in between on 1/W of the data:

    flat_g (all gradients, one contiguous fp32 buffer the .grad tensors are views of)
      -- reduce_scatter (SUM) -->  this rank's 1/W shard of the summed gradient
      -- clip coefficient: sum of squares of the shard, all_reduce of ONE float, computed on the device (no host sync)
      -- fused AdamW on the shard (dsf_adamw_step: one pass over p, g, m, v; moments exist for the shard only: 2 x 60 MB / W)
      -- all_gather -->  flat_p (all parameters, the .data tensors are views of it)

Same bytes on the wire as the all-reduce, but the optimiser arithmetic and its state shrink by W and nothing is computed
redundantly.  The parameters stay ordinary nn.Parameters (views into the flat buffer), so the model, its state_dict and
the training operators of diffsinger_amd/train.py are unchanged; `param_generation()` tells the packed-weight caches that the
raw-pointer update happened (an external kernel does not bump torch's version counters).

Numerics: identical to clip_grad_norm_ + AdamW on the AVERAGED gradient (what DDP + the reference's optimiser compute) up to
fp32 summation order; the world-2 gloo test checks that against torch.optim.AdamW on the full batch.  No CPU path for the
update itself: on the device it is the HIP kernel, and tests inject a torch restatement through `_update`."""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch
import torch.distributed as dist

from . import _lib

_GENERATION = 0


def param_generation() -> int:
    """Bumped by every optimiser step that writes parameters through raw pointers; part of the packed-weight cache tags."""
    return _GENERATION


def _hip_adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, gscale):
    if p.device.type != 'cuda':
        raise RuntimeError('ShardedAdamW: the fused optimiser step has no CPU path - parameters must live on the MI355X')
    lib = _lib.load()
    with torch.cuda.device(p.device):
        _lib.check(lib.dsf_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(beta1), float(beta2),
                                      float(eps), float(weight_decay), int(step), gscale.data_ptr() if gscale is not None else None,
                                      torch.cuda.current_stream(p.device).cuda_stream), 'dsf_adamw_step')


class ShardedAdamW:
    """AdamW over the flattened parameters with the gradient reduce-scatter / parameter all-gather built in (see the module
    docstring).  Single process (no process group): the same flat, fused step without communication.

        opt = ShardedAdamW(model.parameters(), lr=hparams['lr'], betas=(b1, b2), weight_decay=wd, clip_grad_norm=1.0)
        loss.backward(); opt.step(); opt.zero_grad()

    Do NOT wrap the model in DDP as well: the exchange happens here, after backward."""

    ALIGN = 64          # floats: shard boundaries stay 256-byte aligned
    # Known difference from torch.optim.AdamW: every parameter's .grad is a permanent view of the (zeroed) flat gradient, so a parameter
    # that receives NO gradient in a step (an unused branch; fs2.decoder / mel_out under skip_decoder=True) is updated with g = 0 -
    # its moments decay and, with weight_decay > 0, so does the weight - whereas torch skips grad=None parameters entirely (no decay,
    # no state entry).  Hand this optimiser only the parameters the step really trains (the denoiser's), as train.py / the tests do.

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                 clip_grad_norm: Optional[float] = None, group=None, _update: Optional[Callable] = None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('ShardedAdamW: no trainable parameters')
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError('ShardedAdamW: all parameters must be fp32 on one device')
        self.lr, self.betas, self.eps, self.weight_decay, self.clip = lr, tuple(betas), eps, weight_decay, clip_grad_norm
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._update = _update or _hip_adamw
        total = sum(p.numel() for p in self.params)
        unit = self.ALIGN * self.world
        self.total = total
        self.padded = (total + unit - 1) // unit * unit
        self.shard = self.padded // self.world
        self.flat_p = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)       # the parameter now lives in the flat buffer
                p.grad = self.flat_g[off:off + n].view(p.shape)       # autograd accumulates in place into the flat gradient
                off += n
        self.exp_avg = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.step_count = 0
        self.last_grad_norm = None                                    # device scalar: norm of the averaged gradient before clipping

    def zero_grad(self):
        self.flat_g.zero_()
        off = 0
        for p in self.params:                                         # re-attach views a set_to_none=True elsewhere may have dropped
            if p.grad is None:
                p.grad = self.flat_g[off:off + p.numel()].view(p.shape)
            off += p.numel()

    @torch.no_grad()
    def step(self, lr: Optional[float] = None):
        global _GENERATION
        lr = self.lr if lr is None else lr
        lo = self.rank * self.shard
        p_shard = self.flat_p[lo:lo + self.shard]
        if self.world > 1:
            g_shard = torch.empty(self.shard, device=self.flat_g.device, dtype=torch.float32)
            dist.reduce_scatter_tensor(g_shard, self.flat_g, op=dist.ReduceOp.SUM, group=self.group)
        else:
            g_shard = self.flat_g[lo:lo + self.shard]
        gscale = torch.full((1,), 1.0 / self.world, device=g_shard.device, dtype=torch.float32)
        if self.clip is not None and self.clip > 0:                   # the reference clips only when clip_grad_norm > 0 (utils/pl_utils.py:1165-1168;
            sq = (g_shard * g_shard).sum().reshape(1)                 # configs/config_base.yaml ships clip_grad_norm: 0 = no clipping)
            if self.world > 1:
                dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group)
            total_norm = sq.sqrt() / self.world                       # norm of the AVERAGED gradient (what DDP hands clip_grad_norm_)
            self.last_grad_norm = total_norm
            gscale = gscale * torch.clamp(self.clip / (total_norm + 1e-6), max=1.0)
        self.step_count += 1
        self._update(p_shard, g_shard, self.exp_avg, self.exp_avg_sq, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                     self.step_count, gscale)
        if self.world > 1:
            dist.all_gather_into_tensor(self.flat_p, p_shard, group=self.group)       # in place: every rank contributes its own slice
        _GENERATION += 1

    # ---- checkpointing: the state is 1/W of torch.optim.AdamW's; gather it to write a reference-compatible optimizer state ------
    def state_dict(self):
        return {'step': self.step_count, 'exp_avg_shard': self.exp_avg.clone(), 'exp_avg_sq_shard': self.exp_avg_sq.clone(), 'rank': self.rank,
                'world': self.world, 'lr': self.lr}

    def load_state_dict(self, sd):
        if sd['world'] != self.world or sd['rank'] != self.rank:
            raise ValueError('ShardedAdamW: the optimiser shard belongs to a different rank / world size')
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg_shard'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq_shard'])
        if 'lr' in sd:
            self.lr = float(sd['lr'])


class StepLR:
    """torch.optim.lr_scheduler.StepLR(optimizer, decay_steps, gamma=0.5) as usr/task.py:75-84 drives it
    (`scheduler.step(global_step // accumulate_grad_batches)`): lr = base * gamma ** (step // step_size)."""

    def __init__(self, optimizer: ShardedAdamW, step_size: int, gamma: float = 0.5):
        # base rate: the un-decayed `initial_lr` of a resumed torch state (ckpt.adamw_state_to_sharded) if there is one, else the current rate
        self.opt, self.base, self.step_size, self.gamma = optimizer, getattr(optimizer, 'base_lr', optimizer.lr), step_size, gamma

    def step(self, global_step: int):
        self.opt.lr = self.base * self.gamma ** (global_step // self.step_size)

    def get_lr(self):
        return [self.opt.lr]
