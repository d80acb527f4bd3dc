"""Replay a launch-bound forward as ONE hipGraph (`torch.cuda.CUDAGraph`; on ROCm that is hipGraph).

The rows around the hot path are chains of small kernels issued from Python through ctypes - FastSpeech2: ~150 launches per forward,
HiFi-GAN: ~80 (DESIGN.md sections 9, 9b) - whose issue cost (Python + ctypes + `torch.empty`, 10-20 us per operator) is of the order of
the kernels themselves.  The library's operators only ENQUEUE on the stream torch calls current, never allocate or synchronise, so a
forward with fixed shapes is capturable as it is: `GraphedForward(fn)` captures `fn` once per input signature (shapes, dtypes) into a
private memory pool and afterwards copies the inputs into the graph's static buffers and replays it.  torch is plumbing here (stream,
memory pool, capture); every captured node is one of this package's HIP kernels or a torch copy.

Limits (checked, loud): tensor arguments only (plus None / python scalars, which become part of the signature); no data-dependent
shapes inside `fn` (FastSpeech2 in free-running mode sizes the mel axis from predicted durations -> host sync -> not capturable: pass
`mel2ph`); results are STATIC tensors, valid until the next call with the same signature (clone what must survive)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Callable

import torch


def _sig(a):
    if isinstance(a, torch.Tensor):
        return ('T', tuple(a.shape), a.dtype, a.device.index)
    if a is None or isinstance(a, (bool, int, float, str)):
        return ('V', a)
    raise TypeError(f'GraphedForward: unsupported argument type {type(a).__name__} (tensors, None and python scalars only)')


class GraphedForward:
    def __init__(self, fn: Callable[..., Any], max_graphs: int = 8, warmup: int = 2):
        self.fn, self.max_graphs, self.warmup = fn, max_graphs, warmup
        self._cache: 'OrderedDict[tuple, tuple]' = OrderedDict()
        self.captures = 0

    def _capture(self, args, kwargs):
        dev = next((a.device for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor)), None)
        if dev is None or dev.type != 'cuda':
            raise RuntimeError('GraphedForward needs device tensors (there is no CPU path)')
        s_args = [a.clone() if isinstance(a, torch.Tensor) else a for a in args]
        s_kw = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kwargs.items()}
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():               # lazy one-time work (weight packing, workspace growth) happens here
            for _ in range(self.warmup):
                self.fn(*s_args, **s_kw)
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g, capture_error_mode='thread_local'):
            out = self.fn(*s_args, **s_kw)
        self.captures += 1
        return g, s_args, s_kw, out

    @torch.no_grad()
    def __call__(self, *args, **kwargs):
        key = (tuple(_sig(a) for a in args), tuple((k, _sig(v)) for k, v in sorted(kwargs.items())))
        ent = self._cache.get(key)
        if ent is None:
            ent = self._capture(args, kwargs)
            self._cache[key] = ent
            if len(self._cache) > self.max_graphs:
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(key)
        g, s_args, s_kw, out = ent
        for s, a in zip(s_args, args):
            if isinstance(s, torch.Tensor):
                s.copy_(a)
        for k, s in s_kw.items():
            if isinstance(s, torch.Tensor):
                s.copy_(kwargs[k])
        g.replay()
        return out
