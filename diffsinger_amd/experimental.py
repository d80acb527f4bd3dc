"""EXPERIMENTS - nothing here is on a product path ("beyond the fp32-MFMA ceiling").

split-precision convolution: the denoiser's dilated Conv1d(256 -> 512, 3) as an fp32-accurate GEMM on the bf16 matrix pipe
(`dsf_split_conv1d_probe`, csrc/dsd_split.hpp k_split_conv): host-side weight packing and the call wrapper."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .fs2 import padded_frames


def split3(t: torch.Tensor):
    """fp32 -> three bf16 planes with a + b + c == t exactly (8 + 8 + 8 mantissa bits)."""
    a = t.bfloat16()
    r = t - a.float()
    b = r.bfloat16()
    return a, b, (r - b.float()).bfloat16()


def pack_split_weight(w: torch.Tensor) -> torch.Tensor:
    """Conv1d weight [512][256][3] fp32 -> int16 tensor [wave 4][chunk 48][row block 4][plane 3][lane 64][8]: lane (i = lane & 31,
    h = lane >> 5), element e of chunk (g, tap) holds plane(W[128 wave + 32 mb + i][16 g + 8 h + e][tap])."""
    assert tuple(w.shape) == (512, 256, 3), w.shape
    planes = torch.stack([p.view(torch.int16) for p in split3(w.detach().float().cpu())])       # [3][512][256][3]
    x = planes.reshape(3, 4, 4, 32, 16, 2, 8, 3)                                              # [pl][wave][mb][i][g][h][e][tap]
    x = x.permute(1, 4, 7, 2, 0, 5, 3, 6)                                                     # [wave][g][tap][mb][pl][h][i][e]
    return x.reshape(4, 48, 4, 3, 64, 8).contiguous()


def split_conv1d(x_cm: torch.Tensor, wplanes: torch.Tensor, T: int, dil: int, iters: int = 1, timed: bool = False, variant: int = 0):
    """x_cm [B][256][TS] fp32 on the device, wplanes from pack_split_weight (on the device) -> out [B][512][TS] (and the average launch
    time in ms when timed).  variant 0: compiler-scheduled two-stage pipeline; 1 / 2: hand-pinned operand pipeline with three / six weight stages."""
    if x_cm.device.type != 'cuda':
        raise RuntimeError('split_conv1d: no CPU path')
    lib = _lib.load()
    B, Cc, TS = x_cm.shape
    assert Cc == 256 and TS == padded_frames(T) and x_cm.is_contiguous() and wplanes.device == x_cm.device
    out = torch.empty(B, 512, TS, device=x_cm.device, dtype=torch.float32)
    ms = C.c_float(0.0)
    with torch.cuda.device(x_cm.device):
        _lib.check(lib.dsf_split_conv1d_probe(x_cm.data_ptr(), wplanes.data_ptr(), out.data_ptr(), B, T, dil, variant, iters, C.byref(ms) if timed else None,
                                              torch.cuda.current_stream(x_cm.device).cuda_stream), 'dsf_split_conv1d_probe')
    return (out, float(ms.value)) if timed else out
