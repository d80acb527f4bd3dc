"""DenoiserEngine: thin owner of one `dsd_handle` (include/dsd.h).  torch is plumbing only: it provides the
device tensors whose raw pointers, sizes and strides cross the C ABI, and the current HIP stream."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib

STATE_KEYS_GLOBAL = ['input_projection', 'mlp.0', 'mlp.2', 'skip_projection', 'output_projection']
STATE_KEYS_LAYER = ['dilated_conv', 'diffusion_projection', 'conditioner_projection', 'output_projection']


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class DenoiserEngine:
    """Owns the packed weights, tables, workspace and cached hipGraphs for ONE device.

    Mirrors the reference objects it stands behind: `DiffNet` parameters (usr/diff/net.py:91-105), the
    `GaussianDiffusion` schedule buffers (usr/diff/shallow_diffusion_tts.py:103-126) and the inference loop
    (:248-276)."""

    def __init__(self, mel_bins: int, residual_channels: int, encoder_hidden: int, residual_layers: int,
                 dilation_cycle_length: int, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('DenoiserEngine needs a HIP device (torch device type "cuda"); there is no CPU path')
        self.M, self.L = mel_bins, residual_layers
        cfg = _lib.DsdConfig(mel_bins, residual_channels, encoder_hidden, residual_layers, dilation_cycle_length)
        h = C.c_void_p()
        _lib.check(self.lib.dsd_create(C.byref(cfg), self.device.index or 0, C.byref(h)), 'dsd_create')
        self._h = h
        self.prepared_shape = None
        self.prepare_serial = 0           # bumped by every prepare(): lets a cache of 'what is prepared' (DiffNet.bind_cond) notice a direct call
        self.n_sched = 0

    def close(self):
        if getattr(self, '_h', None):
            self.lib.dsd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights / tables ---------------------------------------------------------------------------------
    def load_weights(self, state: Dict[str, torch.Tensor]):
        """state: DiffNet state_dict (keys as in the reference, optionally already on the device)."""
        L = self.L
        keep = []

        def dev(name):
            t = state[name].detach().to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        w = _lib.DsdWeights()
        w.input_projection_w, w.input_projection_b = dev('input_projection.weight'), dev('input_projection.bias')
        w.mlp0_w, w.mlp0_b = dev('mlp.0.weight'), dev('mlp.0.bias')
        w.mlp2_w, w.mlp2_b = dev('mlp.2.weight'), dev('mlp.2.bias')
        arrays = []
        for field, key in [('dilated_conv', 'dilated_conv'), ('diffusion_projection', 'diffusion_projection'),
                           ('conditioner_projection', 'conditioner_projection'), ('output_projection', 'output_projection')]:
            for suffix, part in (('_w', 'weight'), ('_b', 'bias')):
                arr = (C.c_void_p * L)(*[dev(f'residual_layers.{l}.{key}.{part}') for l in range(L)])
                arrays.append(arr)
                setattr(w, field + suffix, C.cast(arr, C.POINTER(C.c_void_p)))
        w.skip_projection_w, w.skip_projection_b = dev('skip_projection.weight'), dev('skip_projection.bias')
        w.final_projection_w, w.final_projection_b = dev('output_projection.weight'), dev('output_projection.bias')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_load_weights(self._h, C.byref(w), _stream_ptr(self.device)), 'dsd_load_weights')
            torch.cuda.current_stream(self.device).synchronize()    # sources may be freed after this
        self.prepared_shape = None

    def set_schedule(self, betas: np.ndarray):
        b = np.ascontiguousarray(np.asarray(betas, dtype=np.float64))
        _lib.check(self.lib.dsd_set_schedule(self._h, b.ctypes.data_as(C.POINTER(C.c_double)), len(b)), 'dsd_set_schedule')
        self.n_sched = len(b)

    def schedule_table(self, which: int) -> np.ndarray:
        out = np.empty(self.n_sched, dtype=np.float32)
        _lib.check(self.lib.dsd_get_schedule_table(self._h, which, out.ctypes.data_as(C.POINTER(C.c_float)), self.n_sched))
        return out

    def set_spec_range(self, spec_min: Sequence[float], spec_max: Sequence[float]):
        lo = np.ascontiguousarray(np.asarray(spec_min, dtype=np.float32).reshape(-1))
        hi = np.ascontiguousarray(np.asarray(spec_max, dtype=np.float32).reshape(-1))
        if len(lo) != self.M or len(hi) != self.M:
            raise ValueError(f'spec_min/spec_max must have {self.M} entries')
        fp = C.POINTER(C.c_float)
        _lib.check(self.lib.dsd_set_spec_range(self._h, lo.ctypes.data_as(fp), hi.ctypes.data_as(fp)), 'dsd_set_spec_range')

    def set_use_graph(self, enable: bool):
        _lib.check(self.lib.dsd_set_use_graph(self._h, int(bool(enable))))

    def set_lat_split(self, g: int):
        """Row split of the latency kernels (csrc/dsd_lat.hpp): -1 by batch size (default), 0 never, 2 / 4 / 8 / 16 forced."""
        _lib.check(self.lib.dsd_set_lat_split(self._h, int(g)), 'dsd_set_lat_split')

    def lat_split(self) -> int:
        """G of the latency kernels the prepared batch runs with, 0 = not on that path."""
        return self.lib.dsd_get_lat_split(self._h)

    def set_conv_mode(self, mode, touch_ahead: int = -1):
        """How the PERSISTENT loop evaluates the dilated convolution (csrc/dsd_loop_wino.hpp): 'winograd' / 1 (default) = Winograd F(2,3)
        along the frame axis, 2/3 of the fp32 multiplications; 'direct' / 0 = the K = 768 contraction, bit-identical to the per-layer
        kernels.  touch_ahead: steps the L2 touch of the transformed-weight stream runs in front (0 = off, -1 = keep)."""
        m = {'direct': 0, 'winograd': 1}.get(mode, mode)
        _lib.check(self.lib.dsd_set_conv_mode(self._h, int(m), int(touch_ahead)), 'dsd_set_conv_mode')

    def conv_mode(self) -> int:
        """1 if the dilated convolution of the prepared batch runs as Winograd F(2,3) - on the persistent loop (loop_mode() == 1) or on the
        latency kernels at G = 2 / 4 / 8 (lat_split()) - else 0 (include/dsd.h dsd_get_conv_mode)."""
        return self.lib.dsd_get_conv_mode(self._h)

    def set_loop_mode(self, mode: int):
        """2 (default): automatic - latency kernels for small batches, else the persistent loop / per-layer kernels by chip occupancy;
        1: the whole K-step loop as one persistent kernel when the batch allows it; 0: per-layer kernels; 3: latency kernels."""
        _lib.check(self.lib.dsd_set_loop_mode(self._h, int(mode)), 'dsd_set_loop_mode')
        self._loop_req = int(mode)

    def requested_loop_mode(self) -> int:
        """The mode last asked for with set_loop_mode (or DSD_LOOP at creation; default 2) - not the path the prepared batch takes (loop_mode())."""
        return getattr(self, '_loop_req', int(os.environ.get('DSD_LOOP', '2') or 2))

    def loop_mode(self) -> int:
        return self.lib.dsd_get_loop_mode(self._h)

    def loop_launches(self) -> int:
        """k_loop launches per sampling call for the prepared batch (chunks of whole utterances); 0 = not on the persistent path."""
        return self.lib.dsd_loop_launches(self._h)

    def set_split_mode(self, on: bool):
        """EXPERIMENT (csrc/dsd_split.hpp; default off): the residual layers as six bf16 plane products per fp32
        product on the bf16 matrix pipe (fp32-class accuracy).  Uses the per-layer kernel path while it is on."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_set_split_mode(self._h, int(bool(on)), _stream_ptr(self.device)), 'dsd_set_split_mode')

    def split_mode(self) -> int:
        return self.lib.dsd_get_split_mode(self._h)

    def debug_layer(self, layer: int, t: int, x_in: torch.Tensor):
        """Debug: one residual layer on x_in [B,C,T] with the prepared batch's conditioner projection -> (x_out [B,C,T] or None for the
        last layer, skip [B,C,T]).  The fp32 kernel, or the split-precision one while set_split_mode(True)."""
        B, Cc, T = x_in.shape
        TS = (T + 31) // 32 * 32
        xi = torch.zeros(B, Cc, TS, device=self.device, dtype=torch.float32)
        xi[:, :, :T] = x_in
        xo = torch.zeros_like(xi)
        sk = torch.zeros_like(xi)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_debug_layer(self._h, int(layer), int(t), xi.data_ptr(), xo.data_ptr(), sk.data_ptr(), _stream_ptr(self.device)),
                       'dsd_debug_layer')
        last = layer == self.L - 1
        return (None if last else xo[:, :, :T]), sk[:, :, :T]

    def loop_timeouts(self) -> int:
        """Synchronises; nonzero = an inter-workgroup wait of the persistent loop hit its spin bound (results invalid)."""
        with torch.cuda.device(self.device):
            rc = self.lib.dsd_loop_timeouts(self._h, _stream_ptr(self.device))
        if rc < 0:
            _lib.check(rc, 'dsd_loop_timeouts')
        return rc

    def check(self, synchronize: bool = False):
        """Raise RuntimeError if a persistent K-step loop that has FINISHED since the last report hit its spin bound (its mel / x tiles are
        NaN).  Reads a pinned host word: no synchronisation unless asked for - then the caller's current stream is drained first, so the
        verdict covers everything enqueued so far.  After the report the handle runs the hipGraph path (repeat the call; set_loop_mode
        re-arms the persistent loop).  Every other engine call makes the same check on entry."""
        if synchronize:
            torch.cuda.current_stream(self.device).synchronize()
        _lib.check(self.lib.dsd_check(self._h), 'persistent loop')

    def parked(self) -> int:
        """Sampling calls this engine still runs on the hipGraph path after a reported persistent-loop timeout (0 = not parked); it
        returns to the persistent loop by itself after that many calls, or at once with set_loop_mode()."""
        return self.lib.dsd_loop_parked(self._h)

    def hold_cus(self, n_workgroups: int, milliseconds: int, stream: Optional[torch.cuda.Stream] = None):
        """Test hook: a foreign kernel that occupies `n_workgroups` compute units for `milliseconds` or until release_cus(), whichever comes
        first, on a side stream that really runs BESIDE the current stream; returns that stream once every holder is resident.
        HIP multiplexes streams onto a few hardware queues: a side stream that shares its queue with the current stream would serialise the
        caller's next kernel BEHIND the holders instead of starving it (seen after many streams had been created in the process).  So
        each candidate stream is probed - an event recorded on the current stream must complete while the holders are resident - and given
        up for the next one if it does not.  The control words live in pinned host memory: counting the resident holders and releasing
        them never goes through a stream."""
        import time
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        cur.synchronize()
        for attempt in range(8):
            side = stream if (stream is not None and attempt == 0) else torch.cuda.Stream(dev)
            ctl = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._hold_ctl = ctl
            with torch.cuda.device(dev):
                _lib.check(self.lib.dsd_debug_hold_cus(dev.index or 0, int(n_workgroups), int(milliseconds), ctl.data_ptr(), side.cuda_stream),
                           'dsd_debug_hold_cus')
            t0 = time.time()
            while int(ctl[0]) < n_workgroups:
                if time.time() - t0 > 5.0:
                    self.release_cus()
                    raise RuntimeError(f'hold_cus: only {int(ctl[0])} of {n_workgroups} holders became resident within 5 s')
                time.sleep(0.002)
            probe = torch.cuda.Event()
            probe.record(cur)
            t0 = time.time()
            while not probe.query() and time.time() - t0 < 0.5:
                time.sleep(0.002)
            if probe.query():
                self._hold_attempts = attempt + 1
                return side
            self.release_cus()                                   # the current stream sits behind the holders: try another stream
            side.synchronize()
        raise RuntimeError('hold_cus: no side stream runs concurrently with the current stream')

    def release_cus(self):
        """Ends the holders of the last hold_cus() (a store into pinned host memory the holders poll)."""
        ctl = getattr(self, '_hold_ctl', None)
        if ctl is not None:
            ctl[1] = 1

    def set_layer_tile(self, frames: int):
        _lib.check(self.lib.dsd_set_layer_tile(self._h, int(frames)))

    def layer_tile(self) -> int:
        return self.lib.dsd_get_layer_tile(self._h)

    def device_bytes(self) -> int:
        return self.lib.dsd_device_bytes(self._h)

    # -- batch ----------------------------------------------------------------------------------------------
    def prepare(self, cond: torch.Tensor):
        """cond [B,H,T] fp32 on the device, any strides (the reference passes a transposed view)."""
        if cond.dim() != 3 or cond.dtype != torch.float32 or cond.device != self.device:
            raise ValueError('cond must be a [B,H,T] fp32 tensor on the engine device')
        B, H, T = cond.shape
        sb, sh, st = cond.stride()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_prepare(self._h, B, T, cond.data_ptr(), sb, sh, st, _stream_ptr(self.device)), 'dsd_prepare')
        self.prepared_shape = (B, T)
        self.prepare_serial += 1

    def _spec(self, x: torch.Tensor, name='x') -> torch.Tensor:
        if self.prepared_shape is None:
            raise RuntimeError('no batch prepared: call prepare(cond) first')
        B, T = self.prepared_shape
        if x.dim() == 4:
            if x.shape[1] != 1:
                raise ValueError(f'{name}: expected [B,1,M,T]')
            x = x[:, 0]
        if tuple(x.shape) != (B, self.M, T) or x.dtype != torch.float32 or x.device != self.device:
            raise ValueError(f'{name}: expected fp32 [{B},{self.M},{T}] on {self.device}, got {tuple(x.shape)} {x.dtype} {x.device}')
        return x

    def denoise(self, x: torch.Tensor, t, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """eps_hat = DiffNet(x, t, cond) for the prepared cond.  x [B,M,T] or [B,1,M,T]; t: int or [B] ints."""
        xs = self._spec(x).contiguous()
        B = xs.shape[0]
        if isinstance(t, torch.Tensor):
            t = t.detach().cpu().reshape(-1).tolist()
        if isinstance(t, int):
            t = [t] * B
        if len(t) != B:
            raise ValueError('t must have one entry per utterance')
        tarr = (C.c_int32 * B)(*[int(v) for v in t])
        eps = out if out is not None else torch.empty_like(xs)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_denoise(self._h, xs.data_ptr(), tarr, eps.data_ptr(), _stream_ptr(self.device)), 'dsd_denoise')
        return eps

    def q_sample(self, x_start: torch.Tensor, noise: torch.Tensor, t: int) -> torch.Tensor:
        xs, z = self._spec(x_start, 'x_start').contiguous(), self._spec(noise, 'noise').contiguous()
        out = torch.empty_like(xs)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_q_sample(self._h, xs.data_ptr(), z.data_ptr(), int(t), out.data_ptr(), _stream_ptr(self.device)), 'dsd_q_sample')
        return out

    def sample_ddpm(self, x: torch.Tensor, noise: Optional[torch.Tensor], k_step: int, seed: Optional[int] = None) -> torch.Tensor:
        """In place: x [B,M,T] contiguous goes from x_{k_step} to x_0.  noise [k_step,B,M,T] (or [k,B,1,M,T]) = the explicit
        N(0,1) draws, or None: drawn inside the kernel (Philox4x32-10 keyed by `seed`, see include/dsd.h)."""
        xs = self._spec(x)
        if not xs.is_contiguous():
            raise ValueError('x must be contiguous (it is updated in place)')
        B, T = self.prepared_shape
        nptr = None
        if noise is not None:
            if noise.dim() == 5:
                noise = noise[:, :, 0]
            if tuple(noise.shape) != (k_step, B, self.M, T) or not noise.is_contiguous() or noise.dtype != torch.float32 \
                    or noise.device != self.device:
                raise ValueError(f'noise must be contiguous fp32 [{k_step},{B},{self.M},{T}] on {self.device}')
            nptr = noise.data_ptr()
        else:
            _lib.check(self.lib.dsd_set_noise_seed(self._h, int(seed or 0) & 0xFFFFFFFFFFFFFFFF), 'dsd_set_noise_seed')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_sample_ddpm(self._h, xs.data_ptr(), nptr, int(k_step), _stream_ptr(self.device)), 'dsd_sample_ddpm')
        return x

    def philox_normal(self, seed: int, step: int, n: int) -> torch.Tensor:
        """The in-kernel N(0,1) draws of p_sample call `step` for elements 0..n-1 (element = flat index into [B][M][T])."""
        out = torch.empty(n, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_philox_normal(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, int(step), out.data_ptr(), n, _stream_ptr(self.device)),
                       'dsd_philox_normal')
        return out

    def p_sample(self, x: torch.Tensor, noise: Optional[torch.Tensor], t: int, seed: Optional[int] = None) -> torch.Tensor:
        xs = self._spec(x)
        z = self._spec(noise, 'noise').contiguous() if noise is not None else None
        if not xs.is_contiguous():
            raise ValueError('x must be contiguous (it is updated in place)')
        if z is None:
            _lib.check(self.lib.dsd_set_noise_seed(self._h, int(seed or 0) & 0xFFFFFFFFFFFFFFFF), 'dsd_set_noise_seed')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_p_sample(self._h, xs.data_ptr(), z.data_ptr() if z is not None else None, int(t), _stream_ptr(self.device)),
                       'dsd_p_sample')
        return x

    def p_sample_ex(self, x: torch.Tensor, noise: torch.Tensor, t, clip_denoised: bool = True, repeat_noise: bool = False) -> torch.Tensor:
        """In place: one p_sample call with a step index PER UTTERANCE, optional clamp, and noise [B,M,T] or - repeat_noise - one
        [M,T] draw shared by the batch (usr/diff/shallow_diffusion_tts.py:159-166, noise_like :38-41)."""
        xs = self._spec(x)
        if not xs.is_contiguous():
            raise ValueError('x must be contiguous (it is updated in place)')
        B, T = self.prepared_shape
        if isinstance(t, torch.Tensor):
            t = t.detach().cpu().reshape(-1).tolist()
        if isinstance(t, int):
            t = [t] * B
        if len(t) != B:
            raise ValueError('t must have one entry per utterance')
        if repeat_noise:
            z = noise.reshape(-1, self.M, T)[0].contiguous()
            if z.dtype != torch.float32 or z.device != self.device:
                raise ValueError('noise must be fp32 on the engine device')
        else:
            z = self._spec(noise, 'noise').contiguous()
        tarr = (C.c_int32 * B)(*[int(v) for v in t])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_p_sample_ex(self._h, xs.data_ptr(), z.data_ptr(), tarr, int(bool(clip_denoised)), int(bool(repeat_noise)),
                                                _stream_ptr(self.device)), 'dsd_p_sample_ex')
        return x

    def sample_plms(self, x: torch.Tensor, k_step: int, interval: int) -> torch.Tensor:
        xs = self._spec(x)
        if not xs.is_contiguous():
            raise ValueError('x must be contiguous (it is updated in place)')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_sample_plms(self._h, xs.data_ptr(), int(k_step), int(interval), _stream_ptr(self.device)), 'dsd_sample_plms')
        return x

    def norm_spec(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B,T,M] -> normalised x [B,M,T]."""
        B, T = self.prepared_shape
        mel = mel.contiguous()
        if tuple(mel.shape) != (B, T, self.M) or mel.dtype != torch.float32 or mel.device != self.device:
            raise ValueError(f'mel must be fp32 [{B},{T},{self.M}] on {self.device}')
        out = torch.empty(B, self.M, T, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_norm_spec(self._h, mel.data_ptr(), out.data_ptr(), _stream_ptr(self.device)), 'dsd_norm_spec')
        return out

    def denorm_spec(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [B,M,T] -> de-normalised mel [B,T,M] (times mask [B,T] if given)."""
        xs = self._spec(x).contiguous()
        B, T = self.prepared_shape
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.float32).contiguous()
            if tuple(mask.shape) != (B, T):
                raise ValueError('mask must be [B,T]')
            mptr = mask.data_ptr()
        out = torch.empty(B, T, self.M, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_denorm_spec(self._h, xs.data_ptr(), mptr, out.data_ptr(), _stream_ptr(self.device)), 'dsd_denorm_spec')
        return out

    def time_layer_kernel(self, layer: int, t: int, iters: int) -> float:
        """Average device milliseconds of one residual-layer kernel launch (HIP events on the launch stream)."""
        ms = C.c_float(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_time_layer_kernel(self._h, layer, t, iters, C.byref(ms), _stream_ptr(self.device)), 'dsd_time_layer_kernel')
        return float(ms.value)

    def loop_timeline(self, x: torch.Tensor, noise: torch.Tensor, k_step: int, phase: int):
        """Debug: per-wave shader-clock stamps [workgroups, 4 waves, 16] of the persistent DDPM loop: 0..7 the layer phase `phase`,
        8..15 the head of its evaluation (include/dsd.h)."""
        import numpy as np
        xs = self._spec(x)
        if noise.dim() == 5:
            noise = noise[:, :, 0]
        max_wg = 1 << 12
        out = np.zeros(max_wg * 64, dtype=np.uint64)
        n = C.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_debug_loop_timeline(self._h, xs.data_ptr(), noise.data_ptr(), int(k_step), int(phase),
                                                        out.ctypes.data_as(C.POINTER(C.c_uint64)), max_wg, C.byref(n), _stream_ptr(self.device)),
                       'dsd_debug_loop_timeline')
        return out[:n.value * 64].reshape(n.value, 4, 16)

    def layer_timeline(self, layer: int, t: int):
        """Debug: per-wave shader-clock stamps [blocks, 4 waves, 8] of one launch of the layer kernel."""
        import numpy as np
        max_blocks = 1 << 16
        out = np.zeros(max_blocks * 32, dtype=np.uint64)
        n = C.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dsd_debug_layer_timeline(self._h, layer, t, out.ctypes.data_as(C.POINTER(C.c_uint64)), max_blocks,
                                                         C.byref(n), _stream_ptr(self.device)), 'dsd_debug_layer_timeline')
        return out[:n.value * 32].reshape(n.value, 4, 8)
