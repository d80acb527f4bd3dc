"""The global mutable `hparams` dict of the reference (utils/hparams.py:23-122), as far as the hot path reads
it.  When this package runs inside the reference tree (`utils.hparams` importable) that very dict object is
used, so `set_hparams()` there configures us too; stand-alone it is our own dict, filled by `use_preset()`
or by the caller.

Keys the hot path reads: hidden_size, residual_layers, residual_channels,
dilation_cycle_length, audio_num_mel_bins, keep_bins, timesteps, K_step, diff_loss_type, diff_decoder_type,
schedule_type, max_beta, spec_min, spec_max, gaussian_start, pndm_speedup, use_midi."""
from __future__ import annotations

try:                                    # inside the reference tree: share its dict
    from utils.hparams import hparams   # type: ignore
except Exception:                       # stand-alone
    hparams = {}


def use_preset(name: str, **overrides) -> dict:
    """Fill `hparams` with the hot-path keys of one of the reference's shipped configs."""
    from .synth import presets
    pre = dict(presets()[name])
    pre.pop('source', None)
    pre.update(overrides)
    hparams.update(pre)
    return hparams
