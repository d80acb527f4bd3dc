"""Drive the REAL reference implementation (only where /root/reference is mounted: the build container).

TEST INFRASTRUCTURE ONLY - used by `oracle/make_golden.py` (fixture generation) and
`tests/test_oracle_vs_reference.py` (proves the restatement bit-equal to the reference).  Nothing here
is reachable from the product path, `bench.py` or the `-m gpu` tests: /root/reference does not exist on
the GPU box.

How the reference is driven (verified end-to-end):
  * `librosa` / `pycwt` are imported (but never used) by utils/cwt.py:1,3 and utils/pitch_utils.py:4 on
    the import chain of usr/diff/net.py -> stub both in sys.modules;
  * YAML `base_config` paths are cwd-relative (utils/hparams.py:47-60) -> chdir to the reference root;
  * `max_beta` / `diff_loss_type` are read at IMPORT time of usr/diff/shallow_diffusion_tts.py (:44,:73)
    -> set_hparams() first, import second.  One python process = one config: use `run_isolated`.
  * the sampler's RNG is the module-global `noise_like` (:38-41) -> replaced by an explicit noise queue.
"""
from __future__ import annotations

import os
import sys
import types
from collections import deque

REFERENCE_ROOT = os.environ.get('DIFFSINGER_REFERENCE', '/root/reference')


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'usr', 'diff', 'net.py'))


class Reference:
    """One loaded reference configuration (process-global: hparams is a module-level dict there)."""

    def __init__(self, config: str, overrides: dict | None = None):
        import torch
        sys.dont_write_bytecode = True                      # the mount is read-only
        for n in ('librosa', 'pycwt'):
            if n not in sys.modules:
                sys.modules[n] = types.ModuleType(n)
        sys.modules['pycwt'].wavelet = types.SimpleNamespace()
        self._cwd = os.getcwd()
        os.chdir(REFERENCE_ROOT)
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
        from utils.hparams import hparams, set_hparams
        set_hparams(config=config, print_hparams=False)
        if overrides:
            hparams.update(overrides)
        self.hparams = hparams
        from usr.diff.net import DiffNet                    # noqa: E402  (after set_hparams on purpose)
        import usr.diff.shallow_diffusion_tts as sdt        # noqa: E402
        from utils.text_encoder import TokenTextEncoder     # noqa: E402
        self.torch, self.DiffNet, self.sdt, self.TokenTextEncoder = torch, DiffNet, sdt, TokenTextEncoder
        os.chdir(self._cwd)

    def build(self, seed: int, final_proj_std: float, timesteps: int, k_step: int, spec_min, spec_max):
        torch = self.torch
        torch.manual_seed(seed)
        net = self.DiffNet(self.hparams['audio_num_mel_bins'])
        if final_proj_std > 0:                              # default is zeros (net.py:105) -> trivial parity
            torch.nn.init.normal_(net.output_projection.weight, std=final_proj_std)
        enc = self.TokenTextEncoder(None, vocab_list=['a', 'b', 'c'], replace_oov=',')
        gd = self.sdt.GaussianDiffusion(enc, self.hparams['audio_num_mel_bins'], net, timesteps=timesteps,
                                        K_step=k_step, loss_type='l1', spec_min=spec_min, spec_max=spec_max).eval()
        return net, gd

    def build_legacy(self, seed: int, final_proj_std: float, timesteps: int, spec_min, spec_max):
        """usr/diff/diffusion.py::GaussianDiffusion (the class usr/task.py::DiffFsTask builds): cosine schedule, no K_step."""
        torch = self.torch
        import usr.diff.diffusion as legacy                # noqa: E402
        self.legacy = legacy
        torch.manual_seed(seed)
        net = self.DiffNet(self.hparams['audio_num_mel_bins'])
        if final_proj_std > 0:
            torch.nn.init.normal_(net.output_projection.weight, std=final_proj_std)
        enc = self.TokenTextEncoder(None, vocab_list=['a', 'b', 'c'], replace_oov=',')
        gd = legacy.GaussianDiffusion(enc, self.hparams['audio_num_mel_bins'], net, timesteps=timesteps, loss_type='l1',
                                      spec_min=spec_min, spec_max=spec_max).eval()
        return net, gd

    def sample_ddpm_legacy(self, gd, x, cond, noises):
        torch = self.torch
        q = list(noises)
        self.legacy.noise_like = lambda shape, device, repeat=False: q.pop(0)
        B = x.shape[0]
        for i in reversed(range(0, gd.num_timesteps)):      # usr/diff/diffusion.py:317-318
            x = gd.p_sample(x, torch.full((B,), i, dtype=torch.long), cond)
        assert not q
        return x

    def sample_ddpm(self, gd, x, cond, noises, k_step):
        torch = self.torch
        q = list(noises)
        self.sdt.noise_like = lambda shape, device, repeat=False: q.pop(0)
        B = x.shape[0]
        for i in reversed(range(0, k_step)):                # shallow_diffusion_tts.py:269-270
            x = gd.p_sample(x, torch.full((B,), i, dtype=torch.long), cond)
        assert not q
        return x

    def sample_plms(self, gd, x, cond, k_step, interval):
        """Reference PLMS only works for B == 1 (builtin max() on a tensor, :192): run per utterance."""
        torch = self.torch
        outs = []
        for b in range(x.shape[0]):
            xb, cb = x[b:b + 1], cond[b:b + 1]
            gd.noise_list = deque(maxlen=4)                 # :262
            for i in reversed(range(0, k_step, interval)):  # :264-267
                xb = gd.p_sample_plms(xb, torch.full((1,), i, dtype=torch.long), interval, cb)
            outs.append(xb)
        return torch.cat(outs, 0)
