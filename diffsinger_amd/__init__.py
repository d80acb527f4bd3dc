"""diffsinger_amd - MI355X-native (gfx950) diffusion-denoiser hot path for DiffSinger / DiffSpeech.

Public surface mirrors the reference's own plugin API for this path:

    DIFF_DECODERS                       the registry the tasks index with hparams['diff_decoder_type']
                                        (usr/task.py:10, usr/diffspeech_task.py:12, usr/diffsinger_task.py:23)
    DiffNet(in_dims)                    usr/diff/net.py:81
    GaussianDiffusion(phone_encoder, out_dims, denoise_fn, timesteps, K_step, loss_type, betas, spec_min, spec_max)
                                        usr/diff/shallow_diffusion_tts.py:71  (+ .inference(cond, ...))
    register(*registries)               rebinds 'wavenet' (and adds 'wavenet_hip') in the reference's registries

Importing the package does not load the HIP library; constructing an engine does, and fails loudly if
libdsdenoise.so is missing (no CPU fallback)."""
from .hparams import hparams, use_preset  # noqa: F401

__all__ = ['DIFF_DECODERS', 'DiffNet', 'GaussianDiffusion', 'OfflineGaussianDiffusion', 'register', 'hparams', 'use_preset']


def __getattr__(name):      # lazy: torch-heavy modules load on first use
    if name in ('DiffNet',):
        from .net import DiffNet
        return DiffNet
    if name in ('GaussianDiffusion', 'OfflineGaussianDiffusion'):
        from . import diffusion
        return getattr(diffusion, name)
    if name in ('DIFF_DECODERS', 'register'):
        from . import registry
        return getattr(registry, name)
    raise AttributeError(name)
