"""The `DIFF_DECODERS` registry (usr/task.py:10-12, usr/diffspeech_task.py:12-14, usr/diffsinger_task.py:23-27):
`DIFF_DECODERS[hparams['diff_decoder_type']](hparams)` -> denoise_fn."""
from __future__ import annotations

from .net import DiffNet

def _fft(hp):
    from .candidate_decoder import FFT          # usr/diffsinger_task.py:25-26
    return FFT(hp['hidden_size'], hp['dec_layers'], hp['dec_ffn_kernel_size'], hp['num_heads'])


DIFF_DECODERS = {
    'wavenet': lambda hp: DiffNet(hp['audio_num_mel_bins']),
    'wavenet_hip': lambda hp: DiffNet(hp['audio_num_mel_bins']),
    'fft': _fft,
}


def register(*registries, override: bool = True):
    """Insert the HIP denoiser into the reference's registries, e.g.

        import usr.task, usr.diffspeech_task, usr.diffsinger_task, diffsinger_amd
        diffsinger_amd.register(usr.task.DIFF_DECODERS, usr.diffspeech_task.DIFF_DECODERS, usr.diffsinger_task.DIFF_DECODERS)

    With override=True the stock key 'wavenet' is rebound too, so shipped YAMLs pick the HIP path unchanged."""
    for reg in registries:
        reg['wavenet_hip'] = DIFF_DECODERS['wavenet_hip']
        if override:
            reg['wavenet'] = DIFF_DECODERS['wavenet']
            if 'fft' in reg:                    # usr/diffsinger_task.py registers the transformer candidate decoder too
                reg['fft'] = DIFF_DECODERS['fft']
    return registries
