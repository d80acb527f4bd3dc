"""The inference chain around the hot path as the reference's SVS inference runs it - acoustic model (FastSpeech2MIDI + diffusion
sampler) -> PitchExtractor -> NSF-HiFi-GAN - with every stage on the HIP modules of this package:

    DiffSingerE2EInfer.forward_model      inference/svs/ds_e2e.py:32-47
    BaseSVSInfer.run_vocoder              inference/svs/base_svs_infer.py:61-70

The text front end (g2p, MIDI / duration parsing: base_svs_infer.py:72-230) is host-side string processing and stays the
reference's; this class starts from the batch dict it produces."""
from __future__ import annotations

import torch

from .hparams import hparams


class DiffSingerE2EInfer:
    def __init__(self, model, vocoder, pe=None, device='cuda'):
        """model: diffsinger_amd.GaussianDiffusion (with its FastSpeech2MIDI); vocoder: diffsinger_amd.vocoder.HifiGanGenerator;
        pe: diffsinger_amd.pe.PitchExtractor or None (hparams['pe_enable'])."""
        self.device = torch.device(device)
        self.model = model.to(self.device).eval()
        self.vocoder = vocoder.to(self.device).eval()
        self.pe = pe.to(self.device).eval() if pe is not None else None

    def run_vocoder(self, c, **kwargs):
        c = c.transpose(2, 1)                                         # [B, 80, T]
        f0 = kwargs.get('f0')                                         # [B, T]
        if f0 is not None and hparams.get('use_nsf'):
            y = self.vocoder(c, f0).view(-1)
        else:
            y = self.vocoder(c).view(-1)
        return y[None]

    @torch.no_grad()
    def forward_model(self, sample):
        """sample: txt_tokens [B,T_t] (+ pitch_midi, midi_dur, is_slur, spk_ids).  Returns the waveform [B * T * hop] as numpy,
        like the reference (utterances concatenated)."""
        dev = self.device
        g = lambda k: sample[k].to(dev) if sample.get(k) is not None else None
        output = self.model(g('txt_tokens'), spk_id=g('spk_ids'), ref_mels=None, infer=True, pitch_midi=g('pitch_midi'),
                            midi_dur=g('midi_dur'), is_slur=g('is_slur'))
        mel_out = output['mel_out']                                   # [B, T, 80]
        if self.pe is not None:
            f0_pred = self.pe(mel_out)['f0_denorm_pred']              # pe predicts from the predicted mel
        else:
            f0_pred = output.get('f0_denorm')
        wav_out = self.run_vocoder(mel_out, f0=f0_pred)
        return wav_out.cpu().numpy()[0]
