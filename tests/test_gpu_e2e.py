"""GPU: the whole inference chain of the reference's SVS end-to-end config (inference/svs/ds_e2e.py:32-47) on the HIP modules -
FastSpeech2MIDI -> PLMS diffusion sampler -> PitchExtractor -> NSF-HiFi-GAN - with synthetic weights: the chain class equals the
stage-by-stage composition of the separately parity-tested modules, bit for bit, and produces a finite waveform of T * 256 samples."""
import numpy as np
import pytest
import torch

from oracle import hifigan_oracle as HO
from oracle import pe_oracle as PO
from oracle.make_golden_hifigan import CONFIG
from tests import fs2_helpers as FH

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_e2e_chain_equals_stagewise_composition():
    import diffsinger_amd
    from diffsinger_amd import hparams
    from diffsinger_amd.infer import DiffSingerE2EInfer
    from diffsinger_amd.pe import PitchExtractor
    from diffsinger_amd.synth import presets
    from diffsinger_amd.vocoder import HifiGanGenerator
    case, fs2m, hp, params, inp = FH.case_setup('fs2_midi_e2e_free')             # opencpop e2e: use_midi, pe_enable, PLMS
    pre = presets()[case['preset']]
    hparams['use_nsf'] = True
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1',
                                          spec_min=pre['spec_min'], spec_max=pre['spec_max'], fs2=fs2m)
    pe = PitchExtractor().eval()
    pe.load_state_dict(PO.synth_extractor_params(hp, 31), strict=True)
    h = dict(CONFIG, use_pitch_embed=True)
    voc = HifiGanGenerator(h)
    voc.load_state_dict(HO.synth_generator_params(h, 32), strict=True)
    chain = DiffSingerE2EInfer(gd, voc, pe, device=DEV)
    torch.manual_seed(0)
    wav = chain.forward_model(inp)
    # the same stages by hand
    torch.manual_seed(0)
    with torch.no_grad():
        out = chain.model(inp['txt_tokens'].to(DEV), infer=True, pitch_midi=inp['pitch_midi'].to(DEV), midi_dur=inp['midi_dur'].to(DEV),
                          is_slur=inp['is_slur'].to(DEV))
        mel = out['mel_out']
        f0 = chain.pe(mel)['f0_denorm_pred']
        y = chain.vocoder(mel.transpose(2, 1), f0)
    B, T, M = mel.shape
    assert M == 80 and wav.shape == (B * T * 256,) and wav.dtype == np.float32
    assert np.isfinite(wav).all() and np.abs(wav).max() <= 1.0 and np.abs(wav).max() > 1e-3
    assert np.array_equal(wav, y.view(-1).cpu().numpy())
    assert f0.shape == (B, T) and bool((f0 >= 0).all())
    print('e2e frames', B, T, 'voiced frac', float((f0 > 0).float().mean()))
