"""FastSpeech2 fixture cases: what oracle/make_golden_fs2.py runs THROUGH THE REFERENCE modules and
what the tests replay through oracle/fs2_oracle.py and the HIP modules.  TEST INFRASTRUCTURE ONLY.

There are no checkpoints: the weights are a seeded synthetic state_dict (`synth_params`) loaded with strict=True into the
reference module, the oracle and the HIP module alike; inputs are seeded (`make_inputs`).
  mode 'teacher': mel2ph, f0, uv supplied (what validation / shallow-diffusion training feed, tasks/tts/fs2.py)
  mode 'free'   : durations and pitch predicted (test-time inference)"""
import torch

VOCAB = 63          # TokenTextEncoder(None, vocab_list=[60 phones]) has 3 reserved symbols; <pad> = 0

CASES = {
    # DiffSpeech / LJSpeech: cwt pitch predictor (Linear + conv stack, 11 outputs), sinusoidal positions
    'fs2_lj_teacher': dict(preset='lj_ds_beta6', mode='teacher', B=3, T_txt=21, seed=201),
    'fs2_lj_free': dict(preset='lj_ds_beta6', mode='free', B=2, T_txt=17, seed=202),
    # DiffSinger / PopCS: frame-level pitch predictor
    'fs2_popcs_teacher': dict(preset='popcs_ds_beta6', mode='teacher', B=2, T_txt=19, seed=203),
    'fs2_popcs_free': dict(preset='popcs_ds_beta6', mode='free', B=2, T_txt=15, seed=204),
    # FastSpeech2MIDI, Opencpop cascade: MIDI / duration / slur embeddings, relative positional encoding, 5-layer predictors
    'fs2_midi_cascade_teacher': dict(preset='opencpop_ds60_rel', mode='teacher', B=2, T_txt=23, seed=205),
    # FastSpeech2MIDI e2e (no pitch embedding), predicted durations
    'fs2_midi_e2e_free': dict(preset='opencpop_ds1000', mode='free', B=3, T_txt=18, seed=206),
    # ... and teacher-forced: the forward of the e2e TRAINING step (BASELINE configs[3]; usr/diffsinger_task.py:273-300)
    'fs2_midi_e2e_teacher': dict(preset='opencpop_ds1000', mode='teacher', B=2, T_txt=20, seed=207),
    # options no shipped DiffSpeech / DiffSinger YAML enables (configs/singing/base.yaml:34 ships use_spk_embed: true, the usr/ configs
    # override it): `overrides` are applied to the reference's hparams and to ours alike
    #   projected speaker d-vector + energy embedding (fs2.py:45-46, :75-82, :107-108, :139-140), energy supplied
    'fs2_popcs_spk_energy_teacher': dict(preset='popcs_ds_beta6', mode='teacher', B=2, T_txt=16, seed=208,
                                         overrides={'use_spk_embed': True, 'use_energy_embed': True}),
    #   speaker ids with separate tables for the duration / pitch predictors (:37-42, :109-119), energy predicted
    'fs2_lj_spkid_energy_free': dict(preset='lj_ds_beta6', mode='free', B=3, T_txt=14, seed=209,
                                     overrides={'use_spk_id': True, 'use_split_spk_id': True, 'num_spk': 5, 'use_energy_embed': True}),
    #   phone-level pitch (:184-196): one f0 per phone, its bin gathered to the frames
    'fs2_popcs_ph_teacher': dict(preset='popcs_ds_beta6', mode='teacher', B=2, T_txt=18, seed=210, overrides={'pitch_type': 'ph'}),
    'fs2_popcs_ph_free': dict(preset='popcs_ds_beta6', mode='free', B=2, T_txt=13, seed=211, overrides={'pitch_type': 'ph'}),
}

OUT_KEYS = ['encoder_out', 'mel2ph', 'dur', 'decoder_inp', 'mel_out', 'pitch_pred', 'cwt', 'f0_denorm', 'energy_pred']


def synth_params(shapes: dict, seed: int) -> dict:
    """shapes: {state_dict key: (shape, dtype)} of the module.  Deterministic, non-trivial values for every tensor."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(shapes):
        shape, dtype = shapes[k]
        if k.endswith('_float_tensor'):
            out[k] = torch.zeros(shape, dtype=dtype)
        elif k.endswith('pos_embed_alpha'):
            out[k] = 1 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            std = (shape[1] ** -0.5) if ('embed' in k and len(shape) == 2 and 'predictor' not in k) else fan_in ** -0.5
            out[k] = torch.randn(shape, generator=g) * std
        elif k.endswith('weight'):                      # LayerNorm gains
            out[k] = 1 + 0.1 * torch.randn(shape, generator=g)
        else:                                           # biases
            out[k] = 0.1 * torch.randn(shape, generator=g)
    for k in ('encoder_embed_tokens.weight', 'encoder.embed_tokens.weight', 'pitch_embed.weight', 'midi_embed.weight', 'energy_embed.weight'):
        if k in out:
            out[k][0] = 0                               # padding_idx row
    if 'encoder.embed_tokens.weight' in out:
        out['encoder.embed_tokens.weight'] = out['encoder_embed_tokens.weight']      # one shared tensor in the module
    out['dur_predictor.linear.bias'] = out['dur_predictor.linear.bias'] + 1.2        # ~2 frames / phone: no empty utterance
    if 'energy_predictor.linear.bias' in out:
        out['energy_predictor.linear.bias'] = out['energy_predictor.linear.bias'] + 2.0   # predicted energy > 0 (a negative one indexes the table out of range, fs2.py:179)
    return out


def make_inputs(case: dict, use_midi: bool) -> dict:
    g = torch.Generator().manual_seed(case['seed'])
    B, Tt = case['B'], case['T_txt']
    tok = torch.randint(1, VOCAB, (B, Tt), generator=g)
    for b in range(1, B):                               # ragged: utterance b is shorter
        tok[b, Tt - 3 * b:] = 0
    inp = {'txt_tokens': tok}
    valid = (tok > 0)
    if use_midi:
        inp['pitch_midi'] = torch.randint(40, 80, (B, Tt), generator=g) * valid
        inp['midi_dur'] = torch.rand(B, Tt, generator=g) * valid
        inp['is_slur'] = torch.randint(0, 2, (B, Tt), generator=g) * valid
    if case['mode'] == 'teacher':
        dur = torch.randint(1, 7, (B, Tt), generator=g) * valid
        T = int(dur.sum(-1).max())
        cs = torch.cumsum(dur, 1)
        pos = torch.arange(T)[None, None]
        mask = (pos >= (cs - dur)[:, :, None]) & (pos < cs[:, :, None])
        inp['mel2ph'] = (torch.arange(1, Tt + 1)[None, :, None] * mask.long()).sum(1)
        uv = (torch.rand(B, T, generator=g) < 0.2).float()
        inp['uv'] = uv
        inp['f0'] = (torch.rand(B, T, generator=g) * 2 + 6.5) * (1 - uv)      # log2 Hz, 0 where unvoiced
    ov = case.get('overrides', {})
    g2 = torch.Generator().manual_seed(case['seed'] + 77)                   # (the base cases' draws stay what they were)
    if ov.get('pitch_type') == 'ph' and case['mode'] == 'teacher':
        del inp['uv']
        inp['f0'] = (torch.rand(B, Tt, generator=g2) * 2 + 6.5) * valid      # one value per phone
    if ov.get('use_spk_embed'):
        inp['spk_embed'] = torch.randn(B, 256, generator=g2)
    if ov.get('use_spk_id'):
        n = ov.get('num_spk', 1) + 1
        inp['spk_embed'] = torch.randint(0, n, (B,), generator=g2)
        if ov.get('use_split_spk_id'):
            inp['spk_embed_dur_id'] = torch.randint(0, n, (B,), generator=g2)
            inp['spk_embed_f0_id'] = torch.randint(0, n, (B,), generator=g2)
    if ov.get('use_energy_embed') and case['mode'] == 'teacher':
        inp['energy'] = torch.rand(B, inp['mel2ph'].shape[1], generator=g2) * 3.5 * (inp['mel2ph'] > 0)
    return inp
