"""CPU oracle for the FastSpeech2 / FastSpeech2MIDI conditioner + aux decoder.
TEST INFRASTRUCTURE ONLY - imported by tests/ and the fixture generator, never by diffsinger_amd/.

A functional restatement, in torch-CPU fp32, of the INFERENCE forward of (paths relative to /root/reference):

  FastSpeech2.forward / add_dur / add_pitch / run_decoder    modules/fastspeech/fs2.py:93-231
  FastSpeech2MIDI.forward, FastspeechMIDIEncoder             modules/diffsinger_midi/fs2.py:10-118
  FFTBlocks / FastspeechEncoder / FastspeechDecoder          modules/fastspeech/tts_modules.py:251-356
  DurationPredictor, LengthRegulator, PitchPredictor         modules/fastspeech/tts_modules.py:58-245
  EncSALayer, TransformerFFNLayer, MultiheadAttention        modules/commons/common_layers.py:542-588, 486-522, 166-263
  SinusoidalPositionalEmbedding, make_positions              modules/commons/common_layers.py:88-143, utils/__init__.py:145-157
  RelPositionalEncoding                                      modules/commons/espnet_positional_embedding.py:86-112
  denorm_f0 / norm_f0 / f0_to_coarse, cwt2f0                 utils/pitch_utils.py:21-77, utils/cwt.py:118-147

It works on a plain state_dict with the reference's parameter names and an hparams dict.  Pinned against the live
reference (tests/test_fs2_oracle_vs_reference.py, build container) and by fixtures generated FROM the reference
(oracle/make_golden_fs2.py -> tests/golden/fs2_*.npz).

Not covered (raise): speaker embeddings (use_spk_id / use_spk_embed: off in every shipped DiffSpeech/DiffSinger config),
energy embedding (use_energy_embed: off), pitch_ar, pitch_type 'ph', dur_loss other than 'mse', ffn_padding 'LEFT', norm 'bn'."""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

f0_bin = 256
f0_max = 1100.0
f0_min = 50.0
f0_mel_min = 1127 * np.log(1 + f0_min / 700)
f0_mel_max = 1127 * np.log(1 + f0_max / 700)


# ----------------------------------------------------------------------------------------------
# positional tables
# ----------------------------------------------------------------------------------------------
def sinusoidal_table(n: int, dim: int, padding_idx: Optional[int]) -> torch.Tensor:
    """common_layers.py:105-124 (tensor2tensor flavour: [sin | cos], padding row zeroed)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(n, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(n, -1)
    if dim % 2 == 1:
        e = torch.cat([e, torch.zeros(n, 1)], dim=1)
    if padding_idx is not None:
        e[padding_idx, :] = 0
    return e


def make_positions(x: torch.Tensor, padding_idx: int) -> torch.Tensor:
    """utils/__init__.py:145-157: non-padding symbols -> 1-based running position + padding_idx."""
    mask = x.ne(padding_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


def sinusoidal_positions(x0: torch.Tensor, dim: int, padding_idx: int, init_size: int) -> torch.Tensor:
    """SinusoidalPositionalEmbedding.forward (common_layers.py:126-147) on `x0` = the [B,T] tensor whose zeros mark padding
    (token ids for the encoder; channel 0 of the activations for FFTBlocks / PitchPredictor, tts_modules.py:227,:297)."""
    B, T = x0.shape[:2]
    n = max(init_size, padding_idx + 1 + T)
    tab = sinusoidal_table(n, dim, padding_idx)
    pos = make_positions(x0, padding_idx)
    return tab.index_select(0, pos.view(-1)).view(B, T, -1)


def rel_positional_table(T: int, dim: int) -> torch.Tensor:
    """espnet_positional_embedding.py:23-46 with reverse=True: pe[t] encodes position T_max-1-t for a table of
    max_len = 5000 rows (extended when T is larger); the module slices the FIRST T rows."""
    n = max(5000, T)
    pe = torch.zeros(n, dim)
    position = torch.arange(n - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe[:T]


# ----------------------------------------------------------------------------------------------
# transformer (FFT) blocks
# ----------------------------------------------------------------------------------------------
def self_attention(x, in_proj_w, out_proj_w, key_padding_mask, heads: int):
    """MultiheadAttention(self_attention=True, bias=False) through F.multi_head_attention_forward
    (common_layers.py:243-263).  x [T,B,C]; key_padding_mask [B,T] bool (True = pad)."""
    T, B, C = x.shape
    hd = C // heads
    q, k, v = F.linear(x, in_proj_w).chunk(3, dim=-1)
    q = q.contiguous().view(T, B * heads, hd).transpose(0, 1)
    k = k.contiguous().view(T, B * heads, hd).transpose(0, 1)
    v = v.contiguous().view(T, B * heads, hd).transpose(0, 1)
    mask = torch.zeros(B, 1, 1, T, dtype=x.dtype).masked_fill(key_padding_mask.view(B, 1, 1, T), float('-inf'))
    mask = mask.expand(-1, heads, -1, -1).reshape(B * heads, 1, T)
    q = q * math.sqrt(1.0 / float(hd))
    w = torch.baddbmm(mask, q, k.transpose(-2, -1))
    w = F.softmax(w, dim=-1)
    o = torch.bmm(w, v)
    o = o.transpose(0, 1).contiguous().view(T * B, C)
    return F.linear(o, out_proj_w).view(T, B, C)


def ffn(p, pre, x, kernel: int, act: str):
    """TransformerFFNLayer (common_layers.py:486-522), padding 'SAME'.  x [T,B,C]."""
    y = F.conv1d(x.permute(1, 2, 0), p[pre + 'ffn_1.weight'], p[pre + 'ffn_1.bias'], padding=kernel // 2).permute(2, 0, 1)
    y = y * kernel ** -0.5
    if act == 'gelu':
        y = F.gelu(y)
    elif act == 'relu':
        y = F.relu(y)
    else:
        raise NotImplementedError(act)
    return F.linear(y, p[pre + 'ffn_2.weight'], p[pre + 'ffn_2.bias'])


def enc_sa_layer(p, pre, x, pad_mask, heads: int, kernel: int, act: str):
    """EncSALayer.forward (common_layers.py:565-588), eval mode.  x [T,B,C], pad_mask [B,T] bool."""
    C = x.shape[-1]
    keep = (1 - pad_mask.float()).transpose(0, 1)[..., None]
    res = x
    y = F.layer_norm(x, (C,), p[pre + 'layer_norm1.weight'], p[pre + 'layer_norm1.bias'], 1e-5)
    y = self_attention(y, p[pre + 'self_attn.in_proj_weight'], p[pre + 'self_attn.out_proj.weight'], pad_mask, heads)
    x = (res + y) * keep
    res = x
    y = F.layer_norm(x, (C,), p[pre + 'layer_norm2.weight'], p[pre + 'layer_norm2.bias'], 1e-5)
    y = ffn(p, pre + 'ffn.', y, kernel, act)
    return (res + y) * keep


def fft_blocks(p, pre, x, n_layers: int, heads: int, kernel: int, act: str, pad_mask=None, use_pos_embed=True,
               return_hiddens=False):
    """FFTBlocks.forward (tts_modules.py:288-314), eval mode.  x [B,T,C] -> [B,T,C]."""
    C = x.shape[-1]
    pad_mask = x.abs().sum(-1).eq(0) if pad_mask is None else pad_mask
    keep_tb = 1 - pad_mask.transpose(0, 1).float()[:, :, None]
    if use_pos_embed:
        x = x + p[pre + 'pos_embed_alpha'] * sinusoidal_positions(x[..., 0], C, 0, 2000)
    x = x.transpose(0, 1) * keep_tb
    hiddens = []
    for l in range(n_layers):
        x = enc_sa_layer(p, f'{pre}layers.{l}.op.', x, pad_mask, heads, kernel, act) * keep_tb
        hiddens.append(x)
    x = F.layer_norm(x, (C,), p[pre + 'layer_norm.weight'], p[pre + 'layer_norm.bias'], 1e-5) * keep_tb
    if return_hiddens:
        return x.transpose(0, 1), [h.transpose(0, 1) for h in hiddens]
    return x.transpose(0, 1)


def encoder(p, hp, txt_tokens, midi=None):
    """FastspeechEncoder.forward (tts_modules.py:330-345) / FastspeechMIDIEncoder (diffsinger_midi/fs2.py:10-37).
    midi = (midi_embedding, midi_dur_embedding, slur_embedding) or None."""
    C = hp['hidden_size']
    pad = txt_tokens.eq(0)
    x = math.sqrt(C) * F.embedding(txt_tokens, p['encoder.embed_tokens.weight'], 0)
    if midi is not None:
        x = x + midi[0] + midi[1] + midi[2]
    if hp['use_pos_embed']:
        if hp.get('rel_pos'):
            if midi is not None:
                x = x * math.sqrt(C) + rel_positional_table(x.shape[1], C)[None]      # RelPositionalEncoding.forward
            else:
                # the plain FastspeechEncoder calls embed_positions(txt_tokens) (tts_modules.py:350): with rel_pos the module
                # is RelPositionalEncoding, which would scale the integer tokens - no shipped non-MIDI config sets rel_pos
                raise NotImplementedError('rel_pos without use_midi')
        else:
            x = x + sinusoidal_positions(txt_tokens, C, 0, 2000)
    return fft_blocks(p, 'encoder.', x, hp['enc_layers'], hp['num_heads'], hp['enc_ffn_kernel_size'], hp['ffn_act'], pad,
                      use_pos_embed=False)


# ----------------------------------------------------------------------------------------------
# variance predictors
# ----------------------------------------------------------------------------------------------
def conv_stack(p, pre, x_bct, n_layers: int, kernel: int, mask_keep=None):
    """[ConstantPad1d 'SAME', Conv1d, ReLU, LayerNorm(dim=1, eps=1e-12), Dropout] x n (tts_modules.py:84-97, :198-209)."""
    for i in range(n_layers):
        w, b = p[f'{pre}conv.{i}.1.weight'], p[f'{pre}conv.{i}.1.bias']
        x_bct = F.conv1d(F.pad(x_bct, [(kernel - 1) // 2, (kernel - 1) // 2]), w, b)
        x_bct = F.relu(x_bct)
        x_bct = F.layer_norm(x_bct.transpose(1, -1), (w.shape[0],), p[f'{pre}conv.{i}.3.weight'], p[f'{pre}conv.{i}.3.bias'],
                             1e-12).transpose(1, -1)
        if mask_keep is not None:
            x_bct = x_bct * mask_keep[:, None, :]
    return x_bct


def duration_predictor_inference(p, hp, xs, x_masks):
    """DurationPredictor.inference (tts_modules.py:107-131), dur_loss 'mse'.  Returns (dur long [B,T], xs [B,T,1])."""
    if hp['dur_loss'] != 'mse':
        raise NotImplementedError(hp['dur_loss'])
    keep = 1 - x_masks.float()
    y = conv_stack(p, 'dur_predictor.', xs.transpose(1, -1), hp['dur_predictor_layers'], hp['dur_predictor_kernel'], keep)
    y = F.linear(y.transpose(1, -1), p['dur_predictor.linear.weight'], p['dur_predictor.linear.bias'])
    y = y * keep[:, :, None]
    dur = torch.clamp(torch.round(y.squeeze(-1).exp() - 1.0), min=0).long()
    return dur, y


def length_regulator(dur, dur_padding):
    """LengthRegulator.forward (tts_modules.py:158-186), alpha = 1."""
    dur = torch.round(dur.float()).long()
    dur = dur * (1 - dur_padding.long())
    token_idx = torch.arange(1, dur.shape[1] + 1)[None, :, None]
    cs = torch.cumsum(dur, 1)
    cs_prev = F.pad(cs, [1, -1], mode='constant', value=0)
    pos = torch.arange(dur.sum(-1).max())[None, None]
    mask = (pos >= cs_prev[:, :, None]) & (pos < cs[:, :, None])
    return (token_idx * mask.long()).sum(1)


def pitch_predictor(p, pre, xs, n_layers: int, kernel: int):
    """PitchPredictor.forward (tts_modules.py:215-229).  xs [B,T,idim] -> [B,T,odim]."""
    idim = xs.shape[-1]
    xs = xs + p[pre + 'pos_embed_alpha'] * sinusoidal_positions(xs[..., 0], idim, 0, 4096)
    y = conv_stack(p, pre, xs.transpose(1, -1), n_layers, kernel)
    return F.linear(y.transpose(1, -1), p[pre + 'linear.weight'], p[pre + 'linear.bias'])


def norm_f0(f0, uv, hp):
    if hp['pitch_norm'] == 'standard':
        f0 = (f0 - hp['f0_mean']) / hp['f0_std']
    if hp['pitch_norm'] == 'log':
        f0 = torch.log2(f0)
    if uv is not None and hp['use_uv']:
        f0[uv > 0] = 0
    return f0


def denorm_f0(f0, uv, hp, pitch_padding=None):
    if hp['pitch_norm'] == 'standard':
        f0 = f0 * hp['f0_std'] + hp['f0_mean']
    if hp['pitch_norm'] == 'log':
        f0 = 2 ** f0
    if uv is not None and hp['use_uv']:
        f0[uv > 0] = 0
    if pitch_padding is not None:
        f0[pitch_padding] = 0
    return f0


def f0_to_coarse(f0):
    f0_mel = 1127 * (1 + f0 / 700).log()
    f0_mel[f0_mel > 0] = (f0_mel[f0_mel > 0] - f0_mel_min) * (f0_bin - 2) / (f0_mel_max - f0_mel_min) + 1
    f0_mel[f0_mel <= 1] = 1
    f0_mel[f0_mel > f0_bin - 1] = f0_bin - 1
    return (f0_mel + 0.5).long()


def cwt2f0_norm(cwt_spec, mean, std, mel2ph, hp):
    """fs2.py:239-245 with utils/cwt.py:118-125,135-142 (10 scales)."""
    b = (torch.arange(0, 10).float()[None, None, :] + 1 + 2.5) ** (-2.5)
    rec = (cwt_spec * b).sum(-1)
    rec = (rec - rec.mean(-1, keepdim=True)) / rec.std(-1, keepdim=True)
    f0 = (rec * std[:, None] + mean[:, None]).exp()
    f0 = torch.cat([f0] + [f0[:, -1:]] * (mel2ph.shape[1] - f0.shape[1]), 1)
    return norm_f0(f0, None, hp)


def scale_grad(x, g):
    """fs2.py:153, :194: `x.detach() + predictor_grad * (x - x.detach())` - the value of x (bit for bit: x - x is 0), predictor_grad times its
    gradient.  Only under autograd (the gradient pin of the training path, oracle/check_fs2_grad.py)."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x
    return x.detach() + g * (x - x.detach())


def add_pitch(p, hp, decoder_inp, f0, uv, mel2ph, ret, encoder_out):
    """FastSpeech2.add_pitch (fs2.py:183-231)."""
    if hp.get('pitch_ar'):
        # fs2.py:215 calls self.pitch_predictor(decoder_inp, f0 ...) - PitchPredictor.forward (tts_modules.py:222) takes one argument:
        # the option raises a TypeError in the reference itself
        raise NotImplementedError('pitch_ar: dead option of the reference (TypeError at fs2.py:215)')
    if hp['pitch_type'] == 'ph':
        # fs2.py:184-196: prediction and quantisation at the PHONE rate, the bins gathered to the frames through mel2ph
        pp_inp = scale_grad(encoder_out, hp['predictor_grad'])
        pitch_padding = encoder_out.sum().abs() == 0
        ret['pitch_pred'] = pp = pitch_predictor(p, 'pitch_predictor.', pp_inp, hp['predictor_layers'], hp['predictor_kernel'])
        if f0 is None:
            f0 = pp[:, :, 0]
        ret['f0_denorm'] = f0_denorm = denorm_f0(f0, None, hp, pitch_padding=pitch_padding)
        pitch = F.pad(f0_to_coarse(f0_denorm.clone()), [1, 0])
        pitch = torch.gather(pitch, 1, mel2ph)
        ret['pitch_coarse'] = pitch
        return F.embedding(pitch, p['pitch_embed.weight'], 0)
    decoder_inp = scale_grad(decoder_inp, hp['predictor_grad'])
    pitch_padding = mel2ph == 0
    given_f0 = f0 is not None
    if hp['pitch_type'] == 'cwt':
        pitch_padding = None
        h = F.linear(decoder_inp, p['cwt_predictor.0.weight'], p['cwt_predictor.0.bias'])
        ret['cwt'] = cwt_out = pitch_predictor(p, 'cwt_predictor.1.', h, hp['predictor_layers'], hp['predictor_kernel'])
        s = encoder_out[:, 0, :]
        s = F.relu(F.linear(s, p['cwt_stats_layers.0.weight'], p['cwt_stats_layers.0.bias']))
        s = F.relu(F.linear(s, p['cwt_stats_layers.2.weight'], p['cwt_stats_layers.2.bias']))
        stats = F.linear(s, p['cwt_stats_layers.4.weight'], p['cwt_stats_layers.4.bias'])
        mean = ret['f0_mean'] = stats[:, 0]
        std = ret['f0_std'] = stats[:, 1]
        if f0 is None:
            std = std * hp['cwt_std_scale']
            f0 = cwt2f0_norm(cwt_out[:, :, :10], mean, std, mel2ph, hp)
            if hp['use_uv']:
                uv = cwt_out[:, :, -1] > 0
    else:
        ret['pitch_pred'] = pp = pitch_predictor(p, 'pitch_predictor.', decoder_inp, hp['predictor_layers'], hp['predictor_kernel'])
        if f0 is None:
            f0 = pp[:, :, 0]
        if hp['use_uv'] and uv is None:
            uv = pp[:, :, 1] > 0
    ret['f0_denorm'] = f0_denorm = denorm_f0(f0, uv, hp, pitch_padding=pitch_padding)
    if pitch_padding is not None:
        # fs2.py:225-226 `f0[pitch_padding] = 0`: when f0 is the view pitch_pred[:, :, 0] this zeroes ret['pitch_pred'] at the
        # padded frames as a side effect (kept: the returned pitch_pred is compared too); a caller-supplied f0 is not touched here
        if given_f0:
            f0 = f0.clone()
        f0[pitch_padding] = 0
    pitch = f0_to_coarse(f0_denorm.clone())
    ret['pitch_coarse'] = pitch
    return F.embedding(pitch, p['pitch_embed.weight'], 0)


# ----------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------
def add_energy(p, hp, decoder_inp, energy, ret):
    """FastSpeech2.add_energy (fs2.py:174-181); EnergyPredictor is PitchPredictor (tts_modules.py:253-254)."""
    decoder_inp = scale_grad(decoder_inp, hp['predictor_grad'])
    ret['energy_pred'] = energy_pred = pitch_predictor(p, 'energy_predictor.', decoder_inp, hp['predictor_layers'], hp['predictor_kernel'])[:, :, 0]
    if energy is None:
        energy = energy_pred
    energy = torch.clamp(energy * 256 // 4, max=255).long()
    ret['energy_coarse'] = energy
    return F.embedding(energy, p['energy_embed.weight'], 0)


def fs2_forward(p: Dict[str, torch.Tensor], hp: dict, txt_tokens, mel2ph=None, f0=None, uv=None, skip_decoder=False,
                pitch_midi=None, midi_dur=None, is_slur=None, spk_embed=None, energy=None, spk_embed_dur_id=None,
                spk_embed_f0_id=None) -> Dict[str, torch.Tensor]:
    """FastSpeech2.forward (fs2.py:93-149) / FastSpeech2MIDI.forward (diffsinger_midi/fs2.py:55-118), infer=True."""
    ret = {}
    if hp.get('use_midi'):
        midi_emb = F.embedding(pitch_midi, p['midi_embed.weight'], 0)
        dur_emb = F.linear(midi_dur[:, :, None], p['midi_dur_layer.weight'], p['midi_dur_layer.bias']) if midi_dur is not None else 0
        slur_emb = F.embedding(is_slur, p['is_slur_embed.weight']) if is_slur is not None else 0
        encoder_out = encoder(p, hp, txt_tokens, (midi_emb, dur_emb, slur_emb))
    else:
        encoder_out = encoder(p, hp, txt_tokens)
    ret['encoder_out'] = encoder_out
    src_nonpadding = (txt_tokens > 0).float()[:, :, None]
    # speaker conditioning (fs2.py:107-121): a projected d-vector, or embedding rows by speaker id (optionally separate tables for the
    # duration and the pitch predictors)
    if hp.get('use_spk_embed'):
        spk_dur = spk_f0 = spk = F.linear(spk_embed, p['spk_embed_proj.weight'], p['spk_embed_proj.bias'])[:, None, :]
    elif hp.get('use_spk_id'):
        sid = spk_embed
        spk_embed_dur_id = sid if spk_embed_dur_id is None else spk_embed_dur_id
        spk_embed_f0_id = sid if spk_embed_f0_id is None else spk_embed_f0_id
        spk = F.embedding(sid, p['spk_embed_proj.weight'])[:, None, :]
        spk_dur = spk_f0 = spk
        if hp.get('use_split_spk_id'):
            spk_dur = F.embedding(spk_embed_dur_id, p['spk_embed_dur.weight'])[:, None, :]
            spk_f0 = F.embedding(spk_embed_f0_id, p['spk_embed_f0.weight'])[:, None, :]
    else:
        spk_dur = spk_f0 = spk = 0
    dur_inp = scale_grad((encoder_out + 0 + spk_dur) * src_nonpadding, hp['predictor_grad'])
    if mel2ph is None:
        dur, xs = duration_predictor_inference(p, hp, dur_inp, txt_tokens == 0)
        ret['dur'], ret['dur_choice'] = xs, dur
        mel2ph = length_regulator(dur, txt_tokens == 0)
    else:
        # DurationPredictor.forward (tts_modules.py:133-142): log-domain durations, squeezed
        keep = 1 - (txt_tokens == 0).float()
        y = conv_stack(p, 'dur_predictor.', dur_inp.transpose(1, -1), hp['dur_predictor_layers'], hp['dur_predictor_kernel'], keep)
        y = F.linear(y.transpose(1, -1), p['dur_predictor.linear.weight'], p['dur_predictor.linear.bias']) * keep[:, :, None]
        ret['dur'] = y.squeeze(-1)
    ret['mel2ph'] = mel2ph
    C = encoder_out.shape[-1]
    decoder_inp = F.pad(encoder_out, [0, 0, 1, 0])
    decoder_inp = torch.gather(decoder_inp, 1, mel2ph[..., None].repeat([1, 1, C]))
    tgt_nonpadding = (mel2ph > 0).float()[:, :, None]
    pitch_inp = (decoder_inp + 0 + spk_f0) * tgt_nonpadding
    if hp['use_pitch_embed']:
        decoder_inp = decoder_inp + add_pitch(p, hp, pitch_inp, f0, uv, mel2ph, ret, (encoder_out + 0 + spk_f0) * src_nonpadding)
    if hp.get('use_energy_embed'):
        decoder_inp = decoder_inp + add_energy(p, hp, pitch_inp, energy, ret)
    ret['decoder_inp'] = decoder_inp = (decoder_inp + spk) * tgt_nonpadding
    if skip_decoder:
        return ret
    x = fft_blocks(p, 'decoder.', decoder_inp, hp['dec_layers'], hp['num_heads'], hp['dec_ffn_kernel_size'], hp['ffn_act'])
    ret['decoder_out'] = x
    ret['mel_out'] = F.linear(x, p['mel_out.weight'], p['mel_out.bias']) * tgt_nonpadding
    return ret
