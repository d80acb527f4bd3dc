"""Developer tool (GPU): time of the HIP FastSpeech2 / FastSpeech2MIDI forward at the bench shape of the
diffusion loop (8 utterances, 128 phones x 8 frames = 1024 mel frames each), teacher-forced durations / pitch, next to the
K-step loop it feeds.  One JSON line per preset.      python tools/bench_fs2.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import diffsinger_amd
from diffsinger_amd import hparams


def flops_per_frame(hp, T):
    H, L, k = hp['hidden_size'], hp['dec_layers'], hp['dec_ffn_kernel_size']
    layer = 2 * 3 * H * H + 2 * H * H + 4 * T * H + 2 * 4 * H * H * k + 2 * 4 * H * H
    f = L * layer + 2 * H * 80
    if hp['use_pitch_embed']:
        f += hp['predictor_layers'] * 2 * H * H * hp['predictor_kernel']
    return f


def run(preset, reps, B=8, T_txt=128, frames_per_phone=8):
    hparams.clear()
    diffsinger_amd.use_preset(preset)
    from diffsinger_amd import fs2
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    cls = fs2.FastSpeech2MIDI if hparams.get('use_midi') else fs2.FastSpeech2
    m = cls(63, 80).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(7)
    tok = torch.randint(1, 63, (B, T_txt), device=dev, generator=g)
    T = T_txt * frames_per_phone
    mel2ph = (torch.arange(T, device=dev) // frames_per_phone + 1)[None].repeat(B, 1)
    kw = dict(mel2ph=mel2ph, f0=torch.rand(B, T, device=dev, generator=g) * 2 + 6.5, uv=torch.zeros(B, T, device=dev))
    if hparams.get('use_midi'):
        kw.update(pitch_midi=torch.randint(40, 80, (B, T_txt), device=dev, generator=g), midi_dur=torch.rand(B, T_txt, device=dev, generator=g),
                  is_slur=torch.randint(0, 2, (B, T_txt), device=dev, generator=g))

    def fwd():
        return m(tok, infer=True, **{k: (v.clone() if k == 'f0' else v) for k, v in kw.items()})

    r = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fwd()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    assert bool(torch.isfinite(r['mel_out']).all())
    f = flops_per_frame(hparams, T)
    print(json.dumps({'preset': preset, 'model': cls.__name__, 'B': B, 'T_txt': T_txt, 'T_mel': T, 'ms_per_forward': sec * 1e3,
                      'mel_frames_per_s': B * T / sec, 'decoder_side_flop_per_frame': f, 'tflops': B * T * f / sec / 1e12,
                      'note': 'wall time incl. the torch index plumbing and ~70 kernel launches from Python (eager, no graph)'}), flush=True)


if __name__ == '__main__':
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for p in ('lj_ds_beta6', 'popcs_ds_beta6', 'opencpop_ds60_rel', 'opencpop_ds1000'):
        run(p, reps)
