"""Extract the hot-path hyper-parameters + spec_min/spec_max statistics of the reference's shipped configs
into diffsinger_amd/presets.json (data only).  Run in the build container: python -m oracle.make_presets
Each config is resolved by the reference's own loader (utils/hparams.py:23-122) in a fresh process."""
import json, os, subprocess, sys

CONFIGS = {
    'lj_ds_beta6': 'usr/configs/lj_ds_beta6.yaml',                       # DiffSpeech, LJSpeech
    'popcs_ds_beta6': 'usr/configs/popcs_ds_beta6.yaml',                 # DiffSinger, PopCS
    'opencpop_ds60_rel': 'usr/configs/midi/cascade/opencs/ds60_rel.yaml',  # DiffSinger cascade, shallow K=60
    'opencpop_ds1000': 'usr/configs/midi/e2e/opencpop/ds1000.yaml',      # DiffSinger e2e + PNDM
}
KEYS = ['audio_num_mel_bins', 'keep_bins', 'hidden_size', 'residual_layers', 'residual_channels',
        'dilation_cycle_length', 'timesteps', 'K_step', 'max_beta', 'schedule_type', 'diff_loss_type',
        'diff_decoder_type', 'pndm_speedup', 'gaussian_start', 'use_midi', 'spec_min', 'spec_max',
        # FastSpeech2 / FastSpeech2MIDI conditioner: modules/fastspeech/fs2.py, tts_modules.py
        'enc_layers', 'dec_layers', 'enc_ffn_kernel_size', 'dec_ffn_kernel_size', 'num_heads', 'ffn_act', 'ffn_padding',
        'use_pos_embed', 'rel_pos', 'encoder_type', 'decoder_type', 'dropout', 'use_pitch_embed', 'pitch_type', 'use_uv',
        'pitch_norm', 'f0_mean', 'f0_std', 'pitch_ar', 'predictor_hidden', 'predictor_layers', 'predictor_kernel',
        'predictor_dropout', 'predictor_grad', 'dur_predictor_layers', 'dur_predictor_kernel', 'dur_loss', 'cwt_hidden_size',
        'cwt_std_scale', 'use_energy_embed', 'use_spk_id', 'use_spk_embed', 'use_split_spk_id', 'num_spk']

CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from oracle.ref_driver import Reference
r = Reference(sys.argv[1])
print('@@' + json.dumps({k: r.hparams.get(k) for k in %r}))
'''

def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for name, cfg in CONFIGS.items():
        res = subprocess.run([sys.executable, '-c', CHILD % (root, KEYS), cfg], capture_output=True, text=True, check=True)
        line = [l for l in res.stdout.splitlines() if l.startswith('@@')][-1]
        out[name] = dict(json.loads(line[2:]), source=cfg)
    path = os.path.join(root, 'diffsinger_amd', 'presets.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', path)

if __name__ == '__main__':
    main()
