"""Developer tool (GPU): how the K=100 DDPM hot path behaves OFF the chip-filling shape of the bench line (8 x 1024 frames =
exactly one 32-frame tile per CU).  Shapes of the reference's own configurations: one utterance per device at inference
(configs/tts/fs2.yaml:70 max_eval_sentences: 1), LJSpeech max_frames 1550 (configs/tts/base.yaml:35-39), singing phrases of
5000-8000 frames (configs/singing/base.yaml:20).  One JSON line per (B, T):

    python tools/shape_sweep.py [reps] > profiles/rNN_shape_sweep.jsonl

A pass = dsd_prepare (hoisted conditioner projection) + the 100-step loop (noise drawn in the kernel) + denorm."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import diffsinger_amd
from diffsinger_amd import hparams
from diffsinger_amd.synth import presets

F_EXEC = 21_053_440            # direct convolution (per-layer / latency kernels, k_loop)
F_EXEC_WINO = 15_810_560       # Winograd F(2,3) convolution (k_loop_wino, the default of the persistent path)
PEAK_TF = 157.3
SHAPES = [(1, 512), (1, 1550), (4, 777), (8, 1000), (8, 1024), (5, 1550), (3, 5000), (2, 8000), (16, 2048)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    shapes = SHAPES
    if len(sys.argv) > 2 and not sys.argv[2].startswith('--'):
        shapes = [tuple(int(v) for v in s.split('x')) for s in sys.argv[2].split(',')]
    pre = presets()['lj_ds_beta6']
    hparams.clear()
    diffsinger_amd.use_preset('lj_ds_beta6')
    torch.manual_seed(1234)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=100, loss_type='l1',
                                          spec_min=pre['spec_min'], spec_max=pre['spec_max']).cuda().eval()
    dev = torch.device('cuda', 0)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    g = torch.Generator(device=dev).manual_seed(7)
    modes = [2] if '--default-only' in sys.argv else [2, 1, 0]       # automatic (the product), then forced persistent loop / per-layer kernels
    for B, T in shapes:
        conds = [torch.randn(B, T, 256, device=dev, generator=g).transpose(1, 2) for _ in range(2)]
        x_T = torch.randn(B, 1, 80, T, device=dev, generator=g)
        k = [0]

        def one():
            k[0] += 1
            return gd.inference(conds[k[0] & 1], x_T=x_T, K_step=100, pndm_speedup=0, noise_seed=5)

        row = {'B': B, 'T': T, 'K': 100, 'tiles32': B * ((T + 31) // 32), 'tiles_per_utt': (T + 31) // 32, 'n_cu': n_cu}
        for mode in modes:
            eng = gd.denoise_fn.engine()
            eng.set_loop_mode(mode)
            out = one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = one()
            torch.cuda.synchronize()
            sec = (time.perf_counter() - t0) / reps
            assert bool(torch.isfinite(out).all()) and eng.loop_timeouts() == 0
            wino = eng.loop_mode() == 1 and eng.conv_mode() == 1
            tf = B * T * 100 * (F_EXEC_WINO if wino else F_EXEC) / sec / 1e12
            path = (f'latency G={eng.lat_split()}' if eng.lat_split()
                    else (('persistent, Winograd conv' if wino else 'persistent, direct conv') if eng.loop_mode() == 1 else f'per-layer tile {eng.layer_tile()}'))
            if mode == 2:
                row.update({'path': path, 'ms_per_pass': round(sec * 1e3, 3), 'mel_frames_per_s': round(B * T / sec, 1),
                            'tflops_executed': round(tf, 2), 'frac_fp32_mfma_peak': round(tf / PEAK_TF, 4)})
            else:
                row['forced_' + ('persistent' if mode == 1 else 'per_layer')] = {'path': path, 'ms_per_pass': round(sec * 1e3, 3),
                                                                                'frac_fp32_mfma_peak': round(tf / PEAK_TF, 4)}
        gd.denoise_fn.engine().set_loop_mode(2)
        if '--lat-splits' in sys.argv:
            # forced row splits of the latency kernels (also with more workgroups than CUs: they are ordinary launches)
            eng = gd.denoise_fn.engine()
            row['forced_lat'] = {}
            for G in (2, 4, 8):
                eng.set_loop_mode(3)
                eng.set_lat_split(G)
                out = one()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    out = one()
                torch.cuda.synchronize()
                row['forced_lat'][f'G={eng.lat_split()}'] = round((time.perf_counter() - t0) / reps * 1e3, 3)
            eng.set_lat_split(-1)
            eng.set_loop_mode(2)
        print(json.dumps(row), flush=True)
        del conds, x_T


if __name__ == '__main__':
    main()
