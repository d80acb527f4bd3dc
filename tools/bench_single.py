"""Developer tool (GPU): the reference's own inference shape - ONE utterance per device (configs/tts/fs2.yaml:70 max_eval_sentences: 1) -
through the three stages: FastSpeech2 (teacher-forced), the diffusion loop, the HiFi-GAN generator.  Eager and as hipGraph replays.

    python tools/bench_single.py [reps] [T_txt] [frames_per_phone] [--conv-ab]  > profiles/rNN_single_utterance.jsonl"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import diffsinger_amd
from diffsinger_amd import hparams
from diffsinger_amd.graphs import GraphedForward
from diffsinger_amd.synth import presets


def timeit(f, reps):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    reps = int(args[0]) if len(args) > 0 else 10
    T_txt = int(args[1]) if len(args) > 1 else 100
    fpp = int(args[2]) if len(args) > 2 else 8
    dev = torch.device('cuda', 0)
    B, T = 1, T_txt * fpp
    m, hp, tok, kw = bench._fs2_setup(B, T_txt, fpp, dev)
    m = m.to(dev)
    tok = tok.to(dev)
    kw = {k: v.to(dev) for k, v in kw.items()}
    from diffsinger_amd import fs2
    row = {'shape': f'1 utterance, {T_txt} phones, {T} mel frames', 'preset': bench.PRESET}
    fwd = lambda: m(tok, infer=True, **{k: (v.clone() if k == 'f0' else v) for k, v in kw.items()})
    row['fs2_eager_ms'] = timeit(fwd, reps)
    gm = GraphedForward(lambda tok_, mel2ph, f0, uv: m(tok_, infer=True, mel2ph=mel2ph, f0=f0, uv=uv))
    row['fs2_graph_ms'] = timeit(lambda: gm(tok, kw['mel2ph'], kw['f0'], kw['uv']), reps)
    if '--conv-ab' in sys.argv:
        for mode in (0, 1):
            fs2.set_conv_split(mode)
            row[f'fs2_eager_ms_conv_split_{mode}'] = timeit(fwd, reps)
        fs2.set_conv_split(-1)
        # the bench shape too: 8 x 128 phones x 8 frames
        m8, _, tok8, kw8 = bench._fs2_setup(8, 128, 8, dev)
        m8 = m8.to(dev)
        tok8 = tok8.to(dev)
        kw8 = {k: v.to(dev) for k, v in kw8.items()}
        f8 = lambda: m8(tok8, infer=True, **{k: (v.clone() if k == 'f0' else v) for k, v in kw8.items()})
        for mode in (-1, 0, 1):
            fs2.set_conv_split(mode)
            row[f'fs2_8x1024_ms_conv_split_{mode}'] = timeit(f8, reps)
        fs2.set_conv_split(-1)
    # diffusion loop on the conditioner output
    pre = presets()[bench.PRESET]
    hparams.clear()
    diffsinger_amd.use_preset(bench.PRESET)
    torch.manual_seed(1234)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=100, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(7)
    cond = torch.randn(B, T, 256, device=dev, generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, device=dev, generator=g)
    row['diffusion_k100_ddpm_ms'] = timeit(lambda: gd.inference(cond, x_T=x_T, K_step=100, pndm_speedup=0, noise_seed=5), max(2, reps // 3))
    row['diffusion_path'] = f'latency G={gd.denoise_fn.engine().lat_split()}' if gd.denoise_fn.engine().lat_split() else 'persistent / per-layer'
    # vocoder
    from diffsinger_amd.vocoder import HifiGanGenerator
    v = HifiGanGenerator(bench.VOC_CONFIG)
    v.remove_weight_norm()
    gg = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for n, p in v.named_parameters():
            if n.endswith('weight'):
                p.copy_(torch.randn(p.shape, generator=gg) / (p[0].numel() if not n.startswith('ups') else p.shape[0] * 2) ** 0.5)
    v = v.to(dev).eval()
    mel = torch.randn(B, 80, T, device=dev, generator=g)
    row['vocoder_eager_ms'] = timeit(lambda: v(mel), reps)
    gv = GraphedForward(v)
    row['vocoder_graph_ms'] = timeit(lambda: gv(mel), reps)
    row['total_graphs_ms'] = row['fs2_graph_ms'] + row['diffusion_k100_ddpm_ms'] + row['vocoder_graph_ms']
    row['audio_seconds'] = T * 256 / 24000
    print(json.dumps({k: (round(val, 3) if isinstance(val, float) else val) for k, val in row.items()}), flush=True)


if __name__ == '__main__':
    main()
