"""Developer tool (GPU): time of the HIP HiFi-GAN / NSF-HiFi-GAN generator on the output of the bench shape
of the diffusion loop (8 utterances x 1024 mel frames -> 8 x 262144 samples), next to a reference-style PyTorch-ROCm eager
generator (MIOpen convolutions + ATen element-wise ops) with the same weights, plus the per-launch time of one dilated resblock
convolution of every stage against its HBM / MFMA roofline.  JSON lines.      python tools/bench_vocoder.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from diffsinger_amd import _lib
from diffsinger_amd.vocoder import HifiGanGenerator, _HipOps, fold_weight, get_padding, padded_samples

CONFIG = dict(resblock='1', upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=128,
              resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], audio_sample_rate=24000)


def torch_generator(sd, h, x):
    """The reference's forward (modules/hifigan/hifigan.py:144-169, ResBlock1 :54-61) written with torch functional ops - what the
    reference runs on a GPU: one library convolution + separate element-wise kernels per layer.  Plain (no NSF) path."""
    nk = len(h['resblock_kernel_sizes'])
    x = F.conv1d(x, sd['conv_pre.weight'], sd['conv_pre.bias'], padding=3)
    for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
        x = F.conv_transpose1d(F.leaky_relu(x, 0.1), sd[f'ups.{i}.weight'], sd[f'ups.{i}.bias'], stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            y = x
            kk = h['resblock_kernel_sizes'][j]
            for q, d in enumerate(h['resblock_dilation_sizes'][j]):
                pre = f'resblocks.{i * nk + j}.'
                t = F.conv1d(F.leaky_relu(y, 0.1), sd[pre + f'convs1.{q}.weight'], sd[pre + f'convs1.{q}.bias'], padding=get_padding(kk, d), dilation=d)
                t = F.conv1d(F.leaky_relu(t, 0.1), sd[pre + f'convs2.{q}.weight'], sd[pre + f'convs2.{q}.bias'], padding=get_padding(kk, 1))
                y = t + y
            xs = y if xs is None else xs + y
        x = xs / nk
    return torch.tanh(F.conv1d(F.leaky_relu(x), sd['conv_post.weight'], sd['conv_post.bias'], padding=3))


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def flops_per_frame(h):
    c0, f, rate = h['upsample_initial_channel'], 2 * 80 * h['upsample_initial_channel'] * 7, 1
    for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
        ch = c0 // 2 ** (i + 1)
        rate *= u
        f += rate * 2 * (2 * ch) * ch * (k // u)
        for kk, dd in zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes']):
            f += rate * len(dd) * 2 * 2 * ch * ch * kk
    return f + rate * 2 * ch * 7


def main():
    """Order: everything on the HIP kernels first (each line is flushed at once), the PyTorch-ROCm eager baseline LAST - on a fresh box its
    first call pays MIOpen's kernel search for ~40 convolution shapes, which can take longer than the rest of this tool together."""
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device('cuda', 0)
    B, T = 8, 1024
    torch.manual_seed(1)
    lib = _lib.load()
    plain = None
    for nsf in (False, True):
        h = dict(CONFIG, use_pitch_embed=nsf)
        m = HifiGanGenerator(h)
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            for n, p in m.named_parameters():                         # fan-in scaled weights: a live signal path
                if n.endswith('weight_v'):
                    fan = p[0].numel() if not n.startswith('ups') else p.shape[0] * 2
                    p.copy_(torch.randn(p.shape, generator=g) / fan ** 0.5)
                    getattr(m.get_submodule(n.rsplit('.', 1)[0]), 'weight_g').copy_(p.flatten(1).norm(dim=1).reshape(-1, 1, 1))
        m.remove_weight_norm()
        m = m.to(dev).eval()
        mel = torch.randn(B, 80, T, device=dev)
        f0 = (torch.rand(B, T, device=dev) * 300 + 80) if nsf else None
        ri, nz = (torch.rand(B, 9, device=dev), torch.randn(B, T * 256, 9, device=dev)) if nsf else (None, None)

        def run(fold):                                                # narrow stages on the folded kernel (1, default) or on k_voc_conv (0)
            lib.dsv_set_fold(fold)
            fixed = m(mel, f0, rand_ini=ri, noise=nz)
            sec_, wav_ = timed(lambda: m(mel, f0), reps)
            return sec_, wav_, fixed

        sec0, _, wav0 = run(0)
        sec, wav, wav1 = run(1)
        assert wav.shape == (B, 1, T * 256) and bool(torch.isfinite(wav).all())
        print(json.dumps({'impl': 'HIP generator (dsv_conv1d & co.)', 'nsf': nsf, 'B': B, 'T_mel': T, 'samples': T * 256, 'ms_per_forward': sec * 1e3,
                          'ms_per_forward_narrow_layers_unfolded': sec0 * 1e3, 'max_abs_diff_folded_vs_unfolded': float((wav1 - wav0).abs().max()),
                          'mel_frames_per_s': B * T / sec, 'x_realtime_24k': B * T * 256 / 24000 / sec, 'flop_per_frame': flops_per_frame(h),
                          'tflops': B * T * flops_per_frame(h) / sec / 1e12}), flush=True)
        if not nsf:
            plain = (m, h, mel, wav)
    # one resblock convolution (kernel 11, dilation 5, leaky_relu in front, residual behind) per stage: bytes = read x + read residual + write
    ops = _HipOps()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def launch_ms(fn):
        fn()
        ev[0].record()
        for _ in range(20):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / 20

    rate = 1
    for i, u in enumerate(CONFIG['upsample_rates']):
        rate *= u
        ch = CONFIG['upsample_initial_channel'] // 2 ** (i + 1)
        L = T * rate
        x = torch.randn(B, ch, padded_samples(L), device=dev)
        x[:, :, L:] = 0
        b = torch.zeros(ch, device=dev)
        byt = 3 * B * ch * L * 4
        for kk, dd in ((11, 5), (11, 1), (3, 1)):
            wraw = torch.randn(ch, ch, kk, device=dev) / (ch * kk) ** 0.5
            w = ops.pack(wraw)
            pad = (kk - 1) * dd // 2
            ms = launch_ms(lambda: ops.conv(x, L, w, b, ch, ch, kk, pad, dd, pre_slope=0.1, residual=x))
            fl = 2 * B * L * ch * ch * kk
            rec = {'stage': i + 1, 'channels': ch, 'samples_per_utt': L, 'k': kk, 'dil': dd, 'k_voc_conv_ms': ms,
                   'k_voc_conv_GBps': byt / ms / 1e6, 'k_voc_conv_tflops_useful': fl / ms / 1e9}
            F = ops.fold_factor(ch, ch, kk, dd)
            if F > 1:
                wf = ops.pack(fold_weight(wraw, F))
                ms = launch_ms(lambda: ops.conv_folded(x, L, wf, b, ch, ch, kk, F, dd, pre_slope=0.1, residual=x))
                rec.update({'fold': F, 'k_voc_conv_fold_ms': ms, 'k_voc_conv_fold_GBps': byt / ms / 1e6, 'frac_hbm_8TBps': byt / ms / 1e6 / 8000,
                            'k_voc_conv_fold_tflops_useful': fl / ms / 1e9})
            rec['note'] = 'per launch incl. the torch.empty of the output and the ctypes call (eager); bytes = input + residual + output'
            print(json.dumps(rec), flush=True)
    m, h, mel, wav = plain
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    with torch.no_grad():
        sec_t, wav_t = timed(lambda: torch_generator(sd, h, mel), max(2, reps // 2))
    print(json.dumps({'impl': 'reference-style PyTorch-ROCm eager generator (MIOpen + ATen)', 'nsf': False, 'B': B, 'T_mel': T,
                      'ms_per_forward': sec_t * 1e3, 'mel_frames_per_s': B * T / sec_t,
                      'max_abs_diff_hip_vs_torch_rocm': float((wav - wav_t).abs().max())}), flush=True)


if __name__ == '__main__':
    main()
