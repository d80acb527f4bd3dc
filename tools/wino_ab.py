"""Developer tool (GPU): A/B of the persistent loop's convolution forms on BASELINE configs[1] (8 x 1024, K = 100 DDPM): the direct K = 768 form
(k_loop) against the Winograd F(2,3) form (k_loop_wino) over the lead of the L2 touch of its weight stream.  One JSON line
per variant: ms per sampling call (HIP events, 3 calls), mel-frames/s of the loop alone, executed TFLOP/s and the fraction of the fp32 MFMA
peak, max-abs difference of the normalised x against the direct form."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B, T, K = 8, 1024, 100
shapes = [(8, 1024)]
if '--shapes' in sys.argv:
    shapes = [(8, 1024), (16, 2048), (3, 1550), (1, 5000), (5, 1024)]
dev = torch.device('cuda', 0)
gd, pre = bench.build_model(dev)
for (B, T) in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    cond = torch.randn(B, T, 256, device=dev, generator=g).transpose(1, 2)
    x = torch.randn(B, 80, T, device=dev, generator=g)
    noise = torch.randn(K, B, 80, T, device=dev, generator=g)
    eng = gd._engine(cond)
    eng.set_loop_mode(1)
    ref = None
    touches = next((tuple(int(v) for v in a[8:].split(',')) for a in sys.argv if a.startswith('--touch=')), (16, 0, 8, 32))
    variants = [('direct', -1)] + [('winograd', t) for t in touches]
    if len(shapes) > 1:
        variants = [('direct', -1), ('winograd', -1)]
    for conv, touch in variants:
        eng.set_conv_mode(conv, touch)
        eng.prepare(cond)
        xs = x.clone()
        eng.sample_ddpm(xs, noise, K)
        out = xs.clone()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(3):
            xs = x.clone()
            eng.sample_ddpm(xs, noise, K)
        ev1.record(); ev1.synchronize()
        ms = ev0.elapsed_time(ev1) / 3
        assert eng.loop_timeouts() == 0 and bool(torch.isfinite(out).all())
        if ref is None:
            ref = out
        wino = eng.loop_mode() == 1 and eng.conv_mode() == 1
        f = bench.F_EVAL_EXEC_WINO if wino else bench.F_EVAL_EXEC
        tf = B * T * K * f / (ms * 1e-3) / 1e12
        print(json.dumps({'shape': [B, T], 'conv': conv, 'touch': touch, 'kernel': 'k_loop_wino' if wino else 'k_loop', 'launches': eng.loop_launches(),
                          'ms_per_call': round(ms, 3), 'mel_frames_per_s': round(B * T / ms * 1e3, 1), 'tflops_executed': round(tf, 2),
                          'frac_fp32_mfma_peak': round(tf / bench.PEAK_FP32_MFMA_TFLOPS, 4), 'max_abs_x_vs_direct': float((out - ref).abs().max())}), flush=True)
    eng.set_conv_mode('winograd', 32)
