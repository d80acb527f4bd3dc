"""Developer tool (GPU): phase timeline of the persistent K-step loop from in-kernel s_memtime stamps.
`--split`: the labelled split-precision loop (k_loop_split; its weight stream by DSD_SPLIT_W) - layer phases only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
B, T, K = 8, 1024, 6
PHASES = None
CONV, TOUCH = 'winograd', -1
for a in list(sys.argv[1:]):
    if a.startswith('--k='):
        K = int(a[4:]); sys.argv.remove(a)
    elif a.startswith('--phases='):
        PHASES = tuple(int(v) for v in a[9:].split(',')); sys.argv.remove(a)
    elif a.startswith('--conv='):
        CONV = a[7:]; sys.argv.remove(a)
    elif a.startswith('--touch='):
        TOUCH = int(a[8:]); sys.argv.remove(a)
dev = torch.device('cuda', 0)
gd, pre = bench.build_model(dev)
g = torch.Generator(device=dev).manual_seed(1)
cond = torch.randn(B, T, 256, device=dev, generator=g).transpose(1, 2)
x = torch.randn(B, 80, T, device=dev, generator=g)
noise = torch.randn(K, B, 80, T, device=dev, generator=g)
eng = gd._engine(cond)
eng.set_loop_mode(1)
eng.set_conv_mode(CONV, TOUCH)
WINO = eng.loop_mode() == 1 and eng.conv_mode() == 1
SPLIT = '--split' in sys.argv
if SPLIT:
    sys.argv.remove('--split')
    eng.set_split_mode(True)
summary = {'phase_cycles': [], 'head_cycles': [], 'mfma_issue_ideal_per_phase': (2048 * 32 + 512 * 64) if WINO and not SPLIT else 2048 * 64,
           'conv': 'winograd F(2,3)' if WINO else 'direct'}
for phase in PHASES or ((43, 44, 63) if not SPLIT else (43, 44, 63, 83)):
    ts = eng.loop_timeline(x.clone(), noise, K, phase).astype(np.int64)
    d = np.diff(ts[:, :, :8], axis=2)
    names = ['weight prefetch issue, own columns of y, barrier', 'conv chunks 0-29 (centre taps) + flag poll, halo loads / writes, 2 barriers',
             'conv chunks 30-95', 'gate + barrier',
             'out-proj K=256', 'residual transpose, x\'', 'publish (drain, barrier, flag) + skip sum']
    if WINO and not SPLIT:
        names = ['weight prefetch issue, own frames of y (pair order), barrier', 'Winograd steps 0-39 of 128 (halo-free half: M1, M2) + flag poll, halo loads / writes, barrier',
                 'steps 40-127: rest of the halo-free half, output transform, M0 / M3 half + conditioner-projection loads', 'gate + barrier',
                 'out-proj K=256', 'residual, x\'', 'publish (drain, barrier, flag) + skip sum']
    if SPLIT:
        names = ['weight prefetch issue, own frames of y as planes, barrier', 'conv chunks 0-11 of 48 (centre taps) + flag poll, halo loads / plane writes, barrier',
                 'conv chunks 12-47 + conditioner-projection loads', 'gate -> planes + barrier', 'out-proj 16 chunks', 'x\'',
                 'publish (drain, barrier, flag) + skip sum']
    print(f'phase {phase} (layer {phase % 20}): {ts.shape[0]} workgroups, shader-clock ticks')
    for i, n in enumerate(names):
        print('  %-76s: mean %8.0f  min %8.0f  max %8.0f' % (n, d[:, :, i].mean(), d[:, :, i].min(), d[:, :, i].max()))
    print('  phase total: mean %.0f ; start skew across workgroups %.0f' % ((ts[:, :, 7] - ts[:, :, 0]).mean(), ts[:, :, 0].max() - ts[:, :, 0].min()))
    summary['phase_cycles'].append(float((ts[:, :, 7] - ts[:, :, 0]).mean()))
    if SPLIT:
        # slots 8 / 9 and 10 / 11: (s_memtime, s_memrealtime at 100 MHz) at this phase and one evaluation (20 layers + a head) later
        d_clk, d_ref = (ts[:, :, 10] - ts[:, :, 8]).astype(np.float64), (ts[:, :, 11] - ts[:, :, 9]).astype(np.float64)
        ok = d_ref > 0
        if ok.any():
            ghz = d_clk[ok] / d_ref[ok] * 0.1
            print('  shader clock over the next evaluation: %.3f GHz (min %.3f, max %.3f); the evaluation: %.0f ticks = %.1f us'
                  % (ghz.mean(), ghz.min(), ghz.max(), d_clk[ok].mean(), d_ref[ok].mean() * 0.01))
            summary.setdefault('shader_clock_ghz', []).append(float(ghz.mean()))
        continue
    hd = ts[:, :, 8:16]
    hn = ['barrier behind the last layer + skip tile (bias, / sqrt(L), stage)', 'skip projection K=256 (2 row blocks / wave) + ReLU tile',
          'final-projection weight prefetch + barrier', 'final projection K=256 (waves 0-2)', 'sampler update: global reads, math, stores (waves 0-2)',
          'barrier', 'next input projection + halo publish']
    print('  head of evaluation %d:' % (phase // 20))
    for i, n in enumerate(hn):
        sel = slice(0, 3) if i in (3, 4) else slice(0, 4)        # wave 3 has no final-projection rows
        a0 = hd[:, sel, i] if i != 5 else hd[:, :3, i]
        dd = hd[:, sel, i + 1] - a0 if i != 5 else hd[:, :3, i + 1] - a0
        print('    %-66s: mean %8.0f  min %8.0f  max %8.0f' % (n, dd.mean(), dd.min(), dd.max()))
    print('    head total (last layer done -> next evaluation\'s x published): mean %.0f' % (hd[:, :, 7] - hd[:, :, 0]).mean())
    summary['head_cycles'].append(float((hd[:, :, 7] - hd[:, :, 0]).mean()))
print('timeouts', eng.loop_timeouts())
if len(sys.argv) > 1:
    import json
    from diffsinger_amd.build import binary_id, kernel_isa
    for kv in sys.argv[2:]:                              # round=<tag> ...
        k_, v_ = kv.split('=', 1)
        summary[k_] = v_
    kname = ('k_loop_wino<1, 4>' if WINO else 'k_loop<1>') if not SPLIT else None
    summary['kernel_tag'], summary['build_id'] = kname, binary_id()
    hits = {n: h for n, h in kernel_isa().items() if kname and n.startswith(kname)}
    if len(hits) == 1:
        (summary['kernel_isa_name'], summary['kernel_isa']), = hits.items()
    summary['phase_cycles_mean'] = sum(summary['phase_cycles']) / len(summary['phase_cycles'])
    summary['head_cycles_mean'] = sum(summary['head_cycles']) / len(summary['head_cycles'])
    summary['shape'] = {'B': B, 'T': T, 'layers': 20}
    json.dump(summary, open(sys.argv[1], 'w'), indent=1)
