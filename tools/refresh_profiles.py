"""Developer tool (CPU): copy the evidence of a `tools/gpu_evidence.sh <tag>` round from gpurun_out/<tag>/ to profiles/<tag>_* and refresh the
tracked summaries bench.py / the docs read: loop_pmc.json, loop_timeline.json, voc_chain_32ch_pmc.json, fs2_ffn1_pmc.json, roofline.json.

    python tools/refresh_profiles.py r57"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ['bench_n1.json', 'bench_n1_kernel_stats.txt', 'bench_row_fs2.json', 'bench_row_train.json', 'bench_row_vocoder.json', 'configs_throughput.jsonl',
         'hbm_probe.txt', 'loop_pmc.txt', 'loop_timeline.txt', 'mfma_probe4.txt', 'shape_sweep.jsonl', 'smoke.txt', 'voc_chain_16ch_pmc.txt',
         'voc_chain_32ch_pmc.txt', 'voc_chain_8ch_pmc.txt', 'voc_chain_timeline.txt', 'vocoder_kernel_stats.txt', 'pytest_gpu_full.txt',
         'single_utterance.jsonl', 'bench_pwg.jsonl', 'fs2_ffn1_pmc.txt', 'fs2_kernel_stats.txt']


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def busy(c):
    return c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8)


def main(tag):
    G, P = os.path.join(ROOT, 'gpurun_out', tag), os.path.join(ROOT, 'profiles')
    for f in FILES:
        if os.path.isfile(os.path.join(G, f)):
            shutil.copy(os.path.join(G, f), os.path.join(P, f'{tag}_{f}'))
    for f in ('loop_pmc.json', 'voc_chain_32ch_pmc.json', 'fs2_ffn1_pmc.json'):
        if os.path.isfile(os.path.join(G, f)):
            d = json.load(open(os.path.join(G, f)))
            d['round'] = tag
            json.dump(d, open(os.path.join(P, f), 'w'), indent=1)
    pm = json.load(open(os.path.join(P, 'loop_pmc.json')))
    tl = json.load(open(os.path.join(G, 'loop_timeline.json')))
    tl['round'], tl['commit'] = tag, pm.get('commit')
    json.dump(tl, open(os.path.join(P, 'loop_timeline.json'), 'w'), indent=1)
    b, voc, fs, tr = (last_json(os.path.join(G, f)) for f in ('bench_n1.json', 'bench_row_vocoder.json', 'bench_row_fs2.json', 'bench_row_train.json'))
    r = json.load(open(os.path.join(P, 'roofline.json')))
    c = pm['counters']
    traffic = c['FETCH_SIZE'] * 2048 + c['WRITE_SIZE'] * 1024
    bare = None
    for ln in open(os.path.join(G, 'mfma_probe4.txt')):
        if ln.startswith('A none'):
            bare = float(ln.split('->')[1].split('TFLOP')[0])
    k = r['k_loop']
    k['launch_ms'][f'{tag}_hip_events'] = b['roofline']['avg_launch_ms']
    k['achieved_tflops'][tag] = b['roofline']['achieved']
    k['frac_of_nominal'][tag] = b['roofline']['frac']
    if bare:
        k['frac_of_measured_bare_mfma_stream'][tag] = b['roofline']['achieved'] / bare
        r['measured']['fp32_mfma_tflops_bare_stream'] = bare
        k['headroom_note'] = (f'the bare MFMA stream tops out at {bare:.1f} TFLOP/s on this box (power-limited clock ~2.2 GHz under full fp32-MFMA load, '
                              f'{tag}_mfma_probe4.txt): k_loop runs at {b["roofline"]["achieved"] / bare:.3f} of that ceiling')
    k.update({'mfma_busy': busy(c), 'hbm_side_bytes_per_launch_pmc': traffic, 'traffic_vs_algorithmic': traffic / k['algorithmic_bytes_per_launch'],
              'traffic_vs_unavoidable': traffic / 7.66e9,
              'phase_cycles': {'layer_phase_mean': tl['phase_cycles_mean'], 'mfma_issue_ideal': 131072, 'head_mean': tl['head_cycles_mean']},
              'source': f'profiles/{tag}_bench_n1.json, {tag}_bench_n1_kernel_stats.txt, {tag}_loop_pmc.txt, {tag}_loop_timeline.txt, loop_pmc.json, loop_timeline.json'})
    kv = r['k_voc_chain']
    kv['row_ms_per_step'] = voc['ms_per_step']
    kv['stages']['32ch'].update({'launch_ms_events': voc['roofline']['avg_launch_ms'], 'useful_tflops': voc['roofline']['achieved'], 'frac_of_nominal': voc['roofline']['frac']})
    for ch in ('8', '16', '32'):
        f = os.path.join(G, f'voc_chain_{ch}ch_pmc.json')
        if os.path.isfile(f):
            d = json.load(open(f))
            cc = d['counters']
            st = kv['stages'][ch + 'ch']
            st.update({'kernel': d.get('kernel', st.get('kernel')), 'mfma_busy': busy(cc), 'hbm_side_bytes_per_launch_pmc': cc['FETCH_SIZE'] * 2048 + cc['WRITE_SIZE'] * 1024})
            st['traffic_vs_algorithmic'] = st['hbm_side_bytes_per_launch_pmc'] / st['algorithmic_bytes_per_launch']
    for ln in open(os.path.join(G, 'vocoder_kernel_stats.txt')):
        for ch in ('8', '16', '32'):
            if f'k_voc_chain<{ch},' in ln:
                kv['stages'][ch + 'ch']['launch_ms_rocprofv3'] = float(ln.split(')')[-1].split()[2]) / 1e3
                kv['stages'][ch + 'ch']['kernel'] = ln.split('(')[0].replace('dsd::', '').strip()
    kv['source'] = f'profiles/{tag}_bench_row_vocoder.json, {tag}_vocoder_kernel_stats.txt, {tag}_voc_chain_{{8,16,32}}ch_pmc.txt, {tag}_voc_chain_timeline.txt'
    r['k_fs_conv'].update({'row_ms_per_step': fs['ms_per_step'], 'achieved_tflops': fs['roofline']['achieved'], 'frac_of_nominal': fs['roofline']['frac'],
                           'launch_ms_events': fs['roofline']['avg_launch_ms'], 'hbm_side_bytes_per_launch_pmc': fs['roofline'].get('traffic'),
                           'source': f'profiles/{tag}_bench_row_fs2.json, fs2_ffn1_pmc.json, {tag}_fs2_kernel_stats.txt, r35_fs2_by_grid.txt'})
    r['k_tr_wgrad'].update({'row_ms_per_step': tr['ms_per_step'], 'achieved_tflops': tr['roofline']['achieved'], 'frac_of_nominal': tr['roofline']['frac'],
                            'launch_ms_events': tr['roofline']['avg_launch_ms'], 'source': f'profiles/{tag}_bench_row_train.json, train_wgrad_pmc.json, r38_train_by_grid.txt'})
    sw = [json.loads(ln) for ln in open(os.path.join(G, 'shape_sweep.jsonl'))]
    r['k_lat'].update({'ms_per_call': [s for s in sw if (s['B'], s['T']) == (1, 512)][0]['ms_per_pass'],
                       'shapes': {f"{s['B']}x{s['T']}": {'path': s['path'], 'ms': s['ms_per_pass'], 'frac': s['frac_fp32_mfma_peak']} for s in sw},
                       'source': f'profiles/{tag}_shape_sweep.jsonl, r18_latency_kernel_stats.txt, r19_lat_splits.jsonl, r47_midsize_paths.jsonl'})
    pw = [json.loads(ln) for ln in open(os.path.join(G, 'bench_pwg.jsonl'))]
    r['k_pwg_layer'] = {'what': 'ParallelWaveGAN generator: one launch per gated residual block (30 per forward), 8 x 1024 mel frames', 'bound': 'mfma',
                        'row_ms_per_step': pw[0]['ms_per_forward'], 'achieved_tflops': pw[0]['tflops'], 'frac_of_nominal': pw[0]['frac_fp32_mfma_peak'],
                        'source': f'profiles/{tag}_bench_pwg.jsonl'}
    json.dump(r, open(os.path.join(P, 'roofline.json'), 'w'), indent=1)
    print(f'{tag}: headline {b["value"]:.0f} frames/s, frac {b["roofline"]["frac"]:.4f}, MFMA busy {busy(c):.3f}, traffic x{traffic / k["algorithmic_bytes_per_launch"]:.2f}; '
          f'vocoder {voc["ms_per_step"]:.2f} ms, fs2 {fs["ms_per_step"]:.2f} ms, train {tr["ms_per_step"]:.2f} ms, pwg {pw[0]["ms_per_forward"]:.1f} ms')


if __name__ == '__main__':
    main(sys.argv[1])
