"""Developer tool (CPU): keep the results of a GPU call - gpurun_out/<tag>/* -> profiles/<tag>_* (text / json summaries only), and the
summaries bench.py reads (`evidence()`: they carry the build id + the kernel's device-code hash) to their fixed names under profiles/.

    python tools/keep.py r6_07 [file ...]        # default: every *.txt / *.json / *.jsonl of the call"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXED = ('loop_pmc.json', 'loop_timeline.json', 'voc_chain_32ch_pmc.json', 'voc_chain_16ch_pmc.json', 'voc_chain_8ch_pmc.json', 'fs2_ffn1_pmc.json',
         'fs2_attn_pmc.json', 'train_wgrad_pmc.json', 'train_trb_fused_w_pmc.json', 'train_stack_fwd_w_pmc.json')
SKIP = ('.log', '.err', '.csv', '.db')


def main(tag, *only):
    G, P = os.path.join(ROOT, 'gpurun_out', tag), os.path.join(ROOT, 'profiles')
    kept = []
    for f in sorted(os.listdir(G)):
        src = os.path.join(G, f)
        if not os.path.isfile(src) or f.endswith(SKIP) or f in ('run.txt',) or (only and f not in only) or os.path.getsize(src) == 0:
            continue
        shutil.copy(src, os.path.join(P, f'{tag}_{f}'))
        kept.append(f)
        if f in FIXED:
            shutil.copy(src, os.path.join(P, f))
    print(f'{len(kept)} files -> profiles/{tag}_*: ' + ' '.join(kept))


if __name__ == '__main__':
    main(*sys.argv[1:])
