"""Developer tool: per-position timing of the sampling graph from a rocprofv3 --kernel-trace csv (kernel_trace.csv):
orders the dispatches of the LAST hipGraph replay by start time and averages, over the K steps, the duration of each
node position (layer 0..L-1, head) and the idle gap in front of it.

    python tools/seq_summary.py <kernel_trace.csv> [L=20] [K=100]"""
import csv
import sys


def main(path, L=20, K=100):
    rows = []
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    per = L + 1
    # the last replay: find the last run of K*(L+1) consecutive k_layer/k_head dispatches
    idx = [i for i, r in enumerate(rows) if 'k_layer' in r[2] or 'k_head' in r[2]]
    seq = idx[-K * per:]
    assert len(seq) == K * per, len(seq)
    assert 'k_head' in rows[seq[-1]][2], rows[seq[-1]][2]
    dur = [[] for _ in range(per)]
    gap = [[] for _ in range(per)]
    for n, i in enumerate(seq):
        pos = n % per
        s, e, name = rows[i]
        dur[pos].append((e - s) / 1e3)
        if n > 0:
            gap[pos].append((s - rows[seq[n - 1]][1]) / 1e3)
    print(f'# {path}: last graph replay, {K} steps x {per} nodes; microseconds')
    print(f'{"pos":<10}{"kernel":<40}{"dur_avg":>9}{"dur_min":>9}{"dur_max":>9}{"gap_avg":>9}{"gap_max":>9}')
    tot = 0.0
    for pos in range(per):
        name = rows[seq[pos]][2].split('(')[0].replace('dsd::', '')
        d, g = dur[pos], gap[pos] or [0.0]
        tot += sum(d) / len(d) + sum(g) / len(g)
        print(f'{("layer %d" % pos) if pos < L else "head":<10}{name:<40}{sum(d) / len(d):>9.2f}{min(d):>9.2f}{max(d):>9.2f}{sum(g) / len(g):>9.2f}{max(g):>9.2f}')
    first, last = rows[seq[0]][0], rows[seq[-1]][1]
    print(f'# one step = {tot:.1f} us (sum of avg durations + gaps); whole replay {(last - first) / 1e6:.3f} ms')


if __name__ == '__main__':
    main(sys.argv[1], *(int(a) for a in sys.argv[2:4]))
