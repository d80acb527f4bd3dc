"""Developer tool: rocprofv3 --kernel-trace csv -> per (kernel, grid size) call counts and durations, for kernels launched with many shapes.

    python tools/trace_by_grid.py <dir with *kernel_trace.csv> [kernel substring] > out.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, kern=''):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                name = row.get('Kernel_Name', '')
                if kern and kern not in name:
                    continue
                grid = tuple(int(row.get(k, 0) or 0) for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z'))
                if not any(grid):
                    grid = (int(row.get('Grid_Size', 0) or 0), 1, 1)
                acc[(name[:60], grid)].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    tot = sum(sum(v) for v in acc.values())
    print(f'# {len(rows)} (kernel, grid) groups, {tot:.1f} us of kernel time')
    print(f'{"kernel":<62}{"grid (threads)":<24}{"calls":>7}{"total_us":>12}{"avg_us":>10}{"min_us":>10}{"pct":>7}')
    for (name, grid), v in rows:
        print(f'{name:<62}{str(grid):<24}{len(v):>7}{sum(v):>12.1f}{sum(v) / len(v):>10.2f}{min(v):>10.2f}{100 * sum(v) / tot:>7.2f}')


if __name__ == '__main__':
    main(*sys.argv[1:])
