"""Developer tool: turn a rocprofv3 (rocpd sqlite) result into the per-kernel stats table committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r1/bench_results.db > profiles/<name>.txt
"""
import sqlite3
import sys


def short(name: str, n: int = 96) -> str:
    name = name.replace('void ', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def main(path: str):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f'# rocprofv3 --kernel-trace --stats summary of {path}')
    print(f'# durations in microseconds (kernel begin->end on the device); {sum(r[1] for r in rows)} dispatches, '
          f'{total / 1e3:.1f} us of kernel time in total')
    print(f'{"kernel":<98}{"calls":>7}{"total_us":>12}{"avg_us":>10}{"min_us":>10}{"max_us":>10}{"pct":>7}{"vgpr":>6}{"agpr":>6}{"sgpr":>6}{"lds":>8}{"grid":>8}{"wg":>5}')
    for name, calls, tot, avg, mn, mx, vg, ag, sg, lds, gx, wx in rows:
        print(f'{short(name):<98}{calls:>7}{tot / 1e3:>12.1f}{avg / 1e3:>10.2f}{mn / 1e3:>10.2f}{mx / 1e3:>10.2f}{100 * tot / total:>7.2f}'
              f'{vg or 0:>6}{ag or 0:>6}{sg or 0:>6}{lds or 0:>8}{gx or 0:>8}{wx or 0:>5}')


if __name__ == '__main__':
    main(sys.argv[1])
