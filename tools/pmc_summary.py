"""Developer tool: reduce rocprofv3 --pmc passes (csv `*counter_collection.csv` or rocpd sqlite) over ONE kernel to
per-dispatch averages, and derive the HBM traffic per launch for bench.py's `roofline.traffic`.

    python tools/pmc_summary.py <dir with the pass outputs> <kernel substring> <out.txt> <out.json> [key=value ...] [min_us=<dur>]

Units / corrections (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots"): FETCH_SIZE and WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) coalesced reads at 64 bytes, i.e. reports half the
bytes - every load of k_layer is a dwordx4 load, so it is doubled.  WRITE_SIZE is taken as reported (it equals the
algorithmic x_out + skip bytes exactly).  Infinity-Cache hits are counted by both (they are fabric-side request counters)."""
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def from_csv(root, kern, min_us=0.0):
    acc = defaultdict(list)
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                if kern not in row.get('Kernel_Name', ''):
                    continue
                if min_us > 0:          # one shape of a kernel that is launched with many: keep the dispatches at least this long
                    try:
                        if (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3 < min_us:
                            continue
                    except (KeyError, ValueError):
                        continue
                name = row['Counter_Name']
                acc[name].append(float(row['Counter_Value']))
                try:
                    dur[name].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
                except (KeyError, ValueError):
                    pass
    return acc, dur


def from_db(root, kern):
    acc = defaultdict(list)
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(root, '**', '*.db'), recursive=True):
        con = sqlite3.connect(path)
        cur = con.cursor()
        names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        cand = [n for n in names if n == 'counters_collection'] or [n for n in names if 'counter' in n.lower() or 'pmc' in n.lower()]
        for tab in cand:
            cols = [r[1] for r in cur.execute(f'pragma table_info("{tab}")')]
            lc = {c.lower(): c for c in cols}
            kcol = next((lc[c] for c in ('kernel_name', 'name') if c in lc), None)
            ccol = next((lc[c] for c in ('counter_name', 'pmc_name', 'symbol') if c in lc), None)
            vcol = next((lc[c] for c in ('value', 'counter_value') if c in lc), None)
            if not (kcol and ccol and vcol):
                continue
            scol, ecol = lc.get('start'), lc.get('end')
            q = f'select "{kcol}", "{ccol}", "{vcol}"' + (f', "{scol}", "{ecol}"' if scol and ecol else '') + f' from "{tab}"'
            try:
                rows = cur.execute(q).fetchall()
            except sqlite3.Error:
                continue
            # one row per (dispatch, counter, [xcc/dimension]): sum the dimensions of one dispatch
            per = defaultdict(float)
            seen = {}
            idcol = next((lc[c] for c in ('dispatch_id', 'id') if c in lc), None)
            if idcol:
                rows2 = cur.execute(f'select "{idcol}", "{kcol}", "{ccol}", "{vcol}"' + (f', "{scol}", "{ecol}"' if scol and ecol else '') + f' from "{tab}"').fetchall()
                for r in rows2:
                    if kern not in (r[1] or ''):
                        continue
                    per[(r[0], r[2])] += float(r[3])
                    if len(r) > 5 and r[4] is not None:
                        seen[(r[0], r[2])] = (r[5] - r[4]) / 1e3
                for (did, cname), v in per.items():
                    acc[cname].append(v)
                    if (did, cname) in seen:
                        dur[cname].append(seen[(did, cname)])
            else:
                for r in rows:
                    if kern in (r[0] or ''):
                        acc[r[1]].append(float(r[2]))
            if acc:
                break
        con.close()
    return acc, dur


def observed_kernels(root, kern):
    """The kernel names (as rocprofv3 wrote them) of the dispatches the counters were averaged over."""
    names = set()
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                if kern in row.get('Kernel_Name', ''):
                    names.add(row['Kernel_Name'])
    return names


def provenance(kern, tag=None, observed=()):
    """Which binary the counters belong to: the build id of the library in this tree (dsd_build_id: sha256 of csrc/ + include/) and the hash
    of THE KERNEL'S device code (diffsinger_amd/kernel_isa.json, written by the build) - bench.py refuses a summary that matches neither the
    library it has loaded nor that kernel in it."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from diffsinger_amd.build import binary_id, kernel_isa
    out = {'build_id': binary_id()}
    hits = {n: h for n, h in kernel_isa().items() if kern in n}
    if len(hits) > 1 and tag:
        # 'k_loop_wino' names two instantiations; the kernel_tag of the summary ('k_loop_wino<1, 4>', possibly followed by a note in
        # parentheses) names the one that ran - r6_16's loop_pmc.json went unstamped without this and the next rebuild orphaned it
        t = tag.split(' (')[0]
        hits = {n: h for n, h in hits.items() if n.startswith(t)}
    if len(hits) > 1 and observed:
        # 'k_voc_chain<32' names three instantiations (two tiles / one tile in place / merged groups); the one that RAN is in the counter csv
        # (round 6: the merged chain launches went unstamped and any later rebuild would have orphaned their PMC figures)
        norm = lambda x: x.replace('void ', '').replace('dsd::', '').replace(' ', '')
        obs = {norm(o) for o in observed}
        ran = {n: h for n, h in hits.items() if any(norm(n).split('(')[0] in o for o in obs)}
        if ran:
            hits = ran
    if len(hits) == 1:
        (out['kernel_isa_name'], out['kernel_isa']), = hits.items()
    return out


def main(root, kern, out_txt, out_json, *extras):
    min_us = 0.0
    for kv in extras:
        if kv.startswith('min_us='):
            min_us = float(kv.split('=', 1)[1])
    acc, dur = from_csv(root, kern, min_us)
    src = 'csv'
    if not acc:
        acc, dur = from_db(root, kern)
        src = 'rocpd sqlite'
    if not acc:
        raise SystemExit(f'no counters for kernel "{kern}" under {root}')
    avg = {k: sum(v) / len(v) for k, v in acc.items()}
    lines = [f'# rocprofv3 --pmc passes over {kern} ({src}), averages per dispatch; tools/gpu_pmc.sh',
             '# FETCH_SIZE / WRITE_SIZE in KiB as reported (FETCH_SIZE under-reports wide reads by 2x on gfx950, MI355X_MICROARCH.md',
             '# section HBM); SQ_* wave counters in quad-cycles summed over all waves, SQ_VALU_MFMA_BUSY_CYCLES in cycles']
    for k in sorted(avg):
        d = dur.get(k)
        extra = f'   (n={len(acc[k])}' + (f', avg kernel duration under this pass {sum(d) / len(d):.1f} us)' if d else ')')
        lines.append(f'{k:<36}{avg[k]:>16.1f}{extra}')
    js = {'kernel': kern, 'source': src, 'counters': avg, 'n': {k: len(v) for k, v in acc.items()}}
    for kv in extras:                       # e.g. frames=8192 kernel_tag='k_layer<1,false>' round=r01b
        k, v = kv.split('=', 1)
        js[k] = int(v) if v.isdigit() else v
    js.update(provenance(kern, js.get('kernel_tag') if isinstance(js.get('kernel_tag'), str) else None, observed_kernels(root, kern) if src == 'csv' else ()))
    if min_us > 0:
        lines.insert(1, f'# only dispatches of at least {min_us:g} us (one shape of the kernel)')
    if 'FETCH_SIZE' in avg and 'WRITE_SIZE' in avg:
        fetch = avg['FETCH_SIZE'] * 1024 * 2
        write = avg['WRITE_SIZE'] * 1024
        js['hbm_bytes_per_launch'] = {'fetch_corrected': fetch, 'write': write, 'total': fetch + write,
                                      'note': 'FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024; '
                                              'fabric-side counters: Infinity-Cache hits included'}
        lines.append(f'# HBM-side traffic per launch: fetch {fetch / 1e6:.1f} MB (corrected x2) + write {write / 1e6:.1f} MB = {(fetch + write) / 1e6:.1f} MB')
    open(out_txt, 'w').write('\n'.join(lines) + '\n')
    json.dump(js, open(out_json, 'w'), indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    main(*sys.argv[1:])
