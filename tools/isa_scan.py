"""Developer tool (CPU): a static pass over the gfx950 code of every kernel in libdsdenoise.so - what round 5 looked for by hand after an
in-kernel timeline had shown a kernel running at half the speed of an identical instruction stream elsewhere (profiles/r5_25_trb_timeline.txt).
Per kernel with a matrix loop, for its DENSEST loop (most MFMAs per instruction):
    MFMAs, vector-ALU instructions per MFMA (beside an fp32 MFMA each costs ~8 cycles of matrix time: tools/mfma_filler_probe.hip),
    full drains (s_waitcnt vmcnt(0) / lgkmcnt(0)) and barriers inside it,
and for the whole kernel: WATERFALL loops around buffer instructions (a descriptor hipcc could not prove uniform: four readfirstlanes, two compares
and a branch per access - tests/test_verified_isa.py forbids them) and scratch (spill) instructions.

    python tools/isa_scan.py [path/to/libdsdenoise.so] > profiles/<tag>_isa_scan.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def disassemble(so):
    with tempfile.TemporaryDirectory() as d:
        lib = os.path.join(d, 'lib.so')
        subprocess.run(['cp', so, lib], check=True)          # objcopy rewrites its input
        subprocess.run([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={d}/fat.bin', lib], check=True, capture_output=True)
        subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={d}/fat.bin', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        f'--output={d}/dev.co'], check=True, capture_output=True)
        return subprocess.run([f'{LLVM}/llvm-objdump', '-d', '--no-show-raw-insn', f'{d}/dev.co'], check=True, capture_output=True, text=True).stdout


def demangle(n):
    for tool in (f'{LLVM}/llvm-cxxfilt', 'c++filt'):
        try:
            r = subprocess.run([tool, n], capture_output=True, text=True)
            if r.returncode == 0 and r.stdout.strip():
                return r.stdout.strip().replace('dsd::', '')
        except FileNotFoundError:
            continue
    return re.sub(r'^_ZN3dsd\d+', '', n)


def scan(txt):
    rows = []
    for name, body in re.findall(r'<(_ZN3dsd[^>]+)>:(.*?)\n\n', txt, re.S):
        lines = body.splitlines()
        ops = [re.sub(r'\s*//.*', '', l).strip() for l in lines]
        nm = sum(o.startswith('v_mfma') for o in ops)
        if nm < 16:
            continue
        addr = []
        for l in lines:
            m = re.search(r'//\s*([0-9A-F]{12}):', l)
            addr.append(int(m.group(1), 16) if m else None)
        amap = {a: i for i, a in enumerate(addr) if a is not None}
        best = None
        for i, o in enumerate(ops):
            m = re.match(r'(s_cbranch_\w+|s_branch)\s+(\d+)', o)
            if m and addr[i] is not None and int(m.group(2)) >= 32768:
                j = amap.get(addr[i] + 4 + (int(m.group(2)) - 65536) * 4)
                if j is None:
                    continue
                seg = ops[j:i + 1]
                mf = sum(x.startswith('v_mfma') for x in seg)
                if mf >= 8 and (best is None or mf / len(seg) > best[0]):
                    best = (mf / len(seg), seg, mf)
        waterfall = sum(1 for a, b in zip(ops, ops[1:]) if a.startswith('s_and_saveexec_b64') and re.match(r'buffer_(load|store|atomic)', b))
        scratch = sum(o.startswith('scratch_') for o in ops)
        if best is None:
            rows.append((demangle(name), nm, None, None, None, None, waterfall, scratch))
            continue
        _, seg, mf = best
        valu = sum(1 for x in seg if x.startswith('v_') and not x.startswith('v_mfma'))
        drains = sum(1 for x in seg if re.match(r's_waitcnt\s+(vmcnt\(0\)|lgkmcnt\(0\))', x) or 'vmcnt(0) lgkmcnt(0)' in x)
        bars = sum(x.startswith('s_barrier') for x in seg)
        rows.append((demangle(name), nm, mf, valu / mf, drains, bars, waterfall, scratch))
    return rows


if __name__ == '__main__':
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'diffsinger_amd', 'libdsdenoise.so')
    rows = scan(disassemble(so))
    print(f'# static scan of {os.path.basename(so)}: kernels with >= 16 MFMAs; densest loop = the loop with the most MFMAs per instruction')
    print(f'# {"kernel":72s} {"MFMA":>5s} | loop: {"MFMA":>5s} {"VALU/MFMA":>9s} {"drains":>6s} {"barriers":>8s} | {"waterfalls":>10s} {"scratch":>7s}')
    for r in sorted(rows, key=lambda r: -(r[3] or 0)):
        name = r[0][:72]
        if r[2] is None:
            print(f'  {name:72s} {r[1]:5d} | loop: {"-":>5s} {"-":>9s} {"-":>6s} {"-":>8s} | {r[6]:10d} {r[7]:7d}')
        else:
            print(f'  {name:72s} {r[1]:5d} | loop: {r[2]:5d} {r[3]:9.2f} {r[4]:6d} {r[5]:8d} | {r[6]:10d} {r[7]:7d}')
