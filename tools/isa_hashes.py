"""Developer tool (CPU): per-kernel hash of the gfx950 device assembly of csrc/dsd.hip (labels normalised, comments dropped) - the guard that
kernels measured on the MI355X are not changed by an edit made while no GPU is at hand.

    python tools/isa_hashes.py                 # print {mangled kernel name: sha1}
    python tools/isa_hashes.py --update        # rewrite tests/golden/kernel_isa_hashes.json (after the kernels ran on the hardware again)
tests/test_verified_isa.py compares the kernels listed in that file with the current sources; kernels not listed (new ones) are free."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffsinger_amd.build import FLAGS as _LIB_FLAGS  # noqa: E402
GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'kernel_isa_hashes.json')
FLAGS = [f for f in _LIB_FLAGS if f != '-shared'] + ['--cuda-device-only', '-S']       # the library's own flags, device-only assembly


_LISTING = None


def kernel_listing():
    """{mangled kernel name: [normalised instruction lines]} (compiled once per process)"""
    global _LISTING
    if _LISTING is not None:
        return _LISTING
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'dsd.s')
        subprocess.run([hipcc] + FLAGS + ['-o', out, os.path.join(ROOT, 'diffsinger_amd', 'csrc', 'dsd.hip')], check=True, capture_output=True, cwd=d)
        funcs, cur = {}, None
        for ln in open(out):
            m = re.match(r'^(_Z\w+):', ln)
            if m:
                cur = m.group(1)
                funcs[cur] = []
                continue
            if ln.startswith('.Lfunc_end'):
                cur = None
                continue
            if cur is not None:
                t = ln.strip()
                if not t or t.startswith(';') or t.startswith('.'):
                    continue
                t = re.sub(r';.*', '', t).strip()
                funcs[cur].append(re.sub(r'\.LBB\d+_(\d+)', r'.LBB_\1', t))
    _LISTING = funcs
    return funcs


def kernel_hashes():
    return {k: hashlib.sha1('\n'.join(v).encode()).hexdigest() for k, v in kernel_listing().items()}


def waterfall_loops():
    """{kernel: count} of buffer loads / stores wrapped in a WATERFALL loop (s_and_saveexec in front of the instruction): hipcc emits one
    per memory instruction whose buffer descriptor it could not prove wave-uniform - four v_readfirstlane, two v_cmp and a branch each, and
    no interleaving with the MFMAs around it (round 5: every weight load of k_trb_fused_w, dsd_kernels.hpp uniform_ptr)."""
    out = {}
    for k, v in kernel_listing().items():
        n = sum(1 for a, b in zip(v, v[1:]) if a.startswith('s_and_saveexec_b64') and re.match(r'buffer_(load|store|atomic)', b))
        if n:
            out[k] = n
    return out


if __name__ == '__main__':
    h = kernel_hashes()
    if '--update' in sys.argv:
        keep = [k for k in sorted(h) if not re.search(os.environ.get('DSD_ISA_UNVERIFIED', r'^$'), k)]     # kernels not yet run on hardware
        json.dump({'note': 'sha1 of the normalised gfx950 assembly of every kernel that has run (and passed its parity tests) on the MI355X; '
                           'tools/isa_hashes.py --update after a GPU run', 'kernels': {k: h[k] for k in keep}}, open(GOLDEN, 'w'), indent=1)
        print(f'{len(keep)} kernels written to {GOLDEN} ({len(h) - len(keep)} not-yet-run kernels left out)')
    else:
        print(json.dumps(h, indent=1))
