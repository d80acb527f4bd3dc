"""Build libdsdenoise.so IN-TREE with hipcc for gfx950 (cross-compiles without a GPU).

    python -m diffsinger_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(PKG, 'csrc', 'dsd.hip')]
# every header under csrc/ and include/ is a dependency (a fixed list went stale when pwg_kernels.hpp was added: its edits did not rebuild)
DEPS = SRC + sorted(os.path.join(PKG, 'csrc', f) for f in os.listdir(os.path.join(PKG, 'csrc')) if f.endswith('.hpp')) + sorted(
    os.path.join(os.path.dirname(PKG), 'include', f) for f in os.listdir(os.path.join(os.path.dirname(PKG), 'include')) if f.endswith('.h'))
LIB = os.path.join(PKG, 'libdsdenoise.so')
# -ffp-contract=off: hipcc's default (fast) fuses a*b+c into FMA wherever it likes, also across the __fmul_rn / __fadd_rn
# "intrinsics" (plain operators to the optimiser) - the sampler arithmetic must round every product like the reference's
# tensor ops do, and the persistent loop and the per-layer kernels must stay bit-identical (explicit fmaf / MFMA are unaffected)
# -amdgpu-mfma-vgpr-form=1 (round 6): MFMA results in the ARCHITECTURAL register file wherever they fit.  hipcc's default for a kernel that may
# use all 512 registers (one wave per SIMD) is the accumulation-register form - every accumulator the vector ALU touches afterwards (output
# transform, gate, residual, epilogues) then crosses between the two files with v_accvgpr_read / _write, 8 cycles of matrix time each beside
# fp32 MFMAs.  Same source, same results; A/B on one box (profiles/r6_09_*): k_loop 127.95 -> 126.19 ms, k_loop_wino 101.9 -> 101.0 ms (and its
# 15 spilled dwords -> 8, k_loop's 43 -> 0), the training step 5.26 -> 5.17 ms, vocoder / FastSpeech2 unchanged.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off', '-mllvm', '-amdgpu-mfma-vgpr-form=1']


_ID_TAG = b'DSD_BUILD_ID='


def source_hash() -> str:
    """sha256 over every source the library is compiled from (csrc/*.hip, csrc/*.hpp, include/*.h: names and bytes, sorted) and the
    compiler flags.  The build bakes it into the .so (`dsd_build_id()`, include/dsd.h): a binary proves which tree it came from."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(PKG)
    for f in sorted(DEPS):
        h.update(os.path.relpath(f, root).encode() + b'\0')
        h.update(open(f, 'rb').read())
        h.update(b'\0')
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def binary_id(path: str = LIB):
    """The build id baked into a built library, read from the file's bytes (no dlopen); None when absent."""
    try:
        blob = open(path, 'rb').read()
    except OSError:
        return None
    i = blob.find(_ID_TAG)
    if i < 0:
        return None
    j = i + len(_ID_TAG)
    return blob[j:j + 64].decode('ascii', 'replace')


def up_to_date() -> bool:
    """The binary is current when the hash it carries equals the hash of the tree - not when its mtime is newer (a snapshot copy, a
    checkout or a touched file says nothing about contents)."""
    return binary_id() == source_hash()


KERNEL_ISA = os.path.join(PKG, 'kernel_isa.json')
_LLVM = '/opt/rocm/lib/llvm/bin'


def kernel_isa_hashes(lib: str = LIB) -> dict:
    """{demangled kernel name: sha1 of its gfx950 instructions} read back from the BUILT library (llvm-objdump of its device code, addresses and
    encodings dropped).  The identity of one kernel's device code: an evidence file under profiles/ that carries it can be matched to the
    library a process has loaded even when an edit elsewhere in csrc/ changed the library's build id."""
    import hashlib
    import re
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        cp = os.path.join(d, 'lib.so')
        shutil.copy(lib, cp)                                          # objcopy rewrites its input
        subprocess.run([f'{_LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={d}/fat.bin', cp], check=True, capture_output=True)
        subprocess.run([f'{_LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={d}/fat.bin',
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={d}/dev.co'], check=True, capture_output=True)
        txt = subprocess.run([f'{_LLVM}/llvm-objdump', '-d', '-C', '--no-show-raw-insn', f'{d}/dev.co'], check=True, capture_output=True, text=True).stdout
    out = {}
    heads = list(re.finditer(r'^[0-9a-f]+ <([^\n]+)>:$', txt, re.M))
    for h, nxt in zip(heads, heads[1:] + [None]):
        body = txt[h.end():nxt.start() if nxt else len(txt)]
        ops = [re.sub(r'\s*//.*', '', ln).strip() for ln in body.splitlines()]
        out[h.group(1).replace('dsd::', '').replace('void ', '')] = hashlib.sha1('\n'.join(o for o in ops if o).encode()).hexdigest()
    return out


def write_kernel_isa(lib: str = LIB) -> str:
    """Sidecar of the built library: diffsinger_amd/kernel_isa.json = {build_id, kernels}.  Git-ignored like the .so, travels with it."""
    import json
    json.dump({'build_id': binary_id(lib), 'kernels': kernel_isa_hashes(lib)}, open(KERNEL_ISA, 'w'), indent=0)
    return KERNEL_ISA


def kernel_isa(build_id=None) -> dict:
    """The sidecar's {kernel: sha1} if it belongs to the library with this build id (default: the built one), else {}."""
    import json
    try:
        js = json.load(open(KERNEL_ISA))
    except (OSError, ValueError):
        return {}
    return js.get('kernels', {}) if js.get('build_id') == (build_id or binary_id()) else {}


def build(force: bool = False, verbose: bool = True) -> str:
    if up_to_date() and not force:
        if not kernel_isa():
            write_kernel_isa()
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + [f'-DDSD_BUILD_ID_STR="{source_hash()}"', '-o', LIB] + SRC
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    assert binary_id() == source_hash(), 'the built library does not carry the hash of the tree it was built from'
    write_kernel_isa()
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
