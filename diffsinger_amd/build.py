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
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off']


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


def build(force: bool = False, verbose: bool = True) -> str:
    if up_to_date() and not force:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + [f'-DDSD_BUILD_ID_STR="{source_hash()}"', '-o', LIB] + SRC
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    assert binary_id() == source_hash(), 'the built library does not carry the hash of the tree it was built from'
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
