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


def up_to_date() -> bool:
    return os.path.isfile(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if up_to_date() and not force:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + ['-o', LIB] + SRC
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
