"""CPU: the C-ABI library is built, loads, and exports exactly the symbols include/dsd.h declares.
No compute calls (there is no GPU here)."""
import ctypes
import os
import re

from diffsinger_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(header='dsd.h', prefix='dsd_'):
    src = open(os.path.join(ROOT, 'include', header)).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(' + prefix + r'[a-z0-9_]+)\s*\(', src)))


def test_library_is_built():
    assert os.path.isfile(_lib.lib_path()), 'run python -m diffsinger_amd.build'


def test_binding_lists_every_header_symbol():
    assert _header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_header_symbol():
    lib = ctypes.CDLL(_lib.lib_path())
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert lib.dsd_abi_version() == _lib.DSD_ABI_VERSION


def test_fs2_header_symbols_bound_and_exported():
    assert _header_symbols('dsf.h', 'dsf_') == sorted(_lib.SYMBOLS_FS2)
    lib = _lib.load()
    for name in _lib.SYMBOLS_FS2:
        assert hasattr(lib, name), name
    assert lib.dsf_padded_frames(33) == 64
    assert lib.dsf_packed_floats(256, 256, 9) == (4 * 32 * 9 * 2 * 64 + 8192) * 4
    assert lib.dsf_packed_floats(1024, 256, 9) == (4 * 4 * 32 * 9 * 2 * 64 + 8192) * 4
    assert lib.dsf_packed_floats(80, 250, 1) == -1                      # input channels must be a multiple of 8
    assert lib.dsf_conv1d(None, None, None, None, 1, 8, 8, 1, 1, 1.0, 0, None, None, None) == -1      # rejected before any HIP call
    assert b'dsf_conv1d' in lib.dsd_last_error()


def test_bad_config_is_rejected_without_a_device():
    lib = _lib.load()
    cfg = _lib.DsdConfig(80, 128, 256, 20, 1)          # residual_channels != 256: rejected before any HIP call
    h = ctypes.c_void_p()
    assert lib.dsd_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1
    assert b'residual_channels' in lib.dsd_last_error()
    assert lib.dsd_create(None, 0, ctypes.byref(h)) == -1


def test_cxx_host_example_builds_against_the_header(tmp_path):
    """examples/dsd_example.cpp: a torch-free, Python-free host of the C ABI compiles and links against include/dsd.h + the .so
    (and refuses to run without a device; on the GPU box tools/gpu_round2_first.sh runs it)."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.isfile(hipcc):
        import pytest
        pytest.skip('no hipcc')
    exe = os.path.join(tmp_path, 'dsd_example')
    lib_dir = os.path.dirname(_lib.lib_path())
    res = subprocess.run([hipcc, '-O1', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'dsd_example.cpp'), '-L', lib_dir,
                          '-ldsdenoise', f'-Wl,-rpath,{lib_dir}', '-o', exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([exe, '1', '32', '2'], capture_output=True, text=True)
        assert run.returncode != 0 and 'hipSetDevice' in run.stderr


def test_build_dependencies_cover_every_included_header():
    """diffsinger_amd/build.py rebuilds when a dependency is newer than the .so: every header dsd.hip includes (transitively) must be one -
    a header missing from a fixed list (pwg_kernels.hpp, round 3) let GPU runs test a stale binary."""
    import os
    import re
    from diffsinger_amd import build as B
    csrc = os.path.join(B.PKG, 'csrc')
    seen, todo = set(), [os.path.join(csrc, 'dsd.hip')]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.isfile(f):
            continue
        seen.add(f)
        for inc in re.findall(r'#include\s+"([^"]+)"', open(f).read()):
            todo.append(os.path.normpath(os.path.join(os.path.dirname(f), inc)))
    deps = {os.path.normpath(d) for d in B.DEPS}
    missing = sorted(os.path.relpath(f, B.PKG) for f in seen if f not in deps)
    assert not missing, missing
