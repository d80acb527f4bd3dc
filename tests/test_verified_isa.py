"""CPU: the device code of every kernel that has run on the MI355X is byte-for-byte (label-normalised) what it was then.  Edits made
while no GPU is available - new experimental kernels in the same translation unit, refactors - must not change measured kernels
silently; after a GPU run `python tools/isa_hashes.py --update` records the new state."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_MOD = None


def _tool():
    global _MOD
    if not os.path.isfile(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('no hipcc')
    if _MOD is None:
        spec = importlib.util.spec_from_file_location('isa_hashes', os.path.join(ROOT, 'tools', 'isa_hashes.py'))
        _MOD = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_MOD)
    return _MOD


def test_measured_kernels_are_unchanged():
    mod = _tool()
    want = json.load(open(mod.GOLDEN))['kernels']
    got = mod.kernel_hashes()
    missing = [k for k in want if k not in got]
    changed = [k for k in want if k in got and got[k] != want[k]]
    assert not missing, f'kernels that ran on the GPU disappeared: {missing}'
    assert not changed, f'device code of GPU-verified kernels changed without a GPU run: {changed}'
    assert len(want) >= 50


def test_no_kernel_wraps_a_buffer_access_in_a_waterfall_loop():
    """Round 5 (profiles/r5_25_trb_timeline.txt): hipcc had moved a scalar address chain of k_trb_fused_w to the vector ALU; the buffer descriptor
    built from it counted as divergent and every weight load became a waterfall loop - the kernel's convolution ran at 59 instead of 37
    cycles per MFMA, invisible in any test (the results are the same).  No kernel of the library may contain the pattern."""
    bad = _tool().waterfall_loops()
    assert not bad, f'waterfall loops around buffer instructions (make the descriptor base provably uniform: uniform_ptr): {bad}'
