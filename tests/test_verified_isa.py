"""CPU: the device code of every kernel that has run on the MI355X is byte-for-byte (label-normalised) what it was then.  Edits made
while no GPU is available - new experimental kernels in the same translation unit, refactors - must not change measured kernels
silently; after a GPU run `python tools/isa_hashes.py --update` records the new state."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_measured_kernels_are_unchanged():
    if not os.path.isfile(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('no hipcc')
    spec = importlib.util.spec_from_file_location('isa_hashes', os.path.join(ROOT, 'tools', 'isa_hashes.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(mod.GOLDEN))['kernels']
    got = mod.kernel_hashes()
    missing = [k for k in want if k not in got]
    changed = [k for k in want if k in got and got[k] != want[k]]
    assert not missing, f'kernels that ran on the GPU disappeared: {missing}'
    assert not changed, f'device code of GPU-verified kernels changed without a GPU run: {changed}'
    assert len(want) >= 50
