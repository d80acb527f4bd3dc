import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests/` here.
    Quarantine hook (no such module exists at present - every kernel in the tree has run on the hardware): tests of a kernel written while
    no GPU is at hand may go to tests/test_gpu_zz_*.py, which stays out of the default `-m gpu` run (a fault in a never-run kernel would
    take the whole verified suite's process down with it) until DSD_RUN_UNVERIFIED=1 is set."""
    if not os.environ.get('DSD_RUN_UNVERIFIED'):
        hold = pytest.mark.skip(reason='kernel not yet run on hardware: set DSD_RUN_UNVERIFIED=1')
        for item in items:
            if os.path.basename(str(item.fspath)).startswith('test_gpu_zz_'):
                item.add_marker(hold)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
