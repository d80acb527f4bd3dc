"""Developer tool (GPU): hunt for run-to-run nondeterminism of the persistent loop on the fixture cases."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_loop import _run

names = sys.argv[1:] or ['plms_opencpop_i40', 'plms_opencpop_i250', 'shallow_opencpop_k60']
for name in names:
    ref, _, _ = _run(name, 0)
    ref2, _, _ = _run(name, 0)
    print(name, 'graph path repeatable:', np.array_equal(ref, ref2), flush=True)
    bad = 0
    for i in range(12):
        a, used, tmo = _run(name, 1)
        eq = np.array_equal(a, ref)
        if not eq:
            bad += 1
            d = np.abs(a - ref)
            nz = np.argwhere(d > 0)
            print(f'  run {i}: MISMATCH {int((d > 0).sum())}/{d.size} elements, max abs {d.max():.3e}; frames (t) touched: '
                  f'{sorted(set(nz[:, 1].tolist()))[:8]}.. utterances {sorted(set(nz[:, 0].tolist()))}', flush=True)
        if tmo:
            print('  timeout', tmo)
    print(name, 'persistent mismatches:', bad, 'of 12', flush=True)
