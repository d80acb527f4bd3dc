#!/bin/bash
# GPU call: quick check of a k_loop_wino change - parity tests of the Winograd loop, hand-off tests under load, A/B against the direct loop, timeline
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
T=${1:-r5_q}
( timeout 900 python -m pytest tests/test_gpu_wino.py -x -q -s 2>&1 | tail -30 ) > gpurun_out/${T}_pytest_wino.txt
( timeout 300 python tools/wino_ab.py 2>&1 | grep "^{" ) > gpurun_out/${T}_wino_ab.jsonl
( timeout 300 python tools/loop_timeline.py gpurun_out/${T}_loop_timeline.json 2>&1 | tail -60 ) > gpurun_out/${T}_loop_timeline.txt
( DSD_CONV=winograd timeout 600 python -m pytest tests/test_gpu_noise.py tests/test_gpu_surfaces.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -8 ) > gpurun_out/${T}_pytest_misc.txt
tail -6 gpurun_out/${T}_pytest_wino.txt; cat gpurun_out/${T}_wino_ab.jsonl; head -12 gpurun_out/${T}_loop_timeline.txt; tail -3 gpurun_out/${T}_pytest_misc.txt
