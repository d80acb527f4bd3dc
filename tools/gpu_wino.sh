#!/bin/bash
# GPU call of round 5: the Winograd loop - parity tests, A/B against the direct loop over the stream knobs, in-kernel timeline, kernel stats
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
T=${1:-r5_01}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_filler_probe tools/mfma_filler_probe.hip && ( timeout 120 /tmp/mfma_filler_probe ) > gpurun_out/${T}_mfma_filler_probe.jsonl 2>&1
( timeout 900 python -m pytest tests/test_gpu_wino.py -x -q -s 2>&1 | tail -60 ) > gpurun_out/${T}_pytest_wino.txt
( timeout 300 python tools/wino_ab.py 2>&1 | tail -40 ) > gpurun_out/${T}_wino_ab.jsonl
( timeout 300 python tools/loop_timeline.py gpurun_out/${T}_loop_timeline.json 2>&1 | tail -60 ) > gpurun_out/${T}_loop_timeline.txt
( timeout 600 python -m pytest tests/test_gpu_loop.py tests/test_gpu_parity.py -x -q 2>&1 | tail -15 ) > gpurun_out/${T}_pytest_loop_parity.txt
cat gpurun_out/${T}_mfma_filler_probe.jsonl; tail -5 gpurun_out/${T}_pytest_wino.txt; cat gpurun_out/${T}_wino_ab.jsonl; tail -3 gpurun_out/${T}_pytest_loop_parity.txt
