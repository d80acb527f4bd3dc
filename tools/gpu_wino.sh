#!/bin/bash
# GPU call of round 5: the Winograd loop - filler price list, parity tests, touch A/B, shape sweep with every path forced, in-kernel timeline,
# the whole GPU suite, the bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
T=${1:-r5_03}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_filler_probe tools/mfma_filler_probe.hip && ( timeout 180 /tmp/mfma_filler_probe ) > gpurun_out/${T}_mfma_filler_probe.jsonl 2>&1
( timeout 300 python tools/wino_ab.py 2>&1 | grep "^{" ) > gpurun_out/${T}_wino_ab.jsonl
( timeout 300 python tools/loop_timeline.py gpurun_out/${T}_loop_timeline.json 2>&1 | tail -60 ) > gpurun_out/${T}_loop_timeline.txt
( timeout 900 python tools/shape_sweep.py 2 3x1550,1x5000,5x1024,1x5200,3x5000,2x2048,4x777,4x1024,6x1024,8x1024,16x2048,1x1550,1x512 --lat-splits 2>&1 | grep "^{" ) > gpurun_out/${T}_shape_sweep.jsonl
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 2>&1 | tail -60 ) > gpurun_out/${T}_pytest_gpu.txt
( timeout 600 python bench.py --steps 10 2>&1 | tail -3 ) > gpurun_out/${T}_bench_n1.json
tail -4 gpurun_out/${T}_pytest_gpu.txt; cat gpurun_out/${T}_wino_ab.jsonl; cat gpurun_out/${T}_shape_sweep.jsonl | cut -c1-400; head -c 600 gpurun_out/${T}_bench_n1.json
