#!/bin/bash
# GPU call: the Winograd forward of the fused training stack (csrc/train_loop_wino.hpp) - its tests, the A/B against the direct persistent
# forward, the bench row and the kernel times of the step
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
T=${1:-r5_t}; R=$(pwd)
( timeout 1200 python -m pytest tests/test_gpu_train_fused.py -x -q -s 2>&1 | tail -60 ) > gpurun_out/${T}_pytest_train_fused.txt
( timeout 300 python tools/bench_train.py 10 --conv-ab 2>&1 | grep "^{" ) > gpurun_out/${T}_train_conv_ab.jsonl
timeout 300 python bench.py --row train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench_row_train.json 2> gpurun_out/${T}_bench_row_train.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o tr -- python $R/tools/bench_train.py 8 --hip-only 8x1024 > $R/gpurun_out/${T}_prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/${T}_prof/*.db $R/gpurun_out/${T}_prof/*/*.db 2>/dev/null | head -1) > $R/gpurun_out/${T}_train_kernel_stats.txt 2>> $R/gpurun_out/${T}_prof.log
cd $R
rm -rf gpurun_out/${T}_prof
tail -8 gpurun_out/${T}_pytest_train_fused.txt; cat gpurun_out/${T}_train_conv_ab.jsonl; cut -c1-400 gpurun_out/${T}_bench_row_train.json; head -12 gpurun_out/${T}_train_kernel_stats.txt | cut -c1-160
