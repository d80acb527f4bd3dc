#!/bin/bash
# GPU call: every training test (operators, fused stack, FastSpeech2 training, gradient exchange) + the A/B of the convolution forms + bench row + kernel times
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
T=${1:-r5_t}; R=$(pwd)
( timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fused.py tests/test_gpu_fs2_train.py tests/test_gpu_train_dist.py tests/test_fft_decoder.py -q -rf 2>&1 | tail -30 ) > gpurun_out/${T}_pytest_train_all.txt
( timeout 300 python tools/bench_train.py 10 --conv-ab 2>&1 | grep "^{" ) > gpurun_out/${T}_train_conv_ab.jsonl
timeout 300 python bench.py --row train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench_row_train.json 2> gpurun_out/${T}_bench_row_train.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o tr -- python $R/tools/bench_train.py 8 --hip-only 8x1024 > $R/gpurun_out/${T}_prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/${T}_prof/*.db $R/gpurun_out/${T}_prof/*/*.db 2>/dev/null | head -1) > $R/gpurun_out/${T}_train_kernel_stats.txt 2>> $R/gpurun_out/${T}_prof.log
cd $R
rm -rf gpurun_out/${T}_prof
tail -8 gpurun_out/${T}_pytest_train_all.txt; grep wino gpurun_out/${T}_train_conv_ab.jsonl | cut -c100-260; cut -c1-400 gpurun_out/${T}_bench_row_train.json; head -14 gpurun_out/${T}_train_kernel_stats.txt | cut -c1-150
