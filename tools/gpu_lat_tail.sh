#!/bin/bash
# GPU call: the next-node prefetch of the latency kernels (GemmPipe TAIL) - tests, A/B of the small shapes, per-node kernel stats
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
T=${1:-r5_lat}
( timeout 900 python -m pytest tests/test_gpu_latency.py tests/test_gpu_parity.py -x -q 2>&1 | tail -8 ) > gpurun_out/${T}_pytest_latency.txt
for v in 1 0 1 0; do
( DSD_LAT_TAIL=$v timeout 300 python tools/shape_sweep.py 5 1x512,1x800,1x1000,1x1550,4x777,2x2048 --default-only 2>/dev/null | grep "^{" | sed "s/^{/{\"lat_tail\": $v, /" ) >> gpurun_out/${T}_shape_sweep_lat.jsonl
done
cd /tmp
for v in 1 0; do
DSD_LAT_TAIL=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_lat$v -o lat -- python $OLDPWD/tools/shape_sweep.py 3 1x512 --default-only > /tmp/prof_lat$v.log 2>&1
python $OLDPWD/tools/rocprof_summary.py $(ls /tmp/prof_lat$v/*.db /tmp/prof_lat$v/*/*.db 2>/dev/null | head -1) | head -12 > $OLDPWD/gpurun_out/${T}_latency_kernel_stats_tail$v.txt 2>&1
done
cd $OLDPWD
tail -3 gpurun_out/${T}_pytest_latency.txt; python3 -c "
import json
for l in open('gpurun_out/${T}_shape_sweep_lat.jsonl'):
    d=json.loads(l); print(d['lat_tail'], d['B'], d['T'], d['path'], d['ms_per_pass'])
"; head -8 gpurun_out/${T}_latency_kernel_stats_tail1.txt | cut -c1-150; head -8 gpurun_out/${T}_latency_kernel_stats_tail0.txt | cut -c1-150
