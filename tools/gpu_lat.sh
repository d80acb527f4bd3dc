# GPU box: the latency path only - its parity tests, the small-batch rows of the shape sweep, rocprofv3 kernel stats of one-utterance runs
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02g}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_latency.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q -rf 2>&1 | tail -30 > $O/pytest_lat.txt
timeout 200 python tools/shape_sweep.py 3 1x512,1x1550,4x777,2x300 --default-only > $O/shape_sweep_small.jsonl 2> $O/shape_sweep_small.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_lat -o lat -- python $R/tools/shape_sweep.py 2 1x512,1x1550,4x777 --default-only > $O/prof_lat.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof_lat/*.db $O/prof_lat/*/*.db 2>/dev/null | head -1) > $O/latency_kernel_stats.txt 2>> $O/prof_lat.log
rm -rf $O/prof_lat
tail -8 $O/pytest_lat.txt | cut -c1-200; cat $O/shape_sweep_small.jsonl; head -12 $O/latency_kernel_stats.txt | cut -c1-150
