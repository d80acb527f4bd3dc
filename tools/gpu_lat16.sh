# GPU box: the G = 16 latency kernels + the tests that failed inside the full suite (complete output)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r16}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_latency.py -m gpu -q -s > $O/pytest_latency.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_loop.py tests/test_gpu_train_fused.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py -m gpu -q > $O/pytest_loop_train.txt 2>&1
timeout 300 python tools/shape_sweep.py 3 1x512,1x300,2x250,1x1000,1x1550 > $O/shape_sweep.jsonl 2> $O/shape_sweep.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o lat -- python $R/tools/shape_sweep.py 2 1x512 --default-only > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/latency_kernel_stats.txt 2>> $O/prof.log
rm -rf $O/prof
cd $R
grep -v amdgpu $O/pytest_latency.txt | tail -25 | cut -c1-250; grep -v amdgpu $O/pytest_loop_train.txt | tail -60 | cut -c1-250; cat $O/shape_sweep.jsonl | cut -c1-300; head -12 $O/latency_kernel_stats.txt | cut -c1-170
