set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r34}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fs2.py tests/test_gpu_fs2_train.py tests/test_gpu_surfaces.py tests/test_gpu_pe.py tests/test_gpu_train.py tests/test_gpu_train_fused.py tests/test_gpu_e2e.py tests/test_gpu_graphs.py tests/test_gpu_widths.py -m gpu -q -s > $O/pytest_fs2.txt 2>&1
grep -v amdgpu $O/pytest_fs2.txt | grep -i "split=\|passed\|failed\|Error" | cut -c1-200 | tail -30
for i in 1 2; do timeout 300 python bench.py --row fs2 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_row_fs2_$i.json 2> $O/bench_row_fs2.err; cut -c1-330 $O/bench_row_fs2_$i.json; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o fs2 -- python $R/bench.py --row fs2 --steps 10 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
python $R/tools/trace_by_grid.py $O/trace > $O/fs2_by_grid.txt
rm -rf $O/trace
head -16 $O/fs2_by_grid.txt | cut -c1-140
