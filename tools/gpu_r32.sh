set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r32}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fused.py tests/test_gpu_surfaces.py tests/test_gpu_fs2_train.py tests/test_gpu_train_dist.py -m gpu -q -s > $O/pytest_train.txt 2>&1
grep -v amdgpu $O/pytest_train.txt | grep -i "linear_rows\|passed\|failed\|Error" | cut -c1-260 | tail -20
for i in 1 2; do timeout 300 python bench.py --row train --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_row_train_$i.json 2> $O/bench_row_train.err; cut -c1-330 $O/bench_row_train_$i.json; done
