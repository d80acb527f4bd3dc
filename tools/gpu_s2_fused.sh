# GPU box: first run of the fused training stack - its unit tests, the existing training tests on the fused path, the per-layer inference path
# (layer_body was refactored), the step benchmark
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02k}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train_fused.py -m gpu -q -s -x 2>&1 | tail -60 > $O/pytest_fused.txt
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_loop.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -15 > $O/pytest_train_loop.txt
timeout 300 python tools/bench_train.py 5 --hip-only 8x1024 > $O/train_step.jsonl 2> $O/train_step.err
timeout 300 python tools/bench_train.py 5 --hip-only 48x512 >> $O/train_step.jsonl 2>> $O/train_step.err
cat $O/pytest_fused.txt | cut -c1-250; tail -5 $O/pytest_train_loop.txt; cat $O/train_step.jsonl | cut -c1-200; tail -3 $O/train_step.err
