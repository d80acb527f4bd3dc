set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -rA 2>&1 | tail -40 > $O/pytest.txt
timeout 400 python tools/bench_train.py 5 > $O/bench_train.jsonl 2> $O/bench_train.err
grep -v "^PASSED\|amdgpu.ids\|^---\|^___" $O/pytest.txt | tail -14; cat $O/bench_train.jsonl
