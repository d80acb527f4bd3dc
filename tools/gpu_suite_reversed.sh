# GPU box: the GPU test files in REVERSE order (order-dependence check: the starved-loop failure of r06 / r15 / r17 only showed inside the whole suite)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-rev}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
FILES=$(ls tests/test_gpu_*.py | sort -r | tr '\n' ' ')
( time timeout 1500 python -m pytest $FILES -m gpu -q -rf -p no:cacheprovider ) > $O/pytest_gpu_reversed.txt 2>&1
grep -v amdgpu $O/pytest_gpu_reversed.txt | grep -i "passed\|failed\|real" | tail -6
