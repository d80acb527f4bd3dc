set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -rA 2>&1 | tail -70 > $O/pytest.txt
grep -v "^PASSED\|amdgpu.ids" $O/pytest.txt | tail -60
