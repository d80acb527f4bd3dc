set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_loop.py -m gpu -q -k "concurrent_load" 2>&1 | tail -25 > $O/pytest.txt
cat $O/pytest.txt | grep -v amdgpu.ids | tail -25
