set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_fft_decoder.py tests/test_gpu_fs2.py -m gpu -q -rA 2>&1 | tail -40 > $O/pytest.txt
grep -v "^PASSED\|amdgpu.ids" $O/pytest.txt | tail -30
