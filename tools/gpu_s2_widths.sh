set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02w}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_widths.py tests/test_fft_decoder.py tests/test_gpu_surfaces.py -m gpu -q -s 2>&1 | grep -v Warning | tail -30 > $O/pytest_widths.txt
cat $O/pytest_widths.txt | cut -c1-220
