set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03j}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_e2e.py tests/test_gpu_pe.py -m gpu -q 2>&1 | tail -6 > $O/pytest_voc.txt
timeout 200 python bench.py --row vocoder --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_row_vocoder.json 2> $O/err.txt
tail -3 $O/pytest_voc.txt; cut -c1-300 $O/bench_row_vocoder.json
