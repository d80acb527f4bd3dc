set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-final}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest_gpu.txt
timeout 300 python bench.py --steps 5 --warmup 2 > $O/bench_n1.json 2> $O/bench_n1.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -3 $O/pytest_gpu.txt; tail -3 $O/smoke.txt; cut -c1-300 $O/bench_n1.json
