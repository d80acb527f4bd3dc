set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02z}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 ) > $O/pytest_gpu.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -12 $O/pytest_gpu.txt | cut -c1-200; tail -3 $O/smoke.txt
