# GPU box: the whole GPU suite with the complete log + smoke()
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-suite}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -rfs -s ) > $O/pytest_gpu_full.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
grep -v amdgpu $O/pytest_gpu_full.txt | grep -i "starved\|passed\|failed\|skipped\|real" | tail -12; tail -3 $O/smoke.txt
