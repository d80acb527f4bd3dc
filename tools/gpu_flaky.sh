set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r01k}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 400 python tools/loop_flaky2.py > $O/flaky2.txt 2>&1
cat $O/flaky2.txt | grep -v amdgpu.ids
