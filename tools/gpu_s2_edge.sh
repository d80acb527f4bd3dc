# GPU box: the fused training stack's shape / loss variants
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03b}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train_fused.py -m gpu -q -s -k "stack_forward or p_losses" 2>&1 | grep -v Warning | tail -30 > $O/pytest.txt
cat $O/pytest.txt | cut -c1-230
