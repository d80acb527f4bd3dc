# GPU box: FastSpeech2 under autograd - operator tests, gradient parity of the whole model, the e2e step; the inference tests that share the ops
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r13}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fs2_train.py -m gpu -q -s 2>&1 | tail -80 > $O/pytest_fs2_train.txt
timeout 900 python -m pytest tests/test_gpu_fs2.py tests/test_gpu_surfaces.py tests/test_gpu_train.py tests/test_gpu_e2e.py -m gpu -q 2>&1 | tail -30 > $O/pytest_fs2_related.txt
cat $O/pytest_fs2_train.txt | cut -c1-260; tail -8 $O/pytest_fs2_related.txt | cut -c1-200
