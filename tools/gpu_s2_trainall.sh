# GPU box: the training step in its four forms (fused stack / operator path / torch conv inside the HIP graph / reference-style PyTorch-ROCm eager) on ONE box
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03o}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python tools/bench_train.py 8 > $O/train_step_all.jsonl 2> $O/err.txt
cut -c1-230 $O/train_step_all.jsonl; tail -2 $O/err.txt
