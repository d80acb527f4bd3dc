# GPU box: A/B of a developer switch of the fused training backward (here: two layers per weight-gradient launch): tests in both modes, the step inside one call
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03l}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_train.py tests/test_gpu_widths.py -m gpu -q 2>&1 | tail -4 > $O/pytest.txt
DSD_TRAIN_WGRAD_PAIR=0 timeout 900 python -m pytest tests/test_gpu_train_fused.py -m gpu -q 2>&1 | tail -2 >> $O/pytest.txt
for rep in 1 2; do for v in 1 0; do
for sh in 8x1024 48x512; do
DSD_TRAIN_WGRAD_PAIR=$v timeout 200 python tools/bench_train.py 10 --hip-only $sh 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'wgrad_pair':$v,'shape':'$sh','ms':d['ms_per_step_fwd_bwd']}))" >> $O/wgrad_pair_ab.jsonl
done; done; done
cat $O/pytest.txt | grep -v Warn; cat $O/wgrad_pair_ab.jsonl; tail -3 $O/err.txt
