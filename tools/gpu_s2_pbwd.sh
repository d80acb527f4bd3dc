# GPU box: the persistent backward chain (k_trb_loop, opt-in) - fused-stack tests, then A/B against the per-layer launches inside one call
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03v}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_train_fused.py -m gpu -q -x 2>&1 | tail -25 > $O/pytest.txt
for rep in ${REPS:-1 2}; do for v in 1 0; do for sh in 8x1024 48x512; do
DSD_TRAIN_PERSIST_BWD=$v timeout 200 python tools/bench_train.py 10 --hip-only $sh 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'persist_bwd':$v,'shape':'$sh','ms':d['ms_per_step_fwd_bwd']}))" >> $O/persist_bwd_ab.jsonl
done; done; done
cat $O/pytest.txt | grep -v Warn; cat $O/persist_bwd_ab.jsonl; tail -3 $O/err.txt
