# GPU box: targeted diagnostics of this round (full tracebacks)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r07}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_vocoder.py -m gpu -q -x 2>&1 | tail -60 > $O/pytest_vocoder.txt
timeout 300 python -m pytest tests/test_gpu_loop.py -m gpu -q -x -s -k starved 2>&1 | tail -80 > $O/pytest_starved.txt
timeout 200 python tools/diag_dcond.py adamw > $O/diag_dcond_adamw.txt 2>&1
timeout 200 python tools/diag_dcond.py sgd > $O/diag_dcond_sgd.txt 2>&1
for mode in off default stage resblock pair; do
timeout 300 python bench.py --row vocoder --chain $mode --steps 10 --warmup 3 --no-cpu-baseline 2>> $O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'chain':'$mode','ms_per_step':d['ms_per_step']}))" >> $O/voc_chain_ab.jsonl
done
cat $O/pytest_vocoder.txt | tail -30; cat $O/pytest_starved.txt | tail -60; cat $O/diag_dcond_adamw.txt; cat $O/diag_dcond_sgd.txt; cat $O/voc_chain_ab.jsonl; tail -5 $O/err.txt
