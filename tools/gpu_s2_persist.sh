# GPU box: the persistent training kernels (csrc/train_loop.hpp) - the whole GPU suite + smoke on the final binary, bench.py --row train, the
# A/B of both switches against the per-layer launches inside the same call, kernel stats of the step
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03z}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 ) > $O/pytest_gpu.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 300 python bench.py --row train --steps 10 --warmup 3 > $O/bench_row_train.json 2> $O/bench_row_train.err
for rep in 1 2; do for v in "1 1" "1 0" "0 0"; do set -- $v; for sh in 8x1024 48x512; do
DSD_TRAIN_PERSIST=$1 DSD_TRAIN_PERSIST_BWD=$2 timeout 200 python tools/bench_train.py 10 --hip-only $sh 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'persist_fwd':$1,'persist_bwd':$2,'shape':'$sh','ms':d['ms_per_step_fwd_bwd']}))" >> $O/persist_ab.jsonl
done; done; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o tr -- python $R/tools/bench_train.py 6 --hip-only 8x1024 > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats_8x1024.txt 2>> $O/prof.log
rm -rf $O/prof
cd $R
tail -12 $O/pytest_gpu.txt | cut -c1-200; tail -3 $O/smoke.txt; cat $O/persist_ab.jsonl; cat $O/bench_row_train.json | cut -c1-400; head -12 $O/train_kernel_stats_8x1024.txt | cut -c1-60,98-160
