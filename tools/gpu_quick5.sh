set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 400 python tools/bench_train.py 5 > $O/bench_train.jsonl 2> $O/bench_train.err
cat $O/bench_train.jsonl; tail -3 $O/bench_train.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o tr -- python $R/tools/bench_train.py 2 > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats.txt 2>> $O/prof.log
rm -rf $O/prof
head -16 $O/train_kernel_stats.txt | cut -c1-200
