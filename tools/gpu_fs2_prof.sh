# GPU box: FastSpeech2 forward row - bench line + rocprofv3 kernel stats of the same command
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-fs2prof}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 python bench.py --row fs2 --steps 20 --warmup 3 > $O/bench_row_fs2.json 2> $O/bench_row_fs2.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o fs2 -- python $R/bench.py --row fs2 --steps 20 --warmup 3 > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/fs2_kernel_stats.txt 2>> $O/prof.log
rm -rf $O/prof
cd $R
cut -c1-400 $O/bench_row_fs2.json; head -60 $O/fs2_kernel_stats.txt | cut -c1-190
