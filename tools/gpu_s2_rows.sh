# GPU box: the three rows around the path under the bench contract + rocprofv3 kernel stats of each
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02r}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 python bench.py --row train --steps 10 --warmup 3 > $O/bench_row_train.json 2> $O/bench_row_train.err
for row in fs2 vocoder train; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$row -o r -- python $R/bench.py --row $row --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_row_${row}_prof.json 2> $O/prof_$row.log
  python $R/tools/rocprof_summary.py $(ls $O/prof_$row/*.db $O/prof_$row/*/*.db 2>/dev/null | head -1) > $O/${row}_kernel_stats.txt 2>> $O/prof_$row.log
  rm -rf $O/prof_$row
done
cd $R
cut -c1-400 $O/bench_row_train.json; head -16 $O/fs2_kernel_stats.txt | cut -c1-80,98-150; head -12 $O/vocoder_kernel_stats.txt | cut -c1-80,98-150
