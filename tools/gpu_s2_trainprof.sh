# GPU box: rocprofv3 kernel stats of the HIP training step alone (one shape per run)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02j}
O=$R/gpurun_out/$TAG
mkdir -p $O
for s in 8x1024 48x512; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$s -o tr -- python $R/tools/bench_train.py 6 --hip-only $s > $O/prof_$s.log 2>&1
  python $R/tools/rocprof_summary.py $(ls $O/prof_$s/*.db $O/prof_$s/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats_$s.txt 2>> $O/prof_$s.log
  rm -rf $O/prof_$s
  head -40 $O/train_kernel_stats_$s.txt | cut -c1-60,98-160
done
