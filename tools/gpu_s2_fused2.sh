# GPU box: fused-stack tests + step benchmark + kernel stats of the step
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02m}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_train.py -m gpu -q -s -x 2>&1 | grep -v Warning | tail -40 > $O/pytest_fused.txt
timeout 300 python tools/bench_train.py 8 --hip-only 8x1024 > $O/train_step.jsonl 2> $O/train_step.err
timeout 300 python tools/bench_train.py 8 --hip-only 48x512 >> $O/train_step.jsonl 2>> $O/train_step.err
for s in 8x1024 48x512; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$s -o tr -- python $R/tools/bench_train.py 6 --hip-only $s > $O/prof_$s.log 2>&1
  python $R/tools/rocprof_summary.py $(ls $O/prof_$s/*.db $O/prof_$s/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats_$s.txt 2>> $O/prof_$s.log
  rm -rf $O/prof_$s
done
cd $R
grep -v "^$" $O/pytest_fused.txt | cut -c1-220 | tail -25; cat $O/train_step.jsonl | cut -c1-200
head -12 $O/train_kernel_stats_8x1024.txt | cut -c1-72,98-140; head -9 $O/train_kernel_stats_48x512.txt | cut -c1-72,98-140
