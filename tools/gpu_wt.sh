set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r01e}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_parity.txt
for m in 0 1; do
  DSD_WT_STORES=$m timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_wt$m.json 2> $O/bench_wt$m.err
done
timeout 300 python tools/bench_fs2.py 20 > $O/bench_fs2.jsonl 2> $O/bench_fs2.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fs2 -o fs2 -- python $R/tools/bench_fs2.py 5 > $O/prof_fs2.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof_fs2/*.db $O/prof_fs2/*/*.db 2>/dev/null | head -1) > $O/fs2_kernel_stats.txt 2>> $O/prof_fs2.log
rm -rf $O/prof_fs2
cat $O/pytest_parity.txt
for m in 0 1; do python -c "
import json; d=json.load(open('$O/bench_wt$m.json')); print('wt=$m', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['parity'])"; done
cat $O/bench_fs2.jsonl; head -12 $O/fs2_kernel_stats.txt
