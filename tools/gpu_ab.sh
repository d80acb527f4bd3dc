# GPU box: A/B of the XCD-aware workgroup map + per-position sequence profile of the sampling graph
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-ab}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_parity.txt
for m in 0 1; do
  DSD_XCD_MAP=$m timeout 300 python tools/layer_sweep.py > $O/sweep_xcd$m.txt 2>&1
  DSD_XCD_MAP=$m timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_xcd$m.json 2> $O/bench_xcd$m.err
  DSD_XCD_MAP=$m timeout 200 python tools/layer_timeline.py 8 1024 32 > $O/timeline_xcd$m.txt 2>&1
done
cd /tmp
for m in 0 1; do
  DSD_XCD_MAP=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/seq$m -o seq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/seq$m.log 2>&1
  python $R/tools/seq_summary.py $(find $O/seq$m -name '*kernel_trace.csv' | head -1) > $O/seq_summary_xcd$m.txt 2>> $O/seq$m.log
  DSD_XCD_MAP=$m timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc$m/fetch -o fetch -- python $R/tools/profile_layer.py 8 1024 32 40 > $O/pmc_fetch$m.log 2>&1
  DSD_XCD_MAP=$m timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc$m/write -o write -- python $R/tools/profile_layer.py 8 1024 32 40 > $O/pmc_write$m.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc$m 'k_layer<1, false>' $O/layer_pmc_xcd$m.txt $O/layer_pmc_xcd$m.json frames=8192 'kernel_tag=k_layer<1,false>' round=$TAG > /dev/null 2>&1
  rm -rf $O/seq$m
done
cat $O/pytest_parity.txt $O/sweep_xcd0.txt $O/sweep_xcd1.txt $O/seq_summary_xcd1.txt; tail -2 $O/layer_pmc_xcd0.txt $O/layer_pmc_xcd1.txt
