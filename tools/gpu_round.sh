set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r1_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err
timeout 300 python tools/layer_sweep.py > gpurun_out/r1_sweep.log 2>&1
timeout 300 python tools/layer_timeline.py 8 1024 32 > gpurun_out/r1_timeline.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1_prof.log 2>&1
ls -R $R/gpurun_out | head -50
