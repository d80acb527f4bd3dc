# GPU box: quick iteration loop - parity tests, layer timeline, layer sweep, short bench (no CPU baseline)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-q}
mkdir -p $R/gpurun_out/$T; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/$T/pytest.log
timeout 200 python tools/layer_timeline.py 8 1024 32 > gpurun_out/$T/timeline.log 2>&1
timeout 200 python tools/layer_sweep.py > gpurun_out/$T/sweep.log 2>&1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
tail -3 gpurun_out/$T/pytest.log; cat gpurun_out/$T/timeline.log gpurun_out/$T/sweep.log; cat gpurun_out/$T/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('parity'))"
