# GPU box: the starved-loop test + forced row splits of the latency kernels on shapes between the G = 16 and the persistent regime
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r19}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_loop.py -m gpu -q -s > $O/pytest_loop.txt 2>&1
timeout 600 python tools/shape_sweep.py 3 1x512,1x1000,1x1550,2x1000,4x777,3x1550,6x1024,8x1024 --lat-splits > $O/lat_splits.jsonl 2> $O/lat_splits.err
grep -v amdgpu $O/pytest_loop.txt | tail -30 | cut -c1-250; cat $O/lat_splits.jsonl | cut -c1-400; tail -5 $O/lat_splits.err
