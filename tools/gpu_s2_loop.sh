# GPU box: does the HEAD k_loop (flags read early, one barrier less) still match the per-layer path bit for bit, and what does it cost?
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02i}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_loop.py tests/test_gpu_parity.py tests/test_gpu_noise.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_loop.txt
timeout 200 python tools/loop_timeline.py > $O/loop_timeline.txt 2>&1
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
tail -5 $O/pytest_loop.txt; tail -30 $O/loop_timeline.txt | cut -c1-200
python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('parity'))"
