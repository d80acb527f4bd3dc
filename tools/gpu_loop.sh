set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r01f}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_loop.py -m gpu -q 2>&1 | tail -30 > $O/pytest_loop.txt
timeout 200 python tools/loop_timeline.py > $O/loop_timeline.txt 2>&1
for m in 1 0; do
  DSD_LOOP=$m timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_loop$m.json 2> $O/bench_loop$m.err
done
cat $O/pytest_loop.txt | tail -12; cat $O/loop_timeline.txt
for m in 0 1; do python -c "
import json; d=json.load(open('$O/bench_loop$m.json')); print('loop=$m', d['value'], d['ms_per_step'], d['parity'])"; done
