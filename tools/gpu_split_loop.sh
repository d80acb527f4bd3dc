# GPU box: the split-precision persistent loop (csrc/dsd_loop_split.hpp): its tests (in their own process, under a timeout), then the
# default bench line with the labelled `secondary` object, the per-layer split tests (the per-layer kernel shares the plane packing).
#   usage: bash tools/gpu_split_loop.sh <tag>
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time DSD_RUN_UNVERIFIED=1 timeout 900 python -m pytest tests/test_gpu_split_loop.py -m gpu -x -q -rf -s > $O/pytest_split_loop.txt 2>&1 ) 2> $O/pytest_split_loop_time.txt
tail -12 $O/pytest_split_loop.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_split_layer.py -m gpu -q -rf -s > $O/pytest_split_layer.txt 2>&1
tail -5 $O/pytest_split_layer.txt | cut -c1-300
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json,sys
d=json.load(open(sys.argv[1])) if len(sys.argv)>1 else None
PY
python -c "
import json
d=json.load(open('$O/bench_n1.json'))
print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','cfg5_shard_n1','secondary')}, indent=1)[:3000])
print(d['parity']['max_abs_mel_err'], d['parity']['kernel'])
"
tail -3 $O/bench_n1.err
