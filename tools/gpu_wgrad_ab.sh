# GPU box: the weight-gradient kernel with column halves (k_tr_wgrad<FIX, 2>: two workgroups per CU) - gradient parity tests, then the
# training-step bench with both forms (DSF_WGRAD_NH = 2 default / 1 whole tile), alternating
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r4_06}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fused.py tests/test_gpu_train_dist.py -m gpu -q -rf -s > $O/pytest_train_nh2.txt 2>&1
tail -4 $O/pytest_train_nh2.txt | cut -c1-200
for nh in 2 1 2 1; do
DSF_WGRAD_NH=$nh timeout 300 python bench.py --row train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_row_train_nh${nh}_$RANDOM.json 2> /dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_row_train_nh*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))
PY
