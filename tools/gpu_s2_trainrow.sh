# GPU box: bench.py --row train (roofline of k_tr_wgrad from the in-step probe) + PMC passes over the training kernels
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03n}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 python bench.py --row train --steps 10 --warmup 3 > $O/bench_row_train.json 2> $O/bench_row_train.err
timeout 300 python -m pytest tests/test_gpu_train_fused.py -m gpu -q -k "stack_forward" 2>&1 | tail -2 > $O/pytest.txt
bash tools/gpu_s2_trainpmc.sh $TAG > $O/pmc.log 2>&1
python -c "
import json; d=json.load(open('$O/bench_row_train.json')); print(d['ms_per_step'], d['roofline'])"
tail -2 $O/pytest.txt; tail -3 $O/bench_row_train.err; cat $O/pmc_k_tr_wgradfalse.txt | tail -12
