set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_noise.py tests/test_gpu_loop.py tests/test_gpu_parity.py -m gpu -q -rA 2>&1 | tail -60 > $O/pytest.txt
timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -v "^PASSED\|amdgpu.ids" $O/pytest.txt | tail -40
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
