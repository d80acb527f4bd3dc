# GPU box: A/B of the chunk rotation in the split-precision loop (DSD_SPLIT_ROT=0 / 1), each in its own process: tests + rate
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r4_05}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rot in 0 1 0 1; do
DSD_SPLIT_ROT=$rot timeout 600 python -m pytest tests/test_gpu_split_loop.py -m gpu -q -rf -s -k "config2 or ddpm_lj or plms_opencpop_i40" > $O/pytest_split_rot${rot}_$RANDOM.txt 2>&1
done
grep -h "rate:\|passed\|failed" $O/pytest_split_rot*.txt | cut -c1-260
