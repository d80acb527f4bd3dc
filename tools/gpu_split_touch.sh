# GPU box: the plane stream of the split loop with the L2 touch off / 8 chunks ahead (DSD_SPLIT_TOUCH; TOUCHES='0 4 8 16' to sweep), beside the register split:
# bit-identity of the weight-stream variants first (the plane buffer is in consumption order now), then one K = 100 call each with stamps.
#   usage: bash tools/gpu_split_touch.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r4_12}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_split_loop.py -m gpu -q -s -k "variants or ddpm_lj" > $O/variants.txt 2>&1
grep "DSD_SPLIT_W\|passed\|failed\|rror" $O/variants.txt | cut -c1-300
for wf in 0; do for t in ${TOUCHES:-0 8}; do
  DSD_SPLIT_W=$wf DSD_SPLIT_TOUCH=$t timeout 200 python tools/loop_timeline.py --split --k=100 --phases=1003,1503 > $O/w${wf}_touch_$t.txt 2>&1
  echo "stream $wf, touch $t:"; grep -v amdgpu $O/w${wf}_touch_$t.txt | sed -n 2,10p | cut -c1-170; grep "timeouts\|rror" $O/w${wf}_touch_$t.txt | head -3
done; done
DSD_SPLIT_W=4 timeout 200 python tools/loop_timeline.py --split --k=100 --phases=1003,1503 > $O/regsplit.txt 2>&1
echo "register split:"; grep "phase total\|shader clock\|timeouts" $O/regsplit.txt | cut -c1-200
