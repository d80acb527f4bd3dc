# GPU box: vocoder chain kernel - bit-identity tests, generator A/B (unfused / fused), kernel stats, workgroup timeline
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r11}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_vocoder.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest_vocoder.txt
for rep in 1 2; do for mode in off stage; do
timeout 300 python bench.py --row vocoder --chain $mode --steps 10 --warmup 3 --no-cpu-baseline 2>> $O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'chain':'$mode','ms_per_step':d['ms_per_step']}))" >> $O/voc_chain_ab.jsonl
done; done
timeout 200 python tools/voc_chain_timeline.py > $O/voc_chain_timeline.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o voc -- python $R/bench.py --row vocoder --steps 5 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/vocoder_kernel_stats.txt 2>> $O/prof.log
rm -rf $O/prof
cd $R
tail -5 $O/pytest_vocoder.txt; cat $O/voc_chain_ab.jsonl; head -10 $O/vocoder_kernel_stats.txt | cut -c1-180; grep -v amdgpu $O/voc_chain_timeline.txt | grep "total\|conv  0\|conv 12\|stage"; tail -3 $O/err.txt
