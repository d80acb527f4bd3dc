export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ka
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/ka/res.txt
  HIP_FORCE_DEV_KERNARG=$v python tools/layer_timeline.py 8 1024 32 2>/dev/null | grep -A8 "layer 3" | grep "stage\|lifetime" >> gpurun_out/ka/res.txt
done
cat gpurun_out/ka/res.txt
