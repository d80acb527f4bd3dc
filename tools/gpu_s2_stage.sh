set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02x}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fs2.py tests/test_gpu_pe.py tests/test_fft_decoder.py tests/test_gpu_train.py tests/test_gpu_widths.py -m gpu -q 2>&1 | tail -4 > $O/pytest_ops.txt
for rep in 1 2; do for v in 1 0; do
DSD_FS_STAGE=$v timeout 200 python bench.py --row fs2 --steps 20 --warmup 3 --no-cpu-baseline 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'row':'fs2','stage':$v,'ms':d['ms_per_step'],'kernel_frac':d['roofline']['frac']}))" >> $O/stage_ab.jsonl
DSD_FS_STAGE=$v timeout 200 python tools/bench_train.py 8 --hip-only 8x1024 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'row':'train','stage':$v,'ms':d['ms_per_step_fwd_bwd']}))" >> $O/stage_ab.jsonl
DSD_FS_STAGE=$v DSD_TRAIN_FUSED=0 timeout 200 python tools/bench_pe.py 2>> $O/err.txt | tail -1 | cut -c1-200 | sed "s/^/stage=$v /" >> $O/stage_ab.jsonl
done; done
tail -2 $O/pytest_ops.txt; cat $O/stage_ab.jsonl; tail -3 $O/err.txt
