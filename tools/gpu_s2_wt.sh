set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02v}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fs2.py tests/test_gpu_vocoder.py tests/test_gpu_pe.py tests/test_gpu_e2e.py tests/test_fft_decoder.py tests/test_gpu_train.py -m gpu -q 2>&1 | tail -6 > $O/pytest_ops.txt
for rep in 1 2; do for v in 1 0; do
DSD_OP_WT=$v timeout 200 python bench.py --row fs2 --steps 20 --warmup 3 --no-cpu-baseline 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'row':'fs2','wt':$v,'ms':d['ms_per_step'],'kernel_frac':d['roofline']['frac']}))" >> $O/wt_ab.jsonl
DSD_OP_WT=$v timeout 200 python bench.py --row vocoder --steps 10 --warmup 2 --no-cpu-baseline 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'row':'vocoder','wt':$v,'ms':d['ms_per_step'],'kernel_frac':d['roofline']['frac']}))" >> $O/wt_ab.jsonl
DSD_OP_WT=$v timeout 200 python tools/bench_train.py 8 --hip-only 8x1024 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'row':'train','wt':$v,'ms':d['ms_per_step_fwd_bwd']}))" >> $O/wt_ab.jsonl
done; done
tail -3 $O/pytest_ops.txt; cat $O/wt_ab.jsonl; tail -3 $O/err.txt
