set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03c}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
DSD_FS_NMB4=1 timeout 900 python -m pytest tests/test_gpu_fs2.py tests/test_gpu_train.py -m gpu -q 2>&1 | tail -3 > $O/pytest.txt
for rep in 1 2; do for v in 1 0; do
DSD_FS_NMB4=$v timeout 200 python bench.py --row fs2 --steps 20 --warmup 3 --no-cpu-baseline 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'row':'fs2','nmb4':$v,'ms':d['ms_per_step'],'kernel_frac':d['roofline']['frac']}))" >> $O/nmb_ab.jsonl
done; done
cat $O/pytest.txt; cat $O/nmb_ab.jsonl; tail -3 $O/err.txt
