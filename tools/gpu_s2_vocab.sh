set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02y}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_e2e.py tests/test_gpu_fs2.py -m gpu -q 2>&1 | tail -4 > $O/pytest_voc.txt
cp diffsinger_amd/libdsdenoise.so /tmp/new.so
for rep in 1 2; do for v in new old; do
if [ $v = old ]; then cp tools/_ab/libdsdenoise_old.so diffsinger_amd/libdsdenoise.so; else cp /tmp/new.so diffsinger_amd/libdsdenoise.so; fi
timeout 200 python bench.py --row vocoder --steps 10 --warmup 2 --no-cpu-baseline 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'row':'vocoder','lib':'$v','ms':d['ms_per_step'],'kernel_frac':d['roofline']['frac']}))" >> $O/voc_ab.jsonl
done; done
cp /tmp/new.so diffsinger_amd/libdsdenoise.so
tail -2 $O/pytest_voc.txt; cat $O/voc_ab.jsonl; tail -3 $O/err.txt
