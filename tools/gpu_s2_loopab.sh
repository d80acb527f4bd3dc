set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03h}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
cp diffsinger_amd/libdsdenoise.so /tmp/new.so
for rep in 1 2 3; do for v in new old; do
if [ $v = old ]; then cp tools/_ab/libdsdenoise_old.so diffsinger_amd/libdsdenoise.so; else cp /tmp/new.so diffsinger_amd/libdsdenoise.so; fi
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'lib':'$v','ms':d['ms_per_step'],'launch_ms':d['roofline']['avg_launch_ms'],'frac':d['roofline']['frac']}))" >> $O/loop_ab.jsonl
done; done
cp /tmp/new.so diffsinger_amd/libdsdenoise.so
cat $O/loop_ab.jsonl; tail -2 $O/err.txt
