# GPU box, FIRST call of the next round: promote the frame-major persistent loop (csrc/dsd_loop_fm.hpp; bit-identical and +1 % in r04c-r04e) -
# the WHOLE GPU suite with DSD_LOOP_FM=1 in the environment (every engine then runs it), smoke, and the headline bench both ways inside one call.
# If green: make it the default in dsd_create, python tools/isa_hashes.py --update, re-run bench + rocprof + PMC evidence.
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r05_fm}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time DSD_LOOP_FM=1 timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 ) > $O/pytest_gpu_loop_fm.txt 2>&1
DSD_LOOP_FM=1 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_loop_fm.txt 2>&1
for rep in 1 2 3; do for v in 0 1; do
DSD_LOOP_FM=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>> $O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'loop_fm':$v,'value':d['value'],'ms_per_step':d['ms_per_step'],'frac':d['roofline']['frac'],'avg_launch_ms':d['roofline']['avg_launch_ms']}))" >> $O/loop_fm_ab.jsonl
done; done
tail -12 $O/pytest_gpu_loop_fm.txt | cut -c1-200; tail -3 $O/smoke_loop_fm.txt; cat $O/loop_fm_ab.jsonl; tail -3 $O/err.txt
