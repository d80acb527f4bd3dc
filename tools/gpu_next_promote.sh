# GPU box, FIRST call of the next round: promote the two opt-in kernels of the last GPU minutes of round 2 - the frame-major persistent loop
# (csrc/dsd_loop_fm.hpp: bit-identical, +1 %, r04c-r04e) and the branch-free K-half conv of the latency path (k_lat_conv<kLatG8BF>: bit-identical,
# 39.4 -> 32.3 ms per 1 x 512 K = 100 call, r04f / r04g): the WHOLE GPU suite with DSD_LOOP_FM=1 DSD_LAT_BF=1 in the environment (every engine
# then runs them) + the held-back module, smoke, the headline bench and the latency probe both ways inside one call.
# If green: make both the default in dsd_create, rename tests/test_gpu_zz_lat_bf.py, python tools/isa_hashes.py --update, refresh the evidence.
# Also measured here: the NOT YET RUN running-pointer convolutions of the vocoder / FastSpeech2 rows (k_voc_conv_inc, k_fs_conv_inc; their held-back
# module tests/test_gpu_zz_conv_inc.py runs with the suite) - bench.py --row vocoder / fs2 with DSV_CONV_INC / DSF_CONV_INC off and on.
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r05_fm}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time DSD_LOOP_FM=1 DSD_LAT_BF=1 DSD_RUN_UNVERIFIED=1 timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 ) > $O/pytest_gpu_loop_fm.txt 2>&1
DSD_LOOP_FM=1 DSD_LAT_BF=1 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_loop_fm.txt 2>&1
for rep in 1 2 3; do for v in 0 1; do
DSD_LOOP_FM=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>> $O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'loop_fm':$v,'value':d['value'],'ms_per_step':d['ms_per_step'],'frac':d['roofline']['frac'],'avg_launch_ms':d['roofline']['avg_launch_ms']}))" >> $O/loop_fm_ab.jsonl
done; done
timeout 120 python tools/lat_bf_probe.py > $O/lat_bf_probe_1x512.json 2>> $O/err.txt
timeout 120 python tools/lat_bf_probe.py opencpop_ds60_rel 1 1550 60 > $O/lat_bf_probe_1x1550.json 2>> $O/err.txt
cat $O/lat_bf_probe_*.json
for rep in 1 2; do for v in 0 1; do
DSV_CONV_INC=$v timeout 300 python bench.py --row vocoder --steps 10 --warmup 3 --no-cpu-baseline 2>> $O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'row':'vocoder','conv_inc':$v,'ms_per_step':d['ms_per_step']}))" >> $O/conv_inc_ab.jsonl
DSF_CONV_INC=$v timeout 300 python bench.py --row fs2 --steps 10 --warmup 3 --no-cpu-baseline 2>> $O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'row':'fs2','conv_inc':$v,'ms_per_step':d['ms_per_step']}))" >> $O/conv_inc_ab.jsonl
done; done
cat $O/conv_inc_ab.jsonl
tail -12 $O/pytest_gpu_loop_fm.txt | cut -c1-200; tail -3 $O/smoke_loop_fm.txt; cat $O/loop_fm_ab.jsonl; tail -3 $O/err.txt
