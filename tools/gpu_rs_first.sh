# GPU box, first call of round 4: (1) the hop probe (what an in-launch all-gather costs: sentinel vs flag protocol), (2) the row-split persistent
# loop's own tests (never run before: under a timeout, in their own process), (3) smoke + the bench line with parity on the timed batch,
# (4) latency of the reference's inference shapes on both paths, (5) the in-kernel timeline of a layer.
#   usage: bash tools/gpu_rs_first.sh <tag>
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
[ -x tools/hop_probe.bin ] && timeout 120 tools/hop_probe.bin 2000 > $O/hop_probe.jsonl 2> $O/hop_probe.err
( time DSD_RUN_UNVERIFIED=1 timeout 900 python -m pytest tests/test_gpu_rs.py -m gpu -x -q -rf -s > $O/pytest_rs.txt 2>&1 ) 2> $O/pytest_rs_time.txt
tail -5 $O/pytest_rs.txt
timeout 300 python tools/rs_timeline.py 1x512 $O/rs_timeline_1x512.json > $O/rs_timeline_1x512.txt 2>&1
timeout 300 python tools/rs_timeline.py 1x1550 $O/rs_timeline_1x1550.json > $O/rs_timeline_1x1550.txt 2>&1
DSD_RS=-1 timeout 400 python tools/shape_sweep.py 5 1x512,1x800,1x1000,1x1550,4x777,2x2048 --default-only > $O/shape_sweep_rs.jsonl 2> $O/shape_sweep_rs.err
timeout 400 python tools/shape_sweep.py 5 1x512,1x800,1x1000,1x1550,4x777,2x2048 --default-only > $O/shape_sweep_lat.jsonl 2> $O/shape_sweep_lat.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rf -s -k "config5_shape_row" > $O/pytest_cfg5_row.txt 2>&1
tail -3 $O/hop_probe.jsonl; tail -15 $O/pytest_rs.txt | cut -c1-250; cat $O/shape_sweep_rs.jsonl | cut -c1-300; cat $O/shape_sweep_lat.jsonl | cut -c1-300
tail -3 $O/smoke.txt; cut -c1-400 $O/bench_n1.json; tail -3 $O/pytest_cfg5_row.txt; tail -14 $O/rs_timeline_1x512.txt | cut -c1-200
