set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03e}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_latency.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_loop.py tests/test_gpu_noise.py tests/test_gpu_surfaces.py -m gpu -q -rf 2>&1 | tail -12 > $O/pytest_lat.txt
for rep in 1 2; do for v in 1 0; do
DSD_LAT_HEAD=$v timeout 200 python tools/shape_sweep.py 3 1x512,2x300,1x200 --default-only 2>> $O/err.txt | sed "s/^{/{\"head_split\": $v, /" >> $O/lat_head_ab.jsonl
done; done
tail -6 $O/pytest_lat.txt | cut -c1-200; cut -c1-200 $O/lat_head_ab.jsonl; tail -3 $O/err.txt
