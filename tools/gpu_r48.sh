set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r48}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_latency.py tests/test_gpu_loop.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py tests/test_gpu_surfaces.py tests/test_gpu_parity.py -m gpu -q > $O/pytest_lat.txt 2>&1
grep -v amdgpu $O/pytest_lat.txt | tail -4 | cut -c1-200
timeout 300 python tools/shape_sweep.py 2 3x1550,1x5000,5x1024,1x4200,1x5200 --default-only > $O/shape_sweep.jsonl 2> $O/shape_sweep.err
cut -c1-230 $O/shape_sweep.jsonl
