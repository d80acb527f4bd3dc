set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r38}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tr -- python $R/bench.py --row train --steps 10 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
python $R/tools/trace_by_grid.py $O/trace > $O/train_by_grid.txt
rm -rf $O/trace
head -45 $O/train_by_grid.txt | cut -c1-150
