set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r37}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
for m in 1 0; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$m -o fs2 -- python $R/bench.py --row fs2 --steps 10 --warmup 2 --no-cpu-baseline --conv-split $m > $O/trace$m.log 2>&1
python $R/tools/trace_by_grid.py $O/trace$m k_fs_conv > $O/fs2_conv_by_grid_split$m.txt
rm -rf $O/trace$m
head -14 $O/fs2_conv_by_grid_split$m.txt | cut -c1-140
done
