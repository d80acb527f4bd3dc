set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in 1 2; do for v in 1 0; do
DSD_WGRAD_PIPE=$v timeout 200 python tools/bench_wgrad.py >> $O/wgrad_ab.jsonl 2>> $O/err.txt
DSD_WGRAD_PIPE=$v timeout 300 python tools/bench_train.py 8 --hip-only 8x1024 2>> $O/err.txt | sed "s/^{/{\"pipe\": $v, /" >> $O/train_ab.jsonl
done; done
cat $O/wgrad_ab.jsonl; cut -c1-160 $O/train_ab.jsonl; tail -3 $O/err.txt
