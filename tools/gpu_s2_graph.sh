set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02u}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in 1 2; do
timeout 200 python tools/bench_train.py 10 --hip-only 8x1024 >> $O/train_graph_ab.jsonl 2>> $O/err.txt
timeout 200 python tools/bench_train.py 10 --hip-only 8x1024 --graph >> $O/train_graph_ab.jsonl 2>> $O/err_graph.txt
done
timeout 200 python tools/bench_train.py 10 --hip-only 48x512 >> $O/train_graph_ab.jsonl 2>> $O/err.txt
timeout 200 python tools/bench_train.py 10 --hip-only 48x512 --graph >> $O/train_graph_ab.jsonl 2>> $O/err_graph.txt
cut -c1-230 $O/train_graph_ab.jsonl; tail -5 $O/err_graph.txt
