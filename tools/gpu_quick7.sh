set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_fs2.py tests/test_fft_decoder.py -m gpu -q 2>&1 | tail -8 > $O/pytest.txt
timeout 400 python tools/bench_train.py 5 > $O/bench_train.jsonl 2> $O/bench_train.err
timeout 300 python tools/bench_fs2.py 20 > $O/fs2_forward.jsonl 2> $O/fs2_forward.err
tail -4 $O/pytest.txt; cat $O/bench_train.jsonl | cut -c1-200; cat $O/fs2_forward.jsonl | cut -c1-170
