set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fs2.py tests/test_fft_decoder.py -m gpu -q -rA 2>&1 | tail -60 > $O/pytest.txt
timeout 300 python tools/bench_fs2.py 20 > $O/fs2_forward.jsonl 2> $O/fs2_forward.err
grep "attention\|passed\|failed\|Error" $O/pytest.txt | head; cat $O/fs2_forward.jsonl | cut -c1-170
