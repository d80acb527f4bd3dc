# GPU box: quick check of the operator families outside the persistent loop - FastSpeech2 / FFT decoder / training tests + their benches
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fs2.py tests/test_fft_decoder.py tests/test_gpu_train.py -m gpu -q 2>&1 | tail -4 > $O/pytest.txt
timeout 300 python tools/bench_fs2.py 20 > $O/fs2_forward.jsonl 2> $O/fs2_forward.err
timeout 400 python tools/bench_train.py 5 > $O/train_step.jsonl 2> $O/train_step.err
tail -2 $O/pytest.txt; cat $O/fs2_forward.jsonl | cut -c1-140; cat $O/train_step.jsonl | cut -c1-200
