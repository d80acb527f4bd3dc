# GPU box: vocoder + pitch extractor (row f2) parity tests + vocoder bench, sized for a ~45 s slot
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-voc}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 30 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_pe.py tests/test_gpu_e2e.py -m gpu -q -s -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
timeout 16 python tools/bench_vocoder.py 3 > $O/vocoder.jsonl 2> $O/vocoder.err
grep -E "err|passed|failed|rc=|Error|assert|FAILED" $O/pytest.txt | cut -c1-200 | tail -70
cut -c1-420 $O/vocoder.jsonl; tail -3 $O/vocoder.err
