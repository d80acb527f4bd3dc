# GPU box: vocoder (row f2) parity tests + bench, sized for a ~1 minute slot
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-voc}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 55 python -m pytest tests/test_gpu_vocoder.py -m gpu -q -s -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
timeout 28 python tools/bench_vocoder.py 3 > $O/vocoder.jsonl 2> $O/vocoder.err
grep -E "err|passed|failed|rc=|Error|assert" $O/pytest.txt | cut -c1-220 | tail -60
cut -c1-330 $O/vocoder.jsonl; tail -3 $O/vocoder.err
