# GPU box: (1) FastSpeech2 operator + model parity tests, (2) does the DVFS perf level explain the gap between the kernel time
# under rocprofv3 (62.9 us) and in a plain run (69.6 us)?  bench with sclk/power sampled, at perf level auto and high.
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r01d}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fs2.py -m gpu -q -rA 2>&1 | tail -150 > $O/pytest_fs2.txt
rocm-smi --showperflevel --showclocks --showpower --showmaxpower --showsclkrange > $O/smi_before.txt 2>&1
sample() {  # $1 = out file; samples until the flag file disappears
  while [ -e $O/.run ]; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -3 >> $1; sleep 0.4; done
}
for lvl in auto high; do
  rocm-smi --setperflevel $lvl > $O/setperf_$lvl.txt 2>&1
  touch $O/.run; sample $O/smi_during_$lvl.csv &
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_$lvl.json 2> $O/bench_$lvl.err
  rm -f $O/.run; wait
done
rocm-smi --setperflevel auto >> $O/setperf_auto.txt 2>&1
rocm-smi --showperflevel > $O/smi_after.txt 2>&1
for lvl in auto high; do python -c "
import json; d=json.load(open('$O/bench_$lvl.json')); print('$lvl', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
tail -40 $O/pytest_fs2.txt
