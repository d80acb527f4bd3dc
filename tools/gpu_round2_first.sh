# GPU box, FIRST call of the next round: what round 1 wrote after its GPU minutes were spent and could not measure.
#   1. the verified -m gpu suite (113 tests), then the never-run kernels (tests/test_gpu_zz_*.py: fused AdamW, split-precision conv prototype,
#      split-precision residual layer) with DSD_RUN_UNVERIFIED=1, one process per file
#   2. bench.py (headline), bench.py --row vocoder (row f2 under the bench contract), bench.py --split (EXPERIMENT: layers on the bf16 pipe)
#   3. tools/bench_vocoder.py in full (per-stage k_voc_conv vs k_voc_conv_fold launches, NSF, PyTorch-ROCm baseline last)
#   4. the torch-free C++ host example; tools/mfma_split_probe (accuracy / matrix-pipe rate / weight-stream ceiling of the split scheme)
#   5. rocprofv3 kernel stats of the vocoder forward and a PMC pass over its folded convolution kernel (separate --pmc run)
# usage: bash tools/gpu_round2_first.sh <tag>          (~6-8 GPU-minutes)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02a}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu.txt
# the never-run kernels, one file per process so that a fault in one does not hide the others
for f in tests/test_gpu_zz_*.py; do DSD_RUN_UNVERIFIED=1 timeout 300 python -m pytest $f -m gpu -q -s -rxX 2>&1 | tail -40 > $O/$(basename $f .py).txt; done
timeout 300 python tools/bench_vocoder.py 5 > $O/vocoder.jsonl 2> $O/vocoder.err
timeout 120 python tools/bench_pe.py 20 > $O/pe_forward.jsonl 2> $O/pe_forward.err
timeout 300 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 200 python tools/shape_sweep.py 2 > $O/shape_sweep.jsonl 2> $O/shape_sweep.err
timeout 200 python bench.py --row vocoder --steps 10 --warmup 2 > $O/bench_vocoder_row.json 2> $O/bench_vocoder_row.err
timeout 300 python bench.py --row train --steps 5 --warmup 2 > $O/bench_train_row.json 2> $O/bench_train_row.err
timeout 200 python bench.py --row fs2 --steps 20 --warmup 3 > $O/bench_fs2_row.json 2> $O/bench_fs2_row.err
timeout 300 python bench.py --split --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_split_experiment.json 2> $O/bench_split_experiment.err
/opt/rocm/bin/hipcc -O2 -I include examples/dsd_example.cpp -L diffsinger_amd -ldsdenoise -Wl,-rpath,$R/diffsinger_amd -o /tmp/dsd_example && timeout 120 /tmp/dsd_example 8 1024 100 > $O/cxx_example.json 2> $O/cxx_example.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_split_probe tools/mfma_split_probe.hip && timeout 60 /tmp/mfma_split_probe > $O/mfma_split_probe.jsonl 2> $O/mfma_split_probe.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_voc -o voc -- python $R/tools/bench_vocoder_quick.py > $O/prof_voc.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof_voc/*.db $O/prof_voc/*/*.db 2>/dev/null | head -1) > $O/vocoder_kernel_stats.txt 2>> $O/prof_voc.log
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_voc/fetch -o fetch -- python $R/tools/bench_vocoder_quick.py > $O/pmc_voc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/pmc_voc/write -o write -- python $R/tools/bench_vocoder_quick.py > $O/pmc_voc_write.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_voc 'k_voc_conv_fold<4>' $O/voc_fold_pmc.txt $O/voc_fold_pmc.json frames=8192 'kernel_tag=k_voc_conv_fold<4>' round=$TAG > $O/pmc_voc_summary.log 2>&1
rm -rf $O/prof_voc
find $O/pmc_voc -name '*.db' -delete
du -sh $O
tail -8 $O/pytest_gpu.txt; for f in $O/test_gpu_zz_*.txt; do echo == $f; grep -E 'err|TFLOP|us per|XPASS|XFAIL|passed|failed|xfailed|xpassed|Error' $f | cut -c1-220 | tail -14; done; cat $O/cxx_example.json; cut -c1-900 $O/bench_split_experiment.json; cat $O/mfma_split_probe.jsonl; cut -c1-700 $O/bench_vocoder_row.json; cut -c1-700 $O/bench_train_row.json; cut -c1-700 $O/bench_fs2_row.json; cut -c1-300 $O/vocoder.jsonl | head -20; cat $O/pe_forward.jsonl; cat $O/bench_n1.json | cut -c1-600; cat $O/shape_sweep.jsonl; tail -3 $O/shape_sweep.err; head -14 $O/vocoder_kernel_stats.txt | cut -c1-180
