# GPU box: the full evidence round - parity tests, bench (+CPU baseline), rocprofv3 kernel stats of the same command,
# PMC passes over the dominant kernel (separate --pmc runs, kernel-trace only), phase timelines, shape sweep, all-config
# throughput, FastSpeech2 forward.   usage: bash tools/gpu_round.sh <tag>
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 200 python tools/loop_timeline.py > $O/loop_timeline.txt 2>&1
timeout 200 python tools/layer_timeline.py 8 1024 32 > $O/layer_timeline.txt 2>&1
timeout 300 python tools/layer_sweep.py > $O/layer_sweep.txt 2>&1
timeout 400 python tools/bench_configs.py 3 > $O/configs_throughput.jsonl 2> $O/configs_throughput.err
timeout 300 python tools/bench_fs2.py 20 > $O/fs2_forward.jsonl 2> $O/fs2_forward.err
timeout 400 python tools/bench_train.py 5 > $O/train_step.jsonl 2> $O/train_step.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/bench_n1_kernel_stats.txt 2>> $O/prof.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_tr -o tr -- python $R/tools/bench_train.py 2 > $O/prof_tr.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof_tr/*.db $O/prof_tr/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats.txt 2>> $O/prof_tr.log
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc/fetch -o fetch -- python $R/tools/profile_loop.py 3 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc/write -o write -- python $R/tools/profile_loop.py 3 > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/pmc/sq -o sq -- python $R/tools/profile_loop.py 3 > $O/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc 'k_loop<1>' $O/loop_pmc.txt $O/loop_pmc.json frames=8192 'kernel_tag=k_loop<1>' round=$TAG > $O/pmc_summary.log 2>&1
rm -rf $O/prof $O/prof_tr
find $O/pmc -name '*.db' -delete
du -sh $O
tail -3 $O/pytest_gpu.txt; cat $O/bench_n1.json; cat $O/train_step.jsonl; head -14 $O/train_kernel_stats.txt | cut -c1-180
