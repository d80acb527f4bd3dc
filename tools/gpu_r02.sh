# GPU box, round 2: the full evidence round of the CURRENT tree - parity tests, headline bench (+CPU baseline), configs[4] at N=1, the
# launch contract under torch.distributed.run, phase timeline of the persistent loop incl. its head, shape sweep (automatic path choice next
# to the forced paths), rocprofv3 kernel stats of the bench command, PMC passes over k_loop (separate --pmc runs, kernel-trace only).
#   usage: bash tools/gpu_r02.sh <tag> [quick]
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02b}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
HEAD=$(cat gpurun_head.txt 2>/dev/null || echo unknown)
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -150 > $O/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 200 python tools/loop_timeline.py $O/loop_timeline.json > $O/loop_timeline.txt 2>&1
timeout 100 python tools/plms_diag.py > $O/plms_diag.txt 2>&1
timeout 400 python tools/shape_sweep.py 2 > $O/shape_sweep.jsonl 2> $O/shape_sweep.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_lat -o lat -- python $R/tools/shape_sweep.py 2 1x512,1x1550 --default-only > $O/prof_lat.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof_lat/*.db $O/prof_lat/*/*.db 2>/dev/null | head -1) > $O/latency_kernel_stats.txt 2>> $O/prof_lat.log
rm -rf $O/prof_lat
cd $R
timeout 200 python bench.py --row vocoder --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_vocoder_row.json 2> $O/bench_vocoder_row.err
timeout 200 python bench.py --row fs2 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_fs2_row.json 2> $O/bench_fs2_row.err
if [ "$2" != "quick" ]; then
timeout 300 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cfg5_n1.json 2> $O/bench_cfg5_n1.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err
timeout 400 python tools/bench_configs.py 3 > $O/configs_throughput.jsonl 2> $O/configs_throughput.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/bench_n1_kernel_stats.txt 2>> $O/prof.log
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc/fetch -o fetch -- python $R/tools/profile_loop.py 3 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc/write -o write -- python $R/tools/profile_loop.py 3 > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/pmc/sq -o sq -- python $R/tools/profile_loop.py 3 > $O/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc 'k_loop<1>' $O/loop_pmc.txt $O/loop_pmc.json frames=8192 'kernel_tag=k_loop<1>' round=$TAG commit=$HEAD > $O/pmc_summary.log 2>&1
rm -rf $O/prof
find $O/pmc -name '*.db' -delete
fi
cd $R
du -sh $O
tail -60 $O/pytest_gpu.txt | cut -c1-220; cut -c1-1800 $O/bench_n1.json; tail -3 $O/bench_n1.err; cat $O/loop_timeline.txt; cat $O/plms_diag.txt; cat $O/shape_sweep.jsonl; tail -3 $O/shape_sweep.err
head -12 $O/latency_kernel_stats.txt | cut -c1-150; python -c "import json,sys; [print(f, json.load(open('$O/'+f)).get('hipgraph_replay'), json.load(open('$O/'+f))['ms_per_step']) for f in ('bench_vocoder_row.json','bench_fs2_row.json')]"; tail -2 $O/bench_vocoder_row.err $O/bench_fs2_row.err; cut -c1-900 $O/bench_cfg5_n1.json; tail -3 $O/bench_cfg5_n1.err; cut -c1-400 $O/bench_torchrun_n1.json; tail -3 $O/bench_torchrun_n1.err; head -12 $O/bench_n1_kernel_stats.txt | cut -c1-160; cat $O/loop_pmc.txt | tail -12
