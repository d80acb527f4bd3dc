# GPU box: targeted diagnostics of this round (full tracebacks)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r08}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_vocoder.py -m gpu -q -x 2>&1 | tail -60 > $O/pytest_vocoder.txt
timeout 300 python -m pytest tests/test_gpu_loop.py tests/test_gpu_train_fused.py -m gpu -q -x -k "starved or dcond_follows" 2>&1 | tail -80 > $O/pytest_fixed.txt
for rep in 1 2; do for mode in off stage; do
timeout 300 python bench.py --row vocoder --chain $mode --steps 10 --warmup 3 --no-cpu-baseline 2>> $O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'chain':'$mode','ms_per_step':d['ms_per_step']}))" >> $O/voc_chain_ab.jsonl
done; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o voc -- python $R/bench.py --row vocoder --steps 5 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/vocoder_kernel_stats.txt 2>> $O/prof.log
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc/fetch -o fetch -- python $R/bench.py --row vocoder --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/pmc/write -o write -- python $R/bench.py --row vocoder --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/pmc/sq -o sq -- python $R/bench.py --row vocoder --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
for k in 'k_voc_chain<8' 'k_voc_chain<16' 'k_voc_chain<32'; do
python $R/tools/pmc_summary.py $O/pmc "$k" "$O/voc_pmc_$k.txt" "$O/voc_pmc_$k.json" "kernel_tag=$k" round=$TAG > $O/pmc_summary.log 2>&1
done
rm -rf $O/prof; find $O/pmc -name '*.db' -delete
cd $R
cat $O/pytest_vocoder.txt | tail -8; cat $O/pytest_fixed.txt | tail -40; cat $O/voc_chain_ab.jsonl; head -14 $O/vocoder_kernel_stats.txt | cut -c1-190; tail -4 $O/voc_pmc_*.txt; tail -5 $O/err.txt
