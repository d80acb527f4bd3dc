# GPU box: the closing evidence of a round for the CURRENT binary - the default bench line (headline + cfg5 shard + the labelled `secondary`),
# rocprofv3 kernel stats of the same command, and three PMC passes over the split loop (FETCH_SIZE, WRITE_SIZE and the SQ counters in separate runs: FETCH_SIZE + WRITE_SIZE in one pass aborts rocprofv3 here) (separate --pmc runs, kernel-trace only).
#   usage: bash tools/gpu_final.sh <tag>
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-final}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg5-shard > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/bench_n1_kernel_stats.txt 2>> $O/prof.log
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_split/fetch -o fetch -- python $R/tools/profile_loop.py 3 --split > $O/pmc_split_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/pmc_split/write -o write -- python $R/tools/profile_loop.py 3 --split > $O/pmc_split_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU -d $O/pmc_split/sq -o sq -- python $R/tools/profile_loop.py 3 --split > $O/pmc_split_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_split 'k_loop_split<1' $O/loop_split_pmc.txt $O/loop_split_pmc.json frames=8192 'kernel_tag=k_loop_split<1, 2>' round=$TAG > $O/pmc_summary.log 2>&1
rm -rf $O/prof; find $O/pmc_split -name '*.db' -delete
python -c "
import json
d=json.load(open('$O/bench_n1.json'))
print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','roofline','cpu_baseline','cfg5_shard_n1')}, indent=1)[:2500])
s=d['secondary']; print(json.dumps({k:s.get(k) for k in ('value','ms_per_step','dtype','format','roofline','parity')}, indent=1)[:3000])
print(d['parity']['max_abs_mel_err'], d['parity']['kernel'])
"
head -12 $O/bench_n1_kernel_stats.txt | cut -c1-200; cat $O/loop_split_pmc.txt | cut -c1-160; tail -2 $O/bench_n1.err
