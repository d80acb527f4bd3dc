# GPU box: PMC passes over the split-precision persistent loop k_loop_split at the bench shape (separate --pmc runs, kernel-trace only) and,
# beside them, the same counters over the fp32 k_loop: what keeps the split loop at 1.22 x (matrix pipe busy vs memory-pipe stalls, L2 traffic).
#   usage: bash tools/gpu_split_pmc.sh <tag>
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r4_04}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
for v in split f32; do
  FLAG=""; [ $v = split ] && FLAG="--split"
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_$v/fetch -o fetch -- python $R/tools/profile_loop.py 3 $FLAG > $O/pmc_${v}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc_$v/write -o write -- python $R/tools/profile_loop.py 3 $FLAG > $O/pmc_${v}_write.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/pmc_$v/sq -o sq -- python $R/tools/profile_loop.py 3 $FLAG > $O/pmc_${v}_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum -d $O/pmc_$v/tc -o tc -- python $R/tools/profile_loop.py 3 $FLAG > $O/pmc_${v}_tc.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc_split 'k_loop_split<1' $O/loop_split_pmc.txt $O/loop_split_pmc.json frames=8192 'kernel_tag=k_loop_split<1, 4>' round=$TAG > $O/pmc_summary.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_f32 'k_loop<1>' $O/loop_f32_pmc.txt $O/loop_f32_pmc.json frames=8192 'kernel_tag=k_loop<1>' round=$TAG >> $O/pmc_summary.log 2>&1
find $O -name '*.db' -delete
cat $O/loop_split_pmc.txt; cat $O/loop_f32_pmc.txt; tail -3 $O/pmc_split_tc.log
[ -x $R/tools/mfma_probe4.bin ] && timeout 120 $R/tools/mfma_probe4.bin > $O/mfma_probe4.txt 2>&1; tail -4 $O/mfma_probe4.txt
