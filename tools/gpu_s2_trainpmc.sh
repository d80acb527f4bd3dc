# GPU box: PMC passes (separate --pmc runs, kernel-trace only) over the kernels of the fused training step, 8 x 1024 frames
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02s}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc/sq -o sq -- python $R/tools/bench_train.py 3 --hip-only 8x1024 > $O/sq.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc/fetch -o fetch -- python $R/tools/bench_train.py 3 --hip-only 8x1024 > $O/fetch.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc/write -o write -- python $R/tools/bench_train.py 3 --hip-only 8x1024 > $O/write.log 2>&1
cd $R
for k in ${KERNELS:-"k_tr_wgrad<false>" "k_trb_fused<false>" "k_tr_stack_fwd"}; do
  python tools/pmc_summary.py $O/pmc "$k" "$O/pmc_$(echo $k | tr -d '<>').txt" "$O/pmc_$(echo $k | tr -d '<>').json" round=$TAG shape=8x1024 > /dev/null 2>> $O/pmc_err.txt
done
rm -rf $O/pmc
ls $O; cat $O/pmc_k_tr_stack_fwd.txt; tail -3 $O/pmc_err.txt
