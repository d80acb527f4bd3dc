# GPU box: MFMA cadence probe + PMC passes over the residual-layer kernel (separate --pmc runs, kernel-trace only)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$1
mkdir -p $O
cd $R
timeout 120 tools/mfma_probe.bin > $O/mfma_probe.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/sq -o sq -- python $R/tools/profile_layer.py 8 1024 32 40 > $O/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/fetch -o fetch -- python $R/tools/profile_layer.py 8 1024 32 40 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/write -o write -- python $R/tools/profile_layer.py 8 1024 32 40 > $O/write.log 2>&1
ls -R $O | head
