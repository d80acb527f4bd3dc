# GPU box: FastSpeech2 forward row - bench line + rocprofv3 kernel stats of the same command
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-fs2prof}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R

timeout 300 python bench.py --row fs2 --steps 20 --warmup 3 > $O/bench_row_fs2.json 2> $O/bench_row_fs2.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o fs2 -- python $R/bench.py --row fs2 --steps 20 --warmup 3 > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/fs2_kernel_stats.txt 2>> $O/prof.log
rm -rf $O/prof
cd $R
cut -c1-400 $O/bench_row_fs2.json; head -60 $O/fs2_kernel_stats.txt | cut -c1-190
# FETCH / WRITE / SQ passes over the mel-rate ffn_1 launches (k_fs_conv<2> dispatches of at least 250 us) -> fs2_ffn1_pmc.json
git_rev=$(cat $R/.git_rev 2>/dev/null || echo unknown)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc/fetch -o fetch -- python $R/bench.py --row fs2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/pmc/write -o write -- python $R/bench.py --row fs2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/pmc/sq -o sq -- python $R/bench.py --row fs2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc 'k_fs_conv<2>' $O/fs2_ffn1_pmc.txt $O/fs2_ffn1_pmc.json 'kernel_tag=k_fs_conv<2> (ffn_1, 256 -> 1024, k = 9, 8 x 1024 frames)' round=$TAG commit=$git_rev min_us=250 > $O/pmc_summary.log 2>&1
rm -rf $O/pmc
cd $R
tail -14 $O/fs2_ffn1_pmc.txt
