# GPU box: the evidence round of the CURRENT binary - whole GPU suite + smoke, headline bench (+ CPU baseline), k_loop timeline, rocprofv3 kernel
# stats of the bench command, the three PMC passes over k_loop (separate --pmc runs, kernel-trace only) -> loop_pmc.json, machine ceilings
# (bare fp32 MFMA stream, HBM read / copy), shape sweep, row benches (vocoder / fs2 / train), all-config throughput, one utterance end to end,
# ParallelWaveGAN, vocoder and FastSpeech2 kernel stats + PMC passes.
#   usage: bash tools/gpu_evidence.sh <tag> [nopytest|pytest] [core]      (core: skip the vocoder / FastSpeech2 PMC passes: those kernels did not change)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r06}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
git_rev=$(cat .git_rev 2>/dev/null || echo unknown)
if [ "$2" != "nopytest" ]; then
( time timeout 1500 python -m pytest tests -m gpu -q -rf --durations=30 > $O/pytest_gpu_full.txt 2>&1 ) 2> $O/pytest_gpu_time.txt; tail -80 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt; cat $O/pytest_gpu_time.txt >> $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
fi
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 200 python tools/loop_timeline.py $O/loop_timeline.json > $O/loop_timeline.txt 2>&1
# machine ceilings: the probes are compiled on the box (no binaries in the tree)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe4 tools/mfma_probe4.hip && timeout 120 /tmp/mfma_probe4 > $O/mfma_probe4.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_probe tools/hbm_probe.hip && timeout 120 /tmp/hbm_probe > $O/hbm_probe.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_filler_probe tools/mfma_filler_probe.hip && timeout 180 /tmp/mfma_filler_probe > $O/mfma_filler_probe.jsonl 2>&1
timeout 200 python tools/wino_ab.py --shapes 2>/dev/null | grep "^{" > $O/wino_ab.jsonl
for row in vocoder fs2 train; do
timeout 300 python bench.py --row $row --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_row_$row.json 2> $O/bench_row_$row.err
done
[ "$3" != "core" ] && timeout 200 python tools/bench_pwg.py 5 > $O/bench_pwg.jsonl 2> $O/bench_pwg.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg5-shard > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) > $O/bench_n1_kernel_stats.txt 2>> $O/prof.log
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc/fetch -o fetch -- python $R/tools/profile_loop.py 3 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc/write -o write -- python $R/tools/profile_loop.py 3 > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/pmc/sq -o sq -- python $R/tools/profile_loop.py 3 > $O/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc 'k_loop_wino' $O/loop_pmc.txt $O/loop_pmc.json frames=8192 'kernel_tag=k_loop_wino<1, 4>' round=$TAG commit=$git_rev > $O/pmc_summary.log 2>&1
# training row: the step with the Winograd / direct convolution kernels on three shapes, kernel times of the step, PMC passes over the step
# (k_tr_wgrad<false> -> train_wgrad_pmc.json for bench.py --row train; the Winograd forward and data-gradient kernels beside it)
timeout 300 python $R/tools/bench_train.py 10 --conv-ab 2>/dev/null | grep "^{" > $O/train_conv_ab.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_tr -o tr -- python $R/tools/bench_train.py 8 --hip-only 8x1024 > $O/prof_tr.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof_tr/*.db $O/prof_tr/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats.txt 2>> $O/prof_tr.log
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_tr/fetch -o fetch -- python $R/tools/bench_train.py 3 --hip-only 8x1024 > $O/pmc_tr_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $O/pmc_tr/write -o write -- python $R/tools/bench_train.py 3 --hip-only 8x1024 > $O/pmc_tr_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/pmc_tr/sq -o sq -- python $R/tools/bench_train.py 3 --hip-only 8x1024 > $O/pmc_tr_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_tr 'k_tr_wgrad<false>' $O/train_wgrad_pmc.txt $O/train_wgrad_pmc.json 'kernel_tag=k_tr_wgrad<false>' shape=8x1024 round=$TAG commit=$git_rev >> $O/pmc_summary.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_tr 'k_trb_fused_w<false, true>' $O/train_trb_fused_w_pmc.txt $O/train_trb_fused_w_pmc.json 'kernel_tag=k_trb_fused_w<false, true>' shape=8x1024 round=$TAG commit=$git_rev >> $O/pmc_summary.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_tr 'k_tr_stack_fwd_w' $O/train_stack_fwd_w_pmc.txt $O/train_stack_fwd_w_pmc.json 'kernel_tag=k_tr_stack_fwd_w' shape=8x1024 round=$TAG commit=$git_rev >> $O/pmc_summary.log 2>&1
rm -rf $O/prof_tr; find $O/pmc_tr -name '*.db' -delete
# the training row once more, now with THIS binary's traffic figure behind it
[ -s $O/train_wgrad_pmc.json ] && cp $O/train_wgrad_pmc.json $R/profiles/train_wgrad_pmc.json && ( cd $R && timeout 300 python bench.py --row train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_row_train.json 2> $O/bench_row_train.err )
cd $R
# (the long sweeps last: a call that runs out of budget has the headline evidence already)
timeout 600 python tools/shape_sweep.py 3 1x512,1x1000,1x1550,4x777,2x2048,3x1550,1x5000,1x8000,6x1024,8x1024,3x5000,16x2048 --lat-splits > $O/shape_sweep.jsonl 2> $O/shape_sweep.err
timeout 400 python tools/bench_configs.py 3 > $O/configs_throughput.jsonl 2> $O/configs_throughput.err
timeout 200 python tools/bench_single.py 10 100 8 > $O/single_utterance.jsonl 2> $O/single_utterance.err
timeout 200 python tools/bench_single.py 10 60 8 >> $O/single_utterance.jsonl 2>> $O/single_utterance.err
cd /tmp
if [ "$3" != "core" ]; then
# vocoder row: kernel stats + FETCH / WRITE / SQ passes over the fused resblock-stage kernels (bench.py --row vocoder reads voc_chain_32ch_pmc.json)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_voc -o voc -- python $R/bench.py --row vocoder --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_voc.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/prof_voc/*.db $O/prof_voc/*/*.db 2>/dev/null | head -1) > $O/vocoder_kernel_stats.txt 2>> $O/prof_voc.log
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_voc/fetch -o fetch -- python $R/bench.py --row vocoder --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_voc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/pmc_voc/write -o write -- python $R/bench.py --row vocoder --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_voc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/pmc_voc/sq -o sq -- python $R/bench.py --row vocoder --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_voc_sq.log 2>&1
for k in 8 16 32; do
python $R/tools/pmc_summary.py $O/pmc_voc "k_voc_chain<$k" $O/voc_chain_${k}ch_pmc.txt $O/voc_chain_${k}ch_pmc.json "kernel_tag=k_voc_chain<$k" round=$TAG commit=$git_rev >> $O/pmc_summary.log 2>&1
done
timeout 200 python $R/tools/voc_chain_timeline.py > $O/voc_chain_timeline.txt 2>&1
# FastSpeech2 row: kernel stats + the PMC passes over the mel-rate ffn_1 launches (bench.py --row fs2 reads fs2_ffn1_pmc.json)
cd $R; bash tools/gpu_fs2_prof.sh $TAG/fs2 > $O/fs2_prof.log 2>&1; cp $O/fs2/fs2_ffn1_pmc.json $O/fs2/fs2_ffn1_pmc.txt $O/fs2/fs2_kernel_stats.txt $O/ 2>/dev/null; rm -rf $O/fs2
fi
rm -rf $O/prof $O/prof_voc
find $O/pmc $O/pmc_voc -name '*.db' -delete
du -sh $O
tail -15 $O/pytest_gpu.txt | cut -c1-220; tail -3 $O/smoke.txt; cut -c1-600 $O/bench_n1.json; cat $O/shape_sweep.jsonl | cut -c1-260; for row in vocoder fs2 train; do cut -c1-330 $O/bench_row_$row.json; echo; done
cat $O/mfma_probe4.txt | tail -8; cat $O/hbm_probe.txt | tail -4; tail -5 $O/loop_pmc.txt
