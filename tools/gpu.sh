# GPU box: ONE script for every measurement run of this repo (round 6: the 21 gpu_*.sh of rounds 1-5 collapsed).
#   usage (from the build container):  gpurun --timeout S -- 'bash tools/gpu.sh <tag> <section> [<section> ...]'
# Results land in gpurun_out/<tag>/; every JSON summary carries the library's build id (dsd_build_id = sha256 of csrc/ + include/) and the tag.
# Sections:
#   suite      the whole GPU test suite + smoke()                        voc        vocoder tests + chain variants A/B + row bench + kernel stats
#   bench      the driver's command (N = 1) + rocprofv3 kernel stats     vocpmc     PMC passes over the vocoder row's chain kernels
#   rows       bench.py --row vocoder | fs2 | train                      fs2        FastSpeech2 tests + row bench + kernel stats by grid
#   looppmc    the three PMC passes over the headline kernel + timeline  train      training tests + row bench + kernel stats
#   sweep      shape sweep + all-config throughput                       probes     machine ceilings (bare MFMA stream, HBM)
#   fs2pmc     PMC passes over the mel-rate ffn_1 launches of FastSpeech2  trainpmc   PMC passes over the training step's kernels
#   parity     whole-batch oracle parity of BASELINE configs[2] / [3] (tools/bench_configs.py --parity: 1-2 min of host time)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r6_00}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
BID=$(python -c "from diffsinger_amd.build import binary_id; print(binary_id())")
echo "build id $BID tag $TAG sections $*" > $O/run.txt

stats() {   # stats <name> <cmd...>: rocprofv3 kernel-trace summary of a command -> $O/<name>_kernel_stats.txt
    local name=$1; shift
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o p -- "$@" > $O/prof_$name.log 2>&1 )
    python tools/rocprof_summary.py $(ls $O/prof_$name/*.db $O/prof_$name/*/*.db 2>/dev/null | head -1) > $O/${name}_kernel_stats.txt 2>> $O/prof_$name.log
    rm -rf $O/prof_$name
}
pmc3() {    # pmc3 <dir> <cmd...>: the three counter passes (separate --pmc runs, kernel-trace only) of a command -> $O/<dir>/{fetch,write,sq}
    local d=$1; shift
    ( cd /tmp
      timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/$d/fetch -o fetch -- "$@" > $O/${d}_fetch.log 2>&1
      timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/$d/write -o write -- "$@" > $O/${d}_write.log 2>&1
      timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/$d/sq -o sq -- "$@" > $O/${d}_sq.log 2>&1 )
    find $O/$d -name '*.db' -delete
}

for sec in "$@"; do case $sec in
suite)
    ( time timeout 1500 python -m pytest tests -m gpu -q -rf --durations=30 > $O/pytest_gpu_full.txt 2>&1 ) 2> $O/pytest_gpu_time.txt
    tail -80 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt; cat $O/pytest_gpu_time.txt >> $O/pytest_gpu.txt
    python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
    tail -6 $O/pytest_gpu.txt | cut -c1-220; tail -3 $O/smoke.txt ;;
bench)
    timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
    stats bench_n1 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg5-shard --no-secondary --no-extras
    cut -c1-900 $O/bench_n1.json; tail -3 $O/bench_n1.err ;;
rows)
    for row in vocoder fs2 train; do
        timeout 300 python bench.py --row $row --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_row_$row.json 2> $O/bench_row_$row.err
        cut -c1-330 $O/bench_row_$row.json; echo
    done ;;
voc)
    timeout 900 python -m pytest tests/test_gpu_vocoder.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_vocoder.txt
    timeout 300 python tools/voc_variants.py 10 > $O/voc_chain_variants.jsonl 2> $O/voc_chain_variants.err
    timeout 300 python bench.py --row vocoder --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_row_vocoder.json 2> $O/bench_row_vocoder.err
    stats vocoder python $R/bench.py --row vocoder --steps 5 --warmup 2 --no-cpu-baseline
    timeout 200 python tools/voc_chain_timeline.py > $O/voc_chain_timeline.txt 2>&1
    tail -4 $O/pytest_vocoder.txt; cat $O/voc_chain_variants.jsonl; tail -3 $O/voc_chain_variants.err; cut -c1-400 $O/bench_row_vocoder.json; echo
    head -12 $O/vocoder_kernel_stats.txt | cut -c1-190 ;;
vocpmc)
    pmc3 pmc_voc python $R/bench.py --row vocoder --steps 3 --warmup 1 --no-cpu-baseline
    for k in 8 16 32; do
        python tools/pmc_summary.py $O/pmc_voc "k_voc_chain<$k" $O/voc_chain_${k}ch_pmc.txt $O/voc_chain_${k}ch_pmc.json "kernel_tag=k_voc_chain<$k" round=$TAG >> $O/pmc_summary.log 2>&1
    done
    tail -12 $O/voc_chain_32ch_pmc.txt ;;
fs2)
    timeout 900 python -m pytest tests/test_gpu_fs2.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_fs2.txt
    timeout 300 python bench.py --row fs2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_row_fs2.json 2> $O/bench_row_fs2.err
    stats fs2 python $R/bench.py --row fs2 --steps 5 --warmup 2 --no-cpu-baseline
    tail -4 $O/pytest_fs2.txt; cut -c1-400 $O/bench_row_fs2.json; echo; head -14 $O/fs2_kernel_stats.txt | cut -c1-190 ;;
train)
    timeout 1200 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_train.txt
    timeout 300 python bench.py --row train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_row_train.json 2> $O/bench_row_train.err
    stats train python $R/tools/bench_train.py 8 --hip-only 8x1024
    tail -4 $O/pytest_train.txt; cut -c1-400 $O/bench_row_train.json; echo; head -14 $O/train_kernel_stats.txt | cut -c1-190 ;;
looppmc)
    timeout 200 python tools/loop_timeline.py $O/loop_timeline.json round=$TAG > $O/loop_timeline.txt 2>&1
    pmc3 pmc python $R/tools/profile_loop.py 3
    python tools/pmc_summary.py $O/pmc 'k_loop_wino' $O/loop_pmc.txt $O/loop_pmc.json frames=8192 'kernel_tag=k_loop_wino<1, 4>' round=$TAG > $O/pmc_summary.log 2>&1
    tail -12 $O/loop_pmc.txt; tail -5 $O/loop_timeline.txt ;;
fs2pmc)
    pmc3 pmc_fs2 python $R/bench.py --row fs2 --steps 3 --warmup 1 --no-cpu-baseline
    python tools/pmc_summary.py $O/pmc_fs2 'k_fs_conv<2>' $O/fs2_ffn1_pmc.txt $O/fs2_ffn1_pmc.json 'kernel_tag=k_fs_conv<2> (ffn_1, 256 -> 1024, k = 9, 8 x 1024 frames)' round=$TAG min_us=250 >> $O/pmc_summary.log 2>&1
    python tools/pmc_summary.py $O/pmc_fs2 'k_fs_attn<128>' $O/fs2_attn_pmc.txt $O/fs2_attn_pmc.json 'kernel_tag=k_fs_attn<128> (mel rate, 8 x 1024 frames)' round=$TAG min_us=30 >> $O/pmc_summary.log 2>&1
    tail -12 $O/fs2_ffn1_pmc.txt ;;
trainpmc)
    pmc3 pmc_tr python $R/tools/bench_train.py 3 --hip-only 8x1024
    python tools/pmc_summary.py $O/pmc_tr 'k_tr_wgrad<false>' $O/train_wgrad_pmc.txt $O/train_wgrad_pmc.json 'kernel_tag=k_tr_wgrad<false>' shape=8x1024 round=$TAG >> $O/pmc_summary.log 2>&1
    python tools/pmc_summary.py $O/pmc_tr 'k_trb_fused_w<false, true>' $O/train_trb_fused_w_pmc.txt $O/train_trb_fused_w_pmc.json 'kernel_tag=k_trb_fused_w<false, true>' shape=8x1024 round=$TAG >> $O/pmc_summary.log 2>&1
    python tools/pmc_summary.py $O/pmc_tr 'k_tr_stack_fwd_w' $O/train_stack_fwd_w_pmc.txt $O/train_stack_fwd_w_pmc.json 'kernel_tag=k_tr_stack_fwd_w' shape=8x1024 round=$TAG >> $O/pmc_summary.log 2>&1
    tail -12 $O/train_wgrad_pmc.txt ;;
parity)
    timeout 1500 python tools/bench_configs.py 1 --parity > $O/configs_parity.jsonl 2> $O/configs_parity.err
    cut -c1-300 $O/configs_parity.jsonl; tail -3 $O/configs_parity.err ;;
touch)
    for rep in 1 2; do timeout 300 python tools/wino_ab.py --touch=16,32,48,24,16,32 2>/dev/null | grep "^{" >> $O/wino_ab_touch.jsonl; done
    cut -c1-210 $O/wino_ab_touch.jsonl ;;
sweep)
    timeout 600 python tools/shape_sweep.py 3 1x512,1x1000,1x1550,4x777,2x2048,3x1550,1x5000,1x8000,6x1024,8x1024,3x5000,16x2048 > $O/shape_sweep.jsonl 2> $O/shape_sweep.err
    timeout 600 python tools/bench_configs.py 3 > $O/configs_throughput.jsonl 2> $O/configs_throughput.err
    cut -c1-260 $O/shape_sweep.jsonl; cut -c1-300 $O/configs_throughput.jsonl ;;
probes)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe4 tools/mfma_probe4.hip && timeout 120 /tmp/mfma_probe4 > $O/mfma_probe4.txt 2>&1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_probe tools/hbm_probe.hip && timeout 120 /tmp/hbm_probe > $O/hbm_probe.txt 2>&1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_filler_probe tools/mfma_filler_probe.hip && timeout 180 /tmp/mfma_filler_probe > $O/mfma_filler_probe.jsonl 2>&1
    tail -8 $O/mfma_probe4.txt; tail -4 $O/hbm_probe.txt ;;
*) echo "unknown section $sec" ;;
esac; done
du -sh $O
