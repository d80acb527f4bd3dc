# GPU box: FastSpeech2 option cases + dsf_linear_rows + FS2 row profile / PMC
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r22}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fs2.py tests/test_gpu_fs2_train.py tests/test_gpu_train.py tests/test_gpu_train_fused.py tests/test_gpu_surfaces.py -m gpu -q -s > $O/pytest_fs2_train.txt 2>&1
grep -v amdgpu $O/pytest_fs2_train.txt | grep -i "linear_rows\|spk\|_ph_\|passed\|failed\|Error" | cut -c1-260 | tail -40
timeout 300 python bench.py --row train --steps 20 --warmup 3 > $O/bench_row_train.json 2> $O/bench_row_train.err; cut -c1-330 $O/bench_row_train.json
bash tools/gpu_fs2_prof.sh $TAG > $O/fs2_prof.log 2>&1
tail -16 $O/fs2_ffn1_pmc.txt; cut -c1-300 $O/bench_row_fs2.json
