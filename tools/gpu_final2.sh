set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_17; mkdir -p $O; cd $R
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp
mkdir -p $O/pmc_split
cp -r $R/gpurun_out/r4_16/pmc_split/sq $O/pmc_split/ 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_split/fetch -o fetch -- python $R/tools/profile_loop.py 3 --split > $O/pmc_split_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/pmc_split/write -o write -- python $R/tools/profile_loop.py 3 --split > $O/pmc_split_write.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_split 'k_loop_split<1' $O/loop_split_pmc.txt $O/loop_split_pmc.json frames=8192 'kernel_tag=k_loop_split<1, 2>' round=r4_17 > $O/pmc_summary.log 2>&1
find $O/pmc_split -name '*.db' -delete
python -c "
import json
d=json.load(open('$O/bench_n1.json'))
print(d['value'], d['roofline']['frac'], d['parity']['max_abs_mel_err'])
s=d['secondary']; print(s['value'], s['roofline']['frac'], json.dumps(s['parity'])[:900])
"
cat $O/loop_split_pmc.txt | cut -c1-160
