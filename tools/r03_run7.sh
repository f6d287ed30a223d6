#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile.sh r03g 2>&1 | tail -60
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03g
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/lds -o lds -- python /tmp/evalonly.py > $OUT/lds.log 2>&1
f=$(find $OUT/lds -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/lds_summary.txt
import sys, csv, collections
agg = collections.defaultdict(float)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if 'uph_solver_kernel' in row['Kernel_Name']:
            agg[(row['Kernel_Name'][:60], row['Counter_Name'])] += float(row['Counter_Value'])
for k, v in sorted(agg.items()):
    print('%-62s %-22s %.6g' % (k[0], k[1], v))
PY
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --gpus 1 --single-process --steps 3 --warmup 1 > $OUT/bench_single_process.json 2> $OUT/bench_single_process.err; tail -c 1500 $OUT/bench_single_process.json
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
