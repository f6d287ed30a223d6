#!/bin/bash
# matrix-core counters of the solve kernel alone (one launch of the bench batch): the pass tools/profile.sh also runs, for adding the fields to an
# existing profiles/pmc_traffic.json of the same kernel sources.  usage (GPU box): bash tools/pmc_mfma.sh <tag>  -> gpurun_out/<tag>/pmc_mfma.txt
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/mfma -o mfma -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $OUT/mfma.json 2> $OUT/mfma.err
f=$(find $OUT/mfma -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/pmc_mfma.txt
import sys, csv, collections, re
agg = collections.defaultdict(float); calls = collections.defaultdict(int)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row['Kernel_Name'].replace(' ', '')
        if re.search(r'uph_solver_kernel<\d+,\d+,2(,(false|true))?>', k) or re.search(r'uph_solver_kernelILi\d+ELi\d+ELi2E', k):
            agg[row['Counter_Name']] += float(row['Counter_Value']); calls[row['Counter_Name']] += 1
for k, v in sorted(agg.items()): print('%-28s %.6g   (dispatch rows %d)' % (k, v, calls[k]))
PY
rm -rf $OUT/mfma
