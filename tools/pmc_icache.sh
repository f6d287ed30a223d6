cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQ_WAIT_INST|SQ_INSTS_SMEM|SQC_" | head -40 > $OUT/counters.txt
cat $OUT/counters.txt | cut -c1-160
run() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/$name.json 2> $OUT/$name.err
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import sys, csv, collections
agg = collections.defaultdict(float)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if 'uph_solver_kernel' in row['Kernel_Name'] and 'Li2EE' in row['Kernel_Name'].replace(' ', '') or ('uph_solver_kernel' in row['Kernel_Name'] and ', 2>' in row['Kernel_Name']):
            agg[row['Counter_Name']] += float(row['Counter_Value'])
for k, v in sorted(agg.items()): print('%-28s %.6g' % (k, v))
PY
}
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run ic2 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_IFETCH SQ_ACTIVE_INST_ANY
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
