#!/bin/bash
# counters of the plane-fit kernel (uph_map_build_kernel) on the hill cloud: what bounds the 5 ms build.  usage (GPU box): bash tools/pmc_map.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/mapbuild.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); xyz = scenes.make_hill_cloud()
m.build(xyz); m.build(xyz)
print("map kernel ms", m.build_stats())
PY
cd /tmp
rm -f $OUT/pmc_map.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_map_$name -o mp -- python /tmp/mapbuild.py > $OUT/pmc_map_$name.log 2>&1
  f=$(find $OUT/pmc_map_$name -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' | tee -a $OUT/pmc_map.txt
import sys, csv, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if 'uph_map_build_kernel' in row['Kernel_Name']:
            agg[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
for k, v in sorted(agg.items()):
    print('%-22s %.6g  per launch (of %d)' % (k, v / max(1, n[k]), n[k]))
PY
done
grep "map kernel" $OUT/pmc_map_FETCH_SIZE.log | tail -1 | tee -a $OUT/pmc_map.txt
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete; find $OUT -name "*agent_info*" -delete
