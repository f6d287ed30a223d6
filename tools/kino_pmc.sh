#!/bin/bash
# counters of the front-end search kernel (one launch of B queries on the hill map); separate rocprofv3 --pmc passes, counters only.
# usage (GPU box): bash tools/kino_pmc.sh <tag> [B]   -> gpurun_out/<tag>/kino_pmc.txt
cd $GRAFT_REPO_ROOT
TAG=$1; B=${2:-8192}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/kino_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import uneven_planner_amd as U
from uneven_planner_amd import scenes
B = int(sys.argv[1])
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
S, G = scenes.random_queries(B, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
ka = U.KinoAstar(m)
r = ka.plan_batch(S, G, path_cap=64)
it = np.array([q["iter_num"] for q in r]); un = np.array([q["use_node_num"] for q in r])
print("B", B, "kernel_ms", ka.stats()["kernel_ms"], "expansions", int(it.sum()), "nodes", int(un.sum()), "slots", ka.slots)
PY
cd /tmp
: > $OUT/kino_pmc.txt
pass() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/kino_pmc_$name -o k -- python /tmp/kino_once.py $B > $OUT/kino_pmc_$name.log 2>&1
  f=$(find $OUT/kino_pmc_$name -name "*counter_collection.csv" | head -1)
  grep "^B " $OUT/kino_pmc_$name.log | tail -1 >> $OUT/kino_pmc.txt
  python - "$f" <<'PY' >> $OUT/kino_pmc.txt
import sys, csv, collections
agg = collections.defaultdict(float)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if 'uph_kino_kernel' in row['Kernel_Name']:
            agg[row['Counter_Name']] += float(row['Counter_Value'])
for k in sorted(agg): print("  %-24s %.6g" % (k, agg[k]))
PY
  rm -rf $OUT/kino_pmc_$name
}
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass lds SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVES
cat $OUT/kino_pmc.txt
