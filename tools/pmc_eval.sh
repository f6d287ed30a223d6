#!/bin/bash
# instruction counters of the penalty kernel alone (uph_eval_batch, MODE 0, `repeat` evaluations per trajectory and launch) and of the solve
# kernel.  usage (GPU box): bash tools/pmc_eval.sh <tag> [variant] [set]   -> gpurun_out/<tag>/pmc_eval_<variant>[_<set>].txt
# counter sets (one rocprofv3 pass each, counters only): base (instruction counts), wait (where a wave's cycles go: SQ_WAIT_ANY = parked at
# s_waitcnt / s_barrier, SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_* = issuing, per pipe), lds (LDS and vector-memory queue levels,
# bank conflicts, matrix-core instructions)
cd $GRAFT_REPO_ROOT
TAG=$1; V=${2:-default}; SET=${3:-base}
case $SET in
  base) CTRS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU";;
  wait) CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA";;
  lds) CTRS="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64";;
  mfma) CTRS="SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU";;
esac
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
if [ "$V" != default ]; then export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$V.so; fi
export TMPDIR=/tmp
cat > /tmp/evalonly.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import uneven_planner_amd as U
from uneven_planner_amd import scenes
B, R = 8192, 20
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(B, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
opt = U.ALMTrajOpt(m); opt.upload(probs); opt.init_scaling_batch()
opt.eval_batch(None, repeat=R)
print("eval kernel ms", opt.stats()["kernel_ms"], "B", B, "R", R, "sumS", sum(s["S"] for s in opt._sizes))
PY
cd /tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_eval_${V}_$SET -o ev -- python /tmp/evalonly.py > $OUT/pmc_eval_${V}_$SET.log 2>&1
f=$(find $OUT/pmc_eval_${V}_$SET -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/pmc_eval_${V}_$SET.txt
import sys, csv, collections, re
agg = collections.defaultdict(float); n = collections.defaultdict(int)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row['Kernel_Name'].replace(' ', '')
        if re.search(r'uph_solver_kernel<\d+,\d+,0(,(false|true))?>', k) or re.search(r'uph_solver_kernelILi\d+ELi\d+ELi0E', k):
            agg[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
B, R = 8192, 20
for k, v in sorted(agg.items()):
    print('%-22s %.6g  (dispatches %d)  per trajectory-evaluation %.1f' % (k, v, n[k], v / max(1, n[k]) / (B * R)))
PY
tail -2 $OUT/pmc_eval_${V}_$SET.log
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete; find $OUT -name "*agent_info*" -delete
