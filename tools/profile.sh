# usage: [WL=astar] bash tools/profile.sh <tag>   (on the GPU box)  -> gpurun_out/prof_<tag>/ ; copy the summaries into profiles/<tag>_*
# WL=astar: the same passes for `bench.py --workload astar` (its own pmc_traffic_astar.json; no calibration / penalty-kernel passes)
# 1. bench line un-profiled  2. rocprofv3 --kernel-trace --stats of the same command (no counters)  3. separate --pmc passes (counters only)
# 4. FETCH_SIZE / WRITE_SIZE calibration on known-byte microkernels (tools/micro/fetch_calib.hip)  5. penalty-kernel-only trace (configs[1])
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
WLF=""; [ "${WL:-hill}" = astar ] && WLF="--workload astar"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
python bench.py $WLF --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py $WLF --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/bench_profiled.json 2> $OUT/rocprof.err
for f in $(find $OUT/kt -name "*kernel_stats*.csv"); do cp $f $OUT/kernel_stats.csv; done
head -6 $OUT/kernel_stats.csv | cut -c1-200
if [ -z "$WLF" ]; then
# penalty kernel alone: 3 launches of uph_eval_batch(repeat = 20) on the batch, then 3 of uph_penalty_batch (A5 alone)
cat > /tmp/evalonly.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(16384, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
opt = U.ALMTrajOpt(m); opt.upload(probs); opt.init_scaling_batch()
for _ in range(3): opt.eval_batch(None, repeat=20)
print("eval kernel ms", opt.stats()["kernel_ms"])
for _ in range(3): opt.penalty_batch(repeat=20, store_residuals=True)
print("A5-only kernel ms", opt.stats()["kernel_ms"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ev -o ev -- python /tmp/evalonly.py > $OUT/eval_only.txt 2> $OUT/eval_only.err
for f in $(find $OUT/ev -name "*kernel_stats*.csv"); do cp $f $OUT/eval_kernel_stats.csv; done
grep "Li0EE\|, 0>\|Li8EE\|, 8>" $OUT/eval_kernel_stats.csv | cut -c1-200
fi
pmc() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $GRAFT_REPO_ROOT/bench.py $WLF --steps 1 --warmup 0 --no-cpu --no-extras > $OUT/$name.json 2> $OUT/$name.err
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python - "$f" "$OUT/pmc_summary.txt" <<'PY'
import sys, csv, collections, re
agg = collections.defaultdict(float); calls = collections.defaultdict(int)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row['Kernel_Name'].replace(' ', '')
        if re.search(r'uph_solver_kernel<\d+,\d+,2(,(false|true))?>', k) or re.search(r'uph_solver_kernelILi\d+ELi\d+ELi2E', k):     # MODE 2 = the solve
            agg[row['Counter_Name']] += float(row['Counter_Value']); calls[row['Counter_Name']] += 1
with open(sys.argv[2], 'a') as out:
    for k, v in sorted(agg.items()):
        line = '%-24s %.6g   (dispatch rows %d)' % (k, v, calls[k]); print(line); out.write(line + '\n')
PY
}
rm -f $OUT/pmc_summary.txt
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
pmc mfma SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES
# calibration: counted vs requested bytes on known patterns
rm -f $OUT/fetch_calibration.txt
[ -n "$WLF" ] || for pat in gather8 rows16 stream16 stream8 write8; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/cal_${pat}_$ctr -o c -- $GRAFT_REPO_ROOT/build/micro/fetch_calib $pat > $OUT/cal_$pat.txt 2>> $OUT/cal.err
    f=$(find $OUT/cal_${pat}_$ctr -name "*counter_collection.csv" | head -1)
    python - "$f" $pat $ctr "$(cat $OUT/cal_$pat.txt)" >> $OUT/fetch_calibration.txt <<'PY'
import sys, csv
tot = 0.0
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if sys.argv[2] in row['Kernel_Name'] and row['Counter_Name'] == sys.argv[3]: tot += float(row['Counter_Value'])
req = float(sys.argv[4].split()[-1])
print('%-9s %-10s counted %.6g KiB = %.6g bytes   requested %.6g bytes   counted/requested %.3f' % (sys.argv[2], sys.argv[3], tot, tot * 1024, req, tot * 1024 / req))
PY
  done
done
[ -n "$WLF" ] || cat $OUT/fetch_calibration.txt
python - $OUT <<'PY'
import sys, json
out = sys.argv[1]
vals = {}
for line in open(out + '/pmc_summary.txt'):
    k = line.split(); vals[k[0]] = float(k[1])
line = json.loads(open(out + '/bench.json').read().strip().split('\n')[-1])
B = int(line['config']['batch_per_gpu'])
import os, subprocess
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
try: head = subprocess.check_output(['git', '-C', os.environ['GRAFT_REPO_ROOT'], 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
except Exception: head = os.environ.get('UPH_GIT_HEAD', 'unknown (the GPU box holds a snapshot without .git)')
json.dump({'batch': B, 'tag': 'profiles/%s_pmc_summary.txt' % os.path.basename(out).replace('prof_', ''), 'fetch_kib': vals.get('FETCH_SIZE', 0.0), 'write_kib': vals.get('WRITE_SIZE', 0.0), 'launches': 1,
           'sq_active_inst_valu': vals.get('SQ_ACTIVE_INST_VALU', 0.0), 'sq_wave_cycles': vals.get('SQ_WAVE_CYCLES', 0.0), 'sq_wait_any': vals.get('SQ_WAIT_ANY', 0.0), 'sq_insts_valu': vals.get('SQ_INSTS_VALU', 0.0),
           'sq_insts_valu_mfma_f64': vals.get('SQ_INSTS_VALU_MFMA_F64', 0.0), 'sq_insts_valu_mfma_mops_f64': vals.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0.0), 'sq_valu_mfma_busy_cycles': vals.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0),
           'launch_ms': line['roofline']['avg_launch_ms'], 'kernel_src_sha': bench.kernel_sources_sha(), 'git_head': head,
           'note': 'ALM/L-BFGS solve kernel (uph_solver_kernel<*,2,2>), one launch of bench.py --steps 1 --warmup 0 (default batch); rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; see <tag>_fetch_calibration.txt for counted/requested on known patterns'},
          open(out + ('/pmc_traffic_astar.json' if os.environ.get('WL') == 'astar' else '/pmc_traffic.json'), 'w'), indent=1)
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace*" -size +2M -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete; find $OUT -name "*agent_info*" -delete
