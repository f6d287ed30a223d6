#!/bin/bash
# The GPU-box command lists of round 4's gpurun calls, one case per call: gpurun -- "bash tools/r04_runs.sh <n>".
# 1 fp64 MFMA probe, GPU tier with the matrix-core xy scatter, A/B against the vector scatter (bench + phases + scatterChunk cycles),
#   barrier-wait profile, wait / LDS counter passes on the penalty kernel
set -u
case "${1:-}" in
1)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04a; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
build/micro/mfma_f64_probe > $OUT/mfma_probe.txt 2>&1; grep -v "4x4x4_4b D lane" $OUT/mfma_probe.txt
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -6 $OUT/gpu_tests.txt
bash tools/gpu_ab.sh r04a default nomfma 2>&1 | tee $OUT/ab.txt
for v in default nomfma; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  echo "== microbench $v"; timeout 200 python tools/microbench.py 2>&1 | tail -3
done | tee $OUT/microbench.txt
for v in barprof barprof_nomfma; do
  export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so
  echo "== $v"; UPH_BAR_PROF=1 timeout 200 python tools/phase_breakdown.py 8192 2>&1 | grep -E "cycles/eval|kernel_ms"
done | tee $OUT/barprof.txt
unset UNEVENHIP_LIB
for set in wait lds; do for v in default nomfma; do bash tools/pmc_eval.sh r04a $v $set; done; done 2>&1 | grep -v "^$" | tee $OUT/pmc_eval_sets.txt
;;
2)
# the batched kinodynamic search against the oracle (new), the tests whose assertions changed (statistical drift test, N4 occupancy vs the
# oracle, ADVICE fixes around the multi-GPU entries), then a short bench line with the new fields
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04b; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_kino.py -x -q > $OUT/kino_tests.txt 2>&1; echo "kino rc $?" >> $OUT/kino_tests.txt
tail -30 $OUT/kino_tests.txt
timeout 900 python -m pytest tests/test_gpu_map.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_buckets.py tests/test_gpu_vocano.py tests/test_gpu_km2.py tests/test_gpu_adapter.py -m gpu -x -q -s > $OUT/changed_tests.txt 2>&1; echo "changed rc $?" >> $OUT/changed_tests.txt
grep -E "drift|passed|failed|rc " $OUT/changed_tests.txt | cut -c1-400 | tail -20
timeout 600 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - $OUT/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", r["value"], "converged/s", r.get("converged_traj_opts_per_s"), "frac", r["roofline"]["frac"], "valu_busy", r["roofline"].get("valu_busy"), r["roofline"].get("valu_busy_source"))
print("front_end", json.dumps(r.get("front_end")))
print("drift", json.dumps(r.get("parity_floor", {}).get("drift")))
PY
tail -3 $OUT/bench.err
;;
3)
# kino tests again, the whole GPU tier (device-side cloud filter, empty-share fix), the stage breakdown of the map build, and the drift statistics at N = 1024
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04c; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -25 $OUT/gpu_tests.txt | cut -c1-300
python - <<'PY' | tee $OUT/map_stages.txt
import time, numpy as np
import uneven_planner_amd as U
from uneven_planner_amd import scenes
xyz = scenes.make_hill_cloud()
m = U.UnevenMap()
for k in range(3):
    t0 = time.perf_counter(); m.build(xyz, download=False); t1 = time.perf_counter(); m.download(); t2 = time.perf_counter()
    print("build %d: uph_map_build %.2f ms (stages %s)  + download of cells / c / occupancy into numpy %.2f ms" % (k, (t1 - t0) * 1e3, {a: round(b, 3) for a, b in m.build_stats()["stages_ms"].items()}, (t2 - t1) * 1e3))
PY
UPH_PB_ONLY_YAML=1 timeout 900 python tools/parity_buckets.py 1024 $OUT/parity_buckets_hill_1024.json hill > $OUT/parity_buckets_hill_1024.txt 2>&1
tail -12 $OUT/parity_buckets_hill_1024.txt | cut -c1-500
;;
4)
# whole GPU tier (no -x), drift statistics at N = 1024, front-end throughput over the kernel's register budgets
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04d; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -12 $OUT/gpu_tests.txt | cut -c1-300
python - <<'PY' | tee $OUT/kino_wps.txt
import time, numpy as np
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
S, G = scenes.random_queries(8192, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
ka = U.KinoAstar(m, slots=8192)
base = None
for wps in (2, 4, 6, 8):
    ka.set_wps(wps)
    ka.plan_batch(S[:256], G[:256], path_cap=1)
    for B in (1, 2048, 8192):
        t0 = time.perf_counter(); r = ka.plan_batch(S[:B], G[:B], path_cap=256); dt = time.perf_counter() - t0
        it = np.array([q["iter_num"] for q in r]); st = np.array([q["status"] for q in r])
        if wps == 2 and B == 8192: base = (it.copy(), st.copy())
        same = "" if base is None or B != 8192 else (" identical to wps 2: %s" % (np.array_equal(it, base[0]) and np.array_equal(st, base[1])))
        print("wps %d B %5d: %8.0f queries/s  kernel %.1f ms  %.2f M expansions/s  max expansions %d%s" % (wps, B, B / dt, ka.stats()["kernel_ms"], it.sum() / dt / 1e6, it.max(), same))
PY
UPH_PB_ONLY_YAML=1 timeout 900 python tools/parity_buckets.py 1024 $OUT/parity_buckets_hill_1024.json hill > $OUT/parity_buckets_hill_1024.txt 2>&1
tail -12 $OUT/parity_buckets_hill_1024.txt | cut -c1-600
;;
5)
# the search kernel's configurations one process each, every one under its own short timeout (call 4 ran into a kernel that never ended)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04e; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
for cfg in "2 0" "2 2" "2 1" "2 3" "4 3" "6 3" "8 3"; do
  timeout 100 python tools/kino_probe.py $cfg 8192 2>&1 | tail -1 | cut -c1-400; echo "  (cfg $cfg rc $?)"
done | tee $OUT/kino_probe.txt
timeout 300 python -m pytest tests/test_gpu_kino.py -q 2>&1 | tail -5 | cut -c1-300 | tee $OUT/kino_tests.txt
;;
6)
# whole GPU tier, front-end throughput at batch sizes beyond the slot count (load balance), XCD-local launch order A/B, single-trajectory phases,
# drift statistics at N = 1024
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04f; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -8 $OUT/gpu_tests.txt | cut -c1-300
for cfg in "2 3" "4 3"; do timeout 200 python tools/kino_probe.py $cfg 65536 2>&1 | tail -1 | cut -c1-600; done | tee $OUT/kino_big.txt
UPH_KINO_SLOTS=2048 timeout 200 python tools/kino_probe.py 4 3 65536 2>&1 | tail -1 | cut -c1-600 | tee -a $OUT/kino_big.txt
python - <<'PY' | tee $OUT/xcd_ab.txt
import time, numpy as np
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(16384, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
for grp in (0, 64, 256, 0, 64, 256):
    o = U.ALMTrajOpt(m); o.set_xcd_locality(grp); o.upload(probs)
    o.set_rho(1.0); o.solve()
    ms = []
    for _ in range(3):
        o.set_rho(1.0); o.solve(); ms.append(o.stats()["kernel_ms"])
    print("xcd_group %3d: solve kernel %.1f ms (B = 16384, %s)" % (grp, np.mean(ms), ["%.1f" % v for v in ms]))
    del o
PY
UPH_LANES=512 timeout 100 python tools/phase_breakdown.py 1 2>&1 | grep -E "cycles/eval|kernel_ms" | tee $OUT/single_phases.txt
UPH_PB_ONLY_YAML=1 timeout 600 python tools/parity_buckets.py 1024 $OUT/parity_buckets_hill_1024.json hill > $OUT/parity_buckets_hill_1024.txt 2>&1
tail -12 $OUT/parity_buckets_hill_1024.txt | cut -c1-700
;;
7)
# end-of-round check: build + smoke, the whole GPU tier, tools/profile.sh <tag> (bench line, kernel trace, counter passes, calibration)
cd $GRAFT_REPO_ROOT
TAG=${2:-r04p}
export UPH_GIT_HEAD=$(cat build/git_head.txt 2>/dev/null || echo unknown)
OUT=gpurun_out/final_$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt | cut -c1-300
timeout 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; tail -4 $OUT/gpu_tests.txt | cut -c1-300
timeout 1200 bash tools/profile.sh $TAG 2>&1 | grep -E "uph_solver_kernel<128, 2, 2|FETCH_SIZE  |WRITE_SIZE  |SQ_INSTS_VALU|SQ_WAIT_ANY|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_VALU" | head -14
python - $TAG <<'PY'
import json, sys
r = json.loads(open("gpurun_out/prof_%s/bench.json" % sys.argv[1]).read().strip().split("\n")[-1])
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r["penalty_kernel"]["batch"]["frac"], r["boundary"]["B%d" % r["config"]["batch_per_gpu"]], r.get("converged_traj_opts_per_s"))
print("front_end", json.dumps(r.get("front_end"))[:900])
print("drift", json.dumps(r.get("parity_floor", {}).get("drift")))
print("map", r.get("map_build_s"), r.get("map_build_stages_ms"))
PY
;;
8)
# one wave per trajectory, eight workgroups per CU, on short problems; then the final bench line of HEAD
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04g; mkdir -p $OUT
timeout 400 python tools/single_wave_class.py 8192 2>&1 | grep -v amdgpu.ids | tee $OUT/single_wave_class.txt
make -C oracle -s 2>&1 | tail -2
timeout 400 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 400 $OUT/bench.json
;;
9)
# single-wave forms (2)-(4) of profiles/r04h_single_wave_forms.txt: A/B on the bench batch (knob on / off), then the tests that depend on lane selection.
# Needs tools/experiments/r04_paired_one_wave.patch applied (uph_ctx_set_single_wave is not in the shipped library).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${2:-r04h}; mkdir -p $OUT
python - <<'PY' | tee $OUT/single_wave_ab.txt
import time, numpy as np
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(16384, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
for B in (16384, 8192):
    for on in (0, 1, 0, 1):
        o = U.ALMTrajOpt(m); o.set_single_wave(on); o.upload(probs[:B])
        o.set_rho(1.0); o.solve()
        ms = []
        for _ in range(3):
            o.set_rho(1.0); o.solve(); ms.append(o.stats()["kernel_ms"])
        ln = o.lanes_per_problem(); r = np.array([q["ret"] for q in o.download(full=False)])
        print("B %5d single_wave %d: solve kernel %.1f ms (%s) -> %.0f traj-opts/s by kernel time; one-wave share %.2f; converged %.3f; prepare %.1f ms" % (
            B, on, np.mean(ms), ["%.1f" % v for v in ms], B / np.mean(ms) * 1e3, (ln == 64).mean(), (r == 0).mean(), o.stats()["prepare_ms"]))
        del o
PY
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_lanes.py tests/test_gpu_parity.py tests/test_gpu_edge.py -q 2>&1 | tail -8 | cut -c1-300 | tee $OUT/tests.txt
;;
10)
# front-end search kernel after a change: the four kino tests, then throughput at 2 and 4 waves per SIMD (every probe under its own timeout)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${2:-r04l}; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_kino.py -x -q 2>&1 | tail -6 | cut -c1-300 | tee $OUT/kino_tests.txt
for w in 4 2; do timeout 150 python tools/kino_probe.py $w 3 ${3:-16384} 2>&1 | grep -v amdgpu.ids | tee -a $OUT/kino_probe.txt; done
;;
*) echo "usage: tools/r04_runs.sh <n>";;
esac
