#!/bin/bash
# The GPU-box command lists of round 6's gpurun calls, one case per call: gpurun -- "bash tools/r06_runs.sh <n>".
set -u
cd $GRAFT_REPO_ROOT
export UPH_GIT_HEAD=$(cat build/git_head.txt 2>/dev/null || echo unknown)
make -C oracle -s 2>&1 | tail -2
case "${1:-}" in
1)
# first contact of the round: the whole GPU tier (new: A5 alone against the oracle, the `.map` cache semantics), smoke, a bench line with the three penalty-kernel figures
OUT=gpurun_out/r06a; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -15 $OUT/gpu_tests.txt | cut -c1-400
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 600 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - $OUT/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value %.0f  launch %.1f ms  frac %.3f  converged %.3f  single %.2f ms  B256 %.0f" % (r["value"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["converged_frac"], r["single_traj_ms"], r.get("traj_opts_per_s_B256", 0)))
print(json.dumps(r["roofline"]["penalty_kernel"])[:1500])
print(json.dumps(r["penalty_kernel"])[:2500])
PY
;;
2)
# register diet of the sample code (pinned kinematic sums, late dual loads) at two and three waves per SIMD: penalty kernel + solve on the bench batch and on short problems
OUT=gpurun_out/r06b; mkdir -p $OUT
for v in default diet2 diet3 diet3e default diet3; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  echo "== $v"
  timeout 400 python tools/ab_eval.py 16384 5.5 2>&1 | tail -3
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
timeout 900 python -m pytest tests/test_gpu_map.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -5 | tee $OUT/tests.txt
;;
3)
# the whole GPU tier, then the new bench forms: default line with the configs block, --workload astar, the RCCL path with one rank (self-test, map hash)
OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -6 $OUT/gpu_tests.txt | cut -c1-400
( time timeout 900 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench rc $?"; tail -3 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value %.0f  launch %.1f ms  frac %.3f  converged %.3f  single %.2f ms  B256 %.0f" % (r["value"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["converged_frac"], r["single_traj_ms"], r.get("traj_opts_per_s_B256", 0)))
c = r.get("configs", {})
print("configs block: %.1f s" % c.get("seconds", -1))
for e in c.get("entries", []):
    print(json.dumps({k: v for k, v in e.items() if k not in ("workload", "config")})[:1400])
PY
( time timeout 600 python bench.py --workload astar --steps 3 --warmup 1 > $OUT/bench_astar.json 2> $OUT/bench_astar.err ) 2>&1 | grep real; tail -3 $OUT/bench_astar.err
python - $OUT/bench_astar.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("astar: value %.0f  launch %.1f ms  frac %.3f  converged %.3f" % (r["value"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["converged_frac"]), r["front_end"], r["config"]["workload"][:80])
print(json.dumps(r.get("parity_floor"))[:600])
PY
UPH_FORCE_DIST=1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu --no-extras > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "dist1 rc $?"; tail -3 $OUT/bench_dist1.err
python - $OUT/bench_dist1.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("dist1:", r["value"], r["rccl_selftest"], r["map_hash"], r["map_hash_identical_on_all_ranks"], r["config"]["rccl_world"], r["per_rank_spread"])
PY
timeout 300 python bench.py --gpus 1 --single-process --steps 2 --warmup 1 > $OUT/bench_sp1.json 2> $OUT/bench_sp1.err; echo "sp1 rc $?"; tail -c 600 $OUT/bench_sp1.json
;;
4)
# where the two waves of a workgroup land (SIMD ids), and the chain-running role dealt to alternating waves; scaling-kernel regrouping
OUT=gpurun_out/r06d; mkdir -p $OUT
./build/micro/wave_placement 4096 20000 2>&1 | tee $OUT/wave_placement.txt | tail -12
for v in default flip1 flip9 flip10 flip16 sc3 default flip16; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  [ -z "${UNEVENHIP_LIB:-}" ] || [ -f "$UNEVENHIP_LIB" ] || continue
  echo "== $v"
  timeout 400 python tools/ab_eval.py 16384 0 2>&1 | tail -1
done 2>&1 | tee $OUT/ab.txt
;;
5)
# initScaling regrouped (3 + 2 + 2 constraints, operator rows in batches of 4 columns): scaling-kernel ms and bit-identity of the results; then the whole GPU tier
OUT=gpurun_out/r06e; mkdir -p $OUT
for v in default sc3w4 sc2w4 sc3 default sc3w4; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  [ -z "${UNEVENHIP_LIB:-}" ] || [ -f "$UNEVENHIP_LIB" ] || continue
  echo "== $v"
  timeout 400 python tools/ab_eval.py 16384 0 2>&1 | tail -1
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -6 $OUT/gpu_tests.txt | cut -c1-400
;;
7)
# blocked two-loop: the state-machine tests first (teacher-forced late states: ring wrap, cautious skip ...), per-evaluation parity, then the A/B against the plain recursion
OUT=gpurun_out/r06f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_forced.py tests/test_gpu_parity.py tests/test_gpu_lanes.py tests/test_gpu_edge.py -m gpu -q -x 2>&1 | tail -12 | cut -c1-300 | tee $OUT/tests.txt
for v in default seq default seq; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  echo "== $v"
  timeout 400 python tools/ab_eval.py 16384 5.5 2>&1 | tail -2
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
timeout 300 python - <<'PY' 2>&1 | tail -4 | tee $OUT/single.txt
import os, sys
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(256, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
for tag, pp in (("single hill", [scenes.hill_problem()]), ("B256", probs)):
    o = U.ALMTrajOpt(m); o.upload(pp)
    for _ in range(3):
        o.set_rho(1.0); o.solve(); st = o.stats()
    print("%s: kernel %.3f ms + scaling %.3f ms, iters %d, %.0f traj/s" % (tag, st["kernel_ms"], st["prepare_ms"], st["lbfgs_iters"], len(pp) / (st["kernel_ms"] + st["prepare_ms"]) * 1e3))
PY
;;
8)
# blocked two-loop, second pass: unconditional prefetch; phase cycles per evaluation (in-kernel counters) blocked vs plain recursion; A/B
OUT=gpurun_out/r06g; mkdir -p $OUT
for v in cyc cycseq; do
  export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so
  echo "== $v (B = 16384, 128 lanes)"; timeout 300 python tools/phase_breakdown.py 16384 2>&1 | grep -E "cycles/eval|kernel_ms|residency"
  echo "== $v (single trajectory class: B = 64, 512 lanes)"; timeout 300 python tools/phase_breakdown.py 64 2>&1 | grep -E "cycles/eval|kernel_ms"
done 2>&1 | tee $OUT/phases.txt
for v in default seq default seq; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  echo "== $v"
  timeout 400 python tools/ab_eval.py 16384 5.5 2>&1 | tail -2
done 2>&1 | tee $OUT/ab.txt
;;
9)
# the default bench line with its own live counter passes (rocprofv3 --pmc children of bench.py)
OUT=gpurun_out/r06h; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench rc $?"; tail -3 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
rf = r["roofline"]
print("value %.0f  launch %.1f ms  frac %.3f" % (r["value"], rf["avg_launch_ms"], rf["frac"]))
print("traffic", rf["traffic"], "|", rf["traffic_source"])
print("valu_busy", rf["valu_busy"], rf["valu_busy_at_measured_fp64_issue_rate"], "wait", rf["wave_wait_frac"], "|", rf["valu_busy_source"])
print("mfma", json.dumps(rf["mfma"])[:300])
PY
;;
10)
# after the disc-walk refactor of the map build (one predicate for the sizing and the staging kernel): the whole GPU tier + the map-build timing
OUT=gpurun_out/r06i; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -4 $OUT/gpu_tests.txt | cut -c1-300
timeout 200 python - <<'PY' 2>&1 | tail -3 | tee $OUT/map_ms.txt
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
for tag, xyz, prm in (("hill", scenes.make_hill_cloud(), None), ("vocano", np.load("tests/golden/vocano_xyz.npz")["xyz"], dict(max_rho=0.08)), ("forest", np.load("tests/golden/forest_xyz.npz")["xyz"], None)):
    m = U.UnevenMap(prm); m.build(xyz, download=False)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); m.build(xyz, download=False); ts.append(time.perf_counter() - t0)
    print(tag, "uph_map_build %.2f ms, stages" % (min(ts) * 1e3), {k: round(v, 3) for k, v in m.build_stats()["stages_ms"].items()})
PY
;;
11)
# latency class: dense knot solve (difference-form operator) at 512 lanes -- the lane-variant tests (1e-9 against the oracle), then single trajectory / B = 256 / desert B = 256 against the Thomas build
OUT=gpurun_out/r06j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_buckets.py -m gpu -q -x 2>&1 | tail -8 | cut -c1-300 | tee $OUT/tests.txt
for v in default thomas512 default thomas512; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  echo "== $v"
  timeout 300 python - <<'PY' 2>&1 | tail -4
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(256, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
for tag, pp in (("single hill", [scenes.hill_problem()]), ("B64", probs[:64]), ("B256", probs)):
    o = U.ALMTrajOpt(m); o.upload(pp)
    ks = []
    for _ in range(3):
        o.set_rho(1.0); o.solve(); st = o.stats(); ks.append(st["kernel_ms"] + st["prepare_ms"])
    out = o.download(full=False)
    print("%s: %.3f ms (kernel + scaling), iters %d, evals %d, %.4f ms / iteration, %.0f traj/s, converged %.2f" % (tag, min(ks), st["lbfgs_iters"], st["evals"], st["kernel_ms"] / st["lbfgs_iters"] * (len(pp) if len(pp) == 1 else 1), len(pp) / min(ks) * 1e3, np.mean([q["ret"] == 0 for q in out])))
o = U.ALMTrajOpt(m); o.upload([scenes.hill_problem()] * 256); o.init_scaling_batch()
o.eval_batch(None, repeat=20); o.eval_batch(None, repeat=20)
print("hill x 256 evaluation: %.2f us" % (o.stats()["kernel_ms"] * 1e3 / 20))
PY
done 2>&1 | tee $OUT/ab.txt
;;
12)
# soak of the final build (mixed batch sizes through every kernel-selection regime + front-end queries: return codes, device-memory growth) and the default bench line
OUT=gpurun_out/r06k; mkdir -p $OUT; export TMPDIR=/tmp
UPH_SOAK_ITERS=600 timeout 900 python tools/soak.py 2>&1 | tail -6 | tee $OUT/soak.txt
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real | tee $OUT/bench_wall.txt; tail -2 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value %.0f  ms/step %.2f  launch %.1f ms  frac %.3f  B32768 %.0f  B8192 %.0f B256 %.0f  traffic %.4g (%s)" % (r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r.get("traj_opts_per_s_B32768", 0), r.get("traj_opts_per_s_B8192", 0), r.get("traj_opts_per_s_B256", 0), r["roofline"]["traffic"] or 0, r["roofline"]["traffic_source"][:60]))
PY
;;
13)
# the other forms of the bench call after this round's surgery: km2 (with its parity floor), fp32 samples + pipelined, the tiled km2 form refuses N = 1 gracefully
OUT=gpurun_out/r06l; mkdir -p $OUT
timeout 900 python bench.py --workload km2 --steps 3 --warmup 1 > $OUT/bench_km2.json 2> $OUT/bench_km2.err; echo "km2 rc $?"; tail -2 $OUT/bench_km2.err
python - $OUT/bench_km2.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("km2 value %.0f frac %.3f converged %.3f" % (r["value"], r["roofline"]["frac"], r["converged_frac"]), json.dumps(r.get("parity_floor"))[:400])
PY
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --pipelined --fp32 > $OUT/bench_fp32_pipelined.json 2> $OUT/bench_fp32.err; echo "fp32 rc $?"; tail -2 $OUT/bench_fp32.err
python - $OUT/bench_fp32_pipelined.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("fp32 samples value %.0f dtype %s pipelined %s traffic_source %s" % (r["value"], r["dtype"], r.get("pipelined"), r["roofline"]["traffic_source"][:90]))
PY
;;
14)
# after the handshake refactor: the RCCL path with one rank, and the plain default line (short)
OUT=gpurun_out/r06m; mkdir -p $OUT; export TMPDIR=/tmp
UPH_FORCE_DIST=1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu --no-extras > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "dist1 rc $?"; tail -2 $OUT/bench_dist1.err
python - $OUT/bench_dist1.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("dist1:", r["value"], r["rccl_selftest"], r["map_hash"], r["map_hash_identical_on_all_ranks"], r["config"]["rccl_world"], r["per_rank_spread"])
PY
timeout 600 python bench.py --steps 3 --warmup 1 --no-configs > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 300 $OUT/bench.json
;;
15)
# parity buckets + drift statistics of the A*-seeded workload at N = 2048 (device | oracle(FMA) against the oracle), all host threads for the two CPU legs
OUT=gpurun_out/r06n; mkdir -p $OUT
UPH_PB_THREADS=$(nproc) timeout 1500 python tools/parity_buckets.py 2048 $OUT/parity_buckets_astar_2048.json astar 2>&1 | grep -v amdgpu.ids | tee $OUT/parity_buckets_astar_2048.txt | tail -14
;;
16)
# the last commit: smoke, the whole GPU tier, the driver's form of the bench call
OUT=gpurun_out/r06zz; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -3 $OUT/gpu_tests.txt | cut -c1-300
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real | tee $OUT/bench_wall.txt; tail -2 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
rf = r["roofline"]
print("value %.0f  ms/step %.2f  launch %.1f ms  frac %.3f  converged %.3f | penalty %s | traffic %.4g (%s)" % (r["value"], r["ms_per_step"], rf["avg_launch_ms"], rf["frac"], r["converged_frac"],
      {k: round(v, 3) for k, v in rf["penalty_kernel"].items() if k.startswith("frac")}, rf["traffic"] or 0, rf["traffic_source"][:50]))
print("configs:", [(round(e.get("value", -1), 1), e.get("error", "")) for e in r["configs"]["entries"]], "astar extra %.0f" % r.get("traj_opts_per_s_astar_seeded", 0))
PY
;;
6)
# end-of-round record on the final sources: smoke, the whole GPU tier, profile.sh (bench line, kernel trace, counter passes, calibration) for the headline and for --workload astar
OUT=gpurun_out/r06z; mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -4 $OUT/gpu_tests.txt | cut -c1-400
# the kernel arithmetic of this round's library against round 5's final library (built from commit 0fb623f): same bits -> round 5's N = 4096 / 16384 drift tables stand
for v in default r05; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  [ -z "${UNEVENHIP_LIB:-}" ] || [ -f "$UNEVENHIP_LIB" ] || continue
  echo "== $v"; timeout 400 python tools/ab_eval.py 16384 0 2>&1 | tail -1
done 2>&1 | tee $OUT/bit_identity_vs_r05.txt
unset UNEVENHIP_LIB
mkdir -p build/micro; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/micro/fetch_calib.hip -o build/micro/fetch_calib 2>/dev/null
bash tools/profile.sh r06z 2>&1 | tail -40
WL=astar bash tools/profile.sh r06z_astar 2>&1 | tail -15
;;
esac
