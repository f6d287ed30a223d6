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
esac
