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
esac
