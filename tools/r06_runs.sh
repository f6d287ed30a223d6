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
esac
