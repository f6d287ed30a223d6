#!/bin/bash
# end-of-round check on the GPU box: build + smoke, the whole GPU tier, then tools/profile.sh <tag> (bench line, kernel trace, counters)
cd $GRAFT_REPO_ROOT
TAG=${1:-r03p}
OUT=gpurun_out/final_$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; tail -4 $OUT/gpu_tests.txt
bash tools/profile.sh $TAG 2>&1 | grep -E "uph_solver_kernel<128, 2, 2|FETCH_SIZE  |WRITE_SIZE  |SQ_INSTS_VALU|SQ_WAIT_ANY|SQ_WAVE_CYCLES" | head -12
python - $TAG <<'PY'
import json, sys
r = json.loads(open("gpurun_out/prof_%s/bench.json" % sys.argv[1]).read().strip().split("\n")[-1])
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r["penalty_kernel"]["batch"]["frac"], r["boundary"]["B%d" % r["config"]["batch_per_gpu"]])
PY
