#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03j; mkdir -p $OUT
timeout 600 python bench.py --workload km2 > $OUT/bench_km2.json 2> $OUT/bench_km2.err; python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03j/bench_km2.json").read().strip().split("\n")[-1])
print(r["value"], r["ms_per_step"], r["converged_frac"], r["parity_floor"])
PY
