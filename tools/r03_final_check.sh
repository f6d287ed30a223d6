cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03m; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; tail -4 $OUT/gpu_tests.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03m/bench.json").read().strip().split("\n")[-1])
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["traffic_source"], r["boundary"], r["parity_floor"]["same_ret"])
PY
