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
*) echo "usage: tools/r04_runs.sh <n>";;
esac
