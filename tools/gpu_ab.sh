#!/bin/bash
# A/B of library variants on the GPU box: bench line (frac / launch ms) + phase breakdown per variant.  usage: bash tools/gpu_ab.sh <tag> <variant...>   ("default" = the in-tree build)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --batch 8192 --steps 4 --warmup 1 --no-cpu > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  launch %.1f ms  frac %.3f  converged %.3f' % (sys.argv[2], r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
  timeout 200 python tools/phase_breakdown.py 8192 2>&1 | grep -E "cycles/eval" | awk '{printf "%s=%s ", $1, $NF} END {print ""}'
done
