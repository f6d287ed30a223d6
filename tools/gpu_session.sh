#!/bin/bash
# One GPU-box session: GPU test tier + A/B bench of library variants + phase breakdown.  usage: bash tools/gpu_session.sh <tag> [variant names...]
cd $GRAFT_REPO_ROOT
TAG=${1:-s}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
make -C oracle -s 2>&1 | tail -2
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
echo "== bench default"; timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 900 $OUT/bench_default.json
timeout 200 python tools/phase_breakdown.py 8192 > $OUT/phase_default.txt 2>&1; cat $OUT/phase_default.txt
for v in "$@"; do
  echo "== bench variant $v"
  UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_$v.json 2> $OUT/bench_$v.err; tail -c 900 $OUT/bench_$v.json
  UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so timeout 200 python tools/phase_breakdown.py 8192 > $OUT/phase_$v.txt 2>&1; cat $OUT/phase_$v.txt
done
echo "== bench default lanes 64"; timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu --lanes 64 > $OUT/bench_l64.json 2> $OUT/bench_l64.err; tail -c 600 $OUT/bench_l64.json
