#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -25 $OUT/gpu_tests.txt
bash tools/pmc_map.sh r03c 2>&1 | tail -25
for scene in desert vocano; do
  timeout 600 python tools/parity_buckets.py 256 $OUT/parity_buckets_$scene.json $scene > $OUT/parity_buckets_$scene.txt 2>&1
  tail -12 $OUT/parity_buckets_$scene.txt
done
