#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03f; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -22 $OUT/gpu_tests.txt
bash tools/gpu_ab.sh r03f default 2>&1 | tee $OUT/ab.txt
