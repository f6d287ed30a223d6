#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03d; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -30 $OUT/gpu_tests.txt
