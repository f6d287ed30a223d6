#!/bin/bash
# round-3 first GPU call: new multi-GPU tests first (fail fast), then the whole GPU tier, then the bench line
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03a; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_multi.py -x -q > $OUT/multi_tests.txt 2>&1; echo "multi rc $?" >> $OUT/multi_tests.txt
tail -15 $OUT/multi_tests.txt
timeout 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -8 $OUT/gpu_tests.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
tail -c 3000 $OUT/bench.json
tail -5 $OUT/bench.err
