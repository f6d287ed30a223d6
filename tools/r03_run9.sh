#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03i; mkdir -p $OUT
timeout 600 python bench.py --workload km2 > $OUT/bench_km2.json 2> $OUT/bench_km2.err; tail -c 1800 $OUT/bench_km2.json; tail -3 $OUT/bench_km2.err
timeout 600 python tools/soak.py > $OUT/soak.txt 2>&1; tail -8 $OUT/soak.txt
