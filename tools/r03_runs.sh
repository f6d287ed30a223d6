#!/bin/bash
# The GPU-box command lists of round 3's gpurun calls, one function per call: gpurun -- "bash tools/r03_runs.sh <n>".
# 1 first tests + bench  2 scheduler A/B + penalty-kernel counters  3/4 full GPU tier, map counters, desert / volcano bucket tables
# 5 scatter-window A/B + bit-identity  6 full tier after the limit lift  7 profile.sh r03g + LDS counters + single-process bench
# 8 two more codegen A/Bs  9/10 km2 bench + soak  11 bit-identity against the round-2 source + hill bucket table
set -u
case "${1:-}" in
1)
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
;;
2)
cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh r03b default bias0 ilp iterilp memcl 2>&1 | tee gpurun_out/r03b_ab.txt
bash tools/pmc_eval.sh r03b default
bash tools/pmc_eval.sh r03b bias0
;;
3)
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
;;
4)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03d; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -30 $OUT/gpu_tests.txt
;;
5)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03e; mkdir -p $OUT
bash tools/gpu_ab.sh r03e default scold yb16 yb12 2>&1 | tee $OUT/ab.txt
python tools/cmp_variant.py $OUT/res_default.npy
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_scold.so python tools/cmp_variant.py $OUT/res_scold.npy
python - <<'PY'
import numpy as np
a = np.load("gpurun_out/r03e/res_default.npy", allow_pickle=True); b = np.load("gpurun_out/r03e/res_scold.npy", allow_pickle=True)
print("bit-identical to the round-2 scatter window:", all(np.array_equal(x, y) for x, y in zip(a, b)))
PY
bash tools/pmc_eval.sh r03e default
;;
6)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03f; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -22 $OUT/gpu_tests.txt
bash tools/gpu_ab.sh r03f default 2>&1 | tee $OUT/ab.txt
;;
7)
cd $GRAFT_REPO_ROOT
bash tools/profile.sh r03g 2>&1 | tail -60
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03g
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/lds -o lds -- python /tmp/evalonly.py > $OUT/lds.log 2>&1
f=$(find $OUT/lds -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/lds_summary.txt
import sys, csv, collections
agg = collections.defaultdict(float)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if 'uph_solver_kernel' in row['Kernel_Name']:
            agg[(row['Kernel_Name'][:60], row['Counter_Name'])] += float(row['Counter_Value'])
for k, v in sorted(agg.items()):
    print('%-62s %-22s %.6g' % (k[0], k[1], v))
PY
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --gpus 1 --single-process --steps 3 --warmup 1 > $OUT/bench_single_process.json 2> $OUT/bench_single_process.err; tail -c 1500 $OUT/bench_single_process.json
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
;;
8)
cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh r03h default trk nounc 2>&1 | tee gpurun_out/r03h_ab.txt
;;
9)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03i; mkdir -p $OUT
timeout 600 python bench.py --workload km2 > $OUT/bench_km2.json 2> $OUT/bench_km2.err; tail -c 1800 $OUT/bench_km2.json; tail -3 $OUT/bench_km2.err
timeout 600 python tools/soak.py > $OUT/soak.txt 2>&1; tail -8 $OUT/soak.txt
;;
10)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03j; mkdir -p $OUT
timeout 600 python bench.py --workload km2 > $OUT/bench_km2.json 2> $OUT/bench_km2.err; python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03j/bench_km2.json").read().strip().split("\n")[-1])
print(r["value"], r["ms_per_step"], r["converged_frac"], r["parity_floor"])
PY
;;
11)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03k; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -1
python tools/cmp_variant.py $OUT/res_head.npy
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_bias0.so python tools/cmp_variant.py $OUT/res_r02.npy
python - <<'PY'
import numpy as np
a = np.load("gpurun_out/r03k/res_head.npy", allow_pickle=True); b = np.load("gpurun_out/r03k/res_r02.npy", allow_pickle=True)
print("HEAD bit-identical to the round-2 source (library variant built before this round's kernel edits):", all(np.array_equal(x, y) for x, y in zip(a, b)))
PY
timeout 900 python tools/parity_buckets.py 256 $OUT/parity_buckets_hill.json hill > $OUT/parity_buckets_hill.txt 2>&1; tail -30 $OUT/parity_buckets_hill.txt
;;
*) echo "usage: bash tools/r03_runs.sh <1..11>"; exit 2 ;;
esac
