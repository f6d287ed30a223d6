#!/bin/bash
# The GPU-box command lists of round 5's gpurun calls, one case per call: gpurun -- "bash tools/r05_runs.sh <n>".
set -u
cd $GRAFT_REPO_ROOT
export UPH_GIT_HEAD=$(cat build/git_head.txt 2>/dev/null || echo unknown)      # (the GPU box holds a snapshot without .git: `git rev-parse --short HEAD > build/git_head.txt` before the call)
make -C oracle -s 2>&1 | tail -2
case "${1:-}" in
1)
# local frames (km^2 parity fix): hill results bit for bit against round 4's library, the whole GPU tier (new far-from-origin test), A/B of the
# solve launch against round 4's library, the km^2 bench line with its parity buckets, LDS bank conflicts of the penalty kernel by phase
OUT=gpurun_out/r05a; mkdir -p $OUT
python tools/cmp_variant.py $OUT/x_default.npy 2>&1 | tail -2
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_r04.so python tools/cmp_variant.py $OUT/x_r04.npy 2>&1 | tail -2
python -c "
import numpy as np
a, b = np.load('$OUT/x_default.npy'), np.load('$OUT/x_r04.npy')
print('hill, 64 solves, this build vs round 4 library: bit-identical', np.array_equal(a, b), 'max diff', np.abs(a - b).max())" | tee $OUT/bit_identity.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -15 $OUT/gpu_tests.txt | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_km2.py -q -s -k far_from_origin 2>&1 | grep -E "far-from|within|passed|failed" | cut -c1-300 | tee $OUT/far_test.txt
for v in default r04; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  launch %.1f ms  frac %.3f  converged %.3f' % (sys.argv[2], r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
timeout 600 python bench.py --workload km2 --steps 3 --warmup 1 > $OUT/bench_km2.json 2> $OUT/bench_km2.err; echo "km2 rc $?"
python - $OUT/bench_km2.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("km2 value", r["value"], "frac", r["roofline"]["frac"], "converged", r["converged_frac"])
print(json.dumps(r.get("parity_floor"), indent=None)[:1500])
PY
bash tools/pmc_phases.sh r05a lds 2>&1 | tee $OUT/pmc_phases_lds.txt
;;
2)
# the two tests call 1 failed (cache write order, cost tolerance of capped solves), the new forest / mountain tier, local frames through per-trajectory
# grid descriptors: hill bit-identity again + A/B against round 4 and against the build without in-kernel timers, the drift statistics at N = 4096
OUT=gpurun_out/r05b; mkdir -p $OUT
python tools/cmp_variant.py $OUT/x_default.npy 2>&1 | tail -1
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_r04.so python tools/cmp_variant.py $OUT/x_r04.npy 2>&1 | tail -1
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_nocyc.so python tools/cmp_variant.py $OUT/x_nocyc.npy 2>&1 | tail -1
python -c "
import numpy as np
a, b, c = np.load('$OUT/x_default.npy'), np.load('$OUT/x_r04.npy'), np.load('$OUT/x_nocyc.npy')
print('hill, 64 solves: this build vs round 4 library bit-identical', np.array_equal(a, b), '; build without in-kernel timers bit-identical', np.array_equal(a, c))" | tee $OUT/bit_identity.txt
timeout 900 python -m pytest tests/test_gpu_km2.py tests/test_gpu_map.py tests/test_gpu_forest.py tests/test_gpu_kino.py tests/test_gpu_adapter.py tests/test_gpu_tiles.py -m gpu -q -s > $OUT/gpu_tests.txt 2>&1; echo "rc $?" >> $OUT/gpu_tests.txt
grep -E "far-from|within 1e-4|forest|passed|failed|rc |Error|assert" $OUT/gpu_tests.txt | cut -c1-400 | tail -40
for v in default r04 nocyc default r04 nocyc; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  launch %.1f ms  frac %.3f  converged %.3f' % (sys.argv[2], r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
UPH_PB_ONLY_YAML=1 UPH_PB_THREADS=96 timeout 900 python tools/parity_buckets.py 4096 $OUT/parity_buckets_hill_4096.json hill > $OUT/parity_buckets_hill_4096.txt 2>&1
tail -12 $OUT/parity_buckets_hill_4096.txt | cut -c1-700
UPH_PB_THREADS=96 timeout 900 python tools/parity_buckets.py 4096 $OUT/parity_buckets_desert_4096.json desert > $OUT/parity_buckets_desert_4096.txt 2>&1
tail -12 $OUT/parity_buckets_desert_4096.txt | cut -c1-700
;;
3)
# K-major LDS layout + timers off + exact staging capacity: bit identity against round 4's library, the whole GPU tier, the branchy far-from-origin problem
# under other lane counts, A/B of the layout alone (nocyc = old layout without timers), LDS counters of the penalty kernel
OUT=gpurun_out/r05c; mkdir -p $OUT
python tools/cmp_variant.py $OUT/x_default.npy 2>&1 | tail -1
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_r04.so python tools/cmp_variant.py $OUT/x_r04.npy 2>&1 | tail -1
python -c "
import numpy as np
a, b = np.load('$OUT/x_default.npy'), np.load('$OUT/x_r04.npy')
print('hill, 64 solves: K-major layout, timers off vs round 4 library: bit-identical', np.array_equal(a, b))" | tee $OUT/bit_identity.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -25 $OUT/gpu_tests.txt | cut -c1-400
timeout 300 python tools/far_outlier_probe.py 2>&1 | grep lanes | tee $OUT/far_outlier_probe.txt
for v in nocyc default nocyc default; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  launch %.1f ms  frac %.3f  converged %.3f  single traj %.2f ms' % (sys.argv[2], r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac'], r['single_traj_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
bash tools/pmc_eval.sh r05c default lds 2>&1 | tail -12
bash tools/pmc_eval.sh r05c nocyc lds 2>&1 | tail -12
;;
4)
# knot buffers padded per lane block (K-major layout reverted: measured 0.3 % slower): bit identity, whole GPU tier, A/B against the packed buffers
# (nocyc = packed, timers off), LDS counters of the penalty kernel
OUT=gpurun_out/r05d; mkdir -p $OUT
python tools/cmp_variant.py $OUT/x_default.npy 2>&1 | tail -1
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_r04.so python tools/cmp_variant.py $OUT/x_r04.npy 2>&1 | tail -1
python -c "
import numpy as np
a, b = np.load('$OUT/x_default.npy'), np.load('$OUT/x_r04.npy')
print('hill, 64 solves: padded knot buffers, timers off vs round 4 library: bit-identical', np.array_equal(a, b))" | tee $OUT/bit_identity.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -12 $OUT/gpu_tests.txt | cut -c1-400
timeout 300 python -m pytest tests/test_gpu_km2.py -q -s -k far_from_origin 2>&1 | grep -E "oracle solves|evaluation|passed|failed" | cut -c1-400 | tee $OUT/far_test.txt
for v in nocyc default nocyc default; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  launch %.1f ms  frac %.3f  converged %.3f  single traj %.2f ms' % (sys.argv[2], r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac'], r['single_traj_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
bash tools/pmc_eval.sh r05d default lds 2>&1 | grep SQ_
;;
5)
# tuning A/B with the timers off: two-loop prefetch depth 4 / 5 (default) / 6, grid descriptor held in SGPRs instead of re-read per sample chunk
OUT=gpurun_out/r05e; mkdir -p $OUT
for v in default pf4 pf6 gridsgpr default pf4 pf6 gridsgpr; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  launch %.1f ms  frac %.3f  converged %.3f  single traj %.2f ms' % (sys.argv[2], r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac'], r['single_traj_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
;;
6)
# end-of-round record: smoke, the whole GPU tier, tools/profile.sh (bench line with cpu legs, rocprofv3 kernel trace of the same command, counter passes, calibration)
TAG=${2:-r05f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 600 python tools/soak.py 2>&1 | tail -4 | tee $OUT/soak.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -4 $OUT/gpu_tests.txt | cut -c1-300
bash tools/profile.sh $TAG 2>&1 | tail -40 | cut -c1-300
;;
7)
# the bench line alone, for the record (profiles/<tag>_bench.json)
TAG=${2:-r05g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
tail -c 400 $OUT/bench.json; echo
timeout 600 python bench.py --workload km2 --steps 3 --warmup 1 > $OUT/bench_km2.json 2> $OUT/bench_km2.err; echo "km2 rc $?"
;;
8)
# initScaling with the operator rows fetched once for all seven constraints of a sample (sg1) / once per group of four and three (default): bit identity,
# scaling-kernel time, init-scaling tests
OUT=gpurun_out/r05h; mkdir -p $OUT
python tools/cmp_variant.py $OUT/x_default.npy 2>&1 | tail -1
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_r04.so python tools/cmp_variant.py $OUT/x_r04.npy 2>&1 | tail -1
python -c "
import numpy as np
a, b = np.load('$OUT/x_default.npy'), np.load('$OUT/x_r04.npy')
print('hill, 64 solves: this build vs round 4 library: bit-identical', np.array_equal(a, b), 'max diff', np.abs(a-b).max())" | tee $OUT/bit_identity.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lanes.py tests/test_gpu_edge.py -m gpu -q 2>&1 | tail -3
for v in sg3 default sg3 default; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  step %.1f ms  launch %.1f ms  scaling kernel %.2f ms  single traj %.2f ms' % (sys.argv[2], r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['scaling_kernel_ms'], r['single_traj_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
;;
9)
# piece-aligned sample chunks (results move at rounding level: no bit identity with round 4 any more): A/B against the plain chunks (noalign), the tests that
# compare with the oracle, the drift / bucket statistics at N = 4096 again
OUT=gpurun_out/r05m; mkdir -p $OUT
for v in noalign default noalign default; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  step %.1f ms  launch %.1f ms  frac %.3f  converged %.3f  evals/traj %.2f  single traj %.2f ms' % (sys.argv[2], r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac'], r['evals_per_traj'], r['single_traj_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -6 $OUT/gpu_tests.txt | cut -c1-300
UPH_PB_ONLY_YAML=1 UPH_PB_THREADS=96 timeout 900 python tools/parity_buckets.py 4096 $OUT/parity_buckets_hill_4096.json hill > $OUT/parity_buckets_hill_4096.txt 2>&1
tail -9 $OUT/parity_buckets_hill_4096.txt | cut -c1-700
;;
10)
# generic A/B of the in-tree library against one variant build: bash tools/r05_runs.sh 10 <variant> [tag]; then the oracle-comparing tests
V=${2:-prev}; OUT=gpurun_out/${3:-r05ab}; mkdir -p $OUT
for v in $V default $V default; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  step %.1f ms  launch %.1f ms  frac %.3f  converged %.3f  evals/traj %.2f  single traj %.2f ms (%s iterations, %.4f ms each)' % (sys.argv[2], r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac'], r['evals_per_traj'], r['single_traj_ms'], r.get('single_traj_lbfgs_iters'), r['ms_per_lbfgs_iter']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forced.py tests/test_gpu_edge.py tests/test_gpu_lanes.py tests/test_gpu_buckets.py tests/test_gpu_km2.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -4 | cut -c1-300
;;
11)
# final record of the round: smoke, soak, whole GPU tier, tools/profile.sh, the bench lines with the fresh counter pass in place, drift / bucket tables at N = 4096
TAG=${2:-r05r}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 600 python tools/soak.py 2>&1 | tail -4 | tee $OUT/soak.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "all rc $?" >> $OUT/gpu_tests.txt
tail -3 $OUT/gpu_tests.txt | cut -c1-300
bash tools/profile.sh $TAG 2>&1 | tail -12 | cut -c1-200
cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
timeout 600 python bench.py --workload km2 --steps 3 --warmup 1 > $OUT/bench_km2.json 2> $OUT/bench_km2.err; echo "km2 rc $?"
UPH_PB_ONLY_YAML=1 UPH_PB_THREADS=96 timeout 900 python tools/parity_buckets.py 4096 $OUT/parity_buckets_hill_4096.json hill > $OUT/parity_buckets_hill_4096.txt 2>&1
tail -9 $OUT/parity_buckets_hill_4096.txt | cut -c1-700
UPH_PB_THREADS=96 timeout 900 python tools/parity_buckets.py 4096 $OUT/parity_buckets_desert_4096.json desert > $OUT/parity_buckets_desert_4096.txt 2>&1
tail -9 $OUT/parity_buckets_desert_4096.txt | cut -c1-700
;;
12)
# the gather without its dead register pair (loadCell<.., WITH_Z = false>): bit identity against the r05r library, A/B of the solve launch (+ the variant that
# requests the next chunk's gather before the scatter, -DUPH_GATHER_AHEAD=1), one full bench line of the new build for the penalty kernel
OUT=gpurun_out/${2:-r05s}; mkdir -p $OUT
python tools/cmp_variant.py $OUT/x_default.npy 2>&1 | tail -1
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_r05r.so python tools/cmp_variant.py $OUT/x_r05r.npy 2>&1 | tail -1
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_ahead.so python tools/cmp_variant.py $OUT/x_ahead.npy 2>&1 | tail -1
python -c "
import numpy as np
a, b, c = np.load('$OUT/x_default.npy'), np.load('$OUT/x_r05r.npy'), np.load('$OUT/x_ahead.npy')
print('hill, 64 solves: this build vs the r05r library bit-identical', np.array_equal(a, b), '; gather-ahead variant bit-identical', np.array_equal(a, c), 'max diff', np.abs(a - c).max())" | tee $OUT/bit_identity.txt
for v in r05r default ahead r05r default ahead; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  step %.1f ms  launch %.1f ms  frac %.3f  converged %.3f  evals/traj %.2f  single traj %.2f ms (%s iterations, %.4f ms each)' % (sys.argv[2], r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac'], r['evals_per_traj'], r['single_traj_ms'], r.get('single_traj_lbfgs_iters'), r['ms_per_lbfgs_iter']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
unset UNEVENHIP_LIB
timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY' | tee -a $OUT/ab.txt
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('new build, full line: value %.0f  launch %.1f ms  frac %.3f  penalty kernel %.3f / %.3f  B8192 %.0f  B4096 %.0f  B256 %.0f  astar %.0f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['roofline']['penalty_kernel']['frac'], r['roofline']['penalty_kernel']['frac_hill_trajectory_x256'], r['traj_opts_per_s_B8192'], r['traj_opts_per_s_B4096'], r['traj_opts_per_s_B256'], r['traj_opts_per_s_astar_seeded']))
PY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_km2.py tests/test_gpu_map.py -m gpu -q 2>&1 | tail -3 | cut -c1-300
;;
13)
# generic: bit identity of the in-tree library against one variant build (64 hill solves), then case 10's A/B and tests: bash tools/r05_runs.sh 13 <variant> [tag]
V=${2:-prev}; OUT=gpurun_out/${3:-r05ab}; mkdir -p $OUT
python tools/cmp_variant.py $OUT/x_default.npy 2>&1 | tail -1
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$V.so python tools/cmp_variant.py $OUT/x_$V.npy 2>&1 | tail -1
python -c "
import numpy as np
a, b = np.load('$OUT/x_default.npy'), np.load('$OUT/x_$V.npy')
print('hill, 64 solves: this build vs the $V library bit-identical', np.array_equal(a, b), 'max diff', np.abs(a - b).max())" | tee $OUT/bit_identity.txt
bash tools/r05_runs.sh 10 $V ${3:-r05ab}
;;
14)
# A/B of one or more variant builds against the in-tree library, nothing else: bash tools/r05_runs.sh 14 <tag> <variant> [<variant> ...]; then the per-evaluation parity
# tests on the first variant
OUT=gpurun_out/${2:-r05ab}; mkdir -p $OUT; shift 2
python tools/cmp_variant.py $OUT/x_default.npy 2>&1 | tail -1
for v in "$@"; do
  UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so python tools/cmp_variant.py $OUT/x_$v.npy 2>&1 | tail -1
  python -c "
import numpy as np
a, b = np.load('$OUT/x_default.npy'), np.load('$OUT/x_$v.npy')
print('hill, 64 solves: $v vs the in-tree library bit-identical', np.array_equal(a, b), 'max diff', np.abs(a - b).max())" | tee -a $OUT/bit_identity.txt
done
for v in default "$@" default "$@"; do
  if [ "$v" = default ]; then unset UNEVENHIP_LIB; else export UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print('%-10s value %.0f traj/s  step %.1f ms  launch %.1f ms  frac %.3f  converged %.3f  evals/traj %.2f  single traj %.2f ms (%s iterations, %.4f ms each)' % (sys.argv[2], r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['converged_frac'], r['evals_per_traj'], r['single_traj_ms'], r.get('single_traj_lbfgs_iters'), r['ms_per_lbfgs_iter']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
done 2>&1 | tee $OUT/ab.txt
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_$1.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q 2>&1 | tail -2 | cut -c1-300
;;
15)
# drift statistics on larger samples (final build): hill N = 16384, volcano N = 4096
OUT=gpurun_out/${2:-r05w}; mkdir -p $OUT
UPH_PB_ONLY_YAML=1 UPH_PB_THREADS=192 timeout 1200 python tools/parity_buckets.py 16384 $OUT/parity_buckets_hill_16384.json hill > $OUT/parity_buckets_hill_16384.txt 2>&1
tail -9 $OUT/parity_buckets_hill_16384.txt | cut -c1-700
UPH_PB_THREADS=192 timeout 900 python tools/parity_buckets.py 4096 $OUT/parity_buckets_vocano_4096.json vocano > $OUT/parity_buckets_vocano_4096.txt 2>&1
tail -9 $OUT/parity_buckets_vocano_4096.txt | cut -c1-700
;;
esac
