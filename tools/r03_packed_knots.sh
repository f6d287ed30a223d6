cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03n; mkdir -p $OUT
make -C oracle -s 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_lanes.py tests/test_gpu_forced.py -q -x > $OUT/tests.txt 2>&1; tail -4 $OUT/tests.txt
python tools/cmp_variant.py $OUT/res_head.npy
UNEVENHIP_LIB=$GRAFT_REPO_ROOT/build/variants/libunevenhip_prepack.so python tools/cmp_variant.py $OUT/res_pre.npy
python - <<'PY'
import numpy as np
a = np.load("gpurun_out/r03n/res_head.npy", allow_pickle=True); b = np.load("gpurun_out/r03n/res_pre.npy", allow_pickle=True)
print("packed knot solve bit-identical to the two-wave one:", all(np.array_equal(x, y) for x, y in zip(a, b)))
PY
bash tools/gpu_ab.sh r03n default prepack 2>&1 | tee $OUT/ab.txt
bash tools/pmc_eval.sh r03n default | head -9
