#!/bin/bash
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
