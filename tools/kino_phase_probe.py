"""where an expansion of the front-end search spends its cycles: python tools/kino_phase_probe.py [B] with a -DUPH_KINO_PROF build
(UNEVENHIP_LIB=build/variants/libunevenhip_kprof.so): per-section shader-clock cycles per expansion, one query alone and B queries at once."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U              # noqa: E402
from uneven_planner_amd import scenes       # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
S, G = scenes.random_queries(B, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
ka = U.KinoAstar(m)
names = ["heap top, node fetch, one-shot test", "stage 0: state transit, keys", "pop", "stage 1: table nodes, terrain, cost", "replay of the table order", "node writes, pushes, loop"]
for nb in (1, B):
    cap = 64
    s, g = np.ascontiguousarray(S[:nb]), np.ascontiguousarray(G[:nb])
    r = ka.plan_batch(s, g, path_cap=cap)
    # the profiling build leaves its six counters in the last two path rows, whatever n_path says: read the raw array
    import ctypes as C
    from uneven_planner_amd import _lib
    paths = np.zeros((nb, cap, 3)); npth, st, it, un = (np.zeros(nb, dtype=np.int32) for _ in range(4))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    _lib.check(ka.L.uph_kino_plan_batch(ka.h, nb, s.ctypes.data_as(_lib.DP), g.ctypes.data_as(_lib.DP), cap, paths.ctypes.data_as(_lib.DP), ip(npth), ip(st), ip(it), ip(un), 0, 0, None), "plan")
    cyc = paths[:, cap - 2:, :].reshape(nb, 6)
    tot_it = it.sum()
    print("B %d: kernel %.2f ms, %d expansions, %.0f cycles per expansion (%.2f us at 2.4 GHz)" % (nb, ka.stats()["kernel_ms"], tot_it, cyc.sum() / tot_it, cyc.sum() / tot_it / 2400.0))
    for k in range(6):
        print("   %-40s %8.0f cycles / expansion  %5.1f %%" % (names[k], cyc[:, k].sum() / tot_it, 100.0 * cyc[:, k].sum() / cyc.sum()))
