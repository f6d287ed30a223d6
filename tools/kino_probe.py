"""front-end search probe: python tools/kino_probe.py <wps> <flags> <B> -- one configuration of the search kernel (waves per SIMD; flags bit 0 dynamic
query hand-out, bit 1 sincosFast) on the hill map: agreement of the first 32 queries with the CPU oracle (status, expansions, nodes) and throughput."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U              # noqa: E402
from oracle import oracle_py as O           # noqa: E402
from uneven_planner_amd import scenes       # noqa: E402

wps, flags, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
S, G = scenes.random_queries(B, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
ka = U.KinoAstar(m, slots=int(os.environ.get("UPH_KINO_SLOTS", 256 * 4 * wps)))
ka.set_wps(wps); ka.set_flags(flags)
r = ka.plan_batch(S[:32], G[:32], path_cap=8)
g = O.OracleGrid(); g.set_cells(m.map_buffer); g.set_occ(m.occ_buffer, m.occ_r2_buffer)
ok = O.OracleKinoAstar(g)
same = sum(int(d["status"] == o["status"] and d["iter_num"] == o["iter_num"] and d["use_node_num"] == o["use_node_num"]) for d, o in ((r[i], ok.plan(S[i], G[i])) for i in range(32)))
out = "wps %d flags %d: %d / 32 queries with the oracle's status, expansions and node count" % (wps, flags, same)
for nb in sorted(set([1, min(B, 2048), min(B, 8192), B])):
    t0 = time.perf_counter(); r = ka.plan_batch(S[:nb], G[:nb], path_cap=64); dt = time.perf_counter() - t0
    it = np.array([q["iter_num"] for q in r])
    out += " | B %d: %.0f queries/s, kernel %.1f ms, %.2f M expansions/s" % (nb, nb / dt, ka.stats()["kernel_ms"], it.sum() / dt / 1e6)
print(out, flush=True)
