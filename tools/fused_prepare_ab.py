"""A/B: reset + initScaling as a launch of its own (default) against inside the solve launch (uph_ctx_set_fused_prepare: needs
tools/experiments/r04_fused_prepare.patch applied), bench batch.
usage: python tools/fused_prepare_ab.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U              # noqa: E402
from uneven_planner_amd import scenes       # noqa: E402

m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(16384, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
ref = None
for B in (16384, 8192):
    for on in (0, 1, 0, 1):
        o = U.ALMTrajOpt(m); o.set_fused_prepare(on); o.upload(probs[:B])
        o.set_rho(1.0); o.solve()
        ts, ks, ps = [], [], []
        for _ in range(3):
            o.set_rho(1.0); t0 = time.perf_counter(); o.solve(); ts.append(time.perf_counter() - t0)
            st = o.stats(); ks.append(st["kernel_ms"]); ps.append(st["prepare_ms"])
        out = o.download(full=False)
        sig = (np.array([q["ret"] for q in out]), np.array([q["cost"] for q in out]))
        same = ""
        if B == 16384:
            if ref is None: ref = sig
            else: same = "  results equal to the first run: %s" % bool(np.array_equal(ref[0], sig[0]) and np.array_equal(ref[1], sig[1]))
        print("B %5d fused %d: step %.1f ms wall (%s) = prepare %.1f + solve %.1f ms by events -> %.0f traj-opts/s%s" % (
            B, on, np.mean(ts) * 1e3, ["%.1f" % (v * 1e3) for v in ts], np.mean(ps), np.mean(ks), B / np.mean(ts), same), flush=True)
        del o
