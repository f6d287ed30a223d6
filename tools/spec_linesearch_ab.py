"""speculative line search (uph_ctx_set_speculation: needs tools/experiments/r04_speculative_linesearch.patch applied) against the plain one on small
batches: bit-equality of the results and solve times.
usage: python tools/spec_linesearch_ab.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U              # noqa: E402
from uneven_planner_amd import scenes       # noqa: E402

m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
rnd = scenes.random_problems(85, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
for tag, probs in (("hill trajectory", [scenes.hill_problem()]), ("8 random", rnd[:8]), ("85 random", rnd)):
    outs = {}
    for on in (0, 1):
        o = U.ALMTrajOpt(m); o.set_speculation(on); o.upload(probs)
        o.set_rho(1.0); o.solve()
        ms = []
        for _ in range(3):
            o.set_rho(1.0); o.solve(); ms.append(o.stats()["kernel_ms"])
        st = o.stats()
        outs[on] = o.download(full=False)
        print("%-16s speculation %d: solve kernel %.2f ms (%s), prepare %.2f ms, evals %d, iterations %d -> %.4f ms per L-BFGS iteration" % (
            tag, on, min(ms), ["%.2f" % v for v in ms], st["prepare_ms"], st["evals"], st["lbfgs_iters"], min(ms) / max(1, st["lbfgs_iters"]) * len(probs)), flush=True)
        del o
    same = all(a["ret"] == b["ret"] and a["cost"] == b["cost"] and a["evals"] == b["evals"] and np.array_equal(a["x"], b["x"]) and np.array_equal(a["c_xy"], b["c_xy"])
               for a, b in zip(outs[0], outs[1]))
    print("%-16s results bit-identical: %s" % (tag, same), flush=True)
