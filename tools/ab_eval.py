"""A/B helper of round 6 (library variant through UNEVENHIP_LIB): the penalty kernel (uph_eval_batch, 20 evaluations per launch) and a full solve on
(a) the bench's hill batch and (b) a batch of SHORT problems only (small LDS footprint: more workgroups per CU where the registers allow them).
usage: python tools/ab_eval.py [B] [dmax of the short class]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dmax = float(sys.argv[2]) if len(sys.argv) > 2 else 5.5
m = U.UnevenMap()
m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
grid = (nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1])
sets = [("batch", scenes.random_problems(B, seed0=1000, occ_r2=m.occ_r2_buffer, grid=grid))]
if dmax > 0:
    sets.append(("short", scenes.random_problems(B, seed0=1000, dmin=3.0, dmax=dmax, occ_r2=m.occ_r2_buffer, grid=grid)))
for tag, probs in sets:
    opt = U.ALMTrajOpt(m)
    opt.set_lanes(128)
    opt.upload(probs)
    opt.init_scaling_batch()
    R = 20
    opt.eval_batch(None, repeat=R)
    ms = []
    for _ in range(3):
        opt.eval_batch(None, repeat=R)
        ms.append(opt.stats()["kernel_ms"])
    S = sum(s["S"] for s in opt._sizes)
    pieces = [s["Nxy"] for s in opt._sizes]
    line = "%-6s pieces mean %.1f max %d | eval x%d: %.3f ms  frac %.3f" % (tag, np.mean(pieces), max(pieces), R, min(ms), S * R * 376 / (min(ms) * 1e-3) / 8e12)
    opt.upload(probs)
    sm = []
    for _ in range(3):
        opt.set_rho(1.0)
        opt.solve()
        st = opt.stats()
        sm.append(st["kernel_ms"])
        pm = st["prepare_ms"]
    out = opt.download(full=False)
    import hashlib
    hx = hashlib.sha1(np.concatenate([o["x"] for o in out]).tobytes()).hexdigest()[:10]
    print(line + " | solve: %.2f ms (%.0f traj/s), scaling %.2f ms, evals %d, converged %.3f, results %s" % (min(sm), B / min(sm) * 1e3, pm, st["evals"], np.mean([o["ret"] == 0 for o in out]), hx), flush=True)
