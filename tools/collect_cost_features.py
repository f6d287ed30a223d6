"""Collects, per trajectory of a hill-scene batch: a-priori features (n, heading change of the initial path, the post-scaling report at
x0 = max |vx|, |ax|, |ay|, curvature, attitude, sigma, non-holonomic error) and the measured solve cost (workgroup cycles, ALM passes,
L-BFGS iterations) -> gpurun_out/cost_features.npz, for fitting the launch-order cost model (uph_batch_upload).
usage (GPU box): python tools/collect_cost_features.py [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
if len(sys.argv) > 2 and sys.argv[2] == "km2":        # BASELINE.json configs[4] scene (256 m square is enough for the statistics)
    from uneven_planner_amd.uneven_map import km2_map, km2_problems
    m = km2_map(256.0)
    probs = km2_problems(m, 256.0, B, 0)
else:
    m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
    nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
    probs = scenes.random_problems(B, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
opt = U.ALMTrajOpt(m); opt.upload(probs)
opt.init_scaling_batch()
rep0 = opt.getMaxVxAxAyCurAttSig()
st0 = opt.download()
scale_fx = np.array([o["scale_fx"] for o in st0])
opt.set_rho(1.0); opt.solve()
cy = opt.cycles().astype(np.float64)
out = opt.download()
turn, dmax_, d2max, kink = [], [], [], []
for p in probs:
    yw = np.concatenate([[p["init_yaw"][0]], p["inner_yaw"], [p["end_yaw"][0]]])
    dy = np.abs(np.diff(yw))
    turn.append(dy.sum()); dmax_.append(dy.max()); d2max.append(np.abs(np.diff(yw, 2)).max() if yw.size > 2 else 0.0)
    xy = np.concatenate([p["init_xy"][:, :1], p["inner_xy"], p["end_xy"][:, :1]], axis=1)
    seg = np.diff(xy, axis=1)
    ang = np.arctan2(seg[1], seg[0])
    da = np.abs(np.arctan2(np.sin(np.diff(ang)), np.cos(np.diff(ang))))
    kink.append(da.max() if da.size else 0.0)
np.savez(os.path.join("gpurun_out", "cost_features.npz"), n=np.array([s["n"] for s in opt._sizes]), turn=np.array(turn), dmax=np.array(dmax_), d2max=np.array(d2max), kink=np.array(kink), rep0=rep0, scale_fx=scale_fx,
         cycles=cy[:, 6], alm=np.array([o["alm_iters"] for o in out]), iters=np.array([o["lbfgs_iters"] for o in out]), ret=np.array([o["ret"] for o in out]),
         total_time=np.array([p["total_time"] for p in probs]), kernel_ms=opt.stats()["kernel_ms"])
print("saved", B, "kernel_ms", opt.stats()["kernel_ms"])
