"""Diagnosis of tests/test_gpu_km2.py::test_far_from_origin...: the 48 far-from-origin problems with inner_max_iter = 8 solved twice on one context, on a fresh
context, and with 128 / 256 / 512 lanes per trajectory; per configuration the three largest way-point deviations from the window oracle and whether two runs
agree bit for bit (a branchy problem moves with the summation order of the lane count; a race would move between identical runs)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U              # noqa: E402
from oracle import oracle_py as O           # noqa: E402
from uneven_planner_amd import scenes       # noqa: E402

big = U.UnevenMap(dict(map_size_x=640.0, map_size_y=640.0, xy_resolution=0.25), storage="f32").fill_fbm()
nx, ny = int(big.voxel_num[0]), int(big.voxel_num[1])
far, seed = [], 7100
while len(far) < 48:
    p = scenes.local_problems(1, seed0=seed, half=315.0, dmin=4.0, dmax=9.0, occ_r2=big.occ_r2_buffer, grid=(nx, ny, big.xy_resolution, big.map_origin[0], big.map_origin[1]))[0]
    seed += 1
    if max(abs(p["init_xy"][0, 0]), abs(p["init_xy"][1, 0])) > 200.0:
        far.append(p)
prm = dict(inner_max_iter=8.0)
ref = []
for p in far:
    og, q, sh = O.window_oracle(big, p)
    r = O.OracleALM(og, prm).optimize(q)
    nin = p["inner_xy"].shape[1]
    xo = np.array(r["x"], dtype=np.float64)
    xo[1:1 + 2 * nin:2] += sh[0]
    xo[2:2 + 2 * nin:2] += sh[1]
    ref.append((xo, nin, r))


def dev_err(out):
    e = []
    for d, (xo, nin, r) in zip(out, ref):
        e.append(np.abs(d["x"][1:] - xo[1:]).max() / max(1.0, np.ptp(xo[1:1 + 2 * nin:2]), np.ptp(xo[2:2 + 2 * nin:2])))
    return np.array(e)


base = None
for lanes in (0, 128, 256, 512):
    o = U.ALMTrajOpt(big, prm)
    if lanes:
        o.set_lanes(lanes)
    o.set_rho(1.0)
    a = o.optimize_batch(far)
    o.set_rho(1.0)
    b = o.optimize_batch(far)
    same = all(np.array_equal(x["x"], y["x"]) for x, y in zip(a, b))
    e = dev_err(a)
    w = np.argsort(-e)[:3]
    if base is None:
        base = a
    print("lanes %3s: two runs bit-identical %s | same as the automatic choice %s | worst way-point errors %s (problems %s, oracle iterations %s)" % (
        lanes or "auto", same, all(np.array_equal(x["x"], y["x"]) for x, y in zip(a, base)), ["%.1e" % v for v in e[w]], w.tolist(), [ref[i][2]["lbfgs_iters"] for i in w]))
