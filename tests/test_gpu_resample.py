"""GPU tier, SURVEY.md 8f row N1: the initial-guess stage feeds the optimiser.  Front-end style paths go through the product's batched
routine (uph_resample_batch) and through the oracle's restatement of the reference stage -- PlanManager::rcvWpsCallBack
(plan_manager/src/plan_manager.cpp:62-132) and the back-end test node's ALMTrajOpt::rcvWpsCallBack (back_end/src/alm_traj_opt.cpp:73-144) --;
the optimizeSE2Traj arguments must agree bit for bit, and what the device makes of the product's arguments must equal what the oracle makes of
its own: first objective evaluation at 1e-9, initScaling at 1e-9, and a full solve on short paths (below the iteration count where the
optimiser's own noise amplification starts to matter) at north_star's 1e-4."""
import math

import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu
KEYS = ("init_xy", "end_xy", "inner_xy", "init_yaw", "end_yaw", "inner_yaw")


def _paths(n, seed, dmin, dmax):
    from uneven_planner_amd import resample as R
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        s = np.array([rng.uniform(-3.5, 3.5), rng.uniform(-3.5, 3.5), rng.uniform(-math.pi, math.pi)])
        g = np.array([rng.uniform(-3.5, 3.5), rng.uniform(-3.5, 3.5), rng.uniform(-math.pi, math.pi)])
        if not (dmin <= np.linalg.norm(g[:2] - s[:2]) <= dmax):
            continue
        p = R.hermite_path(s, g)
        p[:, 2] = np.arctan2(np.sin(p[:, 2]), np.cos(p[:, 2]))          # wrapped yaw column with jumps, as a front-end delivers it
        out.append(p)
    return out


@pytest.mark.parametrize("mode", ["plan_manager", "test_node"])
def test_resampled_inputs_and_first_evaluation_match_oracle(mode, oracle, oracle_grid, analytic_cells):
    import uneven_planner_amd as U
    from uneven_planner_amd import resample as R
    kw = dict(test_mode=1, test_max_vel=0.5) if mode == "test_node" else {}
    ps = _paths(24, 11, 2.0, 7.0)
    prod = R.resample_batch(ps, cap_xy=64, cap_yaw=128, **kw)
    ref = [oracle.resample(p, kw) for p in ps]
    for a, b in zip(prod, ref):                                            # the boundary's inputs, bit for bit
        assert all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in KEYS) and a["total_time"] == b["total_time"]
    if mode == "test_node":                                                # the test node's yaw way-points: yaw comb + one per position node
        assert all(a["inner_yaw"].shape[0] > 2 * a["inner_xy"].shape[1] for a in prod)
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    opt = U.ALMTrajOpt(m)
    opt.upload(prod)
    opt.init_scaling_batch()
    st = opt.download()
    f, g = opt.eval_batch(opt.x0_packed(prod))
    for i, q in enumerate(ref):
        a = oracle.OracleALM(oracle_grid)
        x0 = a.setup(q)
        a.init_scaling(x0)
        so = a.get_state()
        assert rel(so["scale_cx"], st[i]["scale_cx"]) < 1e-9 and abs(so["scale_fx"] - st[i]["scale_fx"]) <= 1e-9 * abs(so["scale_fx"])
        fo, go, _ = a.eval(x0)
        assert abs(f[i] - fo) <= 1e-9 * abs(fo) and rel(go, g[i]) < 1e-9, (mode, i)


def test_short_paths_end_to_end_within_1e_4(oracle, oracle_grid, analytic_cells):
    """front-end path -> product resampler -> uph_optimize_batch against front-end path -> oracle resampler -> oracle solve, on 0.7-2 m goals:
    wherever the oracle's solve takes at most 120 L-BFGS iterations in total (below that the device program stays under 1e-8 on CPU-side
    statistics; the optimiser's noise amplification reaches 1e-4 from ~170 iterations on), final cost and way-points within north_star's 1e-4"""
    import uneven_planner_amd as U
    from uneven_planner_amd import resample as R
    ps = [p for p in _paths(60, 31, 0.7, 2.0)] + _paths(40, 23, 2.0, 3.5)
    ps = [p for p in ps if oracle.resample(p)["inner_xy"].shape[1] >= 1]
    prod = R.resample_batch(ps)
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    opt = U.ALMTrajOpt(m)
    opt.set_rho(1.0)
    out = opt.optimize_batch(prod)
    checked = 0
    for p, d in zip(ps, out):
        r = oracle.OracleALM(oracle_grid).optimize(oracle.resample(p))
        if r["lbfgs_iters"] > 120:
            continue
        checked += 1
        assert d["ret"] == r["ret"]
        assert abs(d["cost"] - r["cost"]) <= 1e-4 * abs(r["cost"]) and rel(r["x"], d["x"]) <= 1e-4, (r["lbfgs_iters"], rel(r["x"], d["x"]))
    assert checked >= 3
