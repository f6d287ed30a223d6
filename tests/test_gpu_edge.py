"""GPU edge cases through the C-ABI: smallest / largest problems, limits and argument errors, the forest parameter set
(use_scaling = false -> fixed curvature / sigma scales, Q6), a different int_K, trajectories leaving the map."""
import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def devmap(analytic_cells):
    import uneven_planner_amd as U
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    return m


def _tiny_problem(n_xy=1, n_yaw=1):
    from uneven_planner_amd import resample
    path = resample.hermite_path((0.0, 0.0, 0.3), (0.3 * (n_xy + 1) - 0.05, 0.1, 0.4))
    p = resample.resample_path(path)
    p["inner_xy"] = p["inner_xy"][:, :n_xy]
    p["inner_yaw"] = p["inner_yaw"][:max(n_yaw, n_xy)]
    return p


def test_smallest_problem_matches_oracle(devmap, oracle, oracle_grid):
    import uneven_planner_amd as U
    p = _tiny_problem(1, 1)
    assert p["inner_xy"].shape[1] == 1 and p["inner_yaw"].shape[0] == 1          # Nxy = 2, Nyaw = 2, n = 4
    opt = U.ALMTrajOpt(devmap)
    opt.upload([p])
    f, g = opt.eval_batch(opt.x0_packed([p]))
    a = oracle.OracleALM(oracle_grid)
    x0 = a.setup(p)
    fo, go, _ = a.eval(x0)
    assert abs(f[0] - fo) / abs(fo) < 1e-9 and rel(go, g[0]) < 1e-9
    out = opt.optimize_batch([p])[0]
    assert out["ret"] in (0, 2) and np.isfinite(out["x"]).all()


def _winding_path(k, amp):
    """a sinusoid across the 10 m map sampled every ~0.06 m like a front-end path; yaw = unwrapped tangent"""
    x = np.linspace(-4.6, 4.6, 4001)
    y = -0.5 + amp * np.sin(k * x)
    arc = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])
    sq = np.linspace(0.0, arc[-1], int(arc[-1] / 0.06) + 1)
    xs, ys = np.interp(sq, arc, x), np.interp(sq, arc, y)
    yaw = np.unwrap(np.arctan2(np.gradient(ys), np.gradient(xs)))
    return np.column_stack([xs, ys, yaw]), arc[-1]


@pytest.mark.parametrize("target", [64, 128])
def test_largest_supported_problem(devmap, oracle, oracle_grid, target):
    """paths at the old and the new compiled limit -- about 19 m (Nxy <= 64, Nyaw <= 128: four registers per lane in the two-loop) and about
    38 m (Nxy <= 128 = UPH_MAX_PIECE_XY, Nyaw <= 256, n up to 511: eight registers per lane, two-row prefetch ring): one evaluation against
    the oracle, then a capped solve (two ALM passes of at most 40 L-BFGS iterations: history up to 39 pairs through the NQ-register two-loop)"""
    import uneven_planner_amd as U
    from uneven_planner_amd import resample
    p = None
    for k in np.arange(0.4, 6.0, 0.02):                       # more periods = longer path: pick the first one that lands just under the limit
        path, length = _winding_path(k, 3.2)
        if length > 0.3 * (target - 6):
            p = resample.resample_path(path)
            break
    nxy, nyaw = p["inner_xy"].shape[1] + 1, p["inner_yaw"].shape[0] + 1
    assert target - 8 <= nxy <= target and nyaw <= 2 * target, (nxy, nyaw)
    opt = U.ALMTrajOpt(devmap)
    opt.upload([p])
    f, g = opt.eval_batch(opt.x0_packed([p]))
    a = oracle.OracleALM(oracle_grid)
    x0 = a.setup(p)
    fo, go, _ = a.eval(x0)
    assert abs(f[0] - fo) / abs(fo) < 1e-9 and rel(go, g[0]) < 1e-8
    prm = dict(inner_max_iter=40.0, max_iter=1.0)
    o2 = U.ALMTrajOpt(devmap, params=prm)
    o2.set_rho(1.0)
    out = o2.optimize_batch([p])[0]
    ro = oracle.OracleALM(oracle_grid, prm).optimize(p)
    assert out["ret"] == ro["ret"] and out["lbfgs_iters"] == ro["lbfgs_iters"] and out["evals"] == ro["evals"]
    assert abs(out["cost"] - ro["cost"]) <= 1e-5 * abs(ro["cost"]) and rel(ro["x"], out["x"]) < 1e-5, (nxy, rel(ro["x"], out["x"]))


def test_limits_and_argument_errors(devmap):
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    opt = U.ALMTrajOpt(devmap)
    p = scenes.random_problems(1, seed0=2000, dmin=3.0, dmax=4.0)[0]
    big = dict(p)
    big["inner_xy"] = np.zeros((2, 130))                     # 131 position pieces > UPH_MAX_PIECE_XY = 128
    big["inner_yaw"] = np.zeros(262)
    with pytest.raises(U._lib.UnevenHipError):
        opt.upload([big])                                       # UPH_ERR_LIMIT
    bad = dict(p)
    bad["inner_yaw"] = p["inner_yaw"][:p["inner_xy"].shape[1] - 2]      # piece_yaw < piece_xy
    with pytest.raises(U._lib.UnevenHipError):
        opt.upload([bad])
    with pytest.raises(U._lib.UnevenHipError):
        U.ALMTrajOpt(devmap, params=dict(mem_size=1024))        # beyond UPH_MAX_MEM
    opt.upload([p])                                             # the context stays usable after an error
    assert opt.optimize_batch([p])[0]["ret"] in (0, 2)


@pytest.mark.parametrize("params", [dict(use_scaling=False, rho_T=500.0, max_sig=0.001),      # run_forest.yaml deltas (Q6 branch)
                                    dict(int_K=8), dict(past=0, mem_size=16)])
def test_parameter_variants_match_oracle(devmap, oracle, oracle_grid, small_problems, params):
    import uneven_planner_amd as U
    opt = U.ALMTrajOpt(devmap, params=params)
    p = small_problems[0]
    opt.upload([p])
    rng = np.random.default_rng(2)
    K1 = int(params.get("int_K", 16)) + 1
    S = (p["inner_xy"].shape[1] + 1) * K1
    lam, mu = rng.normal(size=S) * 0.1, np.abs(rng.normal(size=6 * S)) * 0.1
    opt.set_state(lam=[lam], mu=[mu])
    f, g = opt.eval_batch(opt.x0_packed([p]))
    op = {k: (float(v) if not isinstance(v, bool) else float(v)) for k, v in params.items()}
    a = oracle.OracleALM(oracle_grid, op)
    x0 = a.setup(p)
    a.set_state(lam=lam, mu=mu)
    fo, go, _ = a.eval(x0)
    assert abs(f[0] - fo) / abs(fo) < 1e-9 and rel(go, g[0]) < 1e-9
    out = opt.optimize_batch([p])[0]
    ro = oracle.OracleALM(oracle_grid, op).optimize(p)
    assert out["ret"] == ro["ret"] or max(out["alm_iters"], ro["alm_iters"]) >= 9
    # the unscaled (forest) parameter set is far worse conditioned (fixed 10x / 1000x constraint scales, Q6): its chaotic spread is larger
    assert abs(out["cost"] - ro["cost"]) / abs(ro["cost"]) < (0.3 if params.get("use_scaling") is False else 5e-2)


def test_trajectory_leaving_the_map(devmap, oracle, oracle_grid):
    """samples outside the 10 m x 10 m map read zeros (uneven_map.h:260-265): same on both sides"""
    import uneven_planner_amd as U
    from uneven_planner_amd import resample
    p = resample.make_problem((4.2, 4.2, 0.2), (5.6, 4.8, 0.3))
    opt = U.ALMTrajOpt(devmap)
    opt.upload([p])
    f, g = opt.eval_batch(opt.x0_packed([p]))
    a = oracle.OracleALM(oracle_grid)
    x0 = a.setup(p)
    fo, go, _ = a.eval(x0)
    assert abs(f[0] - fo) / abs(fo) < 1e-9 and rel(go, g[0]) < 1e-9


def test_single_piece_problems_match_oracle(analytic_cells, oracle, oracle_grid):
    """a goal closer than one piece length has no inner position way-point (piece_xy = 1) and, below half a piece length, no inner yaw
    way-point either (piece_yaw = 1): the reference solves the single quintic piece (no knot system), and so must the device --
    evaluation, initScaling and the full solve against the oracle, alone and inside a batch of ordinary problems"""
    import uneven_planner_amd as U
    from uneven_planner_amd import resample, scenes
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    shorts = [resample.make_problem((0.3, -0.2, 0.2), (0.3 + dx, -0.2 + dy, 0.2 + dyaw)) for dx, dy, dyaw in
              [(0.25, 0.02, 0.1), (0.12, 0.0, 0.0), (0.2, -0.1, -0.3), (0.29, 0.0, 0.0)]]
    assert [(p["inner_xy"].shape[1], p["inner_yaw"].shape[0]) for p in shorts] == [(0, 1), (0, 0), (0, 1), (0, 1)]
    # one position piece with MANY yaw pieces (the test node's stage emits extra yaw way-points; a caller may pass any piece_yaw >= piece_xy)
    many = resample.make_problem((0.3, -0.2, 0.2), (0.58, -0.15, 0.5))
    many["inner_yaw"] = np.linspace(0.2, 0.5, 9)[1:-1].copy()
    assert many["inner_xy"].shape[1] == 0 and many["inner_yaw"].shape[0] == 7
    shorts.append(many)
    good = scenes.random_problems(2, seed0=2100, dmin=3.0, dmax=5.0)
    for lanes in (64, 128, 256, 512):
        opt = U.ALMTrajOpt(m)
        opt.set_lanes(lanes)
        opt.upload(shorts)
        f, gs = opt.eval_batch(opt.x0_packed(shorts))
        opt.init_scaling_batch()
        sc = opt.download()
        opt.set_rho(1.0)
        mixed = opt.optimize_batch([good[0]] + shorts + [good[1]])
        for i, p in enumerate(shorts):
            a = oracle.OracleALM(oracle_grid)
            x0 = a.setup(p)
            fo, go, _ = a.eval(x0)
            assert abs(f[i] - fo) / abs(fo) < 1e-9 and rel(go, gs[i]) < 1e-9
            a.init_scaling(x0)
            st = a.get_state()
            assert abs(sc[i]["scale_fx"] - st["scale_fx"]) / st["scale_fx"] < 1e-9 and rel(st["scale_cx"], sc[i]["scale_cx"]) < 1e-9
            ro = oracle.OracleALM(oracle_grid).optimize(p)
            o = mixed[1 + i]
            if p is many:
                # eight variables, ten ALM passes, > 200 iterations: a solve in the optimiser's sensitive regime; its
                # evaluation and scaling are pinned above, the solve itself only has to end like a solve
                assert o["ret"] in (0, 2) and o["evals"] > 50
                continue
            # one or two variables and at most a dozen iterations per pass: no room for chaotic drift, the solves agree tightly
            assert o["ret"] == ro["ret"] and o["alm_iters"] == ro["alm_iters"]
            assert abs(o["cost"] - ro["cost"]) / abs(ro["cost"]) < 1e-8 and np.abs(o["x"] - np.asarray(ro["x"])).max() < 1e-8
    alone = U.ALMTrajOpt(m)
    alone.set_rho(1.0)
    ref = alone.optimize_batch(good)
    assert np.array_equal(ref[0]["x"], mixed[0]["x"]) and np.array_equal(ref[1]["x"], mixed[-1]["x"])


def test_unsupported_problem_does_not_fail_its_neighbours(analytic_cells):
    """a path beyond UPH_MAX_PIECE_XY (or with fewer yaw than position pieces) gets ret_code UPH_RET_UNSUPPORTED; the other problems of
    the batch are solved exactly as they are alone"""
    import uneven_planner_amd as U
    from uneven_planner_amd import resample, scenes
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    good = scenes.random_problems(3, seed0=2100, dmin=3.0, dmax=5.0)
    fewer_yaw = dict(good[0])
    fewer_yaw["inner_yaw"] = good[0]["inner_yaw"][:good[0]["inner_xy"].shape[1] - 2]
    long_ = resample.resample_path(np.column_stack([np.linspace(-4.5, 4.5, 400), np.linspace(-4.4, 4.4, 400) ** 3 / 20.0, np.zeros(400)]), piece_len=0.06)
    assert long_["inner_xy"].shape[1] + 1 > 128
    opt = U.ALMTrajOpt(m)
    opt.set_rho(1.0)
    out = opt.optimize_batch([good[0], fewer_yaw, good[1], long_, good[2]])
    assert [o["ret"] for o in out][1::2] == [4, 4] and out[1]["last_lbfgs_ret"] == -1 and out[3]["last_lbfgs_ret"] == -4
    alone = U.ALMTrajOpt(m)
    alone.set_rho(1.0)
    ref = alone.optimize_batch(good)
    for a, b in zip(ref, out[0::2]):
        assert a["ret"] == b["ret"] and a["cost"] == b["cost"] and np.array_equal(a["x"], b["x"])
    st = opt.stats()
    assert st["evals"] == sum(o["evals"] for o in out)            # the placeholders' work is not counted
    with pytest.raises(U._lib.UnevenHipError):
        opt.eval_batch(None)                                      # packed-array hooks need a clean batch
    with pytest.raises(U._lib.UnevenHipError):
        U.ALMTrajOpt(m).optimize_batch([long_])                   # nothing solvable in the batch: an error, as before


def test_async_solves_on_two_contexts_equal_the_blocking_ones(analytic_cells):
    """uph_batch_solve_async / uph_batch_wait: two contexts in flight at once give exactly what each gives alone; misuse is refused"""
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    pa = scenes.random_problems(300, seed0=8100, dmin=3.0, dmax=6.0)
    pb = scenes.random_problems(300, seed0=8500, dmin=3.0, dmax=6.0)
    ref = []
    for pp in (pa, pb):
        o = U.ALMTrajOpt(m)
        o.set_rho(1.0)
        ref.append(o.optimize_batch(pp))
    a, b = U.ALMTrajOpt(m), U.ALMTrajOpt(m)
    a.upload(pa); b.upload(pb)
    a.set_rho(1.0); b.set_rho(1.0)
    a.solve_async(); b.solve_async()
    with pytest.raises(U._lib.UnevenHipError):
        a.solve_async()                                   # one asynchronous solve per context
    with pytest.raises(U._lib.UnevenHipError):
        a.download()                                      # results only after the wait
    b.wait(); a.wait()
    with pytest.raises(U._lib.UnevenHipError):
        a.wait()
    for got, want in ((a.download(), ref[0]), (b.download(), ref[1])):
        assert all(g["ret"] == w["ret"] and g["cost"] == w["cost"] and np.array_equal(g["x"], w["x"]) for g, w in zip(got, want))
    assert a.stats()["evals"] == sum(o["evals"] for o in ref[0])
