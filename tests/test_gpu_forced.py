"""GPU tier of the teacher-forced late-state tests (see tests/forced_cases.py): the device state machine continues, through the C-ABI
hooks uph_batch_set_lbfgs_state / uph_batch_lbfgs_resume / uph_batch_alm_passes, from states the CPU oracle dumped deep inside
its solves -- wrapped history ring, rejected cautious update, exhausted line search with restore (Q1), hard line-search error,
k = 1001 cancel, ALM passes >= 2 -- and must agree with the oracle's continuation to 1e-9."""
import numpy as np
import pytest

import forced_cases as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def devmap(analytic_cells):
    import uneven_planner_amd as U
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    return m


@pytest.fixture(scope="module")
def probs():
    from uneven_planner_amd import scenes
    return [scenes.hill_problem()] + scenes.random_problems(3, seed0=1000)


def _opt(devmap, params, lanes):
    import uneven_planner_amd as U
    opt = U.ALMTrajOpt(devmap, params)
    opt.set_lanes(lanes)
    return opt


def _load(opt, probs, states):
    opt.upload(probs)
    opt.set_state(lam=[s["lam"] for s in states], mu=[s["mu"] for s in states], scale_cx=[s["scale_cx"] for s in states],
                  scale_fx=[s["scale_fx"] for s in states], rho=[s["rho"] for s in states])
    opt.set_lbfgs_state(states)


@pytest.mark.parametrize("lanes", [128, 256])
@pytest.mark.parametrize("mem", [5, 7, 8])
def test_wrapped_ring_one_and_six_iterations(devmap, oracle_grid, probs, mem, lanes):
    prm = dict(mem_size=mem)
    states = [F.capture(oracle_grid, p, prm, 0, 3 * mem + 2) for p in probs]
    assert all(s is not None and s["bound"] == mem for s in states)
    opt = _opt(devmap, prm, lanes)
    for budget in (1, 6):
        refs = [F.oracle_resume(oracle_grid, p, prm, s, budget) for p, s in zip(probs, states)]
        _load(opt, probs, states)
        opt.lbfgs_resume(budget)
        got = opt.get_lbfgs_state()
        for i, (r, g) in enumerate(zip(refs, got)):
            F.assert_states_match(r, g, 1e-9, "problem %d mem %d budget %d lanes %d" % (i, mem, budget, lanes))
            assert r["code"] == F.LBFGS_RUNNING and r["k"] == states[i]["k"] + budget


def test_long_history_ring_wrap_at_mem_256(devmap, oracle_grid):
    """the shipped mem_size: a pass long enough to wrap the 256-pair ring exists only with a tight delta; captured at k = 300"""
    from uneven_planner_amd import scenes
    prm = dict(delta=1e-9, g_epsilon=1e-9)
    p = scenes.hill_problem()
    st = F.capture(oracle_grid, p, prm, 0, 300)
    if st is None:
        pytest.skip("the oracle's first pass ends before k = 300 on this problem")
    assert st["bound"] == 256 and st["end"] == (300 - 1) % 256
    ref = F.oracle_resume(oracle_grid, p, prm, st, 2)
    opt = _opt(devmap, prm, 128)
    _load(opt, [p], [st])
    opt.lbfgs_resume(2)
    F.assert_states_match(ref, opt.get_lbfgs_state()[0], 1e-9, "mem 256")


def test_cautious_update_rejected(devmap, oracle_grid, probs):
    prm = dict(mem_size=8)
    rng = np.random.default_rng(3)
    states = [F.doctor(F.capture(oracle_grid, p, prm, 0, 20), "skip", rng) for p in probs]
    refs = [F.oracle_resume(oracle_grid, p, prm, s, 1) for p, s in zip(probs, states)]
    opt = _opt(devmap, prm, 128)
    _load(opt, probs, states)
    opt.lbfgs_resume(1)
    for r, g, s in zip(refs, opt.get_lbfgs_state(), states):
        assert r["bound"] == s["bound"] and r["end"] == s["end"] and np.array_equal(r["d"], -r["g"])
        F.assert_states_match(r, g, 1e-9, "skip", skip_keys=("lm_ys",))


def test_line_search_exhausted_restores_and_alm_continues_with_last_trial_residuals(devmap, oracle_grid, probs):
    prm = dict(mem_size=8)
    rng = np.random.default_rng(4)
    states = [F.doctor(F.capture(oracle_grid, p, prm, 1, 3), "ls_fail", rng, p) for p in probs]
    refs = [F.oracle_resume(oracle_grid, p, prm, s, 5, finish=True) for p, s in zip(probs, states)]
    opt = _opt(devmap, prm, 128)
    _load(opt, probs, states)
    opt.lbfgs_resume(5, finish_pass=True)
    for r, g, s in zip(refs, opt.get_lbfgs_state(), states):
        assert r["code"] == F.LBFGSERR_MAXIMUMLINESEARCH and g["code"] == r["code"] and g["accepted"] == 1 and g["converged"] == r["converged"]
        assert np.array_equal(g["x"], s["x"]) and np.array_equal(g["g"], s["g"])                   # restored (lbfgs.hpp:575-582)
        assert abs(g["fx"] - r["fx"]) <= 1e-9 * abs(r["fx"])                                       # f of the last trial (Q1)
        assert F.rel(r["hx"], g["hx"]) < 1e-9 and F.rel(r["gx"], g["gx"]) < 1e-9
        assert F.rel(r["lam"], g["lam"]) < 1e-9 and F.rel(r["mu"], g["mu"]) < 1e-9 and g["rho"] == r["rho"]


def test_ascent_direction_is_a_hard_error(devmap, oracle_grid, probs):
    prm = dict(mem_size=8)
    states = [F.doctor(F.capture(oracle_grid, p, prm, 0, 10), "ascent", None) for p in probs]
    opt = _opt(devmap, prm, 256)
    _load(opt, probs, states)
    opt.lbfgs_resume(3, finish_pass=True)
    for g, s in zip(opt.get_lbfgs_state(), states):
        assert g["code"] == F.LBFGSERR_INCREASEGRADIENT and g["accepted"] == 0 and g["k"] == s["k"]
        assert np.array_equal(g["x"], s["x"]) and np.array_equal(g["lam"], s["lam"]) and g["rho"] == s["rho"]


def test_progress_callback_cancels_at_k_1001(devmap, oracle_grid, probs):
    prm = dict(mem_size=8)
    states = [F.doctor(F.capture(oracle_grid, p, prm, 0, 12), "cancel", None) for p in probs]
    refs = [F.oracle_resume(oracle_grid, p, prm, s, 3, finish=True) for p, s in zip(probs, states)]
    opt = _opt(devmap, prm, 128)
    _load(opt, probs, states)
    opt.lbfgs_resume(3, finish_pass=True)
    for r, g in zip(refs, opt.get_lbfgs_state()):
        assert r["code"] == F.LBFGS_CANCELED and g["code"] == r["code"] and g["accepted"] == 1 and g["k"] == 1001
        assert F.rel(r["x"], g["x"]) < 1e-9 and F.rel(r["lam"], g["lam"]) < 1e-9 and F.rel(r["mu"], g["mu"]) < 1e-9 and g["rho"] == r["rho"]


@pytest.mark.parametrize("lanes", [128, 256])
def test_later_alm_passes_one_at_a_time(devmap, oracle_grid, probs, lanes):
    """ALM passes >= 2 (alm_traj_opt.h:132-151, Q5): every short pass of the oracle's solves is replayed from the oracle's own
    (x, lambda, mu, rho) and must end at the same x, duals, rho and convergence verdict"""
    prm = dict(mem_size=64)
    opt = _opt(devmap, prm, lanes)
    done = 0
    for p in probs:
        st = F.capture(oracle_grid, p, prm, 0, 1)
        for i, ps in enumerate(st["passes"]):
            if i == 0 or ps["k"] > 30:
                continue
            opt.upload([p])
            opt.set_state(lam=[ps["lam_in"]], mu=[ps["mu_in"]], scale_cx=[st["scale_cx"]], scale_fx=[st["scale_fx"]], rho=[ps["rho_in"]])
            opt.set_x([ps["x_in"]])
            opt.alm_passes(1)
            r = opt.download()[0]
            assert r["lbfgs_iters"] == ps["k"] and r["last_lbfgs_ret"] == ps["ret"], (i, r["lbfgs_iters"], ps["k"])
            assert F.rel(ps["x_out"], r["x"]) < 1e-8 and F.rel(ps["lam_out"], r["lam"]) < 1e-8 and F.rel(ps["mu_out"], r["mu"]) < 1e-8
            assert r["rho_final"] == ps["rho_out"] and (r["ret"] == 0) == bool(ps["converged"])
            done += 1
    assert done >= 8


def test_early_exits_of_lbfgs_in_full_solves(devmap, oracle, oracle_grid, probs):
    """LBFGS_CONVERGENCE at the very first gradient test and LBFGSERR_MAXIMUMITERATION (both accepted by the ALM loop,
    alm_traj_opt.cpp:240-245): with these parameters the solves are short, so the final x is compared strictly"""
    for prm in (dict(g_epsilon=1e6), dict(inner_max_iter=4.0, max_iter=3.0)):
        opt = _opt(devmap, prm, 128)
        opt.set_rho(1.0)
        out = opt.optimize_batch(probs)
        for p, o in zip(probs, out):
            ro = oracle.OracleALM(oracle_grid, prm).optimize(p)
            assert o["ret"] == ro["ret"] and o["alm_iters"] == ro["alm_iters"] and o["lbfgs_iters"] == ro["lbfgs_iters"] and o["evals"] == ro["evals"]
            assert o["last_lbfgs_ret"] == ro["last_lbfgs_ret"]
            assert F.rel(ro["x"], o["x"]) < 1e-8 and abs(o["cost"] - ro["cost"]) <= 1e-8 * abs(ro["cost"])
