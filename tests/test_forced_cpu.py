"""CPU tier of the teacher-forced late-state tests: the product's workgroup program (tests/emu, sequential workgroup) continues from
states the oracle dumped -- see tests/forced_cases.py.  The GPU tier (tests/test_gpu_forced.py) runs the same scenarios through the C-ABI."""
import numpy as np
import pytest

import emu_bridge as E
import forced_cases as F


def _emu(oracle, cells, params):
    E.lib().emu_set_lanes(128)
    return E.Emu(cells, oracle.map_params_vec(), oracle.params_vec(params))


def _state_kw(st):
    return dict(lam=st["lam"], mu=st["mu"], scale_cx=st["scale_cx"], rho=st["rho"], scale_fx=st["scale_fx"])


@pytest.fixture(scope="module")
def prob():
    from uneven_planner_amd import scenes
    return scenes.random_problems(1, seed0=1000)[0]


@pytest.mark.parametrize("mem", [5, 7, 8])
def test_wrapped_ring_one_and_six_iterations(oracle, oracle_grid, analytic_cells, prob, mem):
    prm = dict(mem_size=mem)
    st = F.capture(oracle_grid, prob, prm, 0, 3 * mem + 2)
    assert st is not None and st["bound"] == mem and st["k"] == 3 * mem + 2
    emu = _emu(oracle, analytic_cells, prm)
    for budget in (1, 6):
        ref = F.oracle_resume(oracle_grid, prob, prm, st, budget)
        _, got = emu.lbfgs_resume(prob, st, budget, **_state_kw(st))
        F.assert_states_match(ref, got, 1e-9, "mem %d budget %d" % (mem, budget))
        assert ref["code"] == F.LBFGS_RUNNING and ref["k"] == st["k"] + budget


def test_cautious_update_rejected(oracle, oracle_grid, analytic_cells, prob):
    prm = dict(mem_size=8)
    st = F.doctor(F.capture(oracle_grid, prob, prm, 0, 20), "skip", np.random.default_rng(3))
    ref = F.oracle_resume(oracle_grid, prob, prm, st, 1)
    assert ref["bound"] == st["bound"] and ref["end"] == st["end"] and ref["k"] == st["k"] + 1      # the pair was NOT admitted
    assert np.array_equal(ref["d"], -ref["g"])                                                     # direction falls back to -g (lbfgs.hpp:659)
    _, got = _emu(oracle, analytic_cells, prm).lbfgs_resume(prob, st, 1, **_state_kw(st))
    # (y.s of the rejected pair is a sum with 1e12-fold cancellation by construction of the doctored gradient: not comparable, and unused)
    F.assert_states_match(ref, got, 1e-9, "skip", skip_keys=("lm_ys",))


def test_line_search_exhausted_restores_and_alm_continues_with_last_trial_residuals(oracle, oracle_grid, analytic_cells, prob):
    prm = dict(mem_size=8)
    st = F.doctor(F.capture(oracle_grid, prob, prm, 1, 3), "ls_fail", np.random.default_rng(4), prob)
    ref = F.oracle_resume(oracle_grid, prob, prm, st, 5, finish=True)
    assert ref["code"] == F.LBFGSERR_MAXIMUMLINESEARCH and ref["accepted"] == 1
    assert np.array_equal(ref["x"], st["x"]) and np.array_equal(ref["g"], st["g"])                  # restored (lbfgs.hpp:575-582)
    r, got = _emu(oracle, analytic_cells, prm).lbfgs_resume(prob, st, 5, finish=True, **_state_kw(st))
    assert got["code"] == ref["code"] and got["accepted"] == 1 and got["converged"] == ref["converged"]
    assert np.array_equal(got["x"], st["x"]) and np.array_equal(got["g"], st["g"])
    assert abs(got["fx"] - ref["fx"]) <= 1e-9 * abs(ref["fx"])                                      # f of the LAST TRIAL (Q1)
    assert F.rel(ref["hx"], got["hx"]) < 1e-9 and F.rel(ref["gx"], got["gx"]) < 1e-9               # residuals of the last trial ...
    assert F.rel(ref["lam"], got["lam"]) < 1e-9 and F.rel(ref["mu"], got["mu"]) < 1e-9             # ... drive the dual update
    assert got["rho"] == ref["rho"] == min(2.0 * st["rho"], 1000.0)


def test_ascent_direction_is_a_hard_error(oracle, oracle_grid, analytic_cells, prob):
    prm = dict(mem_size=8)
    st = F.doctor(F.capture(oracle_grid, prob, prm, 0, 10), "ascent", np.random.default_rng(5))
    ref = F.oracle_resume(oracle_grid, prob, prm, st, 3, finish=True)
    assert ref["code"] == F.LBFGSERR_INCREASEGRADIENT and ref["accepted"] == 0
    _, got = _emu(oracle, analytic_cells, prm).lbfgs_resume(prob, st, 3, finish=True, **_state_kw(st))
    assert got["code"] == ref["code"] and got["accepted"] == 0 and got["k"] == st["k"]
    assert np.array_equal(got["x"], st["x"]) and F.rel(st["lam"], got["lam"]) == 0.0 and got["rho"] == st["rho"]


def test_progress_callback_cancels_at_k_1001(oracle, oracle_grid, analytic_cells, prob):
    prm = dict(mem_size=8)
    st = F.doctor(F.capture(oracle_grid, prob, prm, 0, 12), "cancel", np.random.default_rng(6))
    ref = F.oracle_resume(oracle_grid, prob, prm, st, 3, finish=True)
    assert ref["code"] == F.LBFGS_CANCELED and ref["accepted"] == 1 and ref["k"] == 1001
    _, got = _emu(oracle, analytic_cells, prm).lbfgs_resume(prob, st, 3, finish=True, **_state_kw(st))
    assert got["code"] == ref["code"] and got["accepted"] == 1 and got["k"] == 1001
    assert F.rel(ref["x"], got["x"]) < 1e-9 and F.rel(ref["lam"], got["lam"]) < 1e-9 and F.rel(ref["mu"], got["mu"]) < 1e-9
    assert got["rho"] == ref["rho"]


def test_later_alm_passes_one_at_a_time(oracle, oracle_grid, analytic_cells, prob):
    """ALM passes >= 2 (alm_traj_opt.h:132-151, Q5): every short pass of the oracle's solve is replayed from the oracle's own
    (x, lambda, mu, rho) and must end at the same x, duals, rho and convergence verdict"""
    prm = dict(mem_size=64)
    st = F.capture(oracle_grid, prob, prm, 0, 1)
    emu = _emu(oracle, analytic_cells, prm)
    done = 0
    for i, ps in enumerate(st["passes"]):
        if i == 0 or ps["k"] > 30:
            continue
        r = emu.alm_passes(prob, ps["x_in"], 1, lam=ps["lam_in"], mu=ps["mu_in"], scale_cx=st["scale_cx"], rho=ps["rho_in"], scale_fx=st["scale_fx"])
        assert r["lbfgs_iters"] == ps["k"] and r["last_lbfgs_ret"] == ps["ret"], (i, r["lbfgs_iters"], ps["k"])
        assert F.rel(ps["x_out"], r["x"]) < 1e-8, (i, F.rel(ps["x_out"], r["x"]))
        assert F.rel(ps["lam_out"], r["lam"]) < 1e-8 and F.rel(ps["mu_out"], r["mu"]) < 1e-8
        assert r["rho"] == ps["rho_out"]
        assert (r["ret"] == 0) == bool(ps["converged"])          # 0: judgeConvergence said yes; 3: stopped by the one-pass cap
        done += 1
    assert done >= 3
