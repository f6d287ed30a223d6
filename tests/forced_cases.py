"""Teacher-forced late-state scenarios for the L-BFGS / ALM state machine (TEST INFRASTRUCTURE).

The CPU oracle solves a problem and dumps its complete L-BFGS state at a chosen loop top (x, g, d, history ring, end, bound, pf,
step, fx, k) together with the duals / scales / rho of that ALM pass.  A scenario optionally doctors that state so that a rarely
taken branch fires, lets the oracle advance from it by a bounded number of iterations (OracleALM.lbfgs_resume) and hands the SAME
state to the implementation under test (the GPU through the C-ABI hooks, or the CPU emulator of the workgroup program).  Both
continue from identical inputs, so everything can be compared at 1e-9 however late in the solve the state was taken -- the chaotic
drift of free-running solves never enters.  Branches covered (reference: back_end/include/utils/lbfgs.hpp, back_end/src/alm_traj_opt.cpp):
  ring wrap   bound == mem with end wrapped, two-loop over a full ring                         lbfgs.hpp:687-710
  skip        cautious update rejected (ys <= cau): history and direction NOT updated          lbfgs.hpp:675-677
  ls_fail     64 failing trials -> LBFGSERR_MAXIMUMLINESEARCH, x / g restored, ALM continues
              with the residuals of the LAST TRIAL (Q1)                                         lbfgs.hpp:349-353, 575-582; alm_traj_opt.cpp:246-257
  ascent      0 < dginit -> LBFGSERR_INCREASEGRADIENT, ALM gives up (ret 1)                     lbfgs.hpp:300-303; alm_traj_opt.cpp:250-254
  cancel      progress callback at k = 1001 -> LBFGS_CANCELED, ALM continues                    alm_traj_opt.cpp:1016, 240-245
"""
import numpy as np

from oracle import oracle_py as O

LBFGS_RUNNING = 999
LBFGS_STOP, LBFGS_CANCELED = 1, 2
LBFGSERR_MAXIMUMLINESEARCH = -1024 + 15
LBFGSERR_INCREASEGRADIENT = -1024 + 19


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(a).max()))


def capture(og, prob, params, snap_pass, snap_k):
    """oracle solve with a state capture; returns (state dict incl. lam / mu / rho, scales dict) or None"""
    a = O.OracleALM(og, params)
    a.set_capture(snap_pass, snap_k, True)
    a.optimize(prob)
    st = a.capture()
    if st is None:
        return None
    sc = a.get_state()
    st["scale_cx"], st["scale_fx"] = sc["scale_cx"], sc["scale_fx"]
    st["passes"] = a.passes()
    return st


def oracle_resume(og, prob, params, st, budget, finish=False):
    """advance the oracle from `st`; returns the new state + what the ALM did with the result when finish is set"""
    a = O.OracleALM(og, params)
    a.setup(prob)
    a.set_state(lam=st["lam"], mu=st["mu"], scale_cx=st["scale_cx"], scale_fx=st["scale_fx"])
    a.set_rho(st["rho"])
    code, new = a.lbfgs_resume(st, budget)
    new["code"] = code
    accepted = code in (0, LBFGS_STOP, LBFGS_CANCELED, -1024 + 16, LBFGSERR_MAXIMUMLINESEARCH)
    new["accepted"], new["converged"] = 0, 0
    if finish and code != LBFGS_RUNNING:
        new["accepted"] = int(accepted)
        if accepted:
            new["converged"] = a.finish_pass()
    s2 = a.get_state()
    new.update(hx=s2["hx"], gx=s2["gx"], lam=s2["lam"], mu=s2["mu"], rho=a.get_rho())
    return new


def doctor(st, kind, rng, prob=None):
    """returns a doctored copy of the captured state (prob: needed by "ls_fail" for the variable layout)"""
    s = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}
    g, d = s["g"], s["d"]
    if kind == "skip":
        # g <- g + c w with w orthogonal to d: the search (g.d, Armijo, curvature) is untouched, but |gp| is huge -> cau >> ys
        w = rng.normal(size=g.size)
        w -= d * (w @ d) / (d @ d)
        w -= d * (w @ d) / (d @ d)
        s["g"] = g + 1e12 * np.abs(g).max() * w / np.linalg.norm(w)
    elif kind == "ls_fail":
        # a direction so long that 64 halvings never reach the basin: only the xy way-points move (tau and yaw untouched)
        dd = np.zeros_like(d)
        n_xy = 2 * prob["inner_xy"].shape[1]
        dd[1:1 + n_xy] = rng.normal(size=n_xy)
        dd *= 1e22 / np.linalg.norm(dd)
        if g @ dd > 0:
            dd = -dd
        s["d"] = dd
        s["step"] = 1.0
    elif kind == "ascent":
        s["d"] = g.copy()
    elif kind == "cancel":
        s["k"] = 1001
    return s


COMPARE_KEYS = ("x", "g", "d", "pf", "lm_ys", "lm_s", "lm_y")


def assert_states_match(ref, got, tol=1e-9, what="", skip_keys=()):
    assert got["code"] == ref["code"], (what, got["code"], ref["code"])
    for key in ("k", "end", "bound"):
        assert got[key] == ref[key], (what, key, got[key], ref[key])
    assert abs(got["fx"] - ref["fx"]) <= tol * max(1.0, abs(ref["fx"])), (what, "fx", got["fx"], ref["fx"])
    assert abs(got["step"] - ref["step"]) <= tol * max(1.0, abs(ref["step"])), (what, "step")
    for key in COMPARE_KEYS:
        if key in skip_keys:
            continue
        r, g_ = np.asarray(ref[key]), np.asarray(got[key])
        if key in ("lm_s", "lm_y", "lm_ys"):
            # only the rows the algorithm has defined: the `bound` newest pairs, and the row written in this iteration
            m = r.shape[0]
            rows = sorted({(ref["end"] - 1 - t) % m for t in range(ref["bound"])} | {(ref["end"]) % m} | {(ref["end"] - 1) % m})
            rows = [j for j in rows if np.isfinite(r[j]).all()]
            r, g_ = r[rows], g_[rows]
        assert rel(r, g_) < tol, (what, key, rel(r, g_))
