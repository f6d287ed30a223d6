"""Generates tests/golden/lbfgs_golden.npz: the complete evaluation record -- every point the optimiser evaluates, line-search trials
included, in order -- of an INDEPENDENT numpy implementation of the reference's L-BFGS (back_end/include/utils/lbfgs.hpp:439-722,
Lewis-Overton line search :276-389 with its non-upstream early accept :327-330, cautious update :675-677, two-loop recursion :687-710),
written here from the header with numpy vector operations, on two analytic functions and several (mem_size, past) settings, among them
mem_size = 3 so that the history ring wraps many times.  The oracle's lbfgs.hpp restatement must evaluate the same points in the same
order (tests/test_oracle_cpu.py).  Run:  python tests/golden/make_lbfgs_golden.py   (numpy only; deterministic)."""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# lbfgs.hpp:76-128 defaults as the optimiser leaves them
# return codes (lbfgs.hpp:138-183: LBFGS_CONVERGENCE 0, LBFGS_STOP 1; errors count up from LBFGSERR_UNKNOWNERROR = -1024)
E_FUNCVAL, E_MINSTEP, E_MAXSTEP, E_MAXLS, E_MAXITER, E_WIDTH, E_PARAM, E_INCREASE = -1012, -1011, -1010, -1009, -1008, -1007, -1006, -1005
PAR = dict(min_step=1e-32, max_step=1e20, f_dec_coeff=1e-4, s_curv_coeff=0.9, cautious_factor=1e-6, machine_prec=1e-16, max_linesearch=64)


def rosenbrock(x):
    t1 = 1.0 - x[0::2]
    t2 = 10.0 * (x[1::2] - x[0::2] ** 2)
    g = np.zeros_like(x)
    g[1::2] = 20.0 * t2
    g[0::2] = -2.0 * (x[0::2] * g[1::2] + t1)
    return float(np.sum(t1 * t1 + t2 * t2)), g


def bowl(x):
    i = np.arange(x.size)
    w = 1.0 + 3.0 * i
    d = x - np.sin(i + 1.0)
    p = x[:-1] * x[1:]
    g = 2.0 * w * d
    g[:-1] += 0.1 * p * x[1:]
    g[1:] += 0.1 * p * x[:-1]
    return float(np.sum(w * d * d) + 0.05 * np.sum(p * p)), g


def lbfgs(fun, x0, mem_size, past, g_eps, delta, max_iter):
    rec = []

    def ev(x):
        f, g = fun(x)
        rec.append(np.concatenate([x, [f]]))
        return f, g

    def line_search(x, f, g, stp, s, xp, gp):
        count, brackt, touched = 0, False, False
        mu, nu = 0.0, PAR["max_step"]
        if not stp > 0.0:
            return E_PARAM, x, f, g, stp
        dginit = float(gp @ s)
        if 0.0 < dginit:
            return E_INCREASE, x, f, g, stp
        finit = f
        dgtest, dstest = PAR["f_dec_coeff"] * dginit, PAR["s_curv_coeff"] * dginit
        while True:
            x = xp + stp * s
            f, g = ev(x)
            count += 1
            if math.isinf(f) or math.isnan(f):
                return E_FUNCVAL, x, f, g, stp
            if past > 0 and abs(finit - f) / (abs(finit) + 1.0) < delta / past:            # :327-330 (not in upstream LBFGS-Lite)
                return count, x, f, g, stp
            if f > finit + stp * dgtest:
                nu, brackt = stp, True
            elif float(g @ s) < dstest:
                mu = stp
            else:
                return count, x, f, g, stp
            if PAR["max_linesearch"] <= count:
                return E_MAXLS, x, f, g, stp
            if brackt and (nu - mu) < PAR["machine_prec"] * nu:
                return E_WIDTH, x, f, g, stp
            stp = 0.5 * (mu + nu) if brackt else stp * 2.0
            if stp < PAR["min_step"]:
                return E_MINSTEP, x, f, g, stp
            if stp > PAR["max_step"]:
                if touched:
                    return E_MAXSTEP, x, f, g, stp
                touched, stp = True, PAR["max_step"]

    n, m = x0.size, mem_size
    x = x0.copy()
    fx, g = ev(x)
    pf = np.zeros(max(1, past))
    pf[0] = fx
    d = -g
    S, Y, YS, AL = np.zeros((m, n)), np.zeros((m, n)), np.zeros(m), np.zeros(m)
    if np.abs(g).max() / max(1.0, np.abs(x).max()) < g_eps:
        return 0, x, fx, 0, np.array(rec)
    step = 1.0 / math.sqrt(float(d @ d))
    k, end, bound = 1, 0, 0
    while True:
        xp, gp = x.copy(), g.copy()
        ls, x, fx, g, step = line_search(x, fx, g, step, d, xp, gp)
        if ls < 0:
            x, ret = xp, ls
            break
        if np.abs(g).max() / max(1.0, np.abs(x).max()) < g_eps:
            ret = 0
            break
        if past > 0:
            if past <= k and abs(pf[k % past] - fx) / max(1.0, abs(fx)) < delta:
                ret = 1
                break
            pf[k % past] = fx
        if max_iter != 0 and max_iter <= k:
            ret = E_MAXITER
            break
        k += 1
        S[end], Y[end] = x - xp, g - gp
        ys, yy = float(Y[end] @ S[end]), float(Y[end] @ Y[end])
        YS[end] = ys
        d = -g
        cau = float(S[end] @ S[end]) * math.sqrt(float(gp @ gp)) * PAR["cautious_factor"]
        if ys > cau:
            bound = min(m, bound + 1)
            end = (end + 1) % m
            j = end
            for _ in range(bound):
                j = (j + m - 1) % m
                AL[j] = float(S[j] @ d) / YS[j]
                d = d + (-AL[j]) * Y[j]
            d = d * (ys / yy)
            for _ in range(bound):
                beta = float(Y[j] @ d) / YS[j]
                d = d + (AL[j] - beta) * S[j]
                j = (j + 1) % m
        step = 1.0
    return ret, x, fx, k, np.array(rec)


def main():
    out = {}
    rng = np.random.default_rng(314)
    cases = [("bowl_m8", bowl, 12, 8, 3, 1e-9, 1e-14, 0), ("bowl_m3_wrap", bowl, 16, 3, 3, 1e-9, 1e-14, 0), ("bowl_past0", bowl, 10, 5, 0, 1e-8, 0.0, 0),
             ("bowl_maxiter", bowl, 12, 8, 3, 1e-12, 1e-16, 9), ("rosen_m8", rosenbrock, 8, 8, 3, 1e-5, 1e-6, 0), ("rosen_m2_wrap", rosenbrock, 6, 2, 3, 1e-5, 1e-8, 0)]
    for name, fun, n, m, past, geps, delta, maxit in cases:
        x0 = rng.uniform(-1.5, 1.5, n)
        ret, x, f, k, rec = lbfgs(fun, x0, m, past, geps, delta, maxit)
        out[name + "/cfg"] = np.array([0 if fun is rosenbrock else 1, n, m, past, geps, delta, maxit])
        out[name + "/x0"] = x0
        out[name + "/ret"] = np.array(ret); out[name + "/k"] = np.array(k); out[name + "/x"] = x; out[name + "/f"] = np.array(f); out[name + "/rec"] = rec
        print(name, "ret", ret, "iterations", k, "evaluations", rec.shape[0], "f", f)
    np.savez_compressed(os.path.join(HERE, "lbfgs_golden.npz"), **out)


if __name__ == "__main__":
    main()
