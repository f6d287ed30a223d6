"""Shared helper (test infrastructure): the optimiser's own sensitivity floor.  Builds the oracle a second time with
-march=native -ffp-contract=fast (FMA contraction = ~1 ulp perturbations per operation) and solves the same problems with
both builds.  See DESIGN.md "Parity": the loose stop rules of the reference amplify rounding noise by ~1.25x per L-BFGS
iteration, so two bit-different but equally correct implementations end 1e-4..1e-2 apart."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def solve_many(make_alm, probs, threads=1):
    """one oracle solve per problem; threads > 1: one trajectory per host thread (the oracle's C entry points release the GIL; every solve has
    its own OracleALM, the grid is only read)"""
    if threads <= 1:
        return [make_alm().optimize(p) for p in probs]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(lambda p: make_alm().optimize(p), probs))


class fma_session:
    """`with fma_session() as O:` -- inside the block the module oracle.oracle_py is bound to the oracle rebuilt with -march=native -ffp-contract=fast
    (grids and solvers created there live in that build); the plain build is restored on exit"""

    def __enter__(self):
        from oracle import oracle_py as O
        self.O = O
        so = "/tmp/liboracle_fma_%d.so" % os.getpid()
        if not os.path.exists(so):
            subprocess.check_call(["g++", "-O3", "-march=native", "-ffp-contract=fast", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "oracle", "oracle_capi.cpp")])
        self.saved, O._LIB = O._LIB, None
        self.real = O.os.path.join
        O.os.path.join = lambda *a, _r=self.real: so if a[-1] == "liboracle.so" else _r(*a)
        return O

    def __exit__(self, *exc):
        self.O.os.path.join = self.real
        self.O._LIB = self.saved
        return False


def solve_with_eigen_order_oracle(cells, probs, params=None, grid_kw=None, threads=1):
    """the oracle rebuilt with -DORACLE_EIGEN_REDUX=1: every dynamic-vector reduction of the L-BFGS (dot, squaredNorm, norm), the gdT sums and the fixed-size
    block products of calGradCTtoQT summed in the order of Eigen 3.3.7's vectorised redux for SSE2 (oracle/eigen_redux.hpp) instead of left to right;
    same compiler flags as the default oracle otherwise"""
    return solve_with_fma_oracle(cells, probs, params, grid_kw, threads, flags=("-O3", "-DORACLE_EIGEN_REDUX=1"), tag="eig")


def solve_with_fma_oracle(cells, probs, params=None, grid_kw=None, threads=1, flags=("-O3", "-march=native", "-ffp-contract=fast"), tag="fma"):
    from oracle import oracle_py as O
    so = "/tmp/liboracle_%s_%d.so" % (tag, os.getpid())
    subprocess.check_call(["g++"] + list(flags) + ["-std=c++17", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "oracle_capi.cpp")])
    saved = O._LIB
    O._LIB = None
    real = O.os.path.join
    O.os.path.join = lambda *a, _r=real: so if a[-1] == "liboracle.so" else _r(*a)
    try:
        g = O.OracleGrid(**(grid_kw or {}))
        g.set_cells(cells)
        out = solve_many(lambda: O.OracleALM(g, params), probs, threads)
    finally:
        O.os.path.join = real
        O._LIB = saved
    return out


def eval_with_fma_oracle(cells, probs, points):
    """(f, grad, c_xy) of one objective evaluation per problem at the given point, after setup + initScaling, from the oracle rebuilt with
    -ffp-contract=fast -march=native"""
    from oracle import oracle_py as O
    so = "/tmp/liboracle_fma_%d.so" % os.getpid()
    subprocess.check_call(["g++", "-O3", "-march=native", "-ffp-contract=fast", "-std=c++17", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "oracle_capi.cpp")])
    saved = O._LIB
    O._LIB = None
    real = O.os.path.join
    O.os.path.join = lambda *a, _r=real: so if a[-1] == "liboracle.so" else _r(*a)
    try:
        g = O.OracleGrid()
        g.set_cells(cells)
        out = []
        for p, x in zip(probs, points):
            a = O.OracleALM(g)
            x0 = a.setup(p)
            a.init_scaling(x0)
            f, gr, _ = a.eval(x)
            out.append((f, gr, np.asarray(a.coeffs()[0]).ravel()))
    finally:
        O.os.path.join = real
        O._LIB = saved
    return out


def spread(res_a, res_b, key_cost_a="cost", key_cost_b="cost"):
    dx = np.array([np.abs(a["x"] - b["x"]).max() / np.abs(a["x"]).max() for a, b in zip(res_a, res_b)])
    dc = np.array([abs(a[key_cost_a] - b[key_cost_b]) / abs(a[key_cost_a]) for a, b in zip(res_a, res_b)])
    same = float(np.mean([a["ret"] == b["ret"] for a, b in zip(res_a, res_b)]))
    return dict(x_median=float(np.median(dx)), x_p90=float(np.percentile(dx, 90)), x_max=float(dx.max()), x_le_1e4=float((dx <= 1e-4).mean()),
                c_median=float(np.median(dc)), c_p90=float(np.percentile(dc, 90)), c_max=float(dc.max()), c_le_1e4=float((dc <= 1e-4).mean()),
                same_ret=same)


BUCKET_EDGES = [0, 40, 80, 120, 180, 260, 400, 100000]


def bucket_table(ref, other, edges=BUCKET_EDGES):
    """agreement of `other` with `ref` (final way-points / cost, relative, infinity norm), bucketed by ref's total L-BFGS iterations"""
    k = np.array([r["lbfgs_iters"] for r in ref])
    dx = np.array([np.abs(a["x"] - b["x"]).max() / np.abs(b["x"]).max() for a, b in zip(other, ref)])
    dc = np.array([abs(a["cost"] - b["cost"]) / abs(b["cost"]) for a, b in zip(other, ref)])
    rows = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (k >= lo) & (k < hi)
        if m.sum() == 0:
            continue
        rows.append(dict(lo=lo, hi=hi, n=int(m.sum()), x_le_1e4=float((dx[m] <= 1e-4).mean()), c_le_1e4=float((dc[m] <= 1e-4).mean()),
                         x_median=float(np.median(dx[m])), x_max=float(dx[m].max())))
    return rows


# ---- directional drift: is the device's disagreement with the oracle one-sided? ---------------------------------------
def _binom_two_sided(k, n):
    """exact two-sided binomial p-value for k successes of n at p = 1/2 (McNemar's exact test / the sign test)"""
    if n == 0:
        return 1.0
    from math import exp, lgamma, log
    k = min(k, n - k)
    tail = sum(exp(lgamma(n + 1) - lgamma(i + 1) - lgamma(n - i + 1) - n * log(2.0)) for i in range(k + 1))      # (log domain: n reaches the thousands)
    return float(min(1.0, 2.0 * tail))


def paired_counts(ref, other):
    """ref / other: result lists of the SAME problems.  Converged = ret == 0 (alm_traj_opt.cpp:259-262).  b = other converged where ref did not,
    c = ref converged where other did not (McNemar's discordant pairs); cost sign test over the pairs whose final costs differ"""
    cr = np.array([r["ret"] == 0 for r in ref]); co = np.array([o["ret"] == 0 for o in other])
    b, c = int((co & ~cr).sum()), int((cr & ~co).sum())
    dcost = np.array([o["cost"] - r["cost"] for r, o in zip(ref, other)])
    lo, hi = int((dcost < 0).sum()), int((dcost > 0).sum())
    return dict(n=len(ref), converged_ref=float(cr.mean()), converged_other=float(co.mean()), other_only=b, ref_only=c, mcnemar_p=_binom_two_sided(b, b + c),
                same_ret=float(np.mean([r["ret"] == o["ret"] for r, o in zip(ref, other)])),
                cost_lower=lo, cost_higher=hi, cost_sign_p=_binom_two_sided(lo, lo + hi))


def drift_stats(ref, fma, dev):
    """the three-way statement: oracle, oracle rebuilt with FMA contraction (its own reproducibility floor) and the device on the same problems"""
    d, f = paired_counts(ref, dev), paired_counts(ref, fma)
    return dict(n=d["n"], converged_frac=dict(oracle=d["converged_ref"], oracle_fma=f["converged_other"], device=d["converged_other"]),
                device_vs_oracle=dict(device_only=d["other_only"], oracle_only=d["ref_only"], mcnemar_p=d["mcnemar_p"], same_ret=d["same_ret"],
                                      cost_lower=d["cost_lower"], cost_higher=d["cost_higher"], cost_sign_p=d["cost_sign_p"]),
                fma_vs_oracle=dict(fma_only=f["other_only"], oracle_only=f["ref_only"], mcnemar_p=f["mcnemar_p"], same_ret=f["same_ret"],
                                   cost_lower=f["cost_lower"], cost_higher=f["cost_higher"], cost_sign_p=f["cost_sign_p"]))


def assert_no_directional_drift(st, what=""):
    """The device may disagree with the oracle as often as the oracle disagrees with its own FMA rebuild -- the optimiser is chaotic -- but not in
    ONE direction.  Converged rate: |device - oracle| <= |fma - oracle| + 2 SE, SE = sqrt(b + c) / n of the paired difference (McNemar);
    same return code: device >= floor - 2 SE of the difference of the two proportions; final cost: the sign test must not reject at 1e-3 unless the
    FMA pair's does too (the oracle drifting against itself is not the device's doing)."""
    n = st["n"]
    dv, fm = st["device_vs_oracle"], st["fma_vs_oracle"]
    d_dev = abs(dv["device_only"] - dv["oracle_only"]) / n
    d_fma = abs(fm["fma_only"] - fm["oracle_only"]) / n
    se = np.sqrt(dv["device_only"] + dv["oracle_only"]) / n
    assert d_dev <= d_fma + 2.0 * se + 1e-12, "%s: converged rate drifts one way: device-only %d vs oracle-only %d of %d (FMA pair: %d vs %d)" % (
        what, dv["device_only"], dv["oracle_only"], n, fm["fma_only"], fm["oracle_only"])
    pd_, pf_ = dv["same_ret"], fm["same_ret"]
    se_ret = np.sqrt((pd_ * (1 - pd_) + pf_ * (1 - pf_)) / n)
    assert pd_ >= pf_ - 2.0 * se_ret - 1e-12, "%s: same return code %.3f (device) vs %.3f (floor), 2 SE = %.3f" % (what, pd_, pf_, 2 * se_ret)
    assert dv["cost_sign_p"] >= 1e-3 or fm["cost_sign_p"] < 1e-3, "%s: final cost is systematically %s than the oracle's (%d lower, %d higher, p = %.1e; FMA pair %d / %d)" % (
        what, "lower" if dv["cost_lower"] > dv["cost_higher"] else "higher", dv["cost_lower"], dv["cost_higher"], dv["cost_sign_p"], fm["cost_lower"], fm["cost_higher"])
