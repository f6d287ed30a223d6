"""Shared helper (test infrastructure): the optimiser's own sensitivity floor.  Builds the oracle a second time with
-march=native -ffp-contract=fast (FMA contraction = ~1 ulp perturbations per operation) and solves the same problems with
both builds.  See DESIGN.md "Parity": the loose stop rules of the reference amplify rounding noise by ~1.25x per L-BFGS
iteration, so two bit-different but equally correct implementations end 1e-4..1e-2 apart."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def solve_with_fma_oracle(cells, probs, params=None, grid_kw=None):
    from oracle import oracle_py as O
    so = "/tmp/liboracle_fma_%d.so" % os.getpid()
    subprocess.check_call(["g++", "-O3", "-march=native", "-ffp-contract=fast", "-std=c++17", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "oracle_capi.cpp")])
    saved = O._LIB
    O._LIB = None
    real = O.os.path.join
    O.os.path.join = lambda *a, _r=real: so if a[-1] == "liboracle.so" else _r(*a)
    try:
        g = O.OracleGrid(**(grid_kw or {}))
        g.set_cells(cells)
        out = [O.OracleALM(g, params).optimize(p) for p in probs]
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
