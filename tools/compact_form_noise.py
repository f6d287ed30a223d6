"""Rounding noise of the L-BFGS direction in the compact (Gram-matrix) form against the reference's two-loop recursion (CPU, numpy).

The two-loop (lbfgs.hpp:687-710) forms every s_i . q_i on the REDUCED vector q_i; the compact form -- the only formulation that takes the
wave-wide reductions off the serial chain -- forms it as s_i . g - sum_j alpha_j (s_i . y_j) from stored products,
a difference of large terms near convergence.  For L-BFGS states captured from oracle solves (iteration k of ALM pass p: g, the history
ring, end, bound), the direction is computed three ways -- two-loop in float64, compact form in float64, two-loop in long double as the
reference -- and the relative deviations are printed.  usage: python tools/compact_form_noise.py [n_problems]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import forced_cases as F                       # noqa: E402
from oracle import oracle_py as O              # noqa: E402
from uneven_planner_amd import scenes          # noqa: E402


def order(st, m):
    """slots newest -> oldest"""
    return [(st["end"] - 1 - a) % m for a in range(st["bound"])]


def two_loop(g, S, Y, ys, slots, dt):
    q = -g.astype(dt)
    al = {}
    for j in slots:
        a = (S[j].astype(dt) @ q) / dt(ys[j])
        al[j] = a
        q = q - a * Y[j].astype(dt)
    j0 = slots[0]
    q = q * (dt(ys[j0]) / (Y[j0].astype(dt) @ Y[j0].astype(dt)))
    for j in reversed(slots):
        b = (Y[j].astype(dt) @ q) / dt(ys[j])
        q = q + (al[j] - b) * S[j].astype(dt)
    return q


def compact(g, S, Y, ys, slots):
    """same direction from the products s_i.y_j, y_i.y_j, s_i.g, y_i.g (float64 dots), substitutions in age order"""
    idx = list(slots)                                   # newest first
    Sm, Ym = S[idx], Y[idx]
    SY = Sm @ Ym.T                                      # [i][j] = s_i . y_j
    YY = Ym @ Ym.T
    q0 = -g
    bs, by = Sm @ q0, Ym @ q0
    n = len(idx)
    al = np.zeros(n)
    for i in range(n):                                  # newest -> oldest: alpha_i = (s_i.q0 - sum_{j newer} alpha_j s_i.y_j) / ys_i
        al[i] = (bs[i] - SY[i, :i] @ al[:i]) / ys[idx[i]]
    gam = ys[idx[0]] / YY[0, 0]
    c = gam * (by - YY @ al)                            # y_i . (gamma q_final)
    de = np.zeros(n)
    for i in range(n - 1, -1, -1):                      # oldest -> newest: beta_i = (c_i + sum_{j older} delta_j s_j.y_i) / ys_i
        beta = (c[i] + SY[i + 1:, i] @ de[i + 1:]) / ys[idx[i]]
        de[i] = al[i] - beta
    return gam * (q0 - Ym.T @ al) + Sm.T @ de


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    og = O.OracleGrid()
    og.set_cells(scenes.analytic_cells())
    probs = scenes.random_problems(N, seed0=1000)
    rows = []
    for p in probs:
        for (ps, k) in ((0, 15), (0, 60), (0, 150), (1, 40), (2, 80), (4, 120), (6, 200)):
            st = F.capture(og, p, None, ps, k)
            if st is None or st["bound"] < 2:
                continue
            m = st["lm_s"].shape[0]
            sl = order(st, m)
            ref = two_loop(st["g"], st["lm_s"], st["lm_y"], st["lm_ys"], sl, np.longdouble)
            d64 = two_loop(st["g"], st["lm_s"], st["lm_y"], st["lm_ys"], sl, np.float64)
            dc = compact(st["g"], st["lm_s"], st["lm_y"], st["lm_ys"], sl)
            nr = float(np.abs(ref).max())
            e2, ec = float(np.abs(d64 - ref).max()) / nr, float(np.abs(dc - ref.astype(np.float64)).max()) / nr
            eo = float(np.abs(st["d"] - ref.astype(np.float64)).max()) / nr
            rows.append((ps, k, st["bound"], e2, ec, eo))
            print("pass %d k %3d bound %3d   two-loop f64 %.1e   compact f64 %.1e   (oracle's own d %.1e)   ratio %.0f" % (ps, k, st["bound"], e2, ec, eo, ec / max(e2, 1e-300)))
    r = np.array(rows)
    print("median relative deviation from the long-double direction: two-loop %.1e, compact form %.1e (x %.0f); worst compact %.1e" % (
        np.median(r[:, 3]), np.median(r[:, 4]), np.median(r[:, 4] / r[:, 3]), r[:, 4].max()))


if __name__ == "__main__":
    main()
