"""Rounding noise of the BLOCKED two-loop recursion (round 6) against the reference's sequential one (lbfgs.hpp:687-710), CPU / numpy.

Blocked form: the history ring is cut into blocks of BS = 8 consecutive slots.  Inside a block the BS dot products s_i . q are all taken on the vector as it
ENTERS the block; the coupling of the block's own pairs goes through the stored products s_i . y_j of that block alone:
    alpha = A p,  A = (I + diag(rho) U)^-1 diag(rho),  U_ij = s_i . y_j (j newer than i, same block),  rho_i = 1 / (y_i . s_i)
and the second loop uses the transpose:  gamma = A^T (alpha o ys - t),  t_i = y_i . r on the vector entering the block.  Same direction in exact arithmetic;
unlike the compact form (tools/compact_form_noise.py) the vector is still updated block by block, so every dot product is taken on a reduced vector at most
BS - 1 pairs stale, not on q_0.  usage: python tools/blocked_two_loop_noise.py [n_problems] [BS]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import forced_cases as F                       # noqa: E402
from oracle import oracle_py as O              # noqa: E402
from uneven_planner_amd import scenes          # noqa: E402
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from compact_form_noise import order, two_loop  # noqa: E402


def block_matrix(S, Y, ys, slots8):
    """A of one block (slots ascending = older -> newer inside a contiguous run), upper triangular in slot order"""
    b = len(slots8)
    rho = 1.0 / ys[slots8]
    U = np.zeros((b, b))
    for i in range(b):
        for j in range(i + 1, b):
            U[i, j] = S[slots8[i]] @ Y[slots8[j]]
    A = np.zeros((b, b))
    for j in range(b):                                   # column j of (I + rho U)^-1 rho by back substitution (the device's order)
        x = np.zeros(b)
        x[j] = rho[j]
        for i in range(j - 1, -1, -1):
            acc = 0.0
            for k in range(i + 1, j + 1):
                acc += U[i, k] * x[k]
            x[i] = -rho[i] * acc
        A[:, j] = x
    return A


def segments(end, bound, m, BS):
    """the slots newest -> oldest as runs [lo, hi) of consecutive slots inside one block (processed hi-1 .. lo in the first loop)"""
    out = []
    j = (end - 1) % m
    left = bound
    while left > 0:
        blk_lo = (j // BS) * BS
        lo = max(blk_lo, j - left + 1)
        out.append((lo, j + 1))
        left -= j + 1 - lo
        j = (lo - 1) % m
    return out


def blocked(g, S, Y, ys, end, bound, m, BS):
    q = -g.copy()
    segs = segments(end, bound, m, BS)
    al = {}
    mats = {}
    for lo, hi in segs:
        sl = list(range(lo, hi))
        A = block_matrix(S, Y, ys, sl)
        mats[(lo, hi)] = A
        p = np.array([S[j] @ q for j in sl])
        a = A @ p
        for t, j in enumerate(sl):
            al[j] = a[t]
        for t in range(len(sl) - 1, -1, -1):             # newest first, like the recursion
            q = q - a[t] * Y[sl[t]]
    j0 = (end - 1) % m
    q = q * (ys[j0] / (Y[j0] @ Y[j0]))
    for lo, hi in reversed(segs):
        sl = list(range(lo, hi))
        A = mats[(lo, hi)]
        t = np.array([Y[j] @ q for j in sl])
        c = np.array([al[j] * ys[j] for j in sl]) - t
        gam = A.T @ c
        for u, j in enumerate(sl):
            q = q + gam[u] * S[j]
    return q


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    BS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    og = O.OracleGrid()
    og.set_cells(scenes.analytic_cells())
    probs = scenes.random_problems(N, seed0=1000)
    rows = []
    for p in probs:
        for (ps, k) in ((0, 15), (0, 60), (0, 150), (0, 300), (1, 40), (2, 80), (4, 120), (6, 200)):
            st = F.capture(og, p, None, ps, k)
            if st is None or st["bound"] < 2:
                continue
            m = st["lm_s"].shape[0]
            sl = order(st, m)
            ref = two_loop(st["g"], st["lm_s"], st["lm_y"], st["lm_ys"], sl, np.longdouble)
            d64 = two_loop(st["g"], st["lm_s"], st["lm_y"], st["lm_ys"], sl, np.float64)
            db = blocked(st["g"], st["lm_s"], st["lm_y"], st["lm_ys"], st["end"], st["bound"], m, BS)
            nr = float(np.abs(ref).max())
            e2, eb = float(np.abs(d64 - ref).max()) / nr, float(np.abs(db - ref.astype(np.float64)).max()) / nr
            rows.append((ps, k, st["bound"], e2, eb))
            print("pass %d k %3d bound %3d   sequential f64 %.1e   blocked(%d) f64 %.1e   ratio %.1f" % (ps, k, st["bound"], e2, BS, eb, eb / max(e2, 1e-300)))
    r = np.array(rows)
    print("median relative deviation from the long-double direction: sequential %.1e, blocked %.1e (x %.1f); worst blocked %.1e, worst sequential %.1e" % (
        np.median(r[:, 3]), np.median(r[:, 4]), np.median(r[:, 4] / r[:, 3]), r[:, 4].max(), r[:, 3].max()))


if __name__ == "__main__":
    main()
