"""does the association of Eigen's reductions move the parity statement?  Solves N hill problems (analytic hill cells, CPU only) with
three builds of the oracle -- default (reductions left to right), -ffp-contract=fast -march=native (the reproducibility floor of DESIGN.md section 6) and
-DORACLE_EIGEN_REDUX=1 (Eigen 3.3.7's vectorised redux order for SSE2, oracle/eigen_redux.hpp) -- and prints the bucket tables of the two variants against the
default build side by side, with the converged rates and the discordant pairs.  usage: python tools/eigen_order_sensitivity.py [N] [threads] [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sensitivity                              # noqa: E402
from oracle import oracle_py as O               # noqa: E402
from uneven_planner_amd import scenes           # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cells = scenes.analytic_cells()
    probs = scenes.random_problems(N, seed0=1000)
    og = O.OracleGrid()
    og.set_cells(cells)
    ref = sensitivity.solve_many(lambda: O.OracleALM(og), probs, T)
    fma = sensitivity.solve_with_fma_oracle(cells, probs, threads=T)
    eig = sensitivity.solve_with_eigen_order_oracle(cells, probs, threads=T)
    tf, te = sensitivity.bucket_table(ref, fma), sensitivity.bucket_table(ref, eig)
    pf, pe = sensitivity.paired_counts(ref, fma), sensitivity.paired_counts(ref, eig)
    print("N = %d hill problems (analytic cells), run_hill.yaml; reference = default oracle build (reductions left to right)" % N)
    print("converged: default %.3f | FMA rebuild %.3f (discordant %d / %d, McNemar p %.2g, same return code %.3f) | Eigen-order build %.3f (discordant %d / %d, p %.2g, same return code %.3f)" % (
        pf["converged_ref"], pf["converged_other"], pf["other_only"], pf["ref_only"], pf["mcnemar_p"], pf["same_ret"],
        pe["converged_other"], pe["other_only"], pe["ref_only"], pe["mcnemar_p"], pe["same_ret"]))
    print("final cost lower / higher than the default build's: FMA %d / %d (sign test p %.2g) | Eigen order %d / %d (p %.2g)" % (
        pf["cost_lower"], pf["cost_higher"], pf["cost_sign_p"], pe["cost_lower"], pe["cost_higher"], pe["cost_sign_p"]))
    print("  L-BFGS iterations     n | FMA rebuild: way-points<=1e-4  median    | Eigen-order build: way-points<=1e-4  median")
    for a, b in zip(tf, te):
        print("  [%4d, %6d) %6d |          %5.1f %%            %.1e |                %5.1f %%            %.1e" % (a["lo"], a["hi"], a["n"], 100 * a["x_le_1e4"], a["x_median"], 100 * b["x_le_1e4"], b["x_median"]))
    if len(sys.argv) > 3:
        json.dump(dict(N=N, fma=dict(buckets=tf, pairs=pf), eigen_order=dict(buckets=te, pairs=pe)), open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
