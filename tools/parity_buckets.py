"""Final-trajectory parity, bucketed by the length of the solve.

For N random hill-cloud problems and a few parameter sets (the shipped run_hill.yaml values and two that cap the L-BFGS iterations
per ALM pass, which yields short solves), three solvers run on identical inputs: the CPU oracle, the same oracle rebuilt with FMA
contraction (the reference's own reproducibility floor: ~1 ulp per operation, nothing else changed) and the device.  Rows = buckets
of the oracle's total L-BFGS iteration count; columns = fraction of problems whose final way-points / cost agree with the oracle
to 1e-4 (relative, infinity norm) and the median deviation.  usage: python tools/parity_buckets.py [N] [out.json] [hill|desert|vocano|astar]
(desert / vocano: the reference's own clouds, fixtures tests/golden/*_xyz.npz, run_hill.yaml / run_vocano.yaml parameters only)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import sensitivity                          # noqa: E402
import uneven_planner_amd as U              # noqa: E402
from oracle import oracle_py as O           # noqa: E402
from uneven_planner_amd import scenes       # noqa: E402

PARAM_SETS = [("run_hill.yaml", None), ("inner_max_iter=8", dict(inner_max_iter=8.0)), ("inner_max_iter=3", dict(inner_max_iter=3.0))]
bucket_table = sensitivity.bucket_table


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    scene = sys.argv[3] if len(sys.argv) > 3 else "hill"
    global PARAM_SETS
    if os.environ.get("UPH_PB_ONLY_YAML"):           # the shipped parameter set only (large-N drift statistics)
        PARAM_SETS = PARAM_SETS[:1]
    if scene in ("hill", "astar"):
        if scene == "astar":
            PARAM_SETS = PARAM_SETS[:1]
        m = U.UnevenMap()
        m.build(scenes.make_hill_cloud())
    else:
        PARAM_SETS = PARAM_SETS[:1]
        m = U.UnevenMap(dict(max_rho=0.08) if scene == "vocano" else None)        # run_vocano.yaml differs in max_rho only
        m.build(np.load(os.path.join(os.getcwd(), "tests", "golden", "%s_xyz.npz" % scene))["xyz"])
    nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
    if scene == "astar":
        # round 6: the A*-seeded workload of `bench.py --workload astar` -- the problems the reference's own chain produces (device KinoAstar::plan + PlanManager's resampling)
        sys.path.insert(0, os.getcwd())
        import bench
        probs, meta = bench.astar_batch(U, scenes, m, (nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]), N, 1000)
        print("A*-seeded problems:", meta)
    else:
        probs = scenes.random_problems(N, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
    og = O.OracleGrid()
    og.set_cells(m.map_buffer)
    report = {}
    for tag, prm in PARAM_SETS:
        T = int(os.environ.get("UPH_PB_THREADS", "1"))          # host threads for the two CPU legs (N >= 4096: minutes on one thread)
        ref = sensitivity.solve_many(lambda: O.OracleALM(og, prm), probs, T)
        fma = sensitivity.solve_with_fma_oracle(m.map_buffer, probs, prm, threads=T)
        opt = U.ALMTrajOpt(m, prm)
        opt.set_lanes(128)
        opt.set_rho(1.0)
        dev = opt.optimize_batch(probs)
        report[tag] = dict(device=bucket_table(ref, dev), floor=bucket_table(ref, fma),
                           same_ret_device=float(np.mean([a["ret"] == b["ret"] for a, b in zip(dev, ref)])),
                           same_ret_floor=float(np.mean([a["ret"] == b["ret"] for a, b in zip(fma, ref)])),
                           drift=sensitivity.drift_stats(ref, fma, dev))
        dr = report[tag]["drift"]
        print("== %s   (same return code: device %.0f %%, floor %.0f %%)" % (tag, 100 * report[tag]["same_ret_device"], 100 * report[tag]["same_ret_floor"]))
        print("   converged: oracle %.3f  oracle(FMA) %.3f  device %.3f | discordant device-only %d / oracle-only %d (McNemar p %.2g), FMA-only %d / oracle-only %d (p %.2g)"
              " | final cost lower / higher than the oracle's: device %d / %d (sign test p %.2g), FMA %d / %d (p %.2g)" % (
                  dr["converged_frac"]["oracle"], dr["converged_frac"]["oracle_fma"], dr["converged_frac"]["device"], dr["device_vs_oracle"]["device_only"],
                  dr["device_vs_oracle"]["oracle_only"], dr["device_vs_oracle"]["mcnemar_p"], dr["fma_vs_oracle"]["fma_only"], dr["fma_vs_oracle"]["oracle_only"],
                  dr["fma_vs_oracle"]["mcnemar_p"], dr["device_vs_oracle"]["cost_lower"], dr["device_vs_oracle"]["cost_higher"], dr["device_vs_oracle"]["cost_sign_p"],
                  dr["fma_vs_oracle"]["cost_lower"], dr["fma_vs_oracle"]["cost_higher"], dr["fma_vs_oracle"]["cost_sign_p"]))
        print("  L-BFGS iterations    n | device: way-points<=1e-4  cost<=1e-4  median   max    | oracle(FMA): way-points<=1e-4  cost<=1e-4  median   max")
        for d, f in zip(report[tag]["device"], report[tag]["floor"]):
            print("  [%4d, %6d) %5d |        %5.0f %%          %5.0f %%   %.1e %.1e |             %5.0f %%          %5.0f %%   %.1e %.1e" % (
                d["lo"], d["hi"], d["n"], 100 * d["x_le_1e4"], 100 * d["c_le_1e4"], d["x_median"], d["x_max"],
                100 * f["x_le_1e4"], 100 * f["c_le_1e4"], f["x_median"], f["x_max"]))
    if len(sys.argv) > 2:
        json.dump(report, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
