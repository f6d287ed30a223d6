"""BASELINE.json's throughput batch at full size (B = 8192, hill scene, 128 lanes per trajectory): size-independent properties.
Every trajectory is solved by its own workgroup from its own state, so the result of a problem may depend on neither the batch around
it, nor its position in the batch, nor the launch order, nor the run -- bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
B = 8192


def test_full_batch_is_deterministic_order_independent_and_equals_small_batches():
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    m = U.UnevenMap()
    m.build(scenes.make_hill_cloud())
    nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
    probs = scenes.random_problems(B, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
    opt = U.ALMTrajOpt(m)

    def solve(pp, lanes=0):
        o = U.ALMTrajOpt(m) if lanes else opt
        if lanes:
            o.set_lanes(lanes)
        o.set_rho(1.0)
        return o.optimize_batch(pp)
    a = solve(probs)
    b = solve(probs)
    same = lambda r, s: r["ret"] == s["ret"] and r["cost"] == s["cost"] and r["evals"] == s["evals"] and np.array_equal(r["x"], s["x"])
    assert all(same(r, s) for r, s in zip(a, b))                                   # run to run
    perm = np.random.default_rng(0).permutation(B)
    c = solve([probs[i] for i in perm])
    assert all(same(a[i], c[k]) for k, i in enumerate(perm))                       # position in the batch / launch order
    idx = np.random.default_rng(1).choice(B, 96, replace=False)
    d = solve([probs[i] for i in idx], lanes=128)                                  # a small batch forced to the same 128-lane kernel
    assert all(same(a[i], d[k]) for k, i in enumerate(idx))
    # what the batch delivers: every solve ended by convergence or by the ALM pass limit, converged ones are feasible
    rets = np.array([r["ret"] for r in a])
    assert set(np.unique(rets)) <= {0, 2} and 0.5 < (rets == 0).mean() < 0.9
    rep = opt.getMaxVxAxAyCurAttSig()
    conv = rets == 0
    assert np.all(np.abs(rep[conv, 0]) < 0.5 * 1.05) and np.all(rep[conv, 5] < 0.05 * 1.1)      # max_vel, max_sig of run_hill.yaml


def test_kernel_selection_boundaries():
    """the batch sizes at which uph_batch_upload switches kernels (512 lanes up to 256 problems, 256 lanes uncapped below 512, 256 lanes with the register cap from 512,
    128 lanes from 2304): every variant must solve, and the objective at the common starting points must agree to rounding"""
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    from conftest import rel
    m = U.UnevenMap()
    m.set_cells(scenes.analytic_cells())
    probs = scenes.random_problems(2304, seed0=3000, dmin=3.0, dmax=6.0)
    ref_f = ref_g = None
    for Bx in (1, 2, 256, 257, 511, 512, 2303, 2304):
        opt = U.ALMTrajOpt(m)
        opt.upload(probs[:Bx])
        f, g = opt.eval_batch(opt.x0_packed(probs[:Bx]))
        if ref_f is None:
            ref_f, ref_g = f[0], g[0]
        assert abs(f[0] - ref_f) / abs(ref_f) < 1e-12 and rel(ref_g, g[0]) < 1e-11, Bx
        opt.set_rho(1.0)
        opt.solve()
        out = opt.download()
        assert all(o["ret"] in (0, 2) for o in out), Bx
