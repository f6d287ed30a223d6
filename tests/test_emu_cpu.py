"""CPU tier: the product's workgroup program (uneven_planner_amd/csrc/solver_program.hpp, compiled for the host with a
sequential stand-in for the GPU workgroup -- tests/emu/, test scaffolding) against the oracle.  Catches logic errors of the
device code path before GPU time is spent; the real parity tests are the -m gpu ones (through the C-ABI on an MI355X)."""
import numpy as np
import pytest

import emu_bridge as E
from conftest import rel


@pytest.fixture(scope="module")
def emu(oracle, analytic_cells):
    return E.Emu(analytic_cells, oracle.map_params_vec(), oracle.params_vec())


@pytest.mark.parametrize("lanes", [64, 256])
def test_emu_terrain_eval_scaling(emu, oracle, oracle_grid, hill_problem, small_problems, lanes):
    E.lib().emu_set_lanes(lanes)
    rng = np.random.default_rng(1)
    pos = np.column_stack([rng.uniform(-5.2, 5.2, 500), rng.uniform(-5.2, 5.2, 500), rng.uniform(-np.pi, np.pi, 500)])
    v0, g0 = oracle_grid.all_with_grad(pos)
    v1, g1 = emu.terrain(pos)
    assert np.abs(v0 - v1).max() < 1e-12 and np.abs(g0 - g1).max() / np.abs(g0).max() < 1e-12
    for prob in [hill_problem, small_problems[0]]:
        a = oracle.OracleALM(oracle_grid)
        x0 = a.setup(prob)
        lam = rng.normal(size=a.S) * 0.1
        mu = np.abs(rng.normal(size=6 * a.S)) * 0.1
        sc = rng.uniform(0.2, 1.0, size=7 * a.S)
        a.set_state(lam=lam, mu=mu, scale_cx=sc, scale_fx=0.37)
        a.set_rho(3.0)
        f, g, _ = a.eval(x0)
        st = a.get_state()
        r = emu.run(0, prob, x0, lam=lam, mu=mu, scale_cx=sc, rho=3.0, scale_fx=0.37)
        assert abs(f - r["f"]) / abs(f) < 1e-11
        assert rel(g, r["g"]) < 1e-10 and rel(st["hx"], r["hx"]) < 1e-10 and rel(st["gx"], r["gx"]) < 1e-10
        assert rel(a.coeffs()[0], r["c_xy"]) < 1e-11 and rel(a.coeffs()[1], r["c_yaw"]) < 1e-10
        a2 = oracle.OracleALM(oracle_grid)
        x0 = a2.setup(prob)
        a2.init_scaling(x0)
        s2 = a2.get_state()
        r2 = emu.run(1, prob, x0)
        assert abs(s2["scale_fx"] - r2["scale_fx"]) / s2["scale_fx"] < 1e-10
        assert rel(s2["scale_cx"], r2["scale_cx"]) < 1e-10


def test_per_evaluation_noise_is_at_the_oracles_own_fma_floor(emu, oracle, analytic_cells, small_problems, hill_problem):
    """What the reduced knot system (Thomas recurrences instead of the reference's banded LU) costs per evaluation: the device program's
    f / gradient / coefficients differ from the oracle's by no more than the oracle differs from itself when recompiled with FMA
    contraction (tools/per_eval_noise.py prints the table)."""
    import sensitivity
    E.lib().emu_set_lanes(128)
    probs = [hill_problem] + list(small_problems)
    g = oracle.OracleGrid()
    g.set_cells(analytic_cells)
    pts, plain, dev = [], [], []
    for i, p in enumerate(probs):
        a = oracle.OracleALM(g)
        x0 = a.setup(p)
        a.init_scaling(x0)
        st = a.get_state()
        x = x0 + 0.01 * np.random.default_rng(i).normal(size=x0.size)
        f, gr, _ = a.eval(x)
        plain.append((f, gr, np.asarray(a.coeffs()[0]).ravel()))
        r = emu.run(0, p, x, lam=np.zeros(a.S), mu=np.zeros(6 * a.S), scale_cx=st["scale_cx"], rho=1.0, scale_fx=st["scale_fx"])
        dev.append((r["f"], r["g"], np.asarray(r["c_xy"]).ravel()))
        pts.append(x)
    fma = sensitivity.eval_with_fma_oracle(analytic_cells, probs, pts)
    noise = lambda other: (np.median([abs(a[0] - b[0]) / abs(a[0]) for a, b in zip(plain, other)]),
                           np.median([rel(a[1], b[1]) for a, b in zip(plain, other)]), np.median([rel(a[2], b[2]) for a, b in zip(plain, other)]))
    nd, nf = noise(dev), noise(fma)
    print("device program vs oracle (f, grad, coeffs)", nd, " oracle(FMA) vs oracle", nf)
    assert nd[0] <= 4 * nf[0] + 1e-15 and nd[1] <= 4 * nf[1] + 1e-15 and nd[2] <= 4 * nf[2] + 1e-15
    assert nd[1] < 1e-13 and nd[2] < 1e-13


def test_emu_f32_cell_storage_equals_oracle_on_rounded_cells(oracle, analytic_cells, hill_problem):
    """BASELINE.json configs[4]'s fp32 mode: cells stored as floats, widened on load, fp64 arithmetic -- so the lookups and the
    objective equal the oracle's on the float-rounded grid to rounding error, not to fp32 precision"""
    E.lib().emu_set_lanes(128)
    rounded = analytic_cells.astype(np.float32).astype(np.float64)
    og = oracle.OracleGrid()
    og.set_cells(rounded)
    e32 = E.Emu(analytic_cells, oracle.map_params_vec(), oracle.params_vec()).store_f32()
    rng = np.random.default_rng(2)
    pos = np.column_stack([rng.uniform(-5.2, 5.2, 500), rng.uniform(-5.2, 5.2, 500), rng.uniform(-np.pi, np.pi, 500)])
    v0, g0 = og.all_with_grad(pos)
    v1, g1 = e32.terrain(pos)
    assert np.abs(v0 - v1).max() < 1e-12 and np.abs(g0 - g1).max() / np.abs(g0).max() < 1e-12
    a = oracle.OracleALM(og)
    x0 = a.setup(hill_problem)
    f, g, _ = a.eval(x0)
    r = e32.run(0, hill_problem, x0)
    assert abs(f - r["f"]) / abs(f) < 1e-11 and rel(g, r["g"]) < 1e-10
    # and it is a different grid from the fp64 one (the rounding is visible at 1e-8, far above the tolerances above)
    assert np.abs(rounded - analytic_cells).max() > 1e-9


def test_emu_full_solve_tracks_oracle(emu, oracle, oracle_grid, small_problems):
    """the workgroup program's state machine (ALM / L-BFGS / line search / two-loop) against the oracle"""
    E.lib().emu_set_lanes(256)
    for prob in small_problems:
        a = oracle.OracleALM(oracle_grid)
        ro = a.optimize(prob)
        to = a.trace()
        x0 = oracle.OracleALM(oracle_grid).setup(prob)
        re = emu.run(3, prob, x0)
        assert re["ret"] == ro["ret"] or max(re["alm_iters"], ro["alm_iters"]) >= 9
        assert abs(re["f"] - ro["cost"]) / abs(ro["cost"]) < 2e-2
        assert abs(re["evals"] - ro["evals"]) < 0.5 * ro["evals"]
        # physical feasibility of the emulated solve equals the oracle's
        rep_o = a.report()
        assert abs(abs(re["report"][0]) - abs(rep_o[0])) < 0.02 and abs(re["report"][4] - rep_o[4]) < 0.02


def test_knot_minco_operator_reproduces_banded_solve(oracle):
    """the MINCO operator of the GPU path (normalised time, interior-knot (v, a) operator from a long-double elimination, quintic
    Hermite expansion per piece) against the oracle's banded LU"""
    import ctypes as C
    rng = np.random.default_rng(9)
    for N, D in [(2, 2), (3, 2), (17, 1), (40, 2)]:
        Wr = np.zeros((2 * (N - 1), N + 5))
        E.lib().emu_minco_op(N, Wr.ctypes.data_as(C.POINTER(C.c_double)))
        T = 0.63
        q = rng.normal(size=(D, N - 1)).cumsum(axis=1)
        head, tail = rng.normal(size=(D, 3)) * 0.3, rng.normal(size=(D, 3)) * 0.3
        c_ref, _ = oracle.minco_generate(N, D, q, np.full(N, T), head, tail)
        beta = np.zeros((N + 5, D))
        beta[0], beta[1], beta[2] = head[:, 0], T * head[:, 1], T * T * head[:, 2]
        beta[3:N + 2] = q.T
        beta[N + 2], beta[N + 3], beta[N + 4] = tail[:, 0], T * tail[:, 1], T * T * tail[:, 2]
        z = Wr @ beta                                                  # (v_1, a_1, ..., v_{N-1}, a_{N-1}) in normalised time
        p = np.vstack([beta[0:1], beta[3:N + 2], beta[N + 2:N + 3]])   # knot positions 0..N
        v = np.vstack([beta[1:2], z[0::2], beta[N + 3:N + 4]])
        a = np.vstack([beta[2:3], z[1::2], beta[N + 4:N + 5]])
        ct = np.zeros((6 * N, D))
        for i in range(N):
            dl, v0, v1, a0, a1 = p[i + 1] - p[i], v[i], v[i + 1], a[i], a[i + 1]
            ct[6 * i + 0], ct[6 * i + 1], ct[6 * i + 2] = p[i], v0, 0.5 * a0
            ct[6 * i + 3] = 10 * dl - 6 * v0 - 4 * v1 - 1.5 * a0 + 0.5 * a1
            ct[6 * i + 4] = -15 * dl + 8 * v0 + 7 * v1 + 1.5 * a0 - a1
            ct[6 * i + 5] = 6 * dl - 3 * v0 - 3 * v1 - 0.5 * a0 + 0.5 * a1
        k = np.tile(np.arange(6), N)
        c = ct / (T ** k)[:, None]
        assert rel(c_ref, c) < 1e-11


def test_single_piece_problems(emu, oracle, oracle_grid):
    """piece_xy = 1 (goal closer than one piece length) and piece_yaw = 1: no knot system at all; evaluation, initScaling and the full solve
    of the workgroup program against the oracle (the GPU tier repeats this through the C-ABI: test_gpu_edge.py)"""
    from uneven_planner_amd import resample
    E.lib().emu_set_lanes(128)
    for dx, dy, dyaw in [(0.25, 0.02, 0.1), (0.12, 0.0, 0.0)]:
        p = resample.make_problem((0.3, -0.2, 0.2), (0.3 + dx, -0.2 + dy, 0.2 + dyaw))
        assert p["inner_xy"].shape[1] == 0
        a = oracle.OracleALM(oracle_grid)
        x0 = a.setup(p)
        fo, go, _ = a.eval(x0)
        r0 = emu.run(0, p, x0)
        assert abs(r0["f"] - fo) / abs(fo) < 1e-11 and rel(go, r0["g"]) < 1e-10
        a.init_scaling(x0)
        st = a.get_state()
        r1 = emu.run(1, p, x0)
        assert abs(st["scale_fx"] - r1["scale_fx"]) / st["scale_fx"] < 1e-10 and rel(st["scale_cx"], r1["scale_cx"]) < 1e-10
        ro = oracle.OracleALM(oracle_grid).optimize(p)
        r2 = emu.run(2, p, x0)
        assert r2["ret"] == ro["ret"] and r2["alm_iters"] == ro["alm_iters"] and np.abs(np.asarray(ro["x"]) - r2["x"]).max() < 1e-9


def test_yaw_scatter_window_never_misses_a_sample(analytic_cells, oracle):
    """scatterChunk sums a yaw piece's records over a candidate-slot window (exact bounds widened by two / three slots) and lets the stored
    tag decide; the window must contain every sample of the piece whatever the piece ratio and the accumulated round-off of the time tables.
    The same program built with a window of +-40 slots (every slot of a chunk is a candidate) must give bit-identical gradients: position /
    yaw piece ratios 1:1, 1:2 (PlanManager), 1:3 (test node), 1:5 and ragged ones, at the initial point and at perturbed times."""
    from uneven_planner_amd import resample, scenes
    mp, op = oracle.map_params_vec(), oracle.params_vec()
    tight = E.Emu(analytic_cells, mp, op)
    wide = E.Emu(analytic_cells, mp, op, flags=("-DUPH_SC_WLO=40", "-DUPH_SC_WHI=40", "-DUPH_SC_YB=10"))
    rng = np.random.default_rng(77)
    probs = scenes.random_problems(24, seed0=8800, dmin=2.0, dmax=9.0)
    extra = []
    for p in probs[:12]:
        path = resample.hermite_path((p["init_xy"][0, 0], p["init_xy"][1, 0], p["init_yaw"][0]), (p["end_xy"][0, 0], p["end_xy"][1, 0], p["end_yaw"][0]))
        extra.append(resample.resample_path(path, test_mode=True))                       # three yaw pieces per position piece
        extra.append(resample.resample_path(path, yaw_piece_times=5.0))
        extra.append(resample.resample_path(path, yaw_piece_times=1.0))
        q = resample.resample_path(path, yaw_piece_times=1.7)                            # ragged ratio
        if q["inner_yaw"].shape[0] >= q["inner_xy"].shape[1]:
            extra.append(q)
    n_checked = 0
    for p in probs + extra:
        if p["inner_yaw"].shape[0] + 1 > 256 or p["inner_xy"].shape[1] + 1 > 128 or p["inner_xy"].shape[1] < 1:
            continue
        a = oracle.OracleALM(oracle.OracleGrid())
        x0 = a.setup(p)
        for k in range(3):
            x = x0.copy()
            if k:
                x[0] += rng.uniform(-0.7, 0.7)                                            # other total time: other sample / boundary alignment
                x[1:] += rng.normal(scale=0.02, size=x.size - 1)
            lanes = (128, 64, 256)[k]                                                     # chunk size = lanes: other chunk / piece alignments
            tight.L.emu_set_lanes(lanes); wide.L.emu_set_lanes(lanes)
            rt, rw = tight.run(0, p, x), wide.run(0, p, x)
            assert rt["f"] == rw["f"] and np.array_equal(rt["g"], rw["g"]), (p["inner_xy"].shape[1], p["inner_yaw"].shape[0], k)
            n_checked += 1
    assert n_checked > 150
    tight.L.emu_set_lanes(256)                                                        # (the default of the scaffolding)
