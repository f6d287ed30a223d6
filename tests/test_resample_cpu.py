"""the initial-guess stage between the front-end and optimizeSE2Traj (plan_manager.cpp:62-132).
Three independently written forms -- the oracle's C++ restatement in the reference's own two-pass order (oracle/resample.hpp),
the product's streaming batched routine behind the C-ABI (uph_resample_batch) and the product's numpy mirror (resample.resample_path)
-- must agree bit for bit, and reproduce answers derived by hand from the reference's lines.  (The reference has no test or fixture
for this stage and cannot be built here, so there is no golden vector to pin against: "parity unpinned" in oracle/resample.hpp.)"""
import math

import numpy as np
import pytest

from uneven_planner_amd import resample as R

KEYS = ("init_xy", "end_xy", "inner_xy", "init_yaw", "end_yaw", "inner_yaw")


def same(a, b):
    return all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in KEYS) and a["total_time"] == b["total_time"]


def paths(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        s = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(-math.pi, math.pi)])
        g = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(-math.pi, math.pi)])
        p = R.hermite_path(s, g, interval=rng.choice([0.03, 0.06, 0.11]))
        if i % 3 == 0:                 # front-end style yaw column: wrapped into (-pi, pi] with jumps, plus multiples of 2 pi
            p[:, 2] = np.arctan2(np.sin(p[:, 2]), np.cos(p[:, 2])) + 2 * math.pi * rng.integers(-2, 3, size=p.shape[0])
        if i % 4 == 1:                 # repeated poses (zero-length segments) and one long jump (several way-points inside one segment)
            p = np.concatenate([p[:5], p[4:5], p[4:5], p[5::7]])
        out.append(p)
    return out


def test_three_forms_agree_bit_for_bit(oracle):
    ps = paths(60, 42)
    for kw in (dict(), dict(piece_len=0.21, yaw_piece_times=3.0, mean_vel=0.8, init_time_times=1.5, init_sig_vel=0.11)):
        native = R.resample_batch(ps, cap_xy=4096, cap_yaw=4096, **kw)
        for p, nat in zip(ps, native):
            assert same(oracle.resample(p, kw), nat) and same(R.resample_path(p, **kw), nat)
    # the unwrapped yaw column (:62-78) from the one-pass form equals the reference-order first pass
    import ctypes as C
    from uneven_planner_amd import _lib
    L = _lib.load()
    p = ps[0]
    un = np.zeros(p.shape[0])
    off = np.array([0, p.shape[0]], dtype=np.int64)
    mp = _lib.ManagerParams(**R.MANAGER_PARAMS)
    z6, z3, big, n1, t1 = np.zeros(6), np.zeros(3), np.zeros(8192), np.zeros(1, dtype=np.int32), np.zeros(1)
    dp = lambda a: a.ctypes.data_as(_lib.DP)
    rc = L.uph_resample_batch(C.byref(mp), 1, dp(np.ascontiguousarray(p)), off.ctypes.data_as(C.POINTER(C.c_int64)), 4096, 4096, dp(z6), dp(z6.copy()), dp(z3), dp(z3.copy()),
                              dp(big), dp(big.copy()), n1.ctypes.data_as(C.POINTER(C.c_int32)), n1.copy().ctypes.data_as(C.POINTER(C.c_int32)), dp(t1), dp(un))
    assert rc == 0 and np.array_equal(un, oracle.resample(p)["yaw_unwrapped"])
    # (the reference's two `while` loops leave a step in [pi/2, 3 pi/2) as it is -- minus 2 pi, then plus 2 pi -- so no bound of pi/2 on the result)
    assert np.abs(np.diff(un)).max() < 1.5 * math.pi


def test_known_answers_derived_by_hand(oracle):
    # straight line along x in eleven 0.25 m steps (exact in binary): temp_len_pos passes 0.3 on segment 2 (0.5 > 0.3: node at
    # x = 0.25 + (1 - 0.2/0.25) * 0.25 = 0.3), leaves 0.2, and so on: nodes every 0.3 m up to 2.7; yaw nodes every 0.15 m;
    # total_time = 2.75 / 0.5 * 1.2 = 6.6
    p = np.column_stack([0.25 * np.arange(12), np.zeros(12), np.zeros(12)])
    for r in (oracle.resample(p), R.resample_batch([p])[0], R.resample_path(p)):
        assert np.allclose(r["inner_xy"][0], 0.3 * np.arange(1, 10), atol=1e-12) and not r["inner_xy"][1].any()
        assert r["inner_yaw"].shape == (18,) and not r["inner_yaw"].any()
        assert abs(r["total_time"] - 6.6) < 1e-12
        assert np.array_equal(r["init_xy"], [[0.0, 0.05, 0.0], [0.0, 0.0, 0.0]]) and np.array_equal(r["end_xy"], [[2.75, 0.05, 0.0], [0.0, 0.0, 0.0]])
    # yaw column 3.0 -> -3.0 (a +0.283 rad turn through the branch cut): unwrapped to 3.0 -> 3.283..; end heading and velocity follow it
    p = np.array([[0.0, 0.0, 3.0], [0.2, 0.0, -3.0], [0.4, 0.0, -3.0]])
    for r in (oracle.resample(p), R.resample_batch([p])[0], R.resample_path(p)):
        assert abs(r["end_yaw"][0] - (2 * math.pi - 3.0)) < 1e-15 and abs(r["init_yaw"][0] - 3.0) == 0.0
        assert abs(r["end_xy"][0, 1] - 0.05 * math.cos(2 * math.pi - 3.0)) < 1e-17
        # yaw nodes at arc length 0.15 (on segment 0, three quarters of the way through the turn) and 0.30 (segment 1, constant)
        assert np.allclose(r["inner_yaw"], [3.0 + 0.75 * (2 * math.pi - 6.0), 2 * math.pi - 3.0], atol=1e-14)
        assert np.allclose(r["inner_xy"], [[0.3], [0.0]], atol=1e-15)


def test_capacity_overflow_is_reported_with_counts(oracle):
    from uneven_planner_amd import _lib
    p = np.column_stack([np.linspace(0, 60, 1200), np.zeros(1200), np.zeros(1200)])   # 60 m: about 200 position way-points > 128
    with pytest.raises(_lib.UnevenHipError, match="more way-points"):
        R.resample_batch([p])
    assert same(R.resample_batch([p], cap_xy=256, cap_yaw=512)[0], oracle.resample(p))
    with pytest.raises(_lib.UnevenHipError):
        R.resample_batch([p[:1]])                                                      # a path needs two poses


def test_test_node_variant_three_forms_and_known_answers(oracle):
    """the OTHER producer of optimizeSE2Traj's arguments in the reference: the back-end test node's ALMTrajOpt::rcvWpsCallBack
    (back_end/src/alm_traj_opt.cpp:73-144) -- literals 0.3 / 2.0 / 0.05 / 1.2, the optimiser's max_vel, `if` instead of `while` in both combs,
    every position node also appended to the yaw nodes -- selected by uph_manager_params.test_mode"""
    ps = paths(40, 7)
    kw = dict(test_mode=1, test_max_vel=0.7, piece_len=9.0, init_sig_vel=3.0, yaw_piece_times=7.0)      # the manager parameters must be ignored in this mode
    native = R.resample_batch(ps, cap_xy=4096, cap_yaw=4096, **kw)
    for p, nat in zip(ps, native):
        assert same(oracle.resample(p, dict(test_mode=1, test_max_vel=0.7)), nat) and same(R.resample_path(p, test_mode=True, test_max_vel=0.7), nat)
    # known answer in exact rational arithmetic: straight line along x in eleven steps of 0.22 m (no comb ever lands within 0.02 m of a
    # threshold, so float rounding cannot flip a comparison), yaw linear in x (0.4 rad / m).  Yaw comb (pitch 0.15): every 0.22 m segment
    # passes it, but `if` emits ONE node and the excess carries over, so node k sits at arc length 0.15 (k + 1) -- behind its own segment from
    # k = 2 on (negative fraction: the reference extrapolates).  Position comb (pitch 0.3): fires on segments 1, 2, 4, 5, 6, 8, 9, 10 at
    # x = 0.3, 0.6, ..., 2.4; each of these appends its yaw AFTER the yaw comb's node of the same segment.
    from fractions import Fraction as Fr
    step, pitch_p, pitch_y = Fr(22, 100), Fr(3, 10), Fr(15, 100)
    cy = cp = Fr(0)
    ex_xy, ex_yaw = [], []
    for k in range(11):
        cy += step; cp += step
        if cy > pitch_y:
            ex_yaw.append(Fr(4, 10) * (k * step + (1 - (cy - pitch_y) / step) * step)); cy -= pitch_y
        if cp > pitch_p:
            x = k * step + (1 - (cp - pitch_p) / step) * step
            ex_xy.append(x); ex_yaw.append(Fr(4, 10) * x); cp -= pitch_p
    assert [float(v) for v in ex_xy] == [0.3 * k for k in range(1, 9)] or np.allclose([float(v) for v in ex_xy], 0.3 * np.arange(1, 9))
    assert len(ex_yaw) == 19
    p = np.column_stack([0.22 * np.arange(12), np.zeros(12), 0.4 * 0.22 * np.arange(12)])
    for r in (oracle.resample(p, dict(test_mode=1)), R.resample_batch([p], test_mode=1)[0], R.resample_path(p, test_mode=True)):
        assert np.allclose(r["inner_xy"][0], [float(v) for v in ex_xy], atol=1e-12) and not r["inner_xy"][1].any()
        assert np.allclose(r["inner_yaw"], [float(v) for v in ex_yaw], atol=1e-12)
        assert abs(r["total_time"] - 11 * 0.22 / 0.5 * 1.2) < 1e-12
        assert np.array_equal(r["init_xy"], [[0.0, 0.05, 0.0], [0.0, 0.0, 0.0]])
    # a long jump (1.0 m in one segment): PlanManager's `while` emits three position nodes inside it, the test node's `if` only one and
    # carries 0.7 on -- the two stages are different functions of the same path
    p = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [1.1, 0.0, 0.0]])
    assert R.resample_path(p)["inner_xy"].shape[1] == 3 and R.resample_path(p, test_mode=True)["inner_xy"].shape[1] == 2
    assert same(oracle.resample(p, dict(test_mode=1)), R.resample_batch([p], test_mode=1)[0])
