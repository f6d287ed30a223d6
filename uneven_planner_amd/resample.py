"""Initial-guess producer for the optimiser boundary.

Mirrors PlanManager::rcvWpsCallBack's post-A* stage, plan_manager/src/plan_manager.cpp:62-132:
yaw unwrapping (:62-78), boundary PVA with init_sig_vel (:87-95), arc-length way-point insertion with `while`
loops (:97-121) and the total time (:122).  Output is exactly the argument list of
ALMTrajOpt::optimizeSE2Traj (back_end/include/back_end/alm_traj_opt.h:92-98).

The reference's front-end (OMPL / kinodynamic A*) is out of scope; `hermite_path` stands in for it when
synthesising inputs: cubic Hermite start->goal sampled every `collision_interval`.
"""
import math

import numpy as np

# plan_manager/params/run_hill.yaml:57-62
MANAGER_PARAMS = dict(piece_len=0.3, mean_vel=0.5, init_time_times=1.2, yaw_piece_times=2.0, init_sig_vel=0.05, test_mode=0, test_max_vel=0.5)


def hermite_path(start, goal, interval=0.06):
    """Cubic Hermite curve from start (x,y,yaw) to goal, tangent magnitude = chord, sampled about every
    `interval` metres; yaw = tangent direction (end yaws pinned to the given ones)."""
    p0, p1 = np.array(start[:2], float), np.array(goal[:2], float)
    chord = float(np.linalg.norm(p1 - p0))
    m0 = chord * np.array([math.cos(start[2]), math.sin(start[2])])
    m1 = chord * np.array([math.cos(goal[2]), math.sin(goal[2])])
    # arc length estimate on a fine lattice, then uniform-in-parameter sampling dense enough for `interval`
    ts = np.linspace(0.0, 1.0, 2001)

    def pt(t):
        h00 = 2 * t ** 3 - 3 * t ** 2 + 1
        h10 = t ** 3 - 2 * t ** 2 + t
        h01 = -2 * t ** 3 + 3 * t ** 2
        h11 = t ** 3 - t ** 2
        return np.outer(h00, p0) + np.outer(h10, m0) + np.outer(h01, p1) + np.outer(h11, m1)

    def dpt(t):
        h00 = 6 * t ** 2 - 6 * t
        h10 = 3 * t ** 2 - 4 * t + 1
        h01 = -6 * t ** 2 + 6 * t
        h11 = 3 * t ** 2 - 2 * t
        return np.outer(h00, p0) + np.outer(h10, m0) + np.outer(h01, p1) + np.outer(h11, m1)

    fine = pt(ts)
    seg = np.linalg.norm(np.diff(fine, axis=0), axis=1)
    s = np.concatenate([[0.0], np.cumsum(seg)])
    n = max(2, int(math.ceil(s[-1] / interval)) + 1)
    tq = np.interp(np.linspace(0.0, s[-1], n), s, ts)
    xy = pt(tq)
    d = dpt(tq)
    yaw = np.arctan2(d[:, 1], d[:, 0])
    yaw[0], yaw[-1] = start[2], goal[2]
    return np.column_stack([xy, yaw])


def resample_path(init_path, piece_len=0.3, mean_vel=0.5, init_time_times=1.2, yaw_piece_times=2.0,
                  init_sig_vel=0.05, test_mode=False, test_max_vel=0.5):
    """plan_manager.cpp:62-132.  init_path: (M,3) [x,y,yaw].  Returns the optimizeSE2Traj argument dict:
    init_xy, end_xy (2x3: P,V,A columns), inner_xy (2 x (Nxy-1)), init_yaw, end_yaw (3), inner_yaw (Nyaw-1), total_time.
    test_mode: the back-end test node's variant instead, back_end/src/alm_traj_opt.cpp:73-144 (literals 0.3 / 2.0 / 0.05 / 1.2, the
    optimiser's max_vel, one node per comb and segment at most, position nodes also feed the yaw way-points)."""
    if test_mode:
        piece_len, yaw_piece_times, init_sig_vel = 0.3, 2.0, 0.05
    path = np.array(init_path, dtype=np.float64).copy()
    # smooth yaw  :62-78
    for i in range(path.shape[0] - 1):
        dyaw = path[i + 1, 2] - path[i, 2]
        while dyaw >= math.pi / 2:
            path[i + 1, 2] -= math.pi * 2
            dyaw = path[i + 1, 2] - path[i, 2]
        while dyaw <= -math.pi / 2:
            path[i + 1, 2] += math.pi * 2
            dyaw = path[i + 1, 2] - path[i, 2]
    init_xy = np.zeros((2, 3))
    end_xy = np.zeros((2, 3))
    init_xy[:, 0] = path[0, :2]
    end_xy[:, 0] = path[-1, :2]
    init_yaw = np.array([path[0, 2], 0.0, 0.0])
    end_yaw = np.array([path[-1, 2], 0.0, 0.0])
    init_xy[:, 1] = [init_sig_vel * math.cos(init_yaw[0]), init_sig_vel * math.sin(init_yaw[0])]   # :94-95
    end_xy[:, 1] = [init_sig_vel * math.cos(end_yaw[0]), init_sig_vel * math.sin(end_yaw[0])]
    temp_len_yaw = temp_len_pos = total_len = 0.0
    piece_len_yaw = piece_len / yaw_piece_times
    inner_xy, inner_yaw = [], []
    for k in range(path.shape[0] - 1):            # :101-121
        dv = path[k + 1] - path[k]
        temp_seg = math.sqrt(dv[0] * dv[0] + dv[1] * dv[1])
        temp_len_yaw += temp_seg
        temp_len_pos += temp_seg
        total_len += temp_seg
        if test_mode:                                # alm_traj_opt.cpp:122-134: `if`, and temp_node.z() joins the yaw nodes
            if temp_len_yaw > piece_len_yaw:
                inner_yaw.append(path[k, 2] + (1.0 - (temp_len_yaw - piece_len_yaw) / temp_seg) * dv[2])
                temp_len_yaw -= piece_len_yaw
            if temp_len_pos > piece_len:
                node = path[k] + (1.0 - (temp_len_pos - piece_len) / temp_seg) * dv
                inner_xy.append(node[:2].copy())
                inner_yaw.append(node[2])
                temp_len_pos -= piece_len
            continue
        while temp_len_yaw > piece_len_yaw:
            inner_yaw.append(path[k, 2] + (1.0 - (temp_len_yaw - piece_len_yaw) / temp_seg) * dv[2])
            temp_len_yaw -= piece_len_yaw
        while temp_len_pos > piece_len:
            node = path[k] + (1.0 - (temp_len_pos - piece_len) / temp_seg) * dv
            inner_xy.append(node[:2].copy())
            temp_len_pos -= piece_len
    total_time = total_len / test_max_vel * 1.2 if test_mode else total_len / mean_vel * init_time_times   # alm_traj_opt.cpp:137 / plan_manager.cpp:122
    return dict(init_xy=init_xy, end_xy=end_xy,
                inner_xy=np.array(inner_xy, dtype=np.float64).reshape(-1, 2).T.copy(),
                init_yaw=init_yaw, end_yaw=end_yaw, inner_yaw=np.array(inner_yaw, dtype=np.float64),
                total_time=float(total_time))


def resample_batch(paths, cap_xy=128, cap_yaw=256, **kw):
    """the same stage for a batch of front-end paths through the native routine (uph_resample_batch, csrc/resample_host.cpp): list of
    (M_i,3) arrays -> list of optimizeSE2Traj argument dicts.  Needs libunevenhip.so (no GPU); raises when a path needs more than
    cap_xy / cap_yaw way-points (defaults: UPH_MAX_PIECE_XY, UPH_MAX_PIECE_YAW)."""
    import ctypes as C

    from . import _lib
    L = _lib.load()
    mk = dict(MANAGER_PARAMS)
    mk.update(kw)
    mp = _lib.ManagerParams(**{k: (int(bool(v)) if k == "test_mode" else float(v)) for k, v in mk.items()})
    B = len(paths)
    arrs = [np.ascontiguousarray(p, dtype=np.float64).reshape(-1, 3) for p in paths]
    off = np.zeros(B + 1, dtype=np.int64)
    off[1:] = np.cumsum([a.shape[0] for a in arrs])
    flat = np.concatenate(arrs, axis=0) if B else np.zeros((0, 3))
    ixy, exy, iyw, eyw = np.zeros((B, 6)), np.zeros((B, 6)), np.zeros((B, 3)), np.zeros((B, 3))
    oxy, oyw = np.zeros((B, 2 * max(cap_xy, 1))), np.zeros((B, max(cap_yaw, 1)))
    nxy, nyw, tt = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32), np.zeros(B)
    dp = lambda a: a.ctypes.data_as(_lib.DP)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    rc = L.uph_resample_batch(C.byref(mp), B, dp(flat), off.ctypes.data_as(C.POINTER(C.c_int64)), int(cap_xy), int(cap_yaw), dp(ixy), dp(exy), dp(iyw), dp(eyw),
                              dp(oxy), dp(oyw), ip(nxy), ip(nyw), dp(tt), None)
    _lib.check(rc, "uph_resample_batch")
    return [dict(init_xy=ixy[b].reshape(3, 2).T.copy(), end_xy=exy[b].reshape(3, 2).T.copy(), inner_xy=oxy[b, :2 * nxy[b]].reshape(-1, 2).T.copy(),
                 init_yaw=iyw[b].copy(), end_yaw=eyw[b].copy(), inner_yaw=oyw[b, :nyw[b]].copy(), total_time=float(tt[b])) for b in range(B)]


def make_problem(start, goal, **kw):
    """start/goal (x,y,yaw) -> optimizeSE2Traj arguments, through the Hermite stand-in front-end and the resampler."""
    mk = dict(MANAGER_PARAMS)
    mk.update(kw)
    return resample_path(hermite_path(start, goal), **mk)
