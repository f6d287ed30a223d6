"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes bridge to oracle/liboracle.so (the CPU restatement of the
reference's back-end optimiser and map build).  PARITY UNPINNED (see oracle/banded.hpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the
checker / reported baseline.  Nothing under uneven_planner_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PARAM_ORDER = ["rho_T", "rho_ter", "max_vel", "max_acc_lon", "max_acc_lat", "max_kap", "min_cxi", "max_sig",
               "use_scaling", "rho", "beta", "gamma", "epsilon_con", "max_iter", "g_epsilon", "min_step",
               "inner_max_iter", "delta", "mem_size", "past", "int_K"]
# plan_manager/params/run_hill.yaml:32-55
HILL_PARAMS = dict(rho_T=100000.0, rho_ter=10.0, max_vel=0.5, max_acc_lon=5.0, max_acc_lat=10.0, max_kap=2.1,
                   min_cxi=0.8, max_sig=0.05, use_scaling=1.0, rho=1.0, beta=1000.0, gamma=1.0, epsilon_con=0.001,
                   max_iter=10.0, g_epsilon=1.0e-3, min_step=1.0e-32, inner_max_iter=10000.0, delta=1.0e-4,
                   mem_size=256, past=3, int_K=16)
MAP_PARAM_ORDER = ["iter_num", "map_size_x", "map_size_y", "ellipsoid_x", "ellipsoid_y", "ellipsoid_z",
                   "xy_resolution", "yaw_resolution", "min_cnormal", "max_rho", "gravity"]
# plan_manager/params/run_hill.yaml:2-14
HILL_MAP_PARAMS = dict(iter_num=2, map_size_x=10.0, map_size_y=10.0, ellipsoid_x=0.2, ellipsoid_y=0.1,
                       ellipsoid_z=0.1, xy_resolution=0.05, yaw_resolution=0.1, min_cnormal=0.8, max_rho=0.05,
                       gravity=9.81)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, dp, ip, d, i = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_double, C.c_int
        L.orc_grid_create.restype = vp
        L.orc_grid_create.argtypes = [d, d, d, d, d]
        L.orc_grid_destroy.argtypes = [vp]
        L.orc_grid_dims.argtypes = [vp, ip]
        L.orc_grid_set_cells.argtypes = [vp, dp]
        L.orc_grid_get_cells.argtypes = [vp, dp, dp]
        L.orc_terrain_all_with_grad.argtypes = [vp, dp, i, dp, dp]
        L.orc_terrain_get.argtypes = [vp, dp, i, dp]
        L.orc_terrain_variables.argtypes = [vp, dp, i, dp]
        L.orc_minco_generate.argtypes = [i, i, dp, dp, dp, dp, dp, dp]
        L.orc_minco_grad_ct_to_qt.argtypes = [i, i, dp, dp, dp, dp, dp, dp, dp]
        L.orc_minco_jerk_grad.argtypes = [i, i, dp, dp, dp, dp, dp, dp]
        L.orc_banded_solve.argtypes = [i, i, i, dp, dp, i, i]
        L.orc_lbfgs_rosenbrock.restype = i
        L.orc_lbfgs_rosenbrock.argtypes = [i, dp, dp, i, i, d, d, ip, ip]
        L.orc_alm_create.restype = vp
        L.orc_alm_create.argtypes = [vp, dp]
        L.orc_alm_destroy.argtypes = [vp]
        L.orc_alm_set_rho.argtypes = [vp, d]
        L.orc_alm_get_rho.restype = d
        L.orc_alm_get_rho.argtypes = [vp]
        L.orc_alm_set_flat_debug.argtypes = [vp, i]
        L.orc_alm_setup.restype = i
        L.orc_alm_setup.argtypes = [vp, dp, dp, dp, i, dp, dp, dp, i, d, dp]
        L.orc_alm_set_state.argtypes = [vp, dp, dp, dp, dp]
        L.orc_alm_get_state.argtypes = [vp, dp, dp, dp, dp, dp, dp]
        L.orc_alm_init_scaling.argtypes = [vp, dp, i]
        L.orc_alm_eval.restype = d
        L.orc_alm_eval.argtypes = [vp, dp, i, dp, dp]
        L.orc_alm_constrain.restype = d
        L.orc_alm_constrain.argtypes = [vp, dp, i, dp, dp, dp, dp]
        L.orc_alm_get_coeffs.argtypes = [vp, dp, dp, dp, dp, dp]
        L.orc_alm_optimize.restype = i
        L.orc_alm_optimize.argtypes = [vp, dp, dp, dp, i, dp, dp, dp, i, d, dp, dp]
        L.orc_alm_report.argtypes = [vp, dp]
        L.orc_alm_set_coeffs.argtypes = [vp, dp, dp, d, d]
        L.orc_alm_set_capture.argtypes = [vp, i, i, i]
        L.orc_alm_get_iter_log.restype = i
        L.orc_alm_get_iter_log.argtypes = [vp, ip, i]
        L.orc_alm_get_capture.restype = i
        L.orc_alm_get_capture.argtypes = [vp] + [dp] * 11
        L.orc_alm_lbfgs_resume.restype = i
        L.orc_alm_lbfgs_resume.argtypes = [vp, i] + [dp] * 8 + [i]
        L.orc_alm_num_passes.restype = i
        L.orc_alm_num_passes.argtypes = [vp]
        L.orc_alm_get_pass.restype = i
        L.orc_alm_get_pass.argtypes = [vp, i] + [dp] * 9
        L.orc_alm_pass.argtypes = [vp, i, dp, dp]
        L.orc_alm_finish_pass.restype = i
        L.orc_alm_finish_pass.argtypes = [vp]
        L.orc_alm_get_trace.restype = i
        L.orc_alm_get_trace.argtypes = [vp, dp, i]
        L.orc_mapbuilder_create.restype = vp
        L.orc_mapbuilder_create.argtypes = [C.POINTER(C.c_float), C.c_long, i]
        L.orc_mapbuilder_from_pcd.restype = vp
        L.orc_mapbuilder_from_pcd.argtypes = [C.c_char_p]
        L.orc_mapbuilder_destroy.argtypes = [vp]
        L.orc_mapbuilder_cloud_size.restype = C.c_long
        L.orc_mapbuilder_cloud_size.argtypes = [vp]
        L.orc_mapbuilder_get_cloud.argtypes = [vp, C.POINTER(C.c_float)]
        L.orc_map_construct.argtypes = [vp, vp, dp, i, i, i]
        L.orc_map_fit_cell.argtypes = [vp, vp, dp, i, i, i, dp, dp]
        L.orc_grid_get_occ.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.orc_plane_filter.argtypes = [dp, i, dp]
        L.orc_map_write_csv.restype = i
        L.orc_map_write_csv.argtypes = [vp, C.c_char_p]
        L.orc_map_read_csv.restype = i
        L.orc_map_read_csv.argtypes = [vp, C.c_char_p]
        L.orc_grid_compute_occ.argtypes = [vp, d, d]
        L.orc_grid_set_occ.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.orc_grid_frontend_query.argtypes = [vp, dp, i, dp, ip, ip]
        L.orc_kino_create.restype = vp
        L.orc_kino_create.argtypes = [vp, dp]
        L.orc_kino_destroy.argtypes = [vp]
        L.orc_kino_plan.restype = i
        L.orc_kino_plan.argtypes = [vp, dp, dp, i, dp, i, ip, ip, i, ip]
        L.orc_dubins.argtypes = [dp, dp, d, dp]
        L.orc_dubins_interpolate.argtypes = [dp, dp, d, dp, i, dp]
        L.orc_kino_state_transit.argtypes = [vp, dp, dp, d, dp]
        L.orc_heap_selfcheck.restype = i
        L.orc_heap_selfcheck.argtypes = [C.c_uint, i, i]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def params_vec(p=None):
    q = dict(HILL_PARAMS)
    if p:
        q.update(p)
    return np.array([float(q[k]) for k in PARAM_ORDER], dtype=np.float64)


def map_params_vec(p=None):
    q = dict(HILL_MAP_PARAMS)
    if p:
        q.update(p)
    return np.array([float(q[k]) for k in MAP_PARAM_ORDER], dtype=np.float64)


class OracleGrid:
    def __init__(self, size_x=10.0, size_y=10.0, xy_res=0.05, yaw_res=0.1, gravity=9.81):
        self.L = lib()
        self.h = self.L.orc_grid_create(size_x, size_y, xy_res, yaw_res, gravity)
        d = (C.c_int * 3)()
        self.L.orc_grid_dims(self.h, d)
        self.dims = tuple(d)
        self.ncell = self.dims[0] * self.dims[1] * self.dims[2]

    def __del__(self):
        try:
            self.L.orc_grid_destroy(self.h)
        except Exception:
            pass

    def set_cells(self, cells):
        cells = _f64(cells).reshape(self.ncell, 4)
        self.L.orc_grid_set_cells(self.h, _dp(cells))

    def get_cells(self):
        cells = np.zeros((self.ncell, 4))
        cb = np.zeros(self.ncell)
        self.L.orc_grid_get_cells(self.h, _dp(cells), _dp(cb))
        return cells, cb

    def get_occ(self):
        occ = np.zeros(self.ncell, dtype=np.int8)
        occ2 = np.zeros(self.dims[0] * self.dims[1], dtype=np.int8)
        self.L.orc_grid_get_occ(self.h, occ.ctypes.data_as(C.c_char_p), occ2.ctypes.data_as(C.c_char_p))
        return occ, occ2

    def compute_occ(self, min_cnormal=0.8, max_rho=0.05):
        """occupancy of directly set cells (uneven_map.cpp:170-179)"""
        self.L.orc_grid_compute_occ(self.h, float(min_cnormal), float(max_rho))

    def set_occ(self, occ=None, occ2=None):
        a = np.ascontiguousarray(occ, dtype=np.int8) if occ is not None else None
        b = np.ascontiguousarray(occ2, dtype=np.int8) if occ2 is not None else None
        self.L.orc_grid_set_occ(self.h, a.ctypes.data_as(C.c_char_p) if a is not None else None, b.ctypes.data_as(C.c_char_p) if b is not None else None)

    def frontend_query(self, pos):
        """getTerrainSig / isOccupancy / isOccupancyXY at pos (n, 3): (sigma, occ, occ_xy), -1 outside (uneven_map.h:389-396, 473-500)"""
        pos = _f64(pos).reshape(-1, 3)
        n = pos.shape[0]
        sg, oc, oxy = np.zeros(n), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self.L.orc_grid_frontend_query(self.h, _dp(pos), n, _dp(sg), oc.ctypes.data_as(C.POINTER(C.c_int)), oxy.ctypes.data_as(C.POINTER(C.c_int)))
        return sg, oc, oxy

    def all_with_grad(self, pos):
        pos = _f64(pos).reshape(-1, 3)
        n = pos.shape[0]
        v = np.zeros((n, 7))
        g = np.zeros((n, 7, 3))
        self.L.orc_terrain_all_with_grad(self.h, _dp(pos), n, _dp(v), _dp(g))
        return v, g

    def terrain(self, pos):
        pos = _f64(pos).reshape(-1, 3)
        out = np.zeros((pos.shape[0], 4))
        self.L.orc_terrain_get(self.h, _dp(pos), pos.shape[0], _dp(out))
        return out

    def terrain_variables(self, pos):
        pos = _f64(pos).reshape(-1, 3)
        out = np.zeros((pos.shape[0], 7))
        self.L.orc_terrain_variables(self.h, _dp(pos), pos.shape[0], _dp(out))
        return out


KINO_PARAM_ORDER = ["yaw_resolution", "lambda_heu", "weight_r2", "weight_so2", "weight_v_change", "weight_delta_change", "weight_sigma",
                    "time_interval", "collision_interval", "oneshot_range", "wheel_base", "max_steer", "max_vel"]
# plan_manager/params/run_hill.yaml:16-30
HILL_KINO_PARAMS = dict(yaw_resolution=3.15, lambda_heu=1.0, weight_r2=1.0, weight_so2=0.5, weight_v_change=0.0, weight_delta_change=0.0,
                        weight_sigma=10.0, time_interval=0.3, collision_interval=0.06, oneshot_range=1.0, wheel_base=0.26, max_steer=0.5, max_vel=0.5)


def kino_params_vec(p=None):
    q = dict(HILL_KINO_PARAMS)
    if p:
        q.update(p)
    return np.array([float(q[k]) for k in KINO_PARAM_ORDER], dtype=np.float64)


class OracleKinoAstar:
    """KinoAstar::plan (front_end/src/kino_astar.cpp:67-236) through the CPU restatement oracle/kino_astar.hpp"""
    STATUS = {0: "ok", 1: "start not free", 2: "goal not free", 3: "no path", 4: "node pool exhausted", 5: "expansion cap (checker)", 6: "key outside the table"}

    def __init__(self, grid, params=None):
        self.L = lib()
        self.grid = grid
        self.kp = kino_params_vec(params)
        self.h = self.L.orc_kino_create(grid.h, _dp(self.kp))

    def __del__(self):
        try:
            self.L.orc_kino_destroy(self.h)
        except Exception:
            pass

    def plan(self, start, goal, max_expand=0, path_cap=4096, exp_cap=40000):
        s, g = _f64(start), _f64(goal)
        path = np.zeros((path_cap, 3))
        stats = (C.c_int * 4)()
        exp = np.zeros((exp_cap, 3), dtype=np.int32)
        ne = C.c_int(0)
        n = self.L.orc_kino_plan(self.h, _dp(s), _dp(g), int(max_expand), _dp(path), path_cap, stats, exp.ctypes.data_as(C.POINTER(C.c_int)), exp_cap, C.byref(ne))
        return dict(status=stats[0], iter_num=stats[1], use_node_num=stats[2], n_shot=stats[3], path=path[:min(n, path_cap)].copy(), n_path=n,
                    expanded=exp[:min(ne.value, exp_cap)].copy(), n_expanded=ne.value)

    def state_transit(self, state0, ctrl, T):
        out = np.zeros(3)
        self.L.orc_kino_state_transit(self.h, _dp(_f64(state0)), _dp(_f64(ctrl)), float(T), _dp(out))
        return out


def dubins(frm, to, rho):
    """OMPL DubinsStateSpace::dubins + distance: dict(type, t, p, q, distance)"""
    out = np.zeros(6)
    lib().orc_dubins(_dp(_f64(frm)), _dp(_f64(to)), float(rho), _dp(out))
    return dict(type=int(out[0]), t=out[1], p=out[2], q=out[3], distance=out[4])


def dubins_interpolate(frm, to, rho, ts):
    ts = _f64(ts).ravel()
    out = np.zeros((ts.size, 3))
    lib().orc_dubins_interpolate(_dp(_f64(frm)), _dp(_f64(to)), float(rho), _dp(ts), ts.size, _dp(out))
    return out


def heap_selfcheck(seed=1, nops=20000, nan_every=0):
    return lib().orc_heap_selfcheck(int(seed), int(nops), int(nan_every))


class OracleALM:
    """Mirror of the reference's ALMTrajOpt driven through the CPU restatement."""

    def __init__(self, grid, params=None):
        self.L = lib()
        self.grid = grid
        self.pv = params_vec(params)
        self.int_K = int(self.pv[PARAM_ORDER.index("int_K")])
        self.h = self.L.orc_alm_create(grid.h, _dp(self.pv))
        self.n = 0

    def __del__(self):
        try:
            self.L.orc_alm_destroy(self.h)
        except Exception:
            pass

    def setup(self, prob):
        self.prob = prob
        nxy, nyaw = prob["inner_xy"].shape[1], prob["inner_yaw"].shape[0]
        self.piece_xy, self.piece_yaw = nxy + 1, nyaw + 1
        self.S = self.piece_xy * (self.int_K + 1)
        x0 = np.zeros(2 * nxy + nyaw + 1)
        self.n = self.L.orc_alm_setup(self.h, _dp(_f64(prob["init_xy"].T)), _dp(_f64(prob["end_xy"].T)),
                                      _dp(_f64(prob["inner_xy"].T)), nxy, _dp(_f64(prob["init_yaw"])),
                                      _dp(_f64(prob["end_yaw"])), _dp(_f64(prob["inner_yaw"])), nyaw,
                                      float(prob["total_time"]), _dp(x0))
        return x0

    def set_rho(self, rho):
        self.L.orc_alm_set_rho(self.h, float(rho))

    def get_rho(self):
        return self.L.orc_alm_get_rho(self.h)

    def set_flat_debug(self, on):
        self.L.orc_alm_set_flat_debug(self.h, int(on))

    def set_state(self, lam=None, mu=None, scale_cx=None, scale_fx=None):
        sf = np.array([scale_fx], dtype=np.float64) if scale_fx is not None else None
        self.L.orc_alm_set_state(self.h, _dp(_f64(lam)) if lam is not None else None,
                                 _dp(_f64(mu)) if mu is not None else None,
                                 _dp(_f64(scale_cx)) if scale_cx is not None else None, _dp(sf))

    def get_state(self):
        lam, mu, sc = np.zeros(self.S), np.zeros(6 * self.S), np.zeros(7 * self.S)
        sf, hx, gx = np.zeros(1), np.zeros(self.S), np.zeros(6 * self.S)
        self.L.orc_alm_get_state(self.h, _dp(lam), _dp(mu), _dp(sc), _dp(sf), _dp(hx), _dp(gx))
        return dict(lam=lam, mu=mu, scale_cx=sc, scale_fx=sf[0], hx=hx, gx=gx)

    def init_scaling(self, x0):
        x0 = _f64(x0)
        self.L.orc_alm_init_scaling(self.h, _dp(x0), x0.size)

    def eval(self, x):
        x = _f64(x)
        g = np.zeros(x.size)
        parts = np.zeros(3)
        f = self.L.orc_alm_eval(self.h, _dp(x), x.size, _dp(g), _dp(parts))
        return f, g, parts

    def constrain(self, x):
        x = _f64(x)
        a1, a2 = np.zeros((6 * self.piece_xy, 2)), np.zeros(self.piece_xy)
        a3, a4 = np.zeros(6 * self.piece_yaw), np.zeros(self.piece_yaw)
        c = self.L.orc_alm_constrain(self.h, _dp(x), x.size, _dp(a1), _dp(a2), _dp(a3), _dp(a4))
        return c, a1, a2, a3, a4

    def coeffs(self):
        cxy, cyaw = np.zeros((6 * self.piece_xy, 2)), np.zeros(6 * self.piece_yaw)
        t1, t2, jc = np.zeros(1), np.zeros(1), np.zeros(1)
        self.L.orc_alm_get_coeffs(self.h, _dp(cxy), _dp(cyaw), _dp(t1), _dp(t2), _dp(jc))
        return cxy, cyaw, t1[0], t2[0], jc[0]

    def optimize(self, prob):
        """== ALMTrajOpt::optimizeSE2Traj (alm_traj_opt.cpp:168-278).  prob: dict as produced by
        uneven_planner_amd.resample.resample_path (init_xy 2x3, end_xy 2x3, inner_xy 2x(Nxy-1), ...)."""
        nxy, nyaw = prob["inner_xy"].shape[1], prob["inner_yaw"].shape[0]
        self.piece_xy, self.piece_yaw = nxy + 1, nyaw + 1
        self.S = self.piece_xy * (self.int_K + 1)
        x = np.zeros(2 * nxy + nyaw + 1)
        stats = np.zeros(6)
        ret = self.L.orc_alm_optimize(self.h, _dp(_f64(prob["init_xy"].T)), _dp(_f64(prob["end_xy"].T)),
                                      _dp(_f64(prob["inner_xy"].T)), nxy, _dp(_f64(prob["init_yaw"])),
                                      _dp(_f64(prob["end_yaw"])), _dp(_f64(prob["inner_yaw"])), nyaw,
                                      float(prob["total_time"]), _dp(x), _dp(stats))
        cxy, cyaw, txy, tyaw, jc = self.coeffs()
        return dict(ret=ret, x=x, alm_iters=int(stats[0]), lbfgs_iters=int(stats[1]), evals=int(stats[2]),
                    last_lbfgs_ret=int(stats[3]), cost=stats[4], wall_ms=stats[5], c_xy=cxy, c_yaw=cyaw,
                    T_xy=txy, T_yaw=tyaw, jerk_cost=jc)

    # ---- teacher-forced test aids ------------------------------------------------------------------------------------
    def set_capture(self, snap_pass=-1, snap_k=-1, record_passes=True):
        """ask the next optimize() to capture the L-BFGS state at the top of iteration snap_k of ALM pass snap_pass and to record every pass"""
        self.L.orc_alm_set_capture(self.h, int(snap_pass), int(snap_k), int(bool(record_passes)))

    def iter_log(self, cap=200000):
        """(rows, 6) int array {pass, k, ls, bound, end, updated} of every completed L-BFGS iteration of the last optimize()"""
        buf = np.zeros((cap, 6), dtype=np.int32)
        n = self.L.orc_alm_get_iter_log(self.h, buf.ctypes.data_as(C.POINTER(C.c_int)), cap)
        return buf[:min(n, cap)]

    def _state_bufs(self):
        n, m, past = self.n_vars, int(self.pv[PARAM_ORDER.index("mem_size")]), max(1, int(self.pv[PARAM_ORDER.index("past")]))
        return dict(x=np.zeros(n), g=np.zeros(n), d=np.zeros(n), pf=np.zeros(past), lm_ys=np.zeros(m), lm_s=np.zeros((m, n)), lm_y=np.zeros((m, n)), scal=np.zeros(5))

    def capture(self):
        """the captured state (dict) or None; includes lambda / mu / rho in force during that pass"""
        self.n_vars = 2 * (self.piece_xy - 1) + (self.piece_yaw - 1) + 1
        b = self._state_bufs()
        lam, mu, rho = np.zeros(self.S), np.zeros(6 * self.S), np.zeros(1)
        ok = self.L.orc_alm_get_capture(self.h, *[_dp(b[k]) for k in ("x", "g", "d", "pf", "lm_ys", "lm_s", "lm_y", "scal")], _dp(lam), _dp(mu), _dp(rho))
        if not ok:
            return None
        b.update(step=b["scal"][0], fx=b["scal"][1], k=int(b["scal"][2]), end=int(b["scal"][3]), bound=int(b["scal"][4]), lam=lam, mu=mu, rho=rho[0])
        return b

    def lbfgs_resume(self, st, budget):
        """continue from state dict `st` (as returned by capture()) for at most `budget` iterations; returns (code, new state); 999 = still running"""
        b = {k: np.ascontiguousarray(st[k], dtype=np.float64).copy() for k in ("x", "g", "d", "pf", "lm_ys", "lm_s", "lm_y")}
        scal = np.array([st["step"], st["fx"], st["k"], st["end"], st["bound"]], dtype=np.float64)
        ret = self.L.orc_alm_lbfgs_resume(self.h, b["x"].size, *[_dp(b[k]) for k in ("x", "g", "d", "pf", "lm_ys", "lm_s", "lm_y")], _dp(scal), int(budget))
        b.update(step=scal[0], fx=scal[1], k=int(scal[2]), end=int(scal[3]), bound=int(scal[4]))
        return ret, b

    def passes(self):
        out = []
        n = 2 * (self.piece_xy - 1) + (self.piece_yaw - 1) + 1
        for i in range(self.L.orc_alm_num_passes(self.h)):
            r = dict(x_in=np.zeros(n), x_out=np.zeros(n), lam_in=np.zeros(self.S), lam_out=np.zeros(self.S), mu_in=np.zeros(6 * self.S), mu_out=np.zeros(6 * self.S),
                     hx=np.zeros(self.S), gx=np.zeros(6 * self.S))
            scal = np.zeros(6)
            self.L.orc_alm_get_pass(self.h, i, *[_dp(r[k]) for k in ("x_in", "x_out", "lam_in", "lam_out", "mu_in", "mu_out", "hx", "gx")], _dp(scal))
            r.update(rho_in=scal[0], rho_out=scal[1], cost=scal[2], ret=int(scal[3]), k=int(scal[4]), converged=int(scal[5]))
            out.append(r)
        return out

    def finish_pass(self):
        """what the ALM loop does after an accepted L-BFGS code: updateDualVars, then judgeConvergence (alm_traj_opt.cpp:257-259)"""
        return int(self.L.orc_alm_finish_pass(self.h))

    def alm_pass(self, x):
        """ONE ALM pass from x with the current duals / scales / rho.  Returns dict(ret, k, accepted, converged, cost, x)"""
        x = _f64(x).copy()
        o = np.zeros(5)
        self.L.orc_alm_pass(self.h, x.size, _dp(x), _dp(o))
        return dict(ret=int(o[0]), k=int(o[1]), accepted=int(o[2]), converged=int(o[3]), cost=o[4], x=x)

    def trace(self, cap=20000):
        """cost after every accepted L-BFGS iteration of the last optimize(); -1 marks the start of an ALM pass"""
        buf = np.zeros(cap)
        n = self.L.orc_alm_get_trace(self.h, _dp(buf), cap)
        return buf[:min(n, cap)]

    def set_coeffs(self, c_xy, c_yaw, T_xy, T_yaw):
        """install a trajectory given by its coefficients (after setup()); report() then evaluates exactly that trajectory"""
        self.L.orc_alm_set_coeffs(self.h, _dp(_f64(c_xy)), _dp(_f64(c_yaw)), float(T_xy), float(T_yaw))

    def report(self):
        out = np.zeros(7)
        self.L.orc_alm_report(self.h, _dp(out))
        return out


class OracleMapBuilder:
    def __init__(self, xyz=None, pcd_path=None, apply_filters=True):
        self.L = lib()
        if pcd_path is not None:
            self.h = self.L.orc_mapbuilder_from_pcd(pcd_path.encode())
            if not self.h:
                raise IOError("cannot read " + pcd_path)
        else:
            xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
            self.h = self.L.orc_mapbuilder_create(xyz.ctypes.data_as(C.POINTER(C.c_float)), xyz.shape[0],
                                                  int(apply_filters))

    def __del__(self):
        try:
            self.L.orc_mapbuilder_destroy(self.h)
        except Exception:
            pass

    def cloud(self):
        n = self.L.orc_mapbuilder_cloud_size(self.h)
        out = np.zeros((n, 3), dtype=np.float32)
        self.L.orc_mapbuilder_get_cloud(self.h, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def construct(self, grid, map_params=None, x0=0, x1=None, do_occ=True):
        mp = map_params_vec(map_params)
        if x1 is None:
            x1 = grid.dims[0]
        self.L.orc_map_construct(self.h, grid.h, _dp(mp), int(x0), int(x1), int(do_occ))

    def fit_cell(self, grid, x, y, yaw, map_params=None):
        mp = map_params_vec(map_params)
        cell, c = np.zeros(4), np.zeros(1)
        self.L.orc_map_fit_cell(self.h, grid.h, _dp(mp), x, y, yaw, _dp(cell), _dp(c))
        return cell, c[0]


def plane_filter(pts):
    pts = _f64(pts).reshape(-1, 3)
    out = np.zeros(4)
    lib().orc_plane_filter(_dp(pts), pts.shape[0], _dp(out))
    return out


def minco_generate(N, D, inPs, ts, head, tail):
    """inPs: D x (N-1); head/tail: D x 3 -> c (6N x D), jerk cost."""
    c = np.zeros((6 * N, D))
    jc = np.zeros(1)
    lib().orc_minco_generate(N, D, _dp(_f64(np.asarray(inPs).reshape(D, N - 1).T)), _dp(_f64(ts)), _dp(_f64(head)),
                             _dp(_f64(tail)), _dp(c), _dp(jc))
    return c, jc[0]


def minco_grad(N, D, inPs, ts, head, tail, gdC, gdT):
    gT = _f64(gdT).copy()
    gP = np.zeros((N - 1, D))
    lib().orc_minco_grad_ct_to_qt(N, D, _dp(_f64(np.asarray(inPs).reshape(D, N - 1).T)), _dp(_f64(ts)),
                                  _dp(_f64(head)), _dp(_f64(tail)), _dp(_f64(gdC)), _dp(gT), _dp(gP))
    return gP.T.copy(), gT


def minco_jerk_grad(N, D, inPs, ts, head, tail):
    gC, gT = np.zeros((6 * N, D)), np.zeros(N)
    lib().orc_minco_jerk_grad(N, D, _dp(_f64(np.asarray(inPs).reshape(D, N - 1).T)), _dp(_f64(ts)), _dp(_f64(head)),
                              _dp(_f64(tail)), _dp(gC), _dp(gT))
    return gC, gT


def banded_solve(A, b, p, q, adjoint=False):
    A = _f64(A)
    n = A.shape[0]
    b = _f64(b).reshape(n, -1).copy()
    lib().orc_banded_solve(n, p, q, _dp(A), _dp(b), b.shape[1], int(adjoint))
    return b


def lbfgs_rosenbrock(x0, mem_size=8, past=3, g_eps=1e-5, delta=1e-6):
    x = _f64(x0).copy()
    f = np.zeros(1)
    it, ev = C.c_int(0), C.c_int(0)
    r = lib().orc_lbfgs_rosenbrock(x.size, _dp(x), _dp(f), mem_size, past, g_eps, delta, C.byref(it), C.byref(ev))
    return r, x, f[0], it.value, ev.value


def lbfgs_trace(kind, x0, mem_size=8, past=3, g_eps=1e-5, delta=1e-6, max_iter=10000, cap=4096):
    """lbfgs_optimize on an analytic test function (0 Rosenbrock, 1 quadratic + quartic coupling) with the record of EVERY evaluation:
    returns (ret, x, f, iters, evals, trace (evals, n + 1))"""
    L = lib()
    x = _f64(x0).copy()
    n = x.size
    tr = np.zeros((cap, n + 1))
    f = np.zeros(1)
    it, ev = C.c_int(0), C.c_int(0)
    L.orc_lbfgs_trace.restype = C.c_int
    L.orc_lbfgs_trace.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double), C.c_int,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    r = L.orc_lbfgs_trace(int(kind), n, _dp(x), int(mem_size), int(past), float(g_eps), float(delta), int(max_iter), _dp(tr), cap, C.byref(ev), C.byref(it), _dp(f))
    return r, x, f[0], it.value, ev.value, tr[:min(ev.value, cap)]


def resample(path, params=None):
    """plan_manager.cpp:62-132 through the C++ restatement (oracle/resample.hpp): path (M,3) -> the optimizeSE2Traj argument dict
    (+ "yaw_unwrapped")"""
    L = lib()
    q = dict(piece_len=0.3, mean_vel=0.5, init_time_times=1.2, yaw_piece_times=2.0, init_sig_vel=0.05)
    if params:
        q.update(params)
    mp5 = np.array([q["piece_len"], q["mean_vel"], q["init_time_times"], q["yaw_piece_times"], q["init_sig_vel"]])
    if q.get("test_mode"):        # the back-end test node's stage (alm_traj_opt.cpp:73-144): literals + the optimiser's max_vel
        mp5[1] = -float(q.get("test_max_vel", 0.5))
    path = _f64(path).reshape(-1, 3)
    M = path.shape[0]
    cap = 4096
    ixy, exy, iyw, eyw = np.zeros(6), np.zeros(6), np.zeros(3), np.zeros(3)
    oxy, oyw, n2, tt, un = np.zeros(2 * cap), np.zeros(cap), (C.c_int * 2)(), np.zeros(1), np.zeros(M)
    L.orc_resample.restype = None
    L.orc_resample.argtypes = [C.POINTER(C.c_double), C.c_int] + [C.POINTER(C.c_double)] * 7 + [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_resample(_dp(path), M, _dp(mp5), _dp(ixy), _dp(exy), _dp(iyw), _dp(eyw), _dp(oxy), _dp(oyw), cap, cap, n2, _dp(tt), _dp(un))
    assert n2[0] <= cap and n2[1] <= cap
    return dict(init_xy=ixy.reshape(3, 2).T.copy(), end_xy=exy.reshape(3, 2).T.copy(), inner_xy=oxy[:2 * n2[0]].reshape(-1, 2).T.copy(),
                init_yaw=iyw, end_yaw=eyw, inner_yaw=oyw[:n2[1]].copy(), total_time=float(tt[0]), yaw_unwrapped=un)


def window_oracle(umap, prob, margin=8.0):
    """Checker for grids too large to copy to the host (BASELINE.json configs[4], 1e9 cells): the cells of the xy window around one
    problem, downloaded from the device map `umap` (UnevenMap.get_window), as an OracleGrid of that window's size, and the problem
    translated into the window's frame.  The translation is a whole number of cells, so it is exact in floating point and the
    window grid holds the same cell values at the same relative positions.  The product solves such problems in a local frame of the
    same kind (uph_common.hpp TrajFrame: the cell corner nearest the middle of the initial path's bounding box -- the window's centre
    here), so both form the lookups' differences, and the ||x||-normalised stop test of lbfgs.hpp:599-606, on numbers of the path's
    own size.  Returns (grid, shifted problem, (sx, sy))."""
    res = float(umap.xy_resolution)
    nx, ny = int(umap.voxel_num[0]), int(umap.voxel_num[1])
    pts = np.concatenate([np.asarray(prob["init_xy"])[:, :1], np.asarray(prob["end_xy"])[:, :1], np.asarray(prob["inner_xy"]).reshape(2, -1)], axis=1)
    ox, oy = float(umap.map_origin[0]), float(umap.map_origin[1])
    half = 0.5 * max(pts[0].max() - pts[0].min(), pts[1].max() - pts[1].min()) + margin
    n = 2 * int(np.ceil(half / res))                      # even number of cells: the window centre is a cell corner
    cx = int(round((0.5 * (pts[0].max() + pts[0].min()) - ox) / res))
    cy = int(round((0.5 * (pts[1].max() + pts[1].min()) - oy) / res))
    if n > nx or n > ny:
        raise ValueError("window larger than the map")
    x0, y0 = min(max(cx - n // 2, 0), nx - n), min(max(cy - n // 2, 0), ny - n)      # slide the window back inside the map near its border
    cx, cy = x0 + n // 2, y0 + n // 2
    cells = umap.get_window(x0, x0 + n, y0, y0 + n)
    kw = dict(size_x=n * res, size_y=n * res, xy_res=res, yaw_res=float(umap.yaw_resolution), gravity=float(umap.params["gravity"]))
    g = OracleGrid(**kw)
    g.kw, g.window_cells = kw, cells.reshape(-1, 4)          # (what a second build of the oracle needs to make the same grid: tests/sensitivity.py fma_session)
    assert g.dims[0] == n and g.dims[1] == n and g.dims[2] == int(umap.voxel_num[2]), (g.dims, n)
    g.set_cells(cells.reshape(-1, 4))
    sx, sy = ox + cx * res, oy + cy * res                 # window centre in map coordinates
    q = {k: (np.array(v, dtype=np.float64).copy() if not np.isscalar(v) else v) for k, v in prob.items()}
    q["init_xy"][0, 0] -= sx; q["init_xy"][1, 0] -= sy
    q["end_xy"][0, 0] -= sx; q["end_xy"][1, 0] -= sy
    if q["inner_xy"].size:
        q["inner_xy"][0] -= sx; q["inner_xy"][1] -= sy
    return g, q, (sx, sy)
