"""Host-side mirror of the reference's ALMTrajOpt (back_end/include/back_end/alm_traj_opt.h:21-120) over the batched
MI355X back-end.  Same public parameter names, same `optimizeSE2Traj` argument list and return codes; `getTraj()`
returns the piece durations and coefficient matrices the reference's SE2Trajectory holds.  `optimize_batch` is the
batched form (B independent goals solved by one kernel launch, one workgroup per trajectory)."""
import ctypes as C

import numpy as np

from . import _lib

# plan_manager/params/run_hill.yaml:32-55
HILL_OPT_PARAMS = dict(rho_T=100000.0, rho_ter=10.0, max_vel=0.5, max_acc_lon=5.0, max_acc_lat=10.0, max_kap=2.1,
                       min_cxi=0.8, max_sig=0.05, use_scaling=True, rho=1.0, beta=1000.0, gamma=1.0,
                       epsilon_con=0.001, max_iter=10.0, g_epsilon=1.0e-3, min_step=1.0e-32, inner_max_iter=10000.0,
                       delta=1.0e-4, mem_size=256, past=3, int_K=16)
_INT_FIELDS = ("use_scaling", "mem_size", "past", "int_K")


def pack_problems(probs, int_K=16, with_sizes=False):
    """problem dicts -> a ctypes array of uph_problem (column-major matrices, as Eigen::MatrixXd holds them) + the numpy arrays its pointers
    refer to (keep them alive while the array is in use) [+ the per-problem sizes]; no device, no context needed"""
    arr = (_lib.Problem * len(probs))()
    keep, sizes = [], []
    for i, pr in enumerate(probs):
        ixy = np.ascontiguousarray(np.asarray(pr["inner_xy"], dtype=np.float64).T)       # column-major 2 x (Nxy-1)
        iyw = np.ascontiguousarray(pr["inner_yaw"], dtype=np.float64)
        keep += [ixy, iyw]
        a = arr[i]
        a.n_inner_xy, a.n_inner_yaw = ixy.shape[0], iyw.shape[0]
        a.init_xy[:] = np.asarray(pr["init_xy"], dtype=np.float64).T.ravel().tolist()
        a.end_xy[:] = np.asarray(pr["end_xy"], dtype=np.float64).T.ravel().tolist()
        a.init_yaw[:] = np.asarray(pr["init_yaw"], dtype=np.float64).ravel().tolist()
        a.end_yaw[:] = np.asarray(pr["end_yaw"], dtype=np.float64).ravel().tolist()
        a.inner_xy, a.inner_yaw = _dp(ixy), _dp(iyw)
        a.total_time = float(pr["total_time"])
        nxy, nyaw = a.n_inner_xy + 1, a.n_inner_yaw + 1
        sizes.append(dict(Nxy=nxy, Nyaw=nyaw, n=2 * (nxy - 1) + (nyaw - 1) + 1, S=nxy * (int(int_K) + 1)))
    return (arr, keep, sizes) if with_sizes else (arr, keep)


def _dp(a):
    return a.ctypes.data_as(_lib.DP)


class SE2Traj:
    """What MINCO_SE2::getTraj() yields (se2traj.hpp:682-695, 844-850): per piece a duration and a D x 6 coefficient
    matrix, highest order first."""

    def __init__(self, c_xy, c_yaw, T_xy, T_yaw):
        self.c_xy, self.c_yaw, self.T_xy, self.T_yaw = c_xy, c_yaw, T_xy, T_yaw
        nxy, nyaw = c_xy.shape[0] // 6, c_yaw.shape[0] // 6
        self.pos_durations = np.full(nxy, T_xy)
        self.yaw_durations = np.full(nyaw, T_yaw)
        self.pos_coeffs = c_xy.reshape(nxy, 6, 2).transpose(0, 2, 1)[:, :, ::-1].copy()      # (piece, dim, 6) descending powers
        self.yaw_coeffs = c_yaw.reshape(nyaw, 6, 1).transpose(0, 2, 1)[:, :, ::-1].copy()

    def getTotalDuration(self):
        return min(self.pos_durations.sum(), self.yaw_durations.sum())

    @staticmethod
    def _locate(durs, t):
        """PolyTrajectory::locatePieceIdx (se2traj.hpp:343-361)"""
        idx = 0
        while idx < len(durs) and t > durs[idx]:
            t -= durs[idx]
            idx += 1
        if idx == len(durs):
            idx -= 1
            t += durs[idx]
        return idx, t

    @staticmethod
    def _value(c_desc, t):
        """Piece::getValue (se2traj.hpp:106-116): coefficients highest order first, value += tn * coeff, tn *= t"""
        v, tn = 0.0, 1.0
        for i in range(5, -1, -1):
            v += tn * c_desc[i]
            tn *= t
        return v

    def getValue(self, t, yaw=False):
        durs = self.yaw_durations if yaw else self.pos_durations
        co = self.yaw_coeffs if yaw else self.pos_coeffs
        total = 0.0
        for d_ in durs:                                  # getTotalDuration's running sum (se2traj.hpp:290-299)
            total += d_
        idx, tl = self._locate(list(durs), total if t is None else t)
        return np.array([self._value(co[idx, d], tl) for d in range(co.shape[1])])

    def to_msg(self):
        """the mpc_controller/SE2Traj message PlanManager publishes (mpc_controller/msg/SE2Traj.msg:1-9, filled as at
        plan_manager.cpp:150-182): pos_pts / angle_pts = piece start points + the end point (geometry_msgs/Point: x, y, z),
        posT_pts / angleT_pts = piece durations, init_v = init_a = 0.  start_time is the caller's (ros::Time::now())."""
        nxy, nyaw = self.pos_durations.size, self.yaw_durations.size
        pos = np.zeros((nxy + 1, 3))
        for i in range(nxy):
            pos[i, 0], pos[i, 1] = self._value(self.pos_coeffs[i, 0], 0.0), self._value(self.pos_coeffs[i, 1], 0.0)
        pos[nxy, :2] = self.getValue(None)
        ang = np.zeros((nyaw + 1, 3))
        for i in range(nyaw):
            ang[i, 0] = self._value(self.yaw_coeffs[i, 0], 0.0)
        ang[nyaw, 0] = self.getValue(None, yaw=True)[0]
        return dict(pos_pts=pos, angle_pts=ang, init_v=np.zeros(3), init_a=np.zeros(3), posT_pts=self.pos_durations.copy(),
                    angleT_pts=self.yaw_durations.copy())

    def waypoints(self):
        """pos_pts (x, y) / angle_pts of the message"""
        m = self.to_msg()
        return m["pos_pts"][:, :2], m["angle_pts"][:, 0]


class ALMTrajOpt:
    def __init__(self, uneven_map=None, params=None):
        self.L = _lib.load()
        q = dict(HILL_OPT_PARAMS)
        if params:
            q.update(params)
        for k, v in q.items():            # public parameter members, like the reference
            setattr(self, k, v)
        self._pnames = list(q.keys())
        self.h = None
        self.uneven_map = None
        self.in_opt = False
        self._B = 0
        self._sizes = []
        self._last = []
        if uneven_map is not None:
            self.setEnvironment(uneven_map)

    def __del__(self):
        try:
            if self.h:
                self.L.uph_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def setEnvironment(self, env):
        """ALMTrajOpt::setEnvironment (alm_traj_opt.h:127-130); creates the device context with the current parameter members."""
        self.uneven_map = env
        if self.h:
            self.L.uph_ctx_destroy(self.h)
            self.h = None
        p = _lib.OptParams(**{k: (int(getattr(self, k)) if k in _INT_FIELDS else float(getattr(self, k))) for k in self._pnames})
        h = C.c_void_p()
        _lib.check(self.L.uph_ctx_create(env.h, C.byref(p), C.byref(h)), "uph_ctx_create")
        self.h = h

    def set_lanes(self, lanes):
        """64 / 128 / 256 lanes (one / two / four waves) per trajectory, 0 = automatic"""
        _lib.check(self.L.uph_ctx_set_lanes(self.h, int(lanes)), "uph_ctx_set_lanes")

    def set_xcd_locality(self, group):
        """experiment knob: per-XCD L2 locality of the launch order (uph_ctx_set_xcd_locality); takes effect at the next upload"""
        _lib.check(self.L.uph_ctx_set_xcd_locality(self.h, int(group)), "uph_ctx_set_xcd_locality")

    def set_wps(self, wps):
        _lib.check(self.L.uph_ctx_set_wps(self.h, int(wps)), "uph_ctx_set_wps")

    def set_sample_precision(self, bits):
        """32: fp32 arithmetic in the sample phase of the objective (configs[4] "fp32"; no 1e-9 parity with the double-only reference); 64: default"""
        _lib.check(self.L.uph_ctx_set_sample_precision(self.h, int(bits)), "uph_ctx_set_sample_precision")

    def set_rho(self, rho):
        _lib.check(self.L.uph_ctx_set_rho(self.h, float(rho)), "uph_ctx_set_rho")

    def get_rho(self):
        r = C.c_double(0)
        _lib.check(self.L.uph_ctx_get_rho(self.h, C.byref(r)), "uph_ctx_get_rho")
        return r.value

    # ---- batch plumbing ---------------------------------------------------------------------------------------------
    def _make_problems(self, probs):
        arr, keep, self._sizes = pack_problems(probs, int(self.int_K), with_sizes=True)
        return arr, keep

    def upload(self, probs):
        arr, keep = self._make_problems(probs)
        self._B = 0
        _lib.check(self.L.uph_batch_upload(self.h, len(probs), arr), "uph_batch_upload")
        self._B = len(probs)
        self._trace_cap_up = getattr(self, "_trace_cap", 0)

    def solve(self):
        """Kernel only (inputs already resident in HBM)."""
        _lib.check(self.L.uph_batch_solve(self.h), "uph_batch_solve")

    def solve_async(self):
        """enqueue the solve on this context's stream and return (uph_batch_solve_async); pair with wait()"""
        _lib.check(self.L.uph_batch_solve_async(self.h), "uph_batch_solve_async")

    def wait(self):
        _lib.check(self.L.uph_batch_wait(self.h), "uph_batch_wait")

    def stats(self):
        ms = C.c_double(0)
        v = [C.c_int64(0) for _ in range(4)]
        _lib.check(self.L.uph_batch_stats(self.h, C.byref(ms), *[C.byref(x) for x in v]), "uph_batch_stats")
        pm = C.c_double(0)
        _lib.check(self.L.uph_batch_prepare_ms(self.h, C.byref(pm)), "uph_batch_prepare_ms")
        return dict(kernel_ms=ms.value, prepare_ms=pm.value, evals=v[0].value, sample_evals=v[1].value, lbfgs_iters=v[2].value, hist_bytes=v[3].value)

    def cycles(self):
        """(B,16) shader-clock cycles per phase of the last solve (generate, samples, scatter, adjoint, two-loop, scaling, total, ...)"""
        out = np.zeros((max(0, self.L.uph_batch_count(self.h)), 16), dtype=np.int64)      # (sized from the context, not from this wrapper's bookkeeping)
        _lib.check(self.L.uph_batch_cycles(self.h, out.ctypes.data_as(C.POINTER(C.c_longlong))), "uph_batch_cycles")
        return out

    def origin(self):
        """caller's index of every problem this context holds (identity unless the batch came through optimize_batch_multi)"""
        idx = np.zeros(max(0, self.L.uph_batch_count(self.h)), dtype=np.int32)
        _lib.check(self.L.uph_batch_origin(self.h, idx.ctypes.data_as(C.POINTER(C.c_int32))), "uph_batch_origin")
        return idx

    def download(self, full=True):
        """full = False pulls only x and the coefficients (what a planner reads); the per-sample arrays then stay on the device"""
        res, bufs = self._result_array(self._sizes, full)
        _lib.check(self.L.uph_batch_download(self.h, res), "uph_batch_download")
        self._last = self._collect(res, bufs)
        return self._last

    def optimize_batch(self, probs):
        self.upload(probs)
        self.solve()
        return self.download()

    def _result_array(self, sizes, full):
        """uph_result array with caller-owned output arrays: full = every array, else what a planner pulls (x, coefficients)"""
        res = (_lib.Result * len(sizes))()
        bufs = []
        for i, sz in enumerate(sizes):
            b = dict(x=np.zeros(sz["n"]), c_xy=np.zeros((6 * sz["Nxy"], 2)), c_yaw=np.zeros(6 * sz["Nyaw"]))
            r = res[i]
            r.x_final, r.c_xy, r.c_yaw = _dp(b["x"]), _dp(b["c_xy"]), _dp(b["c_yaw"])
            if full:
                b.update(hx=np.zeros(sz["S"]), gx=np.zeros(6 * sz["S"]), lam=np.zeros(sz["S"]), mu=np.zeros(6 * sz["S"]), scale_cx=np.zeros(7 * sz["S"]))
                r.hx, r.gx, r.lambda_, r.mu, r.scale_cx = _dp(b["hx"]), _dp(b["gx"]), _dp(b["lam"]), _dp(b["mu"]), _dp(b["scale_cx"])
            bufs.append(b)
        return res, bufs

    @staticmethod
    def _collect(res, bufs):
        out = []
        for i, b in enumerate(bufs):
            r = res[i]
            b.update(ret=r.ret_code, alm_iters=r.alm_iters, lbfgs_iters=r.lbfgs_iters, evals=r.evals, last_lbfgs_ret=r.last_lbfgs_ret,
                     cost=r.cost, jerk_cost=r.jerk_cost, T_xy=r.piece_T_xy, T_yaw=r.piece_T_yaw, rho_final=r.rho_final, scale_fx=r.scale_fx)
            out.append(b)
        return out

    def prepare_boundary(self, probs, full=False):
        arr, keep = self._make_problems(probs)
        res, bufs = self._result_array(self._sizes, full)
        return arr, keep, res, bufs

    def optimize_boundary(self, probs, full=False, prepared=None):
        """ONE uph_optimize_batch call -- upload + solve + download, the reference's synchronous contract (alm_traj_opt.h:92-98, result
        pulled afterwards :165-168) -- with pageable host arrays.  prepared = prepare_boundary(probs) keeps the ctypes packing out of a
        timed region."""
        arr, keep, res, bufs = prepared if prepared is not None else self.prepare_boundary(probs, full)
        self._B = 0
        import time
        t0 = time.perf_counter()
        rc = self.L.uph_optimize_batch(self.h, len(probs), arr, res)
        self.last_boundary_s = time.perf_counter() - t0
        _lib.check(rc, "uph_optimize_batch")
        self._B = len(probs)
        self._last = self._collect(res, bufs)
        return self._last

    @staticmethod
    def optimize_batch_multi(opts, probs, full=False):
        """uph_optimize_batch_multi: one batch over the contexts `opts` (one per device), results in the caller's order"""
        lead = opts[0]
        arr, keep = lead._make_problems(probs)
        sizes = list(lead._sizes)
        res, bufs = lead._result_array(sizes, full)
        hs = (C.c_void_p * len(opts))(*[o.h for o in opts])
        for o in opts:                     # whatever happens below, no wrapper keeps describing a batch its context no longer holds
            o._B, o._sizes, o._last = 0, [], None
        _lib.check(lead.L.uph_optimize_batch_multi(hs, len(opts), len(probs), arr, res), "uph_optimize_batch_multi")
        out = ALMTrajOpt._collect(res, bufs)
        # every context now holds its share: bring the wrappers' bookkeeping in step with the C contexts (sizes in share order)
        for o in opts:
            n = o.L.uph_batch_count(o.h)
            if n <= 0:
                continue
            idx = o.origin()
            o._B, o._sizes, o._last = int(n), [sizes[i] for i in idx], [out[i] for i in idx]
        return out

    # ---- the reference's entry point -------------------------------------------------------------------------------
    def optimizeSE2Traj(self, initStateXY, endStateXY, innerPtsXY, initYaw, endYaw, innerPtsYaw, totalTime):
        """ALMTrajOpt::optimizeSE2Traj (alm_traj_opt.h:92-98, alm_traj_opt.cpp:168-278).  Returns 0 / 1 / 2."""
        self.in_opt = True
        try:
            out = self.optimize_batch([dict(init_xy=initStateXY, end_xy=endStateXY, inner_xy=innerPtsXY, init_yaw=initYaw, end_yaw=endYaw,
                                            inner_yaw=innerPtsYaw, total_time=totalTime)])
        finally:
            self.in_opt = False
        return out[0]["ret"]

    def getTraj(self, i=0):
        r = self._last[i]
        return SE2Traj(r["c_xy"], r["c_yaw"], r["T_xy"], r["T_yaw"])

    def getTrajJerkCost(self, i=0):
        return self._last[i]["jerk_cost"]

    def getMaxVxAxAyCurAttSig(self):
        """Batched post-solve report (alm_traj_opt.h:170-229 + getNonHolError): (B,7) max vx, ax, ay, cur, att, sigma, non-hol error."""
        out = np.zeros((max(0, self.L.uph_batch_count(self.h)), 7))      # rows = this context's problems (see origin() after optimize_batch_multi)
        _lib.check(self.L.uph_report_batch(self.h, _dp(out)), "uph_report_batch")
        return out

    # ---- test / bench hooks -----------------------------------------------------------------------------------------
    def x0_packed(self, probs):
        xs = []
        for pr in probs:
            T = float(pr["total_time"])
            tau = (np.sqrt(2.0 * T - 1.0) - 1.0) if T > 1.0 else (1.0 - np.sqrt(2.0 / T - 1.0))      # logC2, alm_traj_opt.h:239-242
            xs.append(np.concatenate([[tau], np.asarray(pr["inner_xy"], dtype=np.float64).T.ravel(), np.asarray(pr["inner_yaw"], dtype=np.float64)]))
        return xs

    def eval_batch(self, xs=None, repeat=1):
        """One innerCallback evaluation per uploaded trajectory at xs (list of arrays; None = resident x)."""
        n_tot = sum(s["n"] for s in self._sizes)
        xp = np.concatenate(xs) if xs is not None else None
        f, g = np.zeros(self._B), np.zeros(n_tot)
        _lib.check(self.L.uph_eval_batch(self.h, _dp(xp) if xp is not None else None, _dp(f), _dp(g), int(repeat)), "uph_eval_batch")
        gs, o = [], 0
        for s in self._sizes:
            gs.append(g[o:o + s["n"]].copy())
            o += s["n"]
        return f, gs

    def penalty_batch(self, repeat=1, store_residuals=True):
        """calConstrainCostGrad alone (alm_traj_opt.cpp:663-991) on the resident coefficients / durations / duals / scales: per trajectory
        (cost, gdCxy (6 Nxy, 2), gdCyaw (6 Nyaw,), sum gdTxy, sum gdTyaw)"""
        ncx, ncy = sum(12 * s["Nxy"] for s in self._sizes), sum(6 * s["Nyaw"] for s in self._sizes)
        cost, gx, gy, gt = np.zeros(self._B), np.zeros(ncx), np.zeros(ncy), np.zeros((self._B, 2))
        _lib.check(self.L.uph_penalty_batch(self.h, int(repeat), int(bool(store_residuals)), _dp(cost), _dp(gx), _dp(gy), _dp(gt)), "uph_penalty_batch")
        out, ox, oy = [], 0, 0
        for b, s in enumerate(self._sizes):
            out.append(dict(cost=cost[b], gdCxy=gx[ox:ox + 12 * s["Nxy"]].reshape(-1, 2).copy(), gdCyaw=gy[oy:oy + 6 * s["Nyaw"]].copy(), gdTxy_sum=gt[b, 0], gdTyaw_sum=gt[b, 1]))
            ox += 12 * s["Nxy"]; oy += 6 * s["Nyaw"]
        return out

    def init_scaling_batch(self):
        _lib.check(self.L.uph_init_scaling_batch(self.h), "uph_init_scaling_batch")

    def set_state(self, lam=None, mu=None, scale_cx=None, scale_fx=None, rho=None):
        cat = lambda v: np.ascontiguousarray(np.concatenate(v), dtype=np.float64) if v is not None else None
        a, b, c = cat(lam), cat(mu), cat(scale_cx)
        d = np.ascontiguousarray(scale_fx, dtype=np.float64) if scale_fx is not None else None
        e = np.ascontiguousarray(rho, dtype=np.float64) if rho is not None else None
        p = lambda v: _dp(v) if v is not None else None
        _lib.check(self.L.uph_batch_set_state(self.h, p(a), p(b), p(c), p(d), p(e)), "uph_batch_set_state")

    # teacher-forced late-state hooks (states as oracle.OracleALM.capture() returns them) ---------------------------------
    MAX_PAST = 8

    def set_x(self, xs):
        xp = np.ascontiguousarray(np.concatenate(xs), dtype=np.float64)
        _lib.check(self.L.uph_batch_set_x(self.h, _dp(xp)), "uph_batch_set_x")

    def alm_passes(self, max_passes=0):
        """the ALM loop from the resident x / duals / scales / rho (no reset, no initScaling), at most max_passes passes"""
        _lib.check(self.L.uph_batch_alm_passes(self.h, int(max_passes)), "uph_batch_alm_passes")

    def set_lbfgs_state(self, states):
        """states: one dict per uploaded trajectory with x, g, d, pf, lm_ys, lm_s (mem, n), lm_y, step, fx, k, end, bound"""
        cat = lambda key: np.ascontiguousarray(np.concatenate([np.asarray(st[key], dtype=np.float64).ravel() for st in states]))
        pf = np.zeros((self._B, self.MAX_PAST))
        for i, st in enumerate(states):
            pf[i, :len(st["pf"])] = st["pf"]
        scal = np.array([[st["step"], st["fx"], st["k"], st["end"], st["bound"]] for st in states], dtype=np.float64)
        self.set_x([st["x"] for st in states])
        g, d, ls, ly, ys = cat("g"), cat("d"), cat("lm_s"), cat("lm_y"), cat("lm_ys")
        _lib.check(self.L.uph_batch_set_lbfgs_state(self.h, _dp(g), _dp(d), _dp(pf), _dp(ls), _dp(ly), _dp(ys), _dp(scal)), "uph_batch_set_lbfgs_state")

    def lbfgs_resume(self, budget, finish_pass=False):
        _lib.check(self.L.uph_batch_lbfgs_resume(self.h, int(budget), int(bool(finish_pass))), "uph_batch_lbfgs_resume")

    def get_lbfgs_state(self):
        """list of state dicts (same keys as set_lbfgs_state + code, accepted, converged); x and the duals via download()"""
        mem = int(self.mem_size)
        n_tot = sum(s["n"] for s in self._sizes)
        g, d = np.zeros(n_tot), np.zeros(n_tot)
        pf = np.zeros((self._B, self.MAX_PAST))
        ls, ly, ys = np.zeros(mem * n_tot), np.zeros(mem * n_tot), np.zeros((self._B, mem))
        scal = np.zeros((self._B, 8))
        _lib.check(self.L.uph_batch_get_lbfgs_state(self.h, _dp(g), _dp(d), _dp(pf), _dp(ls), _dp(ly), _dp(ys), _dp(scal)), "uph_batch_get_lbfgs_state")
        x = self.download()
        out, o, oh = [], 0, 0
        for i, s in enumerate(self._sizes):
            n = s["n"]
            out.append(dict(x=x[i]["x"], g=g[o:o + n].copy(), d=d[o:o + n].copy(), pf=pf[i, :max(1, int(self.past))].copy(), lm_ys=ys[i].copy(),
                            lm_s=ls[oh:oh + mem * n].reshape(mem, n).copy(), lm_y=ly[oh:oh + mem * n].reshape(mem, n).copy(),
                            step=scal[i, 0], fx=scal[i, 1], k=int(scal[i, 2]), end=int(scal[i, 3]), bound=int(scal[i, 4]), code=int(scal[i, 5]),
                            accepted=int(scal[i, 6]), converged=int(scal[i, 7]), hx=x[i]["hx"], gx=x[i]["gx"], lam=x[i]["lam"], mu=x[i]["mu"],
                            rho=x[i]["rho_final"]))
            o += n
            oh += mem * n
        return out

    def set_trace(self, cap):
        _lib.check(self.L.uph_ctx_set_trace(self.h, int(cap)), "uph_ctx_set_trace")
        self._trace_cap = int(cap)

    def get_trace(self):
        out = np.zeros((self._B, self._trace_cap_up))
        _lib.check(self.L.uph_ctx_get_trace(self.h, _dp(out)), "uph_ctx_get_trace")
        return out
