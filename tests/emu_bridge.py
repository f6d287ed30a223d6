"""TEST SCAFFOLDING: ctypes bridge to tests/emu/libemu.so (the product's workgroup program compiled with a sequential
stand-in for the GPU workgroup).  Used only by the CPU test tier to debug logic before GPU time is spent."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = {}


def lib(flags=()):
    """flags: extra -D switches (a variant of the workgroup program, e.g. a wider scatter window); each set is its own shared object"""
    key = tuple(flags)
    if key not in _LIB:
        tag = "" if not key else "_" + "_".join(f.replace("-D", "").replace("=", "") for f in key)
        so = os.path.join(_HERE, "emu", "libemu%s.so" % tag)
        src = os.path.join(_HERE, "emu", "emu_solver.cpp")
        csrc = os.path.join(_HERE, "..", "uneven_planner_amd", "csrc")
        deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared"] + list(key) + ["-o", so, src])
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        L.emu_create.restype = C.c_void_p
        L.emu_create.argtypes = [dp, dp, dp]
        L.emu_destroy.argtypes = [C.c_void_p]
        L.emu_store_f32.argtypes = [C.c_void_p]
        L.emu_terrain.argtypes = [C.c_void_p, dp, C.c_int, dp, dp]
        L.emu_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [dp] * 14 + [C.POINTER(C.c_longlong), dp]
        L.emu_minco_op.argtypes = [C.c_int, dp]
        L.emu_set_hook.argtypes = [C.c_int, C.c_int, C.c_int] + [dp] * 7
        _LIB[key] = L
    return _LIB[key]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Emu:
    def __init__(self, cells, map_params_vec, opt_params_vec, flags=()):
        self.L = lib(flags)
        self.cells = np.ascontiguousarray(cells, dtype=np.float64)
        self.mp = np.ascontiguousarray(map_params_vec, dtype=np.float64)
        self.op = np.ascontiguousarray(opt_params_vec, dtype=np.float64)
        self.K = int(self.op[20])
        self.h = self.L.emu_create(_dp(self.mp), _dp(self.cells), _dp(self.op))

    def __del__(self):
        try:
            self.L.emu_destroy(self.h)
        except Exception:
            pass

    def store_f32(self):
        """switch the emulated grid to fp32 cell storage (GridDev::cells32)"""
        self.L.emu_store_f32(self.h)
        return self

    def terrain(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        v = np.zeros((pos.shape[0], 7))
        g = np.zeros((pos.shape[0], 7, 3))
        self.L.emu_terrain(self.h, _dp(pos), pos.shape[0], _dp(v), _dp(g))
        return v, g

    def alm_passes(self, prob, x, cap, **state):
        """mode 4: ALM passes from (x, duals, scales, rho) without reset / initScaling, at most `cap` passes"""
        z = np.zeros(8)
        self.L.emu_set_hook(int(cap), 0, 0, _dp(z), _dp(z), _dp(z), _dp(z), _dp(z), _dp(z), _dp(z))
        return self.run(4, prob, x, **state)

    def lbfgs_resume(self, prob, st, budget, finish=False, **state):
        """mode 5: continue the L-BFGS loop from state dict `st` (oracle.OracleALM.capture() layout); returns (result dict, new state)"""
        mem = self.op[18].astype(int) if hasattr(self.op[18], "astype") else int(self.op[18])
        h = dict(g=np.ascontiguousarray(st["g"], dtype=np.float64).copy(), d=np.ascontiguousarray(st["d"], dtype=np.float64).copy(),
                 pf=np.zeros(8), lm_s=np.ascontiguousarray(st["lm_s"], dtype=np.float64).copy(), lm_y=np.ascontiguousarray(st["lm_y"], dtype=np.float64).copy(),
                 lm_ys=np.ascontiguousarray(st["lm_ys"], dtype=np.float64).copy(), scal=np.zeros(8))
        h["pf"][:len(st["pf"])] = st["pf"]
        h["scal"][:5] = [st["step"], st["fx"], st["k"], st["end"], st["bound"]]
        self._hook = h
        self.L.emu_set_hook(0, int(budget), int(bool(finish)), _dp(h["g"]), _dp(h["d"]), _dp(h["pf"]), _dp(h["lm_s"]), _dp(h["lm_y"]), _dp(h["lm_ys"]), _dp(h["scal"]))
        r = self.run(5, prob, st["x"], **state)
        sc = h["scal"]
        new = dict(x=r["x"], g=h["g"], d=h["d"], pf=h["pf"][:max(1, len(st["pf"]))].copy(), lm_s=h["lm_s"], lm_y=h["lm_y"], lm_ys=h["lm_ys"], step=sc[0], fx=sc[1],
                   k=int(sc[2]), end=int(sc[3]), bound=int(sc[4]), code=int(sc[5]), accepted=int(sc[6]), converged=int(sc[7]),
                   hx=r["hx"], gx=r["gx"], lam=r["lam"], mu=r["mu"], rho=r["rho"])
        return r, new

    def run(self, mode, prob, x, lam=None, mu=None, scale_cx=None, rho=1.0, scale_fx=1.0):
        nxy, nyaw = prob["inner_xy"].shape[1], prob["inner_yaw"].shape[0]
        S = (nxy + 1) * (self.K + 1)
        n = 2 * nxy + nyaw + 1
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        g = np.zeros(n)
        lam = np.zeros(S) if lam is None else np.ascontiguousarray(lam, dtype=np.float64).copy()
        mu = np.zeros(6 * S) if mu is None else np.ascontiguousarray(mu, dtype=np.float64).copy()
        sc = np.ones(7 * S) if scale_cx is None else np.ascontiguousarray(scale_cx, dtype=np.float64).copy()
        hx, gx = np.zeros(S), np.zeros(6 * S)
        cxy, cyaw = np.zeros((6 * (nxy + 1), 2)), np.zeros(6 * (nyaw + 1))
        scal = np.zeros(8)
        scal[0], scal[1] = rho, scale_fx
        ist = np.zeros(6, dtype=np.int64)
        rep = np.zeros(7)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        ixy, exy = f64(prob["init_xy"].T), f64(prob["end_xy"].T)
        iy, ey = f64(prob["init_yaw"]), f64(prob["end_yaw"])
        self.L.emu_run(self.h, mode, nxy, nyaw, _dp(ixy), _dp(exy), _dp(iy), _dp(ey), _dp(x), _dp(g), _dp(lam), _dp(mu),
                       _dp(sc), _dp(hx), _dp(gx), _dp(cxy), _dp(cyaw), _dp(scal),
                       ist.ctypes.data_as(C.POINTER(C.c_longlong)), _dp(rep))
        if mode == 5:
            g = self._hook["g"]
        return dict(x=x, g=g, lam=lam, mu=mu, scale_cx=sc, hx=hx, gx=gx, c_xy=cxy, c_yaw=cyaw, rho=scal[0],
                    scale_fx=scal[1], f=scal[2], jerk_cost=scal[3], T_xy=scal[4], T_yaw=scal[5], ret=int(ist[0]),
                    alm_iters=int(ist[1]), lbfgs_iters=int(ist[2]), evals=int(ist[3]), last_lbfgs_ret=int(ist[4]),
                    hist_reads=int(ist[5]), report=rep)
