"""Host-side mirror of the reference's KinoAstar (front_end/include/front_end/kino_astar.h:99-168) backed by the batched device search of
libunevenhip.so (csrc/kino_search.hip): same member names and argument meaning, Eigen vectors become numpy arrays.

`plan(start_state, end_state)` is KinoAstar::plan (front_end/src/kino_astar.cpp:67-236) for one query; `plan_batch` runs many queries at
once, one wave64 each.  There is no CPU fall-back."""
import ctypes as C

import numpy as np

from . import _lib

# plan_manager/params/run_hill.yaml:16-30
HILL_KINO_PARAMS = dict(yaw_resolution=3.15, lambda_heu=1.0, weight_r2=1.0, weight_so2=0.5, weight_v_change=0.0, weight_delta_change=0.0,
                        weight_sigma=10.0, time_interval=0.3, collision_interval=0.06, oneshot_range=1.0, wheel_base=0.26, max_steer=0.5,
                        max_vel=0.5)
STATUS = {0: "ok", 1: "start is not free", 2: "goal is not free", 3: "Kino Astar Failed, No path", 4: "run out of memory", 5: "expansion cap (test hook)", 6: "internal"}


def _dp(a):
    return a.ctypes.data_as(_lib.DP)


class KinoAstar:
    def __init__(self, uneven_map=None, params=None, slots=0):
        """init(nh) of the reference reads rosparam kino_astar/...; here `params` overrides run_hill.yaml's values"""
        self.L = _lib.load()
        q = dict(HILL_KINO_PARAMS)
        if params:
            q.update(params)
        for k, v in q.items():
            setattr(self, k, float(v))
        self._slots = int(slots)
        self.h = None
        self.uneven_map = None
        self.front_end_path = np.zeros((0, 3))
        self.last = None
        if uneven_map is not None:
            self.setEnvironment(uneven_map)

    def setEnvironment(self, env):
        """kino_astar.h:170-178: binds the map and allocates the node pools (getXYNum nodes per concurrent query)"""
        self.close()
        self.uneven_map = env
        kp = _lib.KinoParams(**{k: getattr(self, k) for k in HILL_KINO_PARAMS})
        h = C.c_void_p()
        _lib.check(self.L.uph_kino_create(env.h, C.byref(kp), self._slots, C.byref(h)), "uph_kino_create")
        self.h = h
        self.slots = self.L.uph_kino_slots(self.h)
        self.n_primitives = self.L.uph_kino_primitives(self.h)

    def set_wps(self, wps):
        """experiment knob: waves per SIMD of the search kernel (2, 4, 6, 8)"""
        _lib.check(self.L.uph_kino_set_wps(self.h, int(wps)), "uph_kino_set_wps")

    def set_flags(self, flags):
        """experiment knob: bit 0 dynamic query hand-out, bit 1 sincosFast (both on by default)"""
        _lib.check(self.L.uph_kino_set_flags(self.h, int(flags)), "uph_kino_set_flags")

    def close(self):
        if getattr(self, "h", None):
            self.L.uph_kino_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def plan_batch(self, starts, goals, path_cap=1024, max_expand=0, exp_cap=0, complete=True):
        """B queries at once: list of dict(status, path (n,3), n_path, iter_num, use_node_num[, expanded (k,3)]).  complete: a front_end_path longer than
        path_cap is searched again with room for all its poses (the reference returns the whole path); False leaves it clipped at path_cap poses with
        n_path telling its real length (throughput measurements that bound the download)"""
        s = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
        g = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 3)
        B = s.shape[0]
        paths = np.zeros((B, max(1, path_cap), 3))
        npth, st, it, un = (np.zeros(B, dtype=np.int32) for _ in range(4))
        exp = np.zeros((B, max(1, exp_cap), 3), dtype=np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        _lib.check(self.L.uph_kino_plan_batch(self.h, B, _dp(s), _dp(g), int(path_cap), _dp(paths), ip(npth), ip(st), ip(it), ip(un), int(max_expand),
                                              int(exp_cap), ip(exp) if exp_cap > 0 else None), "uph_kino_plan_batch")
        out = []
        for b in range(B):
            r = dict(status=int(st[b]), n_path=int(npth[b]), path=paths[b, :min(int(npth[b]), path_cap)].copy(), iter_num=int(it[b]), use_node_num=int(un[b]))
            if exp_cap > 0:
                r["expanded"] = exp[b, :min(int(it[b]), exp_cap)].copy()
            out.append(r)
        # a front_end_path longer than path_cap was clipped by the library (the node chain comes first, the Dubins shot that reaches the goal
        # last): the reference returns the whole path, so those queries are searched again with room for all their poses
        longer = [b for b in range(B) if st[b] == 0 and npth[b] > path_cap] if (complete and path_cap > 1 and max_expand == 0) else []
        if longer:
            again = self.plan_batch(s[longer], g[longer], path_cap=int(npth[longer].max()), max_expand=0, exp_cap=0)
            for b, r2 in zip(longer, again):
                out[b]["path"], out[b]["n_path"] = r2["path"], r2["n_path"]
        self.last = out
        return out

    def plan(self, start_state, end_state):
        """KinoAstar::plan: the front-end path as an (n, 3) array of poses, empty when the reference returns an empty vector"""
        r = self.plan_batch([start_state], [end_state])[0]
        self.front_end_path = r["path"] if r["status"] == 0 else np.zeros((0, 3))
        return self.front_end_path

    def stats(self):
        ms = C.c_double(0)
        _lib.check(self.L.uph_kino_stats(self.h, C.byref(ms)), "uph_kino_stats")
        return dict(kernel_ms=ms.value)
