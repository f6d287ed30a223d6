"""ctypes binding of libunevenhip.so (include/uneven_hip.h).  There is no CPU fall-back: if the HIP library is
missing or no gfx950 device is visible the import / first call fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNEVENHIP_LIB selects another build of the same library (tools/build_variants.sh: A/B measurements of kernel variants)
LIB_PATH = os.environ.get("UNEVENHIP_LIB") or os.path.join(_HERE, "libunevenhip.so")


class MapParams(C.Structure):
    _fields_ = [("iter_num", C.c_int32), ("map_size_x", C.c_double), ("map_size_y", C.c_double),
                ("ellipsoid_x", C.c_double), ("ellipsoid_y", C.c_double), ("ellipsoid_z", C.c_double),
                ("xy_resolution", C.c_double), ("yaw_resolution", C.c_double), ("min_cnormal", C.c_double),
                ("max_rho", C.c_double), ("gravity", C.c_double)]


class OptParams(C.Structure):
    _fields_ = [("rho_T", C.c_double), ("rho_ter", C.c_double), ("max_vel", C.c_double), ("max_acc_lon", C.c_double),
                ("max_acc_lat", C.c_double), ("max_kap", C.c_double), ("min_cxi", C.c_double), ("max_sig", C.c_double),
                ("use_scaling", C.c_int32), ("rho", C.c_double), ("beta", C.c_double), ("gamma", C.c_double),
                ("epsilon_con", C.c_double), ("max_iter", C.c_double), ("g_epsilon", C.c_double),
                ("min_step", C.c_double), ("inner_max_iter", C.c_double), ("delta", C.c_double),
                ("mem_size", C.c_int32), ("past", C.c_int32), ("int_K", C.c_int32)]


class ManagerParams(C.Structure):
    _fields_ = [("piece_len", C.c_double), ("mean_vel", C.c_double), ("init_time_times", C.c_double), ("yaw_piece_times", C.c_double),
                ("init_sig_vel", C.c_double), ("test_mode", C.c_int32), ("test_max_vel", C.c_double)]


class FbmParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("hurst", C.c_double), ("lambda_min", C.c_double), ("lambda_max", C.c_double),
                ("amplitude", C.c_double), ("max_slope_deg", C.c_double), ("n_waves", C.c_int32), ("rough_amp", C.c_double),
                ("rough_lambda", C.c_double), ("patch_lambda", C.c_double), ("rough_threshold", C.c_double)]


class KinoParams(C.Structure):
    _fields_ = [("yaw_resolution", C.c_double), ("lambda_heu", C.c_double), ("weight_r2", C.c_double), ("weight_so2", C.c_double),
                ("weight_v_change", C.c_double), ("weight_delta_change", C.c_double), ("weight_sigma", C.c_double), ("time_interval", C.c_double),
                ("collision_interval", C.c_double), ("oneshot_range", C.c_double), ("wheel_base", C.c_double), ("max_steer", C.c_double),
                ("max_vel", C.c_double)]


FBM_MAX_WAVES = 48
FBM_TABLE_DOUBLES = 4 * FBM_MAX_WAVES + 12 + 9
DP = C.POINTER(C.c_double)


class Problem(C.Structure):
    _fields_ = [("n_inner_xy", C.c_int32), ("n_inner_yaw", C.c_int32), ("init_xy", C.c_double * 6),
                ("end_xy", C.c_double * 6), ("init_yaw", C.c_double * 3), ("end_yaw", C.c_double * 3),
                ("inner_xy", DP), ("inner_yaw", DP), ("total_time", C.c_double)]


class Result(C.Structure):
    _fields_ = [("ret_code", C.c_int32), ("alm_iters", C.c_int32), ("lbfgs_iters", C.c_int32), ("evals", C.c_int32),
                ("last_lbfgs_ret", C.c_int32), ("cost", C.c_double), ("jerk_cost", C.c_double),
                ("piece_T_xy", C.c_double), ("piece_T_yaw", C.c_double), ("rho_final", C.c_double),
                ("scale_fx", C.c_double), ("x_final", DP), ("c_xy", DP), ("c_yaw", DP), ("hx", DP), ("gx", DP),
                ("lambda_", DP), ("mu", DP), ("scale_cx", DP)]


# every symbol include/uneven_hip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_I32, _I64 = C.c_int32, C.c_int64
SYMBOLS = {
    "uph_last_error": (C.c_char_p, []),
    "uph_device_count": (C.c_int, []),
    "uph_version": (C.c_char_p, []),
    "uph_resample_batch": (C.c_int, [C.POINTER(ManagerParams), _I32, DP, C.POINTER(_I64), _I32, _I32, DP, DP, DP, DP, DP, DP, C.POINTER(_I32), C.POINTER(_I32), DP, DP]),
    "uph_map_create": (C.c_int, [C.POINTER(MapParams), C.c_int, C.POINTER(_VP)]),
    "uph_map_create_f32": (C.c_int, [C.POINTER(MapParams), C.c_int, C.POINTER(_VP)]),
    "uph_map_storage_bytes": (C.c_int, [_VP]),
    "uph_map_create_tile": (C.c_int, [C.POINTER(MapParams), C.c_int, _I32, _I32, _I32, C.POINTER(_VP)]),
    "uph_map_tile": (C.c_int, [_VP, C.POINTER(_I32), C.POINTER(_I32)]),
    "uph_map_fill_fbm": (C.c_int, [_VP, C.POINTER(FbmParams), _I32, _I32]),
    "uph_fbm_table": (C.c_int, [C.POINTER(FbmParams), DP]),
    "uph_map_get_window": (C.c_int, [_VP, _I32, _I32, _I32, _I32, DP]),
    "uph_map_save_csv": (C.c_int, [C.c_char_p, DP, C.POINTER(_I32)]),
    "uph_map_load_csv": (C.c_int, [C.c_char_p, C.POINTER(_I32), DP, DP, C.POINTER(_I64)]),
    "uph_map_save_bin": (C.c_int, [C.c_char_p, DP, C.POINTER(_I32)]),
    "uph_map_load_bin": (C.c_int, [C.c_char_p, C.POINTER(_I32), DP]),
    "uph_map_save_cache": (C.c_int, [_VP, C.c_char_p, C.c_char_p]),
    "uph_map_load_cache": (C.c_int, [_VP, C.c_char_p, C.c_char_p, C.POINTER(_I32)]),
    "uph_map_destroy": (None, [_VP]),
    "uph_map_dims": (C.c_int, [_VP, C.POINTER(_I32)]),
    "uph_map_set_cells": (C.c_int, [_VP, DP]),
    "uph_map_get_cells": (C.c_int, [_VP, DP, DP, C.c_char_p, C.c_char_p]),
    "uph_map_build": (C.c_int, [_VP, C.POINTER(C.c_float), _I64, _I32, _I32]),
    "uph_map_build_multi": (C.c_int, [C.POINTER(_VP), _I32, C.POINTER(C.c_float), _I64]),
    "uph_map_fill_fbm_multi": (C.c_int, [C.POINTER(_VP), _I32, C.POINTER(FbmParams)]),
    "uph_multi_slab_plan": (C.c_int, [_I32, _I32, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "uph_multi_batch_plan": (C.c_int, [_I32, _I32, C.POINTER(Problem), C.POINTER(_I32), DP]),
    "uph_map_multi_stats": (C.c_int, [_VP, DP, DP, DP, DP, C.POINTER(_I32)]),
    "uph_multi_shutdown": (None, []),
    "uph_rccl_selftest": (C.c_int, [_I32, C.c_char_p, _I32]),
    "uph_optimize_batch_multi": (C.c_int, [C.POINTER(_VP), _I32, _I32, C.POINTER(Problem), C.POINTER(Result)]),
    "uph_batch_count": (C.c_int, [_VP]),
    "uph_batch_origin": (C.c_int, [_VP, C.POINTER(_I32)]),
    "uph_map_cells_device": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_I64)]),
    "uph_map_commit": (C.c_int, [_VP]),
    "uph_map_export_slab_dev": (C.c_int, [_VP, _I32, _I32, _VP]),
    "uph_map_import_cells_dev": (C.c_int, [_VP, _VP]),
    "uph_terrain_query": (C.c_int, [_VP, DP, _I32, DP, DP]),
    "uph_frontend_query": (C.c_int, [_VP, DP, _I32, DP, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "uph_terrain_pose_query": (C.c_int, [_VP, DP, _I32, DP]),
    "uph_frontend_query_ms": (C.c_int, [_VP, DP]),
    "uph_map_filter_cloud": (_I64, [C.POINTER(C.c_float), _I64, C.POINTER(C.c_float), _I64]),
    "uph_map_build_stats": (C.c_int, [_VP, DP, C.POINTER(_I64), C.POINTER(_I64)]),
    "uph_map_build_stages": (C.c_int, [_VP, DP]),
    "uph_map_built_cloud": (_I64, [_VP, C.POINTER(C.c_float), _I64]),
    "uph_ctx_create": (C.c_int, [_VP, C.POINTER(OptParams), C.POINTER(_VP)]),
    "uph_ctx_destroy": (None, [_VP]),
    "uph_ctx_set_lanes": (C.c_int, [_VP, _I32]),
    "uph_ctx_set_wps": (C.c_int, [_VP, _I32]),
    "uph_ctx_set_xcd_locality": (C.c_int, [_VP, _I32]),
    "uph_ctx_set_sample_precision": (C.c_int, [_VP, _I32]),
    "uph_ctx_set_rho": (C.c_int, [_VP, C.c_double]),
    "uph_ctx_get_rho": (C.c_int, [_VP, DP]),
    "uph_ctx_set_trace": (C.c_int, [_VP, _I32]),
    "uph_ctx_get_trace": (C.c_int, [_VP, DP]),
    "uph_optimize_batch": (C.c_int, [_VP, _I32, C.POINTER(Problem), C.POINTER(Result)]),
    "uph_batch_upload": (C.c_int, [_VP, _I32, C.POINTER(Problem)]),
    "uph_batch_solve": (C.c_int, [_VP]),
    "uph_batch_solve_async": (C.c_int, [_VP]),
    "uph_batch_wait": (C.c_int, [_VP]),
    "uph_batch_download": (C.c_int, [_VP, C.POINTER(Result)]),
    "uph_batch_stats": (C.c_int, [_VP, DP, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    "uph_batch_prepare_ms": (C.c_int, [_VP, DP]),
    "uph_batch_cycles": (C.c_int, [_VP, C.POINTER(C.c_longlong)]),
    "uph_build_id": (C.c_char_p, []),
    "uph_eval_batch": (C.c_int, [_VP, DP, DP, DP, _I32]),
    "uph_penalty_batch": (C.c_int, [_VP, _I32, _I32, DP, DP, DP, DP]),
    "uph_init_scaling_batch": (C.c_int, [_VP]),
    "uph_microbench_batch": (C.c_int, [_VP, _I32]),
    "uph_batch_set_state": (C.c_int, [_VP, DP, DP, DP, DP, DP]),
    "uph_report_batch": (C.c_int, [_VP, DP]),
    "uph_batch_set_x": (C.c_int, [_VP, DP]),
    "uph_batch_alm_passes": (C.c_int, [_VP, _I32]),
    "uph_batch_set_lbfgs_state": (C.c_int, [_VP, DP, DP, DP, DP, DP, DP, DP]),
    "uph_batch_lbfgs_resume": (C.c_int, [_VP, _I32, _I32]),
    "uph_batch_get_lbfgs_state": (C.c_int, [_VP, DP, DP, DP, DP, DP, DP, DP]),
    "uph_kino_create": (C.c_int, [_VP, C.POINTER(KinoParams), _I32, C.POINTER(_VP)]),
    "uph_kino_destroy": (None, [_VP]),
    "uph_kino_slots": (C.c_int, [_VP]),
    "uph_kino_set_wps": (C.c_int, [_VP, _I32]),
    "uph_kino_set_flags": (C.c_int, [_VP, _I32]),
    "uph_kino_primitives": (C.c_int, [_VP]),
    "uph_kino_plan_batch": (C.c_int, [_VP, _I32, DP, DP, _I32, DP, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32), _I32, _I32, C.POINTER(_I32)]),
    "uph_kino_stats": (C.c_int, [_VP, DP]),
}

_LIB = None


UPH_ERR_NO_CACHE = -5      # include/uneven_hip.h: uph_map_load_cache found no readable cache (build the map)


class UnevenHipError(RuntimeError):
    pass


def load():
    """Load libunevenhip.so and bind every exported symbol.  Raises if the library has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise UnevenHipError("libunevenhip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                                 "there is no CPU fall-back")
        # PyTorch-ROCm wheels bundle their own copy of the HIP runtime (same soname, different file).  A process must run on ONE
        # runtime: if libunevenhip.so pulls in /opt/rocm's first, torch later loads its bundled one next to it and finds "no HIP
        # GPUs".  Loading torch first makes both resolve to the same library.  (Only where torch is installed; the C++ adapter has no
        # such concern.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        variant = bool(os.environ.get("UNEVENHIP_LIB"))
        for name, (res, args) in SYMBOLS.items():
            if variant and not hasattr(L, name):
                continue                # an A/B build of an earlier source state (tools/build_variants.sh) may predate a symbol; calling it raises
            fn = getattr(L, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def sources_id():
    """the hash csrc/Makefile compiles into the library (uph_build_id), computed over the tree: csrc/*.hip, *.hpp, *.cpp sorted by name, then include/uneven_hip.h"""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".cpp")))
    h = hashlib.sha1()
    for f in files:
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(here, "..", "include", "uneven_hip.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def build_id():
    L = load()
    return L.uph_build_id().decode() if hasattr(L, "uph_build_id") else None


def check(rc, what=""):
    if rc != 0:
        msg = load().uph_last_error()
        raise UnevenHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))


def require_device():
    n = load().uph_device_count()
    if n <= 0:
        raise UnevenHipError("no HIP device visible: the uneven_planner_amd back-end needs an MI355X (gfx950)")
    return n
