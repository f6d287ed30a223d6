"""GPU tier: the single-process multi-GPU entries of the C-ABI (uph_map_build_multi, uph_map_fill_fbm_multi, uph_optimize_batch_multi) and the
`bench.py --gpus N` launcher.  With G >= 2 visible devices the maps / contexts sit on distinct devices and the slab exchange is an RCCL
all-gather inside the library; on a one-GPU box the same entries run as a "world" of several maps / contexts on device 0 (the exchange is then
device-to-device copies -- RCCL refuses a device twice) and the RCCL binding itself is exercised by uph_rccl_selftest on a clique of one.
Either way the result must equal the single-device result bit for bit: sharding may not change a cell or a trajectory."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL_MAP = dict(map_size_x=3.0, map_size_y=3.0)       # 60 x 60 x 64 cells


def _ndev():
    import uneven_planner_amd as U
    return U._lib.load().uph_device_count()


def test_rccl_binds_and_gathers_in_this_process():
    import uneven_planner_amd as U
    L = U._lib.load()
    n = min(2, L.uph_device_count())
    info = C.create_string_buffer(256)
    U._lib.check(L.uph_rccl_selftest(n, info, 256), "uph_rccl_selftest")
    assert b"rccl" in info.value.lower()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_map_build_multi_equals_single_build(world):
    """60 rows over 2 / 3 / 4 slabs (3 does... 60 / 4 = 15, 60 / 3 = 20; a 7-slab world with a ragged last slab is covered below)"""
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    xyz = scenes.make_hill_cloud(n_side=150, half=2.0)
    one = U.UnevenMap(SMALL_MAP)
    one.build(xyz)
    G = _ndev()
    maps = [U.UnevenMap(SMALL_MAP, device=(g % G) if G >= world else 0) for g in range(world)]
    st = U.UnevenMap.build_multi(maps, xyz)
    assert st["via_rccl"] == (G >= world)
    for m in maps:
        assert np.array_equal(m.map_buffer, one.map_buffer) and np.array_equal(m.occ_buffer, one.occ_buffer) and np.array_equal(m.occ_r2_buffer, one.occ_r2_buffer)
        assert np.array_equal(m.c_buffer, one.c_buffer)


def test_map_build_multi_ragged_slabs_and_rebuild():
    """nx = 60 over 7 maps: per = 9, the last slab holds 6 rows (the RCCL form pads it; the copy form just moves fewer rows); building twice
    into the same maps gives the same grid (every build starts from fresh cells)"""
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    xyz = scenes.make_hill_cloud(n_side=150, half=2.0)
    one = U.UnevenMap(SMALL_MAP)
    one.build(xyz)
    G = _ndev()
    maps = [U.UnevenMap(SMALL_MAP, device=g if G >= 7 else 0) for g in range(7)]
    U.UnevenMap.build_multi(maps, xyz)
    U.UnevenMap.build_multi(maps, xyz)
    for m in maps:
        assert np.array_equal(m.map_buffer, one.map_buffer)


def test_map_build_multi_on_distinct_devices_uses_rccl():
    """the real thing where the box has it: n_gpus = min(2, devices) maps on distinct devices.  One device: n_gpus = 1 must equal uph_map_build"""
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    xyz = scenes.make_hill_cloud(n_side=150, half=2.0)
    n = min(2, _ndev())
    one = U.UnevenMap(SMALL_MAP)
    one.build(xyz)
    maps = [U.UnevenMap(SMALL_MAP, device=g) for g in range(n)]
    st = U.UnevenMap.build_multi(maps, xyz)
    assert st["via_rccl"] == (n > 1)
    for m in maps:
        assert np.array_equal(m.map_buffer, one.map_buffer)


def test_fill_fbm_multi_float_slabs():
    import uneven_planner_amd as U
    kp = dict(map_size_x=40.0, map_size_y=40.0, xy_resolution=0.25)
    one = U.UnevenMap(kp, storage="f32")
    one.fill_fbm()
    G = _ndev()
    maps = [U.UnevenMap(kp, device=g if G >= 3 else 0, storage="f32") for g in range(3)]
    U.UnevenMap.fill_fbm_multi(maps)
    for m in maps:
        assert np.array_equal(m.map_buffer, one.map_buffer) and np.array_equal(m.occ_r2_buffer, one.occ_r2_buffer)


def test_multi_entries_reject_mismatched_maps():
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    xyz = scenes.make_hill_cloud(n_side=60, half=1.0)
    a, b = U.UnevenMap(SMALL_MAP), U.UnevenMap(dict(map_size_x=2.0, map_size_y=2.0))
    with pytest.raises(U._lib.UnevenHipError, match="differ"):
        U.UnevenMap.build_multi([a, b], xyz)
    with pytest.raises(U._lib.UnevenHipError, match="twice"):
        U.UnevenMap.build_multi([a, a], xyz)


def test_optimize_batch_multi_equals_single_context(analytic_cells):
    """one batch over three contexts (distinct devices where available): every problem's result equals the single-context solve bit for bit, in
    the caller's order; an out-of-limit problem comes back UNSUPPORTED without disturbing the others"""
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    G = _ndev()
    world = 3
    probs = scenes.random_problems(41, seed0=4100, dmin=3.0, dmax=6.0)
    long_ = dict(probs[5])
    long_["inner_xy"] = np.zeros((2, 140)); long_["inner_yaw"] = np.zeros(300)       # 141 position pieces > UPH_MAX_PIECE_XY
    probs.insert(17, long_)
    maps = []
    for g in range(world):
        m = U.UnevenMap(device=g if G >= world else 0)
        m.set_cells(analytic_cells)
        maps.append(m)
    single = U.ALMTrajOpt(maps[0]); single.set_lanes(256); single.set_rho(1.0)
    ref = single.optimize_batch(probs)
    opts = []
    for m in maps:
        o = U.ALMTrajOpt(m); o.set_lanes(256); o.set_rho(1.0)
        opts.append(o)
    out = U.ALMTrajOpt.optimize_batch_multi(opts, probs)
    assert out[17]["ret"] == 4 and ref[17]["ret"] == 4
    for i, (a, b) in enumerate(zip(ref, out)):
        assert a["ret"] == b["ret"], i
        if a["ret"] != 4:
            assert a["cost"] == b["cost"] and a["evals"] == b["evals"] and np.array_equal(a["x"], b["x"]) and np.array_equal(a["c_xy"], b["c_xy"]), i
    # fewer problems than contexts, and a batch of one (rho persists on the context that solved it: Q7)
    two = U.ALMTrajOpt.optimize_batch_multi(opts, probs[:2])
    assert [r["ret"] for r in two] == [r["ret"] for r in ref[:2]] and np.array_equal(two[1]["x"], ref[1]["x"])


def test_download_copies_only_what_was_asked(analytic_cells, small_problems):
    import uneven_planner_amd as U
    m = U.UnevenMap(); m.set_cells(analytic_cells)
    opt = U.ALMTrajOpt(m); opt.set_rho(1.0)
    opt.upload(small_problems); opt.solve()
    full = opt.download()
    slim = opt.download(full=False)
    for a, b in zip(full, slim):
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["c_xy"], b["c_xy"]) and np.array_equal(a["c_yaw"], b["c_yaw"]) and a["cost"] == b["cost"]
        assert "hx" not in b and np.abs(a["scale_cx"]).max() > 0
    # the boundary call (upload + solve + download in one uph_optimize_batch) gives the same trajectories
    opt.set_rho(1.0)
    bd = opt.optimize_boundary(small_problems)
    assert all(np.array_equal(a["x"], b["x"]) for a, b in zip(full, bd)) and opt.last_boundary_s > 0


def _bench(*flags, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    if env:
        e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), capture_output=True, text=True, env=e, timeout=900)


def test_bench_gpus_flag_is_live():
    """`python bench.py --gpus 2` without a launcher starts two ranks: on a box with >= 2 GPUs the line says n_gpus 2 (RCCL world 2), on a
    one-GPU box it must fail loudly instead of silently reporting a one-GPU number"""
    r = _bench("--gpus", "2", "--batch", "512", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-extras")
    if _ndev() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["config"]["rccl_world"] == 2 and len(line["per_rank_ms_per_step"]) == 2
    else:
        assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


def test_bench_single_process_mode():
    n = min(2, _ndev())
    r = _bench("--gpus", str(n), "--single-process", "--batch", "512", "--steps", "1", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == n and line["value"] > 0 and len(line["per_gpu_kernel_ms"]) == n
