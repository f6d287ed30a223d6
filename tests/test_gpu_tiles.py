"""a grid that does not fit one GPU is held as x-slab tiles with a halo, problems are routed on the host to the
tile that owns them.  One GPU is visible here, so the test plays a world of four on one device: every tile is filled separately
(analytic terrain, fp32 cells), must hold exactly the cells of the whole grid, and its routed problems must solve bit for bit as they
do on the whole grid -- the tile uses the whole grid's index arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PARAMS = dict(map_size_x=320.0, map_size_y=320.0, xy_resolution=0.25)        # 1280 x 1280 x 64 cells: host copies are occupancy only
WORLD, HALO_M = 4, 20.0


def test_tiles_hold_the_whole_grids_cells_and_solve_identically():
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    from uneven_planner_amd.uneven_map import route_problems, slab_bounds, tile_rows
    full = U.UnevenMap(PARAMS, storage="f32").fill_fbm()
    nx, ny, nyaw = (int(v) for v in full.voxel_num)
    res, ox = full.xy_resolution, float(full.map_origin[0])
    probs = scenes.local_problems(96, seed0=5000, half=150.0, occ_r2=full.occ_r2_buffer, grid=(nx, ny, res, ox, float(full.map_origin[1])))
    opt = U.ALMTrajOpt(full)
    opt.set_rho(1.0)
    ref = opt.optimize_batch(probs)
    routes = route_problems(probs, nx, WORLD, res, ox)
    assert sorted(i for r in routes for i in r) == list(range(len(probs))) and all(len(r) > 0 for r in routes)
    halo = int(round(HALO_M / res))
    for rank in range(WORLD):
        x0, x1 = tile_rows(nx, rank, WORLD, halo)
        _, a, b = slab_bounds(nx, rank, WORLD)
        assert x0 <= a and b <= x1 and (x1 - x0) < nx
        t = U.UnevenMap(PARAMS, storage="f32", tile=(x0, x1)).fill_fbm()
        assert tuple(int(v) for v in t.voxel_num) == (nx, ny, nyaw) and t.occ_r2_buffer.shape == ((x1 - x0) * ny,)
        import ctypes as C
        a0, a1 = C.c_int32(-1), C.c_int32(-1)
        assert t.L.uph_map_tile(t.h, C.byref(a0), C.byref(a1)) == 0 and (a0.value, a1.value) == (x0, x1)
        assert full.L.uph_map_tile(full.h, C.byref(a0), C.byref(a1)) == 0 and (a0.value, a1.value) == (0, nx)
        # the tile's rows are the whole grid's rows, bit for bit (cells and occupancy)
        for xa in (x0, (x0 + x1) // 2, x1 - 3):
            assert np.array_equal(t.get_window(xa, xa + 3, 100, 140), full.get_window(xa, xa + 3, 100, 140))
        assert np.array_equal(t.occ_r2_buffer, full.occ_r2_buffer.reshape(nx, ny)[x0:x1].ravel())
        with pytest.raises(U._lib.UnevenHipError):
            t.get_window(max(0, x0 - 1) if x0 > 0 else x1, (x0 if x0 > 0 else x1 + 1), 0, 4)      # a row the tile does not hold
        # lookups and front-end queries inside / outside the held rows
        rng = np.random.default_rng(rank)
        xs = rng.uniform(ox + (x0 + 2) * res, ox + (x1 - 2) * res, 500)
        pos = np.column_stack([xs, rng.uniform(-150, 150, 500), rng.uniform(-3.1, 3.1, 500)])
        vt, gt = t.getAllWithGrad(pos)
        vf, gf = full.getAllWithGrad(pos)
        assert np.array_equal(vt, vf) and np.array_equal(gt, gf)
        if x0 > 0:
            outside = np.array([[ox + (x0 - 5) * res, 0.0, 0.0]])
            assert t.frontend_query(outside)[1][0] == -1 and full.frontend_query(outside)[1][0] >= 0
        # the problems routed here solve exactly as on the whole grid
        mine = [probs[i] for i in routes[rank]]
        o = U.ALMTrajOpt(t)
        o.set_rho(1.0)
        out = o.optimize_batch(mine)
        for i, r in zip(routes[rank], out):
            assert r["ret"] == ref[i]["ret"] and r["cost"] == ref[i]["cost"] and np.array_equal(r["x"], ref[i]["x"]), (rank, i)
        # a problem of another owner is refused, not solved on clamped cells
        other = routes[(rank + 2) % WORLD][0]
        mixed = o.optimize_batch([mine[0], probs[other]])
        assert mixed[0]["ret"] == ref[routes[rank][0]]["ret"] and mixed[1]["ret"] == 4


def test_bench_km2_tiled_setup_for_a_middle_rank():
    """what `bench.py --workload km2 --tiled` builds on rank 2 of 4 (the tile path has no collective, so one GPU can play any rank): the
    tile, problems that start in the rank's own slab, all of them accepted and solved"""
    import uneven_planner_amd as U
    from uneven_planner_amd.uneven_map import km2_map, km2_problems, slab_bounds
    m = km2_map(320.0, rank=2, world=4, device=0, tiled=True)
    nx = int(m.voxel_num[0])
    _, a, b = slab_bounds(nx, 2, 4)
    assert m.tile == (a - 80, b + 80)
    probs = km2_problems(m, 320.0, 48, 100, rank=2, world=4)
    lo, hi = m.map_origin[0] + a * m.xy_resolution, m.map_origin[0] + b * m.xy_resolution
    assert all(lo <= p["init_xy"][0, 0] <= hi for p in probs)
    opt = U.ALMTrajOpt(m)
    opt.set_rho(1.0)
    out = opt.optimize_batch(probs)
    assert all(o["ret"] in (0, 2) for o in out)               # none refused (4), none left the tile (5)
