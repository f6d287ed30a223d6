"""CPU tier: host-side logic of the package -- the PlanManager resampler (N1), scene generators, PCD reader."""
import math
import os
import struct

import numpy as np

from uneven_planner_amd import resample, scenes


def test_resampler_counts_and_boundary_states():
    p = scenes.hill_problem()
    nxy, nyaw = p["inner_xy"].shape[1] + 1, p["inner_yaw"].shape[0] + 1
    # path length L: Nxy = floor(L/0.3)+1, Nyaw = floor(L/0.15)+1 (plan_manager.cpp:99-121)
    path = resample.hermite_path((4.3, -4.3, 1.57), (-3.5, 3.5, 2.36))
    L = np.linalg.norm(np.diff(path[:, :2], axis=0), axis=1).sum()
    assert nxy == int(L / 0.3) + 1 and nyaw == int(L / 0.15) + 1
    assert nyaw in (2 * nxy - 1, 2 * nxy)
    assert abs(p["total_time"] - L / 0.5 * 1.2) < 1e-9
    assert np.allclose(p["init_xy"][:, 1], [0.05 * math.cos(1.57), 0.05 * math.sin(1.57)])
    assert np.allclose(p["init_xy"][:, 2], 0) and np.allclose(p["end_yaw"][1:], 0)
    # way-points are ~piece_len apart along the path
    pts = np.column_stack([p["init_xy"][:, 0], p["inner_xy"], p["end_xy"][:, 0]]).T
    d = np.linalg.norm(np.diff(pts, axis=0), axis=1)
    assert np.all(d[:-1] < 0.31) and np.all(d[:-1] > 0.25)


def test_resampler_yaw_unwrap():
    path = np.array([[0, 0, 3.0], [0.1, 0, 3.1], [0.2, 0, -3.1], [0.3, 0, -3.0], [0.4, 0, -2.9]])
    p = resample.resample_path(path, piece_len=0.2)
    yaws = np.concatenate([[p["init_yaw"][0]], p["inner_yaw"], [p["end_yaw"][0]]])
    assert np.all(np.abs(np.diff(yaws)) < 1.0)          # no 2*pi jump survives
    assert p["end_yaw"][0] > 3.0


def test_random_problems_are_reproducible_and_in_range():
    a = scenes.random_problems(4, seed0=1234)
    b = scenes.random_problems(4, seed0=1234)
    for pa, pb in zip(a, b):
        assert np.array_equal(pa["inner_xy"], pb["inner_xy"]) and pa["total_time"] == pb["total_time"]
        d = np.linalg.norm(pa["end_xy"][:, 0] - pa["init_xy"][:, 0])
        assert 3.0 <= d <= 10.0
        assert pa["inner_yaw"].shape[0] + 1 >= pa["inner_xy"].shape[1] + 1


def test_hill_cloud_matches_spec():
    xyz = scenes.make_hill_cloud()
    assert xyz.dtype == np.float32 and xyz.shape == (316 * 316, 3)
    assert xyz[:, 0].min() >= -6 and xyz[:, 0].max() <= 6
    assert np.allclose(xyz[:, 2], scenes.hill_height(xyz[:, 0].astype(np.float64), xyz[:, 1].astype(np.float64)), atol=1e-5)
    assert np.array_equal(xyz, scenes.make_hill_cloud())      # deterministic


def test_pcd_reader(tmp_path):
    pts = np.random.default_rng(0).normal(size=(7, 6)).astype(np.float32)
    path = str(tmp_path / "t.pcd")
    with open(path, "wb") as f:
        f.write(b"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z\nSIZE 4 4 4 4 4 4\n"
                b"TYPE F F F F F F\nCOUNT 1 1 1 1 1 1\nWIDTH 7\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 7\nDATA binary\n")
        f.write(pts.tobytes())
    got = scenes.read_pcd(path)
    assert np.array_equal(got, pts[:, :3])


def test_host_grid_view_matches_oracle(oracle, oracle_grid, analytic_cells, tmp_path):
    """value-only lookups (getTerrain / getTerrainVariables / getTerrainSig / getTerrainPos) and the `.map` cache round trip"""
    from uneven_planner_amd.host_map import HostGridView
    hv = HostGridView(analytic_cells)
    rng = np.random.default_rng(3)
    pos = np.column_stack([rng.uniform(-5.2, 5.2, 200), rng.uniform(-5.2, 5.2, 200), rng.uniform(-np.pi, np.pi, 200)])
    pos[:3] = [[0.0, 0.0, -3.095], [4.99995, 0.0, 0.0], [5.5, 0.0, 0.0]]
    want = oracle_grid.terrain(pos)
    got = np.array([hv.getTerrain(p) for p in pos])
    assert np.abs(want - got).max() < 1e-12
    inmap = np.array([hv.isInMap(p) for p in pos])
    wv = oracle_grid.terrain_variables(pos[inmap])
    gv = np.array([hv.getTerrainVariables(p) for p in pos[inmap]])
    assert np.abs(wv - gv).max() < 1e-12
    R, p3 = hv.getTerrainPos(pos[10])
    assert np.allclose(R.T @ R, np.eye(3), atol=1e-12) and abs(p3[2] - want[10, 0]) < 1e-12
    assert hv.getTerrainSig(pos[10]) == got[10, 1]
    # cache round trip on a small grid
    small = HostGridView(rng.uniform(-0.3, 0.3, size=(10 * 10 * 64, 4)), map_size_x=0.5, map_size_y=0.5)
    path = str(tmp_path / "s.map")
    small.write_map_file(path)
    back = HostGridView.read_map_file(path, map_size_x=0.5, map_size_y=0.5)
    assert 0 < np.abs(back.cells - small.cells).max() < 1e-5


# ---- M1: crop box + voxel grid of the PRODUCT's host path against the oracle's restatement of PCL (uneven_map.cpp:133-143)
def test_product_cloud_filter_merges_and_rejects_like_the_oracle(oracle):
    import uneven_planner_amd as U
    rng = np.random.default_rng(17)
    base = rng.uniform(-3.0, 3.0, size=(4000, 3)).astype(np.float32)
    base[:, 2] = np.abs(base[:, 2]) * 0.5
    dup = base[:1500] + rng.uniform(-0.004, 0.004, size=(1500, 3)).astype(np.float32)       # several points per 1 cm voxel
    trip = base[:400] + rng.uniform(-0.003, 0.003, size=(400, 3)).astype(np.float32)
    outside = np.array([[10.5, 0, 1], [0, -10.2, 1], [0, 0, 5.5], [0, 0, -0.02], [-11, -11, 0], [np.nan, 0, 0], [0, np.inf, 0]], dtype=np.float32)
    edge = np.array([[10.0, 10.0, 5.0], [-10.0, -10.0, -0.01]], dtype=np.float32)             # on the faces of the crop box: kept (inclusive)
    cloud = np.vstack([base, dup, outside, trip, edge])
    cloud = cloud[rng.permutation(cloud.shape[0])]
    got = U.UnevenMap.filter_cloud(cloud)
    want = oracle.OracleMapBuilder(xyz=cloud).cloud()
    assert want.shape[0] < cloud.shape[0] - 500                  # the filter really merged (several points per voxel) and rejected
    assert got.shape == want.shape and np.array_equal(got, want)     # bit for bit, same order (leaf-index order)


def test_product_cloud_filter_on_a_cloud_without_merges(oracle):
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    xyz = scenes.make_hill_cloud(n_side=80, half=2.0)
    assert np.array_equal(U.UnevenMap.filter_cloud(xyz), oracle.OracleMapBuilder(xyz=xyz).cloud())


# ---- M5 / N3: `.map` CSV cache both ways between the product's host view and the oracle, and the bit-exact binary side-car
def test_map_csv_cross_reads_and_binary_sidecar(tmp_path, oracle):
    import ctypes as C
    from uneven_planner_amd.host_map import HostGridView
    rng = np.random.default_rng(3)
    g = oracle.OracleGrid(size_x=1.0, size_y=1.0)
    cells = np.column_stack([rng.normal(size=g.ncell), rng.uniform(0, 0.2, g.ncell), rng.uniform(-0.3, 0.3, g.ncell), rng.uniform(-0.3, 0.3, g.ncell)])
    g.set_cells(cells)
    L = oracle.lib()
    # oracle writes (reference format: default ostream precision = 6 significant digits), product reads
    p1 = str(tmp_path / "oracle.map")
    assert L.orc_map_write_csv(g.h, p1.encode()) == 0
    v = HostGridView.read_map_file(p1, 1.0, 1.0)
    assert np.abs(v.cells.reshape(-1, 4) - cells).max() < 1e-5 * max(1.0, np.abs(cells).max())
    # product writes, oracle reads: both must land on the same 6-digit values
    p2 = str(tmp_path / "product.map")
    HostGridView(cells, 1.0, 1.0).write_map_file(p2)
    g2 = oracle.OracleGrid(size_x=1.0, size_y=1.0)
    assert L.orc_map_read_csv(g2.h, p2.encode()) == 0
    assert np.array_equal(g2.get_cells()[0], v.cells.reshape(-1, 4))            # identical text -> identical doubles (incl. the reference's stold double rounding)
    assert open(p1).read() == open(p2).read()                                  # the two writers produce the same file
    # binary side-car: bit-exact round trip, refuses a grid of another shape
    p3 = str(tmp_path / "product.map.bin")
    HostGridView(cells, 1.0, 1.0).write_map_binary(p3)
    assert np.array_equal(HostGridView.read_map_binary(p3, 1.0, 1.0).cells.reshape(-1, 4), cells)
    import pytest
    with pytest.raises(ValueError):
        HostGridView.read_map_binary(p3, 2.0, 1.0)


CSV_CPP = r"""
// the `.map` cache from the reference host's own language: what UnevenMap::constructMap / constructMapInput would call after the swap
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "uneven_hip.h"
int main(int argc, char** argv) {
    // argv: in.map  out.map  out.bin  nx ny nyaw     -- read the CSV as constructMapInput does, write it back, write + re-read the side-car
    const int32_t d[3] = {atoi(argv[4]), atoi(argv[5]), atoi(argv[6])};
    const size_t ncell = (size_t)d[0] * d[1] * d[2];
    std::vector<double> cells(4 * ncell), c(ncell), back(4 * ncell);
    int64_t lines = 0;
    if (uph_map_load_csv(argv[1], d, cells.data(), c.data(), &lines) != UPH_OK) { std::printf("load failed: %s\n", uph_last_error()); return 2; }
    if (uph_map_save_csv(argv[2], cells.data(), d) != UPH_OK || uph_map_save_bin(argv[3], cells.data(), d) != UPH_OK) return 3;
    if (uph_map_load_bin(argv[3], d, back.data()) != UPH_OK) return 4;
    for (size_t i = 0; i < 4 * ncell; i++) if (back[i] != cells[i]) return 5;
    const int32_t other[3] = {d[0] + 1, d[1], d[2]};
    if (uph_map_load_bin(argv[3], other, back.data()) != UPH_ERR_LIMIT) return 6;
    if (uph_map_load_csv("/nonexistent/dir/x.map", d, back.data(), nullptr, nullptr) != UPH_ERR_INVALID) return 7;      // -> the caller builds the map
    std::printf("%lld lines\n", (long long)lines);
    return 0;
}
"""


def test_map_cache_in_the_c_abi_against_the_oracle_and_the_mirror(tmp_path, oracle):
    """VERDICT r04 missing 3 (N3 in the host's language): uph_map_save_csv / uph_map_load_csv / uph_map_save_bin / uph_map_load_bin (host functions of
    the C-ABI, csrc/map_io_host.cpp) against the oracle's restatement of uneven_map.cpp:270-315, 400-412 and the Python mirror -- same text
    both ways, same doubles after the stold-then-double parse, cells the file does not mention stay RXS2() zeros with c = 1, lines with an
    index outside the grid dropped, any line order, side-car bit exact; exercised from ctypes AND from a C++ program"""
    import ctypes as C
    import subprocess
    import uneven_planner_amd as U
    from uneven_planner_amd.host_map import HostGridView
    L = U._lib.load()
    OL = oracle.lib()
    rng = np.random.default_rng(9)
    g = oracle.OracleGrid(size_x=1.0, size_y=1.0)
    dims = (C.c_int32 * 3)(*[int(v) for v in g.dims])
    cells = np.column_stack([rng.normal(size=g.ncell) * 10.0 ** rng.integers(-7, 3, g.ncell), rng.uniform(0, 0.2, g.ncell), rng.uniform(-0.3, 0.3, g.ncell), rng.uniform(-0.3, 0.3, g.ncell)])
    cells[:7, 0] = [0.0, -0.0, 1e-5, 123456.5, -1e-310, 0.1, 1.0 / 3.0]             # %g corner cases: exponent switch, rounding up a digit, a denormal
    g.set_cells(cells)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    # 1. the three writers (reference ostream form in the oracle, the Python mirror's "%.6g", the C-ABI's) produce the same file
    p_or, p_py, p_c = (str(tmp_path / n) for n in ("oracle.map", "mirror.map", "cabi.map"))
    assert OL.orc_map_write_csv(g.h, p_or.encode()) == 0
    HostGridView(cells, 1.0, 1.0).write_map_file(p_py)
    assert L.uph_map_save_csv(p_c.encode(), dp(np.ascontiguousarray(cells)), dims) == 0
    assert open(p_c).read() == open(p_or).read() == open(p_py).read()
    # 2. the three readers land on the same doubles (two roundings: stold, then double)
    got, cbuf, nl = np.full((g.ncell, 4), 7.0), np.zeros(g.ncell), C.c_int64(0)
    assert L.uph_map_load_csv(p_or.encode(), dims, dp(got), dp(cbuf), C.byref(nl)) == 0 and nl.value == g.ncell
    g2 = oracle.OracleGrid(size_x=1.0, size_y=1.0)
    assert OL.orc_map_read_csv(g2.h, p_c.encode()) == 0
    want, want_c = g2.get_cells()[0], g2.get_cells()[1]
    assert np.array_equal(got, want) and np.array_equal(got, HostGridView.read_map_file(p_c, 1.0, 1.0).cells.reshape(-1, 4))
    assert np.array_equal(cbuf, want_c)
    assert np.abs(got - cells).max() <= 5e-6 * np.abs(cells).max() and not np.array_equal(got, cells)          # six significant digits
    # 3. a partial, shuffled file with out-of-range and short lines: untouched cells stay zero / c = 1, later lines win
    lines = open(p_c).read().splitlines()
    keep = [lines[i] for i in rng.permutation(len(lines))[: len(lines) // 3]]
    keep += ["%d,0,0,1,2,0.1,0.2" % g.dims[0], "-1,0,0,1,2,0.1,0.2", "0,0,%d,1,2,0.1,0.2" % g.dims[2], "3,4", "", keep[0].rsplit(",", 4)[0] + ",9.5,0.125,0.25,-0.5"]
    p_part = str(tmp_path / "partial.map")
    open(p_part, "w").write("\n".join(keep) + "\n")
    got2, c2 = np.full((g.ncell, 4), 7.0), np.zeros(g.ncell)
    assert L.uph_map_load_csv(p_part.encode(), dims, dp(got2), dp(c2), C.byref(nl)) == 0
    g3 = oracle.OracleGrid(size_x=1.0, size_y=1.0)
    assert OL.orc_map_read_csv(g3.h, p_part.encode()) == 0
    assert np.array_equal(got2, g3.get_cells()[0]) and np.array_equal(c2, g3.get_cells()[1])
    assert nl.value == len(lines) // 3 + 1 and (got2 == 0.0).all(axis=1).sum() >= g.ncell - len(lines) // 3 - 1
    x0, y0, w0 = (int(v) for v in keep[0].split(",")[:3])
    a0 = (x0 * int(g.dims[1]) + y0) * int(g.dims[2]) + w0
    assert list(got2[a0]) == [9.5, 0.125, 0.25, -0.5] and c2[a0] == np.sqrt(1.0 - 0.25 ** 2 - 0.5 ** 2)
    # 4. side-car: bit exact, same bytes as the mirror's, refuses another grid
    p_b, p_bpy = str(tmp_path / "cabi.map.bin"), str(tmp_path / "mirror.map.bin")
    assert L.uph_map_save_bin(p_b.encode(), dp(np.ascontiguousarray(cells)), dims) == 0
    HostGridView(cells, 1.0, 1.0).write_map_binary(p_bpy)
    assert open(p_b, "rb").read() == open(p_bpy, "rb").read()
    back = np.zeros((g.ncell, 4))
    assert L.uph_map_load_bin(p_b.encode(), dims, dp(back)) == 0 and np.array_equal(back, cells) and np.array_equal(np.signbit(back), np.signbit(cells))
    assert L.uph_map_load_bin(p_b.encode(), (C.c_int32 * 3)(int(g.dims[0]), int(g.dims[1]) + 1, int(g.dims[2])), dp(back)) == -4      # UPH_ERR_LIMIT
    assert L.uph_map_load_bin(p_c.encode(), dims, dp(back)) == -1                          # a CSV is not a side-car
    # 5. the same entry points from C++ (the language of the reference host)
    src = tmp_path / "csv_consumer.cpp"
    src.write_text(CSV_CPP)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "uneven_planner_amd")
    exe = str(tmp_path / "csv_consumer")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", exe, "-L", libdir, "-lunevenhip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out_map, out_bin = str(tmp_path / "cpp.map"), str(tmp_path / "cpp.map.bin")
    r = subprocess.run([exe, p_or, out_map, out_bin] + [str(int(v)) for v in g.dims], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout)
    assert open(out_map).read() == open(p_or).read()                  # six-digit text is a fixed point of read -> write
    assert np.array_equal(HostGridView.read_map_binary(out_bin, 1.0, 1.0).cells.reshape(-1, 4), got)


def test_tile_rows_and_owner_routing_cover_every_problem():
    """SURVEY.md 8e row 3, host side: x-slab tiles with a halo and the owner rule -- every problem has exactly one owner and lies inside
    the owner's tile with room to spare when the halo exceeds the longest local goal"""
    from uneven_planner_amd import scenes
    from uneven_planner_amd.uneven_map import owner_of, route_problems, slab_bounds, tile_rows
    nx, res, ox = 4000, 0.25, -500.0
    probs = scenes.local_problems(200, seed0=5000, half=495.0)
    for world in (1, 3, 8):
        routes = route_problems(probs, nx, world, res, ox)
        assert sorted(i for r in routes for i in r) == list(range(len(probs)))
        halo = int(round(20.0 / res))
        for rank, idx in enumerate(routes):
            x0, x1 = tile_rows(nx, rank, world, halo)
            _, a, b = slab_bounds(nx, rank, world)
            assert 0 <= x0 <= a < b <= x1 <= nx
            lo_m, hi_m = ox + x0 * res, ox + x1 * res
            for i in idx:
                p = probs[i]
                xs = np.concatenate([p["init_xy"][0, :1], p["end_xy"][0, :1], p["inner_xy"][0]])
                assert owner_of(p, nx, world, res, ox) == rank
                assert (xs.min() >= lo_m + 2.0 or x0 == 0) and (xs.max() <= hi_m - 2.0 or x1 == nx)      # UPH_TILE_MARGIN of the upload check


def test_isa_scan_flags_a_load_with_a_dead_destination(tmp_path):
    """tools/isa_waw_waits.py on two synthetic kernels: the round-5 terrain gather as it was (a 16-byte load whose first register pair is never
    read and is reused as the next address temporary: the wait in front of the overwrite is reported) and as it is (an 8-byte and a 16-byte load,
    every destination read: nothing to report)."""
    import subprocess, sys
    dead = """
0000000000001000 <kernel_dead_pair>:
	global_load_dwordx4 v[38:41], v[42:43], off                // 000000001000: DC5C8000
	global_load_dwordx4 v[46:49], v[42:43], off offset:16      // 000000001008: DC5C8010
	v_mov_b32_e32 v57, v133                                    // 000000001010: 7E720385
	s_waitcnt vmcnt(1)                                         // 000000001014: BF8C0F71
	v_lshlrev_b64 v[38:39], 5, v[56:57]                        // 000000001018: D28F0026
	global_load_dwordx4 v[42:45], v[38:39], off                // 000000001020: DC5C8000
	s_waitcnt vmcnt(0)                                         // 000000001028: BF8C0F70
	v_add_f64 v[50:51], v[40:41], v[46:47]                     // 00000000102C: D2800032
	s_endpgm                                                   // 000000001034: BF810000
"""
    clean = """
0000000000002000 <kernel_clean>:
	global_load_dwordx2 v[168:169], v[38:39], off offset:24    // 000000002000: DC548018
	global_load_dwordx4 v[38:41], v[38:39], off offset:8       // 000000002008: DC5C8008
	global_load_dwordx2 v[186:187], v[42:43], off offset:24    // 000000002010: DC548018
	global_load_dwordx4 v[42:45], v[42:43], off offset:8       // 000000002018: DC5C8008
	s_waitcnt vmcnt(2)                                         // 000000002020: BF8C0F72
	v_add_f64 v[50:51], v[168:169], v[38:39]                   // 000000002024: D2800032
	s_waitcnt vmcnt(0)                                         // 00000000202C: BF8C0F70
	v_add_f64 v[52:53], v[186:187], v[42:43]                   // 000000002030: D2800034
	s_endpgm                                                   // 000000002038: BF810000
"""
    f = tmp_path / "k.s"
    f.write_text("\nx.co:\tfile format elf64-amdgpu\n\nDisassembly of section .text:\n" + dead + clean)
    tool = os.path.join(os.path.dirname(__file__), "..", "tools", "isa_waw_waits.py")
    out = subprocess.run([sys.executable, tool, str(f), "kernel_dead_pair", "vm"], capture_output=True, text=True, check=True).stdout
    assert "WAW-suspect wait at 3" in out and "v_lshlrev_b64 v[38:39]" in out, out
    out = subprocess.run([sys.executable, tool, str(f), "kernel_clean", "vm"], capture_output=True, text=True, check=True).stdout
    assert "WAW-suspect" not in out, out
