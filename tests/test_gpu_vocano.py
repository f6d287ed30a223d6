"""BASELINE.json configs[3]: volcano scene (the reference's vocano.pcd as fixture tests/golden/vocano_xyz.npz, parameters of
plan_manager/params/run_vocano.yaml: max_rho = max_sig = 0.08), plane-fit build sharded over x-slabs with one all-gather.

One GPU is visible here, so the sharded path is exercised (a) through RCCL at world size 1 -- UnevenMap.build_sharded end to end:
slab build, export, torch.distributed all_gather_into_tensor on the NCCL(=RCCL) backend, import -- and (b) as a "fake world of 4":
four slab builds on one device stitched by the same export / gather / import calls.  Both must equal the single build bit for bit
(the fit of a cell reads only the cloud and its own cell, uneven_map.cpp:329-391), and slabs of the result are compared with the
CPU oracle's constructMap.  The 8-GPU run itself is the driver's (bench.py --gpus 8 builds its map this way)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VOCANO_MAP = dict(max_rho=0.08)
VOCANO_OPT = dict(max_sig=0.08)


@pytest.fixture(scope="module")
def vocano():
    import uneven_planner_amd as U
    xyz = np.load(os.path.join(G, "vocano_xyz.npz"))["xyz"]
    m = U.UnevenMap(VOCANO_MAP)
    m.build(xyz)
    return xyz, m


def test_vocano_map_slabs_match_oracle(vocano, oracle):
    xyz, m = vocano
    assert m.build_stats()["cloud_points"] == 100000        # every point in its own 1 cm voxel
    g = oracle.OracleGrid()
    b = oracle.OracleMapBuilder(xyz=xyz)
    nx, ny, nyaw = g.dims
    for (x0, x1) in ((50, 53), (100, 103), (160, 162)):     # flank, crater region, far flank (the full CPU build takes a minute)
        b.construct(g, map_params=VOCANO_MAP, x0=x0, x1=x1, do_occ=True)
        co, _ = g.get_cells()
        sl = slice(x0 * ny * nyaw, x1 * ny * nyaw)
        d = np.abs(m.map_buffer[sl] - co[sl]).max(axis=1)
        assert (d > 1e-9).mean() < 1e-3 and np.median(d) < 1e-12, ((d > 1e-9).mean(), d.max())
        occ_o, _ = g.get_occ()
        assert (m.occ_buffer[sl] == occ_o[sl]).mean() > 0.999       # occupancy with max_rho 0.08 (uneven_map.cpp:170-179)


def test_sharded_build_fake_world_of_four_is_bit_identical(vocano):
    import torch
    import uneven_planner_amd as U
    from uneven_planner_amd.uneven_map import gather_slabs, slab_bounds
    xyz, m = vocano
    nx, ny, nyaw = (int(v) for v in m.voxel_num)
    row = ny * nyaw * 4
    for world in (4, 3):                                     # 3: the zero-padded last slab (nx = 200 does not divide)
        slabs = []
        for rank in range(world):
            per, x0, x1 = slab_bounds(nx, rank, world)
            w = U.UnevenMap(VOCANO_MAP)
            w.build(xyz, x0=x0, x1=x1, download=False)
            slab = torch.zeros(per * row, dtype=torch.float64, device="cuda:0")
            U._lib.check(w.L.uph_map_export_slab_dev(w.h, x0, x1, __import__("ctypes").c_void_p(slab.data_ptr())), "export")
            slabs.append(slab)
        torch.cuda.synchronize()
        full = gather_slabs(slabs[0], nx, row, world, lambda f, s_: f.copy_(torch.cat(slabs))).contiguous()
        r = U.UnevenMap(VOCANO_MAP)
        U._lib.check(r.L.uph_map_import_cells_dev(r.h, __import__("ctypes").c_void_p(full.data_ptr())), "import")
        r.download()
        assert np.array_equal(r.map_buffer, m.map_buffer) and np.array_equal(r.occ_buffer, m.occ_buffer) and np.array_equal(r.c_buffer, m.c_buffer)


def test_sharded_build_over_rccl_world_one(vocano):
    import torch
    import torch.distributed as dist
    import uneven_planner_amd as U
    xyz, m = vocano
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        s = U.UnevenMap(VOCANO_MAP)
        s.build_sharded(xyz, 0, 1, lambda full, slab: dist.all_gather_into_tensor(full, slab))
    finally:
        dist.destroy_process_group()
    assert np.array_equal(s.map_buffer, m.map_buffer) and np.array_equal(s.occ_r2_buffer, m.occ_r2_buffer)


def test_vocano_batch_64(vocano, oracle):
    """B = 64 random start/goal solves on the volcano map with run_vocano.yaml's max_sig: first evaluations to 1e-9, solves like the oracle's"""
    import uneven_planner_amd as U
    from conftest import rel
    from uneven_planner_amd import scenes
    xyz, m = vocano
    nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
    probs = scenes.random_problems(64, seed0=4000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
    opt = U.ALMTrajOpt(m, VOCANO_OPT)
    opt.upload(probs)
    f, gs = opt.eval_batch(opt.x0_packed(probs))
    og = oracle.OracleGrid()
    og.set_cells(m.map_buffer)
    for i in range(0, 64, 4):
        a = oracle.OracleALM(og, VOCANO_OPT)
        x0 = a.setup(probs[i])
        fo, go, _ = a.eval(x0)
        assert abs(f[i] - fo) / abs(fo) < 1e-9 and rel(go, gs[i]) < 1e-9
    opt.set_rho(1.0)
    out = opt.optimize_batch(probs)
    rep = opt.getMaxVxAxAyCurAttSig()
    # final costs: the volcano's steep flanks make the solves long and chaotic, so the yardstick is the oracle's own reproducibility on
    # the same problems (FMA-contracted rebuild, tests/sensitivity.py), as in test_gpu_buckets.py
    import sensitivity
    sub = [probs[i] for i in range(0, 64, 4)]
    ref = [oracle.OracleALM(og, VOCANO_OPT).optimize(p) for p in sub]
    fma = sensitivity.solve_with_fma_oracle(m.map_buffer, sub, VOCANO_OPT)
    floor = sensitivity.spread(ref, fma)
    got = sensitivity.spread(ref, [out[i] for i in range(0, 64, 4)])
    print("volcano floor", floor, "device", got)
    assert got["c_median"] <= 3.0 * floor["c_median"] + 1e-3 and got["x_median"] <= 3.0 * floor["x_median"] + 1e-3
    st = sensitivity.drift_stats(ref, fma, [out[i] for i in range(0, 64, 4)])
    print("volcano drift", st)
    sensitivity.assert_no_directional_drift(st, "volcano")            # converged rate (McNemar, 2 SE), same return code (2 SE), cost sign test -- no fixed slack
    assert all(o["ret"] in (0, 2) for o in out)
    conv = np.array([o["ret"] == 0 for o in out])
    assert np.all(rep[conv, 5] < 0.08 * 1.1) and np.all(np.abs(rep[conv, 0]) < 0.5 * 1.1)        # converged => within max_sig / max_vel
