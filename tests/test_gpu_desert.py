"""GPU parity on a REAL scene (BASELINE.json configs[2]: desert scene, batch of 256 random start/goal optimisations): the
reference's desert cloud (fixture tests/golden/desert_xyz.npz) -> plane-fit map on the device vs the oracle's constructMap,
then a B = 256 batch on the device-built map: every problem's first evaluation matches the oracle (on the SAME map) to 1e-9
and the batch converges like the oracle does on a sample of it."""
import os

import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def desert():
    import uneven_planner_amd as U
    xyz = np.load(os.path.join(G, "desert_xyz.npz"))["xyz"]
    m = U.UnevenMap()
    m.build(xyz)
    return xyz, m


def test_desert_map_build_matches_oracle(desert, oracle):
    xyz, m = desert
    st = m.build_stats()
    assert st["cloud_points"] == 100000          # every desert point sits in its own 1 cm voxel
    g = oracle.OracleGrid()
    b = oracle.OracleMapBuilder(xyz=xyz)
    nx, ny, nyaw = g.dims
    for (x0, x1) in ((20, 24), (120, 124)):      # two x-slabs (the full CPU build takes about a minute)
        b.construct(g, x0=x0, x1=x1, do_occ=False)
        co, _ = g.get_cells()
        sl = slice(x0 * ny * nyaw, x1 * ny * nyaw)
        d = np.abs(m.map_buffer[sl] - co[sl]).max(axis=1)
        assert (d > 1e-9).mean() < 1e-3 and np.median(d) < 1e-12, ((d > 1e-9).mean(), d.max())


def test_desert_cells_match_the_independent_numpy_fit(desert):
    """the device plane fit against the numpy / eigh restatement of constructMap (tests/golden/mapcells_golden.npz): a second derivation,
    not the oracle's Jacobi solver"""
    _, m = desert
    z = np.load(os.path.join(G, "mapcells_golden.npz"))
    nx, ny, nyaw = (int(v) for v in m.voxel_num)
    cells = m.map_buffer.reshape(nx, ny, nyaw, 4)
    d = np.array([np.abs(cells[ix, iy, iw] - want).max() for (ix, iy, iw), want in zip(z["idx"], z["cells"])])
    assert (d > 1e-9).sum() == 0 and np.median(d) < 1e-13, ((d > 1e-9).sum(), d.max())


def test_desert_batch_256(desert, oracle):
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    xyz, m = desert
    nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
    probs = scenes.random_problems(256, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
    opt = U.ALMTrajOpt(m)
    opt.upload(probs)
    f, gs = opt.eval_batch(opt.x0_packed(probs))
    og = oracle.OracleGrid()
    og.set_cells(m.map_buffer)                   # the oracle optimiser runs on the device-built map
    for i in range(0, 256, 16):
        a = oracle.OracleALM(og)
        x0 = a.setup(probs[i])
        fo, go, _ = a.eval(x0)
        assert abs(f[i] - fo) / abs(fo) < 1e-9 and rel(go, gs[i]) < 1e-9
    opt.set_rho(1.0)
    out = opt.optimize_batch(probs)
    rets = np.array([o["ret"] for o in out])
    assert set(rets.tolist()) <= {0, 2}
    ref = [oracle.OracleALM(og).optimize(probs[i]) for i in range(0, 256, 16)]
    dc = np.array([abs(out[i]["cost"] - r["cost"]) / abs(r["cost"]) for i, r in zip(range(0, 256, 16), ref)])
    # at the optimiser's own reproducibility (DESIGN.md "Parity"): the solve amplifies rounding-level differences, so an
    # occasional trajectory settles in a neighbouring local solution a few percent away -- the oracle does the same against
    # itself when recompiled with / without FMA contraction.  One such outlier per 16 is tolerated, none beyond 25 %.
    assert np.median(dc) < 5e-3 and np.sort(dc)[-2] < 5e-2 and dc.max() < 0.25
    rep = opt.getMaxVxAxAyCurAttSig()
    ok = rets == 0
    assert np.all(np.abs(rep[ok, 0]) < 0.5 * 1.05) and np.all(-rep[ok, 4] > 0.8 * 0.98)     # converged solves respect v_max and cos xi_min
