"""GPU tier, SURVEY row N4 second half: the batched device search (uph_kino_plan_batch, csrc/kino_search.hip) against the CPU oracle's
restatement of KinoAstar::plan (oracle/kino_astar.hpp <- front_end/src/kino_astar.cpp:67-236) on the same grid and occupancy.

Bar: INTEGER-EXACT -- status, iter_num, use_node_num and the whole sequence of expanded lattice cells are equal, i.e. the device pops the
same nodes in the same order as the reference's std::priority_queue would, ties, in-place relaxations and the NaN nodes of the v = 0
primitives included; the returned poses agree to 1e-9 (device sin / cos / atan2 vs glibc's: an ulp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hill(oracle):
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    m = U.UnevenMap()
    m.build(scenes.make_hill_cloud())
    g = oracle.OracleGrid()
    g.set_cells(m.map_buffer)
    g.set_occ(m.occ_buffer, m.occ_r2_buffer)          # the device's own occupancy layers: this test is about the search
    return m, g


def _queries(m, n, seed0, **kw):
    from uneven_planner_amd import scenes
    nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
    return scenes.random_queries(n, seed0=seed0, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]), **kw)


def _same(dev, orc, tag=""):
    assert dev["status"] == orc["status"], (tag, dev["status"], orc["status"])
    assert dev["iter_num"] == orc["iter_num"] and dev["use_node_num"] == orc["use_node_num"], (tag, dev["iter_num"], orc["iter_num"], dev["use_node_num"], orc["use_node_num"])
    if "expanded" in dev:
        k = min(len(dev["expanded"]), len(orc["expanded"]))
        assert k == orc["n_expanded"] or k == len(dev["expanded"])
        neq = np.nonzero((dev["expanded"][:k] != orc["expanded"][:k]).any(axis=1))[0]
        assert neq.size == 0, "%s: expansion sequences part at step %d of %d" % (tag, neq[0], k)
    assert dev["n_path"] == orc["n_path"], (tag, dev["n_path"], orc["n_path"])
    if orc["n_path"]:
        assert np.abs(dev["path"] - orc["path"][:len(dev["path"])]).max() < 1e-9, tag


def test_expansion_sequences_are_the_references(hill, oracle):
    import uneven_planner_amd as U
    m, g = hill
    S, G = _queries(m, 24, 4000)
    ka = U.KinoAstar(m)
    assert ka.n_primitives == 15                       # kino_astar.cpp:138-145 with run_hill.yaml
    dev = ka.plan_batch(S, G, path_cap=1024, exp_cap=40000)
    ok = oracle.OracleKinoAstar(g)
    n_ok = 0
    for b in range(len(S)):
        o = ok.plan(S[b], G[b])
        _same(dev[b], o, "query %d" % b)
        n_ok += o["status"] == 0
    assert n_ok >= 20                                  # the hill scene is mostly traversable: these are real searches (thousands of expansions each)
    assert np.mean([d["iter_num"] for d in dev]) > 500


def test_short_range_and_capped_searches(hill, oracle):
    """goals inside the one-shot range (the first popped node already shoots), and the expansion cap (test hook) at several depths"""
    import uneven_planner_amd as U
    m, g = hill
    ka = U.KinoAstar(m)
    ok = oracle.OracleKinoAstar(g)
    S, G = _queries(m, 6, 4100, dmin=0.2, dmax=0.9)
    for d, s, gl in zip(ka.plan_batch(S, G, exp_cap=64), S, G):
        _same(d, ok.plan(s, gl), "short")
    S, G = _queries(m, 4, 4200)
    for cap in (1, 2, 17, 300):
        for d, s, gl in zip(ka.plan_batch(S, G, max_expand=cap, exp_cap=512), S, G):
            o = ok.plan(s, gl, max_expand=cap)
            assert o["status"] == 5 or (cap == 300 and o["status"] == 0)          # (a short search may end before the largest cap)
            _same(d, o, "cap %d" % cap)


def test_refusals_dead_ends_and_slot_reuse(oracle, analytic_cells):
    """occupied start / goal (kino_astar.cpp:86-95), a goal walled in (the open set runs empty or the pool runs out -- whichever the reference
    does), and more queries than slots (a wave searches several queries one after the other in the same workspace)"""
    import uneven_planner_amd as U
    m = U.UnevenMap(device=0)
    cells = analytic_cells.reshape(200, 200, 64, 4).copy()
    cells[118:123, 118:123, :, 1] = 1.0                # sigma = 1 > max_rho: a 5 x 5 block of occupied columns around (1.0, 1.0) ...
    # ... and a closed wall around (-1.5, -1.5), FOUR cells = 0.2 m thick: the reference samples a primitive for collisions every 0.06 m of arc only
    # (kino_astar.cpp:173-183) and never its end state, so it tunnels through a one-cell wall
    cells[58:83, 58:62, :, 1] = 1.0; cells[58:83, 79:83, :, 1] = 1.0; cells[58:62, 58:83, :, 1] = 1.0; cells[79:83, 58:83, :, 1] = 1.0
    m.set_cells(cells.reshape(-1, 4))
    g = oracle.OracleGrid()
    g.set_cells(m.map_buffer)
    g.set_occ(m.occ_buffer, m.occ_r2_buffer)
    assert m.occ_r2_buffer.reshape(200, 200)[120, 120] == 1 and m.occ_r2_buffer.reshape(200, 200)[70, 70] == 0
    ok = oracle.OracleKinoAstar(g)
    S = np.array([[1.02, 1.02, 0.0], [3.0, 3.0, 0.5], [3.0, 3.0, 0.5], [3.5, -3.0, 2.0], [-1.5, -1.5, 0.3], [2.0, -2.0, 1.0], [-3.0, 3.0, -1.0]])
    G = np.array([[3.0, 3.0, 0.0], [1.02, 1.02, 0.0], [-1.5, -1.5, 0.0], [0.0, 0.0, 0.0], [-1.2, -1.8, 1.0], [-4.0, 4.0, 0.0], [4.0, -4.0, 3.0]])
    ka = U.KinoAstar(m, slots=3)
    assert ka.slots == 3
    dev = ka.plan_batch(S, G, exp_cap=40000)
    want = [ok.plan(s, gl) for s, gl in zip(S, G)]
    assert [w["status"] for w in want][:2] == [1, 2]
    assert want[2]["status"] in (3, 4)                 # the walled-in goal is never reached
    assert want[4]["status"] in (0, 3)                 # inside the wall, start and goal together: a 0.85 m yard -- the open set soon runs empty
    for b, (d, w) in enumerate(zip(dev, want)):
        _same(d, w, "query %d" % b)
    # the same batch with one query per wave: identical results (workspaces are fully re-initialised between queries)
    dev2 = U.KinoAstar(m).plan_batch(S, G, exp_cap=40000)
    for d, d2 in zip(dev, dev2):
        assert d["status"] == d2["status"] and d["iter_num"] == d2["iter_num"] and np.array_equal(d["expanded"], d2["expanded"]) and np.array_equal(d["path"], d2["path"])


def test_plan_is_the_reference_call(hill, oracle):
    """KinoAstar::plan(start_state, end_state) -> front_end_path: one query, launch pose of run_hill.launch, and its use as the producer of the
    back-end's input (PlanManager::rcvWpsCallBack, plan_manager.cpp:59-132): the path goes through the resampler into the optimiser"""
    import uneven_planner_amd as U
    m, g = hill
    ka = U.KinoAstar()
    ka.setEnvironment(m)
    path = ka.plan([4.3, -4.3, 1.57], [-3.5, 3.5, 2.36])
    o = oracle.OracleKinoAstar(g).plan([4.3, -4.3, 1.57], [-3.5, 3.5, 2.36])
    assert o["status"] == 0 and len(path) == o["n_path"] and np.abs(path - o["path"]).max() < 1e-9
    from uneven_planner_amd import resample
    prob = resample.resample_path(path)
    opt = U.ALMTrajOpt(m)
    out = opt.optimize_batch([prob])[0]
    assert out["ret"] in (0, 2) and np.isfinite(out["cost"])


@pytest.mark.parametrize("params", [dict(collision_interval=0.02),                      # 7 collision samples per primitive: each primitive's lane walks its samples itself
                                    dict(time_interval=0.5, collision_interval=0.05),   # longer arcs, 5 samples (same generic layout)
                                    dict(max_vel=0.8, max_steer=0.3, weight_v_change=0.5, weight_delta_change=0.3),      # other primitives, 4 samples
                                    dict(collision_interval=0.08, oneshot_range=2.0)],  # one sample per primitive (the spread layout: end state + samples over four lane groups)
                         ids=["7-samples", "long-arcs", "other-primitives", "1-sample"])
def test_other_primitive_tables(hill, oracle, params):
    """the kernel has two layouts of an expansion -- end state and collision samples of a primitive spread over lanes p, 16 + p, 32 + p, 48 + p
    (<= 16 primitives of <= 3 samples, run_hill.yaml's case) or one lane per primitive walking its samples -- and hands stateTransit host-made
    constants per (primitive, duration): other parameter sets exercise both against the oracle, expansion sequence included"""
    import uneven_planner_amd as U
    m, g = hill
    ka = U.KinoAstar(m, params=params)
    ok = oracle.OracleKinoAstar(g, params)
    S, G = _queries(m, 6, 4700)
    n_ok = 0
    for d, s, gl in zip(ka.plan_batch(S, G, path_cap=1024, exp_cap=40000), S, G):
        o = ok.plan(s, gl)
        _same(d, o, str(params))
        n_ok += o["status"] == 0
    assert n_ok >= 3


def test_every_kernel_instantiation(hill, oracle):
    """the search kernel is compiled per register budget (2 / 4 / 6 / 8 waves per SIMD: 1023 / 511 / 255 heap entries in LDS) and with or without
    sincosFast; static or dynamic query hand-out is a run-time flag: every combination pops the oracle's sequence"""
    import uneven_planner_amd as U
    m, g = hill
    ok = oracle.OracleKinoAstar(g)
    S, G = _queries(m, 5, 4900)
    ref = [ok.plan(s, gl) for s, gl in zip(S, G)]
    ka = U.KinoAstar(m, slots=3)                       # fewer workspaces than queries: slots are reused inside the launch
    for wps in (2, 4, 6, 8):
        for flags in (0, 1, 2, 3):
            ka.set_wps(wps); ka.set_flags(flags)
            for d, o in zip(ka.plan_batch(S, G, path_cap=1024, exp_cap=40000), ref):
                _same(d, o, "wps %d flags %d" % (wps, flags))
