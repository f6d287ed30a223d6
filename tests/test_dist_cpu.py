"""CPU tier, world_size 2 and 3 over gloo: the N > 1 paths -- (a) the x-slab sharded map build + one all-gather (SURVEY.md 8e;
RCCL on the GPUs, gloo here) through the PRODUCT's slab rule and exchange (uneven_map.slab_bounds / gather_slabs) reproduces the
single-process build, with the CPU oracle standing in for the device kernel;
(b) the batch sharding of bench.py gives every rank a disjoint, reproducible set of problems and a max-over-ranks time."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_py as O
    from uneven_planner_amd import scenes
    xyz = scenes.make_hill_cloud(n_side=100, half=1.5)
    mp_ = dict(map_size_x=2.0, map_size_y=2.0)
    from uneven_planner_amd.uneven_map import gather_slabs, slab_bounds      # the PRODUCT's slab rule and exchange ...
    g = O.OracleGrid(size_x=2.0, size_y=2.0)
    nx, ny, nyaw = g.dims
    row = ny * nyaw * 4
    per, x0, x1 = slab_bounds(nx, rank, world)
    b = O.OracleMapBuilder(xyz=xyz)
    b.construct(g, map_params=mp_, x0=x0, x1=x1, do_occ=False)         # ... around the oracle's fit instead of the device kernel
    cells, _ = g.get_cells()
    slab = torch.zeros(per * row, dtype=torch.float64)
    slab[:(x1 - x0) * row] = torch.from_numpy(cells.reshape(nx, -1)[x0:x1].copy().ravel())
    full = gather_slabs(slab, nx, row, world, lambda f, s_: dist.all_gather_into_tensor(f, s_))
    # fp32 storage (configs[4]): the same exchange on float slabs
    slab32 = slab.to(torch.float32)
    full32 = gather_slabs(slab32, nx, row, world, lambda f, s_: dist.all_gather_into_tensor(f, s_))
    assert full32.dtype == torch.float32 and torch.equal(full32, full.to(torch.float32))
    # strong-scaling split of one batch (bench.py --workload km2): the shares tile the batch
    lo, cnt = scenes.batch_share(11, rank, world)
    share = torch.tensor([lo, cnt], dtype=torch.int64)
    shares = [torch.empty(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(shares, share)
    covered = sorted(i for lo_, c_ in (s_.tolist() for s_ in shares) for i in range(lo_, lo_ + c_))
    assert covered == list(range(11))
    # batch sharding: seeds 1000 + rank*B + i
    B = 3
    probs = scenes.random_problems(B, seed0=1000 + rank * B)
    key = torch.tensor([p["total_time"] for p in probs], dtype=torch.float64)
    keys = [torch.empty(B, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(keys, key)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.savez(os.path.join(out_dir, "r0.npz"), full=full.numpy(), keys=torch.stack(keys).numpy(), tmax=t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 7, 8])       # nx = 40: 3 and 7 do not divide it (the padded last slab; 7 x 6 = 42 rows staged), 8 x 5 = 40 is the in-place shape of an 8-GPU node
def test_sharded_map_build_and_batch_split(tmp_path, oracle, world):
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z = np.load(str(tmp_path / "r0.npz"))
    from uneven_planner_amd import scenes
    xyz = scenes.make_hill_cloud(n_side=100, half=1.5)
    g = oracle.OracleGrid(size_x=2.0, size_y=2.0)
    b = oracle.OracleMapBuilder(xyz=xyz)
    b.construct(g, map_params=dict(map_size_x=2.0, map_size_y=2.0), do_occ=False)
    cells, _ = g.get_cells()
    assert np.array_equal(z["full"].reshape(-1, 4), cells)          # slab build + all-gather == single build, bit for bit
    keys = z["keys"]
    assert keys.shape == (world, 3) and len(set(keys.ravel().tolist())) == 3 * world   # disjoint problem sets
    if world < 2:
        return
    ref = [p["total_time"] for p in scenes.random_problems(3, seed0=1003)]
    assert np.allclose(keys[1], ref)
    assert abs(float(z["tmax"][0]) - 0.1 * world) < 1e-12                  # max over ranks


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python bench.py --gpus N` outside a launcher re-executes itself as N ranks with the torch.distributed.run environment contract
    (bench.spawn_ranks).  The ranks are asked to echo their environment instead of touching a GPU (UPH_BENCH_SPAWN_ECHO)."""
    import json
    import subprocess
    env = dict(os.environ, UPH_BENCH_SPAWN_ECHO="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["WORLD_SIZE"] == "4" and line["RANK"] == "0" and line["LOCAL_RANK"] == "0" and line["MASTER_ADDR"] == "127.0.0.1" and line["gpus"] == 4
    # under a launcher (WORLD_SIZE set) the script does not spawn again
    env2 = dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env2, timeout=120)
    assert json.loads(r2.stdout.strip().splitlines()[-1])["RANK"] == "1"


# ---- every host-side decision an 8-GPU run takes, on a box without GPUs -------------------------------------------------
def _plan(L, nx, n):
    import ctypes as C
    x0, x1, per, inpl = (C.c_int32 * n)(), (C.c_int32 * n)(), C.c_int32(0), C.c_int32(0)
    assert L.uph_multi_slab_plan(nx, n, x0, x1, C.byref(per), C.byref(inpl)) == 0
    return [(int(a), int(b)) for a, b in zip(x0, x1)], int(per.value), bool(inpl.value)


@pytest.mark.parametrize("nx", [200, 4000, 40, 7, 1])
def test_slab_plan_of_the_multi_gpu_build_for_worlds_1_to_8(nx):
    """uph_multi_slab_plan = the slab rule and the in-place / staged decision uph_map_build_multi and uph_map_fill_fbm_multi execute (the same
    functions, csrc/map_build.hip slabOf / slabsInPlace): slabs tile [0, nx) in order, equal the Python mirror's slab_bounds and bench.py's
    sharded build, in place exactly when the world divides the rows (hill grid: 8 | 200 in place, 7 staged through 7 x 29 = 203 padded rows)"""
    import uneven_planner_amd as U
    from uneven_planner_amd.uneven_map import slab_bounds
    L = U._lib.load()
    for n in range(1, 9):
        slabs, per, inpl = _plan(L, nx, n)
        assert per == -(-nx // n) and slabs == [slab_bounds(nx, g, n)[1:] for g in range(n)]
        assert slabs[0][0] == 0 and all(a[1] == b[0] for a, b in zip(slabs, slabs[1:])) and slabs[-1][1] == nx        # contiguous cover, in device order
        assert all(0 <= b - a <= per for a, b in slabs) and n * per >= nx
        assert inpl == (n > 1 and nx % n == 0)
        if not inpl and n > 1:
            assert n * per > nx or nx % n != 0                  # staging holds n x per rows >= nx: the gathered prefix is the whole grid
    if nx == 200:
        assert _plan(L, 200, 8) == ([(25 * g, 25 * g + 25) for g in range(8)], 25, True)           # BASELINE.json configs[3]: rank r owns [25 r, 25 r + 25)
        assert _plan(L, 200, 7)[1:] == (29, False) and _plan(L, 200, 7)[0][-1] == (174, 200)
    import ctypes as C
    assert L.uph_multi_slab_plan(0, 2, (C.c_int32 * 2)(), (C.c_int32 * 2)(), None, None) != 0


def test_batch_plan_of_the_multi_gpu_solve_for_worlds_2_to_8():
    """uph_multi_batch_plan = the split uph_optimize_batch_multi executes (the same function, csrc/unevenhip.hip dealShares): a partition of the
    batch, sizes within one of each other, descending predicted cost dealt round-robin (so the devices' predicted work agrees to within one
    problem's cost), deterministic, unreadable problems tolerated (their context's upload rejects them)"""
    import ctypes as C
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    from uneven_planner_amd.alm_traj_opt import pack_problems
    L = U._lib.load()
    probs = scenes.random_problems(203, seed0=4000)
    arr, keep = pack_problems(probs)
    B = len(probs)
    for n in range(2, 9):
        share, cost = (C.c_int32 * B)(), np.zeros(B)
        assert L.uph_multi_batch_plan(n, B, arr, share, cost.ctypes.data_as(C.POINTER(C.c_double))) == 0
        sh = np.array(share[:])
        sizes = np.bincount(sh, minlength=n)
        assert sh.min() == 0 and sh.max() == n - 1 and sizes.sum() == B and sizes.max() - sizes.min() <= 1
        order = np.argsort(-cost, kind="stable")
        assert np.array_equal(sh[order], np.arange(B) % n)                      # rank k of the cost order goes to device k mod n
        work = np.array([cost[sh == g].sum() for g in range(n)])
        assert work.max() - work.min() <= cost.max() + 1e-9                      # round-robin over a sorted list: the shares differ by less than one problem
        share2 = (C.c_int32 * B)()
        assert L.uph_multi_batch_plan(n, B, arr, share2, None) == 0 and share2[:] == share[:]
    assert cost.min() > 0 and cost.max() / cost.min() > 3.0                     # the batch really mixes short and long solves
    # fewer problems than devices: the first B devices get one each; a problem without arrays costs nothing and is dealt last
    share = (C.c_int32 * 3)()
    assert L.uph_multi_batch_plan(8, 3, arr, share, None) == 0 and sorted(share[:]) == [0, 1, 2]
    bad = (U._lib.Problem * 2)(arr[0], arr[1])
    bad[0].n_inner_yaw = 5
    bad[0].inner_yaw = None
    share = (C.c_int32 * 2)()
    assert L.uph_multi_batch_plan(2, 2, bad, share, None) == 0 and share[:] == [1, 0]
    assert L.uph_multi_batch_plan(0, 2, bad, share, None) != 0


def test_bench_single_process_dry_run_for_8_gpus():
    """`python bench.py --gpus 8 --single-process` (one host process driving eight devices through the C-ABI's multi-GPU entries): argument handling,
    the slab plan it will execute and the shape of its JSON line, with the echo hook instead of devices"""
    import json
    import subprocess
    env = dict(os.environ, UPH_BENCH_SPAWN_ECHO="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for n, inplace, last in ((8, True, [175, 200]), (7, False, [174, 200]), (2, True, [100, 200])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--single-process", "--steps", "3", "--warmup", "1", "--batch", "512"],
                           capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0, r.stderr
        line = json.loads(r.stdout.strip().splitlines()[-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                    "per_rank_ms_per_step", "per_gpu_kernel_ms", "converged_frac", "map_build_multi"):
            assert key in line, key
        assert line["n_gpus"] == n and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 512
        assert line["config"]["parallelism"] == "dp%d" % n and line["config"]["rccl_world"] == n and len(line["per_rank_ms_per_step"]) == n == len(line["per_gpu_kernel_ms"])
        d = line["dry_run"]
        assert d["all_gather_in_place"] == inplace and d["slabs"][-1] == last and len(d["slabs"]) == n and d["seed0_per_device"] == [1000 + 512 * g for g in range(n)]


@pytest.mark.parametrize("world", [2, 8])
def test_bench_rank_handshake_dry_run_over_gloo(world):
    """the Python of bench.py's N > 1 path -- world == --gpus, the collective self-test with known answers, the all-gathered map hash, the
    MAX-over-ranks timing -- run for worlds of 2 and 8 over gloo on CPU tensors (UPH_BENCH_DRY_DIST=1: the same helper functions the real run calls, no device),
    self-spawned ranks (bench.spawn_ranks).  A rank that holds another grid must end the job with a non-zero exit, and so must a launcher world that is not --gpus."""
    import json
    import subprocess
    base = dict(os.environ, UPH_BENCH_DRY_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "UPH_BENCH_SPAWN_ECHO", "UPH_BENCH_DRY_BAD_RANK"):
        base.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2"], capture_output=True, text=True, env=base, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == world and line["rccl_selftest"]["ok"] and line["rccl_selftest"]["world"] == world
    assert line["map_hash_identical_on_all_ranks"] is True and len(line["per_rank_ms_per_step"]) == world
    # the job's time is the slowest rank's (rank r sleeps 10 (1 + r) ms between the barriers)
    assert abs(line["ms_per_step"] - max(line["per_rank_ms_per_step"])) < 1e-9 and line["ms_per_step"] >= 10.0 * world / 2 * 0.9
    if world == 2:
        bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=dict(base, UPH_BENCH_DRY_BAD_RANK="1"), timeout=300)
        assert bad.returncode != 0 and "differs between ranks" in bad.stderr
        # a launcher whose world is not --gpus: refused by every rank
        env2 = dict(base, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
        mm = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True, env=env2, timeout=120)
        assert mm.returncode != 0 and "disagree" in (mm.stderr + mm.stdout)
