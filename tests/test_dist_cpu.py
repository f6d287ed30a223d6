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


@pytest.mark.parametrize("world", [2, 3])       # 3: nx = 40 does not divide -> the padded last slab
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
