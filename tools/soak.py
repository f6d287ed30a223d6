import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd())
import torch
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.set_cells(scenes.analytic_cells())
probs = scenes.random_problems(3000, seed0=7000, dmin=3.0, dmax=7.0)
opt = U.ALMTrajOpt(m)
# the front end on the same map, interleaved with the solves: searches of every batch size through one handle (more queries than workspaces included)
mh = U.UnevenMap(); mh.build(scenes.make_hill_cloud())
nx, ny = int(mh.voxel_num[0]), int(mh.voxel_num[1])
QS, QG = scenes.random_queries(6000, seed0=9000, occ_r2=mh.occ_r2_buffer, grid=(nx, ny, mh.xy_resolution, mh.map_origin[0], mh.map_origin[1]))
ka = U.KinoAstar(mh, slots=1024)
# automatic workspaces (slots = 0, what plan() of the adapters uses): allocated for the batch that arrives, grown for a larger one, same results
ka_auto = U.KinoAstar(mh)
grown = []
for nq in (1, 300, 5000):
    ra = ka_auto.plan_batch(QS[:nq], QG[:nq], path_cap=256)
    grown.append(ka_auto.L.uph_kino_slots(ka_auto.h))
    rb = ka.plan_batch(QS[:nq], QG[:nq], path_cap=256)
    assert all(a["status"] == b["status"] and a["iter_num"] == b["iter_num"] and np.array_equal(a["path"], b["path"]) for a, b in zip(ra, rb)), nq
print("automatic search workspaces after batches of 1 / 300 / 5000 queries:", grown)
assert grown[0] == 16 and grown[1] == 300 and grown[2] >= 4096 and grown[2] <= 5000, grown
n_found = n_q = 0
free0 = torch.cuda.mem_get_info()[0]
rng = np.random.default_rng(0)
t0 = time.time()
for it in range(int(os.environ.get("UPH_SOAK_ITERS", "40"))):      # UPH_SOAK_ITERS=1500: a three-minute soak
    B = int(rng.choice([1, 2, 3, 64, 200, 256, 257, 511, 512, 1000, 2303, 2304, 3000]))
    idx = rng.choice(3000, B, replace=False)
    opt.set_rho(1.0)
    out = opt.optimize_batch([probs[i] for i in idx])
    assert all(o["ret"] in (0, 2) for o in out), (it, B)
    if it % 4 == 0:
        nq = int(rng.choice([1, 7, 300, 1024, 1025, 5000]))
        qi = rng.choice(6000, nq, replace=False)
        r = ka.plan_batch(QS[qi], QG[qi], path_cap=256)
        assert all(q["status"] in (0, 1, 2, 3, 4) for q in r), it
        n_found += sum(q["status"] == 0 for q in r); n_q += nq
    if it % 5 == 0:
        m.frontend_query(np.column_stack([rng.uniform(-5, 5, 5000), rng.uniform(-5, 5, 5000), rng.uniform(-3, 3, 5000)]))
    if it == 10:
        free1 = torch.cuda.mem_get_info()[0]
free2 = torch.cuda.mem_get_info()[0]
print("%s mixed batches in %.1f s; free memory start %.0f MB, after 10 %.0f MB, end %.0f MB" % (os.environ.get("UPH_SOAK_ITERS", "40"), time.time() - t0, free0 / 2**20, free1 / 2**20, free2 / 2**20))
assert free1 - free2 < 64 * 2**20, "device memory keeps growing"
print("front end: %d of %d queries found a path" % (n_found, n_q))
assert n_found > 0.9 * n_q
print("soak ok")
