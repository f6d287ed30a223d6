import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd())
import torch
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.set_cells(scenes.analytic_cells())
probs = scenes.random_problems(3000, seed0=7000, dmin=3.0, dmax=7.0)
opt = U.ALMTrajOpt(m)
free0 = torch.cuda.mem_get_info()[0]
rng = np.random.default_rng(0)
t0 = time.time()
for it in range(40):
    B = int(rng.choice([1, 2, 3, 64, 200, 256, 257, 511, 512, 1000, 2303, 2304, 3000]))
    idx = rng.choice(3000, B, replace=False)
    opt.set_rho(1.0)
    out = opt.optimize_batch([probs[i] for i in idx])
    assert all(o["ret"] in (0, 2) for o in out), (it, B)
    if it % 5 == 0:
        m.frontend_query(np.column_stack([rng.uniform(-5, 5, 5000), rng.uniform(-5, 5, 5000), rng.uniform(-3, 3, 5000)]))
    if it == 10:
        free1 = torch.cuda.mem_get_info()[0]
free2 = torch.cuda.mem_get_info()[0]
print("40 mixed batches in %.1f s; free memory start %.0f MB, after 10 %.0f MB, end %.0f MB" % (time.time() - t0, free0 / 2**20, free1 / 2**20, free2 / 2**20))
assert free1 - free2 < 64 * 2**20, "device memory keeps growing"
print("soak ok")
