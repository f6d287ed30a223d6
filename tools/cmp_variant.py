import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.set_cells(scenes.analytic_cells())
probs = scenes.random_problems(64, seed0=1000)
opt = U.ALMTrajOpt(m); opt.set_lanes(128)
opt.set_rho(1.0); out = opt.optimize_batch(probs)
opt.set_rho(1.0); out2 = opt.optimize_batch(probs)
print("deterministic:", all(np.array_equal(a["x"], b["x"]) for a, b in zip(out, out2)))
np.save(sys.argv[1], np.concatenate([o["x"] for o in out]))
print("evals", sum(o["evals"] for o in out))
