import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(8192, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
res = {}
for bits in (64, 32):
    opt = U.ALMTrajOpt(m); opt.set_sample_precision(bits); opt.upload(probs)
    opt.init_scaling_batch()
    f, g = opt.eval_batch(None, repeat=20)
    ms_eval = opt.stats()["kernel_ms"]
    opt.upload(probs)
    opt.set_rho(1.0); opt.solve()
    opt.set_rho(1.0); opt.solve()
    st = opt.stats(); out = opt.download(); rep = opt.getMaxVxAxAyCurAttSig()
    rets = np.array([o["ret"] for o in out])
    res[bits] = (f, g, out, rets, rep)
    print("bits", bits, "eval x20 %.2f ms" % ms_eval, "solve %.1f ms" % st["kernel_ms"], "evals/traj %.1f" % (st["evals"] / 8192), "converged %.3f" % (rets == 0).mean(),
          "median cost %.4f" % np.median([o["cost"] for o in out]))
f64, g64 = res[64][0], res[64][1]
f32, g32 = res[32][0], res[32][1]
print("eval: f rel diff median %.2e max %.2e" % (np.median(np.abs(f32 - f64) / np.abs(f64)), (np.abs(f32 - f64) / np.abs(f64)).max()))
gd = np.array([np.abs(a - b).max() / np.abs(b).max() for a, b in zip(g32, g64)])
print("eval: grad rel diff median %.2e max %.2e" % (np.median(gd), gd.max()))
c64 = np.array([o["cost"] for o in res[64][2]]); c32 = np.array([o["cost"] for o in res[32][2]])
print("solve: cost rel diff median %.2e p90 %.2e" % (np.median(np.abs(c32 - c64) / np.abs(c64)), np.percentile(np.abs(c32 - c64) / np.abs(c64), 90)))
for bits in (64, 32):
    rets, rep = res[bits][3], res[bits][4]; conv = rets == 0
    print("bits", bits, "feasible among converged: vx<=0.5*1.05 %.4f  sigma<=0.055 %.4f  att %.4f" % ((np.abs(rep[conv, 0]) < 0.525).mean(), (rep[conv, 5] < 0.055).mean(), (-rep[conv, 4] > 0.8 * 0.98).mean()))
