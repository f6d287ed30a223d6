import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
for lanes in (512, 256, 128, 64):
    opt = U.ALMTrajOpt(m); opt.set_lanes(lanes); opt.upload([scenes.hill_problem()])
    for _ in range(2):
        opt.set_rho(1.0); opt.solve()
    st = opt.stats(); cy = opt.cycles().astype(np.float64)[0]
    names = ['generate', 'samples', 'scatter', 'adjoint', 'twoloop', 'after', 'total', 'eval->twoloop']
    print('lanes', lanes, 'kernel_ms %.2f' % st['kernel_ms'], 'evals', st['evals'], 'iters', st['lbfgs_iters'], ' '.join('%s=%.0f' % (n_, cy[k] / st['evals']) for k, n_ in enumerate(names)))
