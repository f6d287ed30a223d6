"""phase microbenchmark of ONE trajectory alone on the GPU, per lane count (MODE 5: 50 evaluations at the initial point, cycles per sub-step):
python tools/microbench_single.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U              # noqa: E402
from uneven_planner_amd import scenes, _lib  # noqa: E402

m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
names = ['generate', 'expand', 'evalConsts', 'samples(chunk0)', 'scatterChunk0', 'adjoint', 'bookkeeping', 'total', 'gen:rhs', 'gen:knots', 'expand:pass', '-', 'adj:herm^T', 'adj:knots', 'adj:gamma', '-']
p = scenes.hill_problem()
print("hill problem: pieces", p["inner_xy"].shape[1] + 1, "yaw pieces", p["inner_yaw"].shape[0] + 1)
for lanes in (512, 256, 128, 64):
    opt = U.ALMTrajOpt(m); opt.set_lanes(lanes); opt.upload([p])
    _lib.check(opt.L.uph_microbench_batch(opt.h, 50), 'microbench')
    cy = opt.cycles().astype(np.float64)
    print('lanes', lanes, ' '.join('%s=%.0f' % (n_, v) for n_, v in zip(names, cy[0, :16]) if n_ != '-'), flush=True)
