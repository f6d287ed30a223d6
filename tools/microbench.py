import sys, os, numpy as np, ctypes as C
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes, _lib
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(8192, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
names = ['generate', 'expand', 'evalConsts', 'samples(chunk0)', 'scatterChunk0', 'adjoint', 'bookkeeping', 'total', 'gen:rhs', 'gen:knots', 'expand:pass', '-', 'adj:herm^T', 'adj:knots', 'adj:gamma', '-']
for B, wps in ((1, 2), (1024, 2), (8192, 2)):
    opt = U.ALMTrajOpt(m); opt.set_wps(wps); opt.set_lanes(128); opt.upload(probs[:B])
    _lib.check(opt.L.uph_microbench_batch(opt.h, 50), 'microbench')
    cy = opt.cycles().astype(np.float64)
    print('B', B, 'wps', wps, ' '.join('%s=%.0f' % (n_, v) for n_, v in zip(names, np.median(cy[:, :16], axis=0))))
