"""A/B for a would-be "small" residency class: a batch of SHORT problems only (footprint <= 160 KiB / 6), solved with the library given in
UNEVENHIP_LIB.  usage: python tools/short_class.py [B] [dmax]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dmax = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(B, seed0=1000, dmin=3.0, dmax=dmax, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
print('pieces: max', max(p['inner_xy'].shape[1] + 1 for p in probs), 'mean', np.mean([p['inner_xy'].shape[1] + 1 for p in probs]))
opt = U.ALMTrajOpt(m); opt.set_lanes(128); opt.upload(probs)
for _ in range(3):
    opt.set_rho(1.0); opt.solve(); st = opt.stats()
    print('kernel_ms %.2f  evals %d  traj/s %.0f' % (st['kernel_ms'], st['evals'], B / st['kernel_ms'] * 1e3))
