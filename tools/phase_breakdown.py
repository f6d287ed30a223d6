import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U
from uneven_planner_amd import scenes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
probs = scenes.random_problems(B, seed0=1000, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
opt = U.ALMTrajOpt(m); opt.set_lanes(int(os.environ.get('UPH_LANES', '0'))); opt.upload(probs)
for _ in range(2):
    opt.set_rho(1.0); opt.solve()
st = opt.stats(); cy = opt.cycles().astype(np.float64)
names = ['generate', 'samples', 'scatter', 'adjoint', 'twoloop', 'after twoloop/eval -> next eval', 'total']
tot = cy[:, 6].sum()
print('B', B, 'kernel_ms', st['kernel_ms'], 'evals', st['evals'], 'iters', st['lbfgs_iters'])
print('max total cycles', cy[:, 6].max(), ' => clock MHz ~', cy[:, 6].max() / (st['kernel_ms'] * 1e3))
for k, nme in enumerate(names[:6]):
    print('%-28s %5.1f %%   cycles/eval %9.0f' % (nme, 100 * cy[:, k].sum() / tot, cy[:, k].sum() / st['evals']))
print('eval end -> twoloop start cycles/eval %9.0f' % (cy[:, 7].sum() / st['evals']))
# packing: sum of workgroup cycles over the 1024 resident slots (256 CUs x 4) against the launch duration (100 MHz-free estimate: the
# shader clock is read from the longest-lived workgroup of an unloaded run; here the ratio to the measured launch is what matters)
slots = 1024 if B >= 2304 else min(B, 512)
clk = float(os.environ.get('UPH_CLK_MHZ', '2400')) * 1e3
print('packing: sum(cycles)/slots = %.1f ms at %.0f MHz, longest workgroup %.1f ms, launch %.1f ms' % (tot / slots / clk, clk / 1e3, cy[:, 6].max() / clk, st['kernel_ms']))
