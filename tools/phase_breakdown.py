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
# residency timeline from the 100 MHz start / end stamps of every workgroup: effective shader clock, and how long the launch runs
# below full residency (the tail)
t0, t1 = cy[:, 14], cy[:, 15]
ok = t1 > t0
if ok.any():
    wall = (t1[ok] - t0[ok]) / 100e6
    print('effective shader clock %.0f MHz (median over workgroups)' % np.median(cy[ok, 6] / wall / 1e6))
    span = (t1[ok].max() - t0[ok].min()) / 100e6
    ev = np.concatenate([np.stack([t0[ok], np.ones(ok.sum())], 1), np.stack([t1[ok], -np.ones(ok.sum())], 1)])
    ev = ev[np.argsort(ev[:, 0], kind='stable')]
    res = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0]) / 100e6
    full = res[:-1].max()
    busy = (res[:-1] * dt).sum() / full
    print('launch span %.1f ms, peak residency %d workgroups, residency-weighted time %.1f ms (= %.1f %% of the span), time below 90 %% residency %.1f ms'
          % (span * 1e3, full, busy * 1e3, 100 * busy / span, dt[res[:-1] < 0.9 * full].sum() * 1e3))
# two-loop profile (library built with -DUPH_TL_PROF): per call, cycles of loop 1 / loop 2 / tail, first-row latency, mean bound
if cy[:, 12].sum() > 0:
    calls = cy[:, 12].sum()
    print('two-loop per call: loop1 %.0f loop2 %.0f tail %.0f  first-row wait %.0f  mean bound %.1f  calls/traj %.0f'
          % (cy[:, 8].sum() / calls, cy[:, 9].sum() / calls, cy[:, 10].sum() / calls, cy[:, 13].sum() / calls, cy[:, 11].sum() / calls, calls / len(cy)))
# barrier profile (library built with -DUPH_BAR_PROF): cycles per evaluation a wave spends between reaching a workgroup barrier and leaving it
if os.environ.get('UPH_BAR_PROF'):
    ev = st['evals']
    print('barrier wait, wave 0, cycles/eval: loops+reductions %.0f  two-loop %.0f  knot solve %.0f  scatter %.0f' % tuple(cy[:, 8 + q].sum() / ev for q in range(4)))
    print('barrier wait, wave 1, cycles/eval: loops+reductions+scatter %.0f  two-loop+knot solve (chains it only waits for) %.0f' % (cy[:, 12].sum() / ev, cy[:, 13].sum() / ev))
    print('whole solve cycles/eval %.0f' % (cy[:, 6].sum() / ev))
