"""Would one wave per trajectory with EIGHT workgroups per CU beat two waves x four workgroups?  (round 4.)  With one wave nothing idles at barriers while
another wave walks a chain (41.6 % of wave 1's cycles today, profiles/r04a_wait_decomposition.txt); the price is LDS: eight workgroups per CU need
<= 20 KB each.  This runs batches of SHORT problems (whose footprint allows eight per CU already with today's layout) with 128 lanes and with 64 lanes
(register-capped kernel, two waves per SIMD) on identical inputs.  usage: python tools/single_wave_class.py [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import uneven_planner_amd as U              # noqa: E402
from uneven_planner_amd import scenes       # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = U.UnevenMap(); m.build(scenes.make_hill_cloud())
nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
for dmax in (4.2, 5.5, 7.0):
    probs = scenes.random_problems(B, seed0=1000, dmin=3.0, dmax=dmax, occ_r2=m.occ_r2_buffer, grid=(nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]))
    pc = np.array([p["inner_xy"].shape[1] + 1 for p in probs])
    lds64 = 8 * (72 * pc.max() + 1169) + 128
    print("goals 3 .. %.1f m: pieces max %d mean %.1f; 64-lane footprint of the longest %d B -> %d workgroups per CU" % (dmax, pc.max(), pc.mean(), lds64, 163840 // lds64))
    res = {}
    for lanes, wps in ((128, 0), (64, 2)):
        opt = U.ALMTrajOpt(m); opt.set_lanes(lanes); opt.set_wps(wps); opt.upload(probs)
        ms = []
        for _ in range(3):
            opt.set_rho(1.0); opt.solve(); st = opt.stats(); ms.append(st["kernel_ms"])
        out = opt.download(full=False)
        res[lanes] = (np.mean(ms[1:]), st["evals"], np.array([o["cost"] for o in out]), np.array([o["ret"] for o in out]))
        print("   %3d lanes: solve kernel %.1f ms (%s)  evals %d  converged %.3f" % (lanes, np.mean(ms[1:]), ["%.1f" % v for v in ms], st["evals"], (res[lanes][3] == 0).mean()))
        del opt
    print("   64 lanes / 128 lanes: %.3f in time; median |cost difference| %.1e" % (res[64][0] / res[128][0], np.median(np.abs(res[64][2] - res[128][2]) / np.abs(res[128][2]))))
