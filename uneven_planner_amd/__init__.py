"""uneven_planner_amd -- MI355X (gfx950) back-end for the trajectory optimiser of ZJU-FAST-Lab/uneven_planner:
hand-written HIP kernels behind a C-ABI (include/uneven_hip.h) that drop in behind ALMTrajOpt::optimizeSE2Traj and
UnevenMap.  This package holds the kernels (csrc/), the ctypes binding (_lib) and host-side mirrors of the two reference
interfaces (alm_traj_opt.ALMTrajOpt, uneven_map.UnevenMap) and of their front-end caller (kino_astar.KinoAstar)."""
from . import _lib  # noqa: F401
from .alm_traj_opt import ALMTrajOpt, HILL_OPT_PARAMS, SE2Traj  # noqa: F401
from .uneven_map import UnevenMap, HILL_MAP_PARAMS  # noqa: F401
from .kino_astar import KinoAstar, HILL_KINO_PARAMS  # noqa: F401
