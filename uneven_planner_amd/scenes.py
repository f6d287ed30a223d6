"""Synthetic scenes and problem batches, plus the PCD reader used by UnevenMap.init.

The reference's hill.pcd is absent (/root/reference/.MISSING_LARGE_BLOBS:13), so the "hill" scene is a
synthetic cloud with the same count / extent / density as desert.pcd: an analytic two-bump heightfield on
[-6,6]^2 sampled on a jittered 316x316 lattice (PCG64 seed 20230906), stored as float32.
"""
import math

import numpy as np

from .resample import make_problem

HILL_SEED = 20230906


def hill_height(x, y):
    return (0.4 + 0.6 * np.exp(-((x - 1.0) ** 2 + (y + 0.5) ** 2) / 6.0)
            + 0.35 * np.exp(-((x + 2.0) ** 2 + (y - 2.0) ** 2) / 3.0) + 0.05 * np.sin(1.3 * x) * np.cos(0.9 * y))


def hill_grad(x, y):
    e1 = 0.6 * np.exp(-((x - 1.0) ** 2 + (y + 0.5) ** 2) / 6.0)
    e2 = 0.35 * np.exp(-((x + 2.0) ** 2 + (y - 2.0) ** 2) / 3.0)
    hx = e1 * (-(x - 1.0) / 3.0) + e2 * (-2.0 * (x + 2.0) / 3.0) + 0.05 * 1.3 * np.cos(1.3 * x) * np.cos(0.9 * y)
    hy = e1 * (-(y + 0.5) / 3.0) + e2 * (-2.0 * (y - 2.0) / 3.0) - 0.05 * 0.9 * np.sin(1.3 * x) * np.sin(0.9 * y)
    return hx, hy


def make_hill_cloud(seed=HILL_SEED, n_side=316, half=6.0, height=hill_height):
    """(n_side^2, 3) float32 points: jittered lattice on [-half, half]^2 (n_side=316 -> 99 856 points)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    step = 2.0 * half / n_side
    ii, jj = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij")
    x = -half + (ii + rng.random(ii.shape)) * step
    y = -half + (jj + rng.random(jj.shape)) * step
    z = height(x, y)
    return np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1).astype(np.float32)


def grid_dims(size_x=10.0, size_y=10.0, xy_res=0.05, yaw_res=0.1):
    """uneven_map.cpp:96-110."""
    span = 2.0 * math.pi + 5e-2
    return (int(math.ceil(size_x / xy_res)), int(math.ceil(size_y / xy_res)), int(math.ceil(span / yaw_res)))


def analytic_cells(height=hill_height, grad=hill_grad, size_x=10.0, size_y=10.0, xy_res=0.05, yaw_res=0.1,
                   sigma_scale=0.02):
    """Cells (nx*ny*nyaw, 4) {z, sigma, zbx, zby} from an analytic heightfield WITHOUT the plane fit: normal from the
    height gradient, a smooth yaw-dependent pseudo sigma.  Test helper for the optimiser (fast, smooth, yaw-varying);
    real maps come from UnevenMap.build (the plane-fit kernel)."""
    nx, ny, nyaw = grid_dims(size_x, size_y, xy_res, yaw_res)
    ox, oy, oyaw = -size_x / 2.0, -size_y / 2.0, -(2.0 * math.pi + 5e-2) / 2.0
    xs = (np.arange(nx) + 0.5) * xy_res + ox
    ys = (np.arange(ny) + 0.5) * xy_res + oy
    yaws = (np.arange(nyaw) + 0.5) * yaw_res + oyaw
    X, Y, W = np.meshgrid(xs, ys, yaws, indexing="ij")
    px, py = X + 0.12 * np.cos(W), Y + 0.12 * np.sin(W)
    z = height(px, py)
    hx, hy = grad(px, py)
    nrm = np.sqrt(hx * hx + hy * hy + 1.0)
    zbx, zby = -hx / nrm, -hy / nrm
    sig = sigma_scale * (hx * hx + hy * hy) * (1.0 + 0.3 * np.cos(W - np.arctan2(hy, hx + 1e-12)) ** 2)
    return np.stack([z, sig, zbx, zby], axis=-1).reshape(-1, 4)


def random_problems(n, seed0=1000, half=4.5, dmin=3.0, dmax=10.0, occ_r2=None, grid=None, **mk):
    """SURVEY.md 8d config 3 protocol: one PCG64 stream per problem (seed0+i): start, goal ~ U([-half,half]^2),
    yaw ~ U(-pi,pi); accept when dmin <= |goal-start| <= dmax (and both cells free when an occupancy layer is given)."""
    out = []
    for i in range(n):
        rng = np.random.Generator(np.random.PCG64(seed0 + i))
        while True:
            s = np.array([rng.uniform(-half, half), rng.uniform(-half, half), rng.uniform(-math.pi, math.pi)])
            g = np.array([rng.uniform(-half, half), rng.uniform(-half, half), rng.uniform(-math.pi, math.pi)])
            d = math.hypot(g[0] - s[0], g[1] - s[1])
            if not (dmin <= d <= dmax):
                continue
            if occ_r2 is not None and grid is not None:
                nx, ny, res, ox, oy = grid
                ok = True
                for p in (s, g):
                    ix, iy = int(math.floor((p[0] - ox) / res)), int(math.floor((p[1] - oy) / res))
                    if ix < 0 or iy < 0 or ix >= nx or iy >= ny or occ_r2[ix * ny + iy]:
                        ok = False
                if not ok:
                    continue
            out.append(make_problem(s, g, **mk))
            break
    return out


def random_queries(n, seed0=1000, half=4.5, dmin=3.0, dmax=10.0, occ_r2=None, grid=None):
    """start / goal poses of random_problems' protocol (same streams, same acceptance) without the initial-path stage: (n, 3), (n, 3) --
    the inputs of the front-end search (KinoAstar::plan)"""
    S, G = np.zeros((n, 3)), np.zeros((n, 3))
    for i in range(n):
        rng = np.random.Generator(np.random.PCG64(seed0 + i))
        while True:
            s = np.array([rng.uniform(-half, half), rng.uniform(-half, half), rng.uniform(-math.pi, math.pi)])
            g = np.array([rng.uniform(-half, half), rng.uniform(-half, half), rng.uniform(-math.pi, math.pi)])
            d = math.hypot(g[0] - s[0], g[1] - s[1])
            if not (dmin <= d <= dmax):
                continue
            if occ_r2 is not None and grid is not None:
                nx, ny, res, ox, oy = grid
                ok = True
                for p in (s, g):
                    ix, iy = int(math.floor((p[0] - ox) / res)), int(math.floor((p[1] - oy) / res))
                    if ix < 0 or iy < 0 or ix >= nx or iy >= ny or occ_r2[ix * ny + iy]:
                        ok = False
                if not ok:
                    continue
            S[i], G[i] = s, g
            break
    return S, G


def batch_share(total, rank, world):
    """strong-scaling split of ONE batch over the ranks (configs[4]): rank r solves problems [lo, lo + count) of the batch"""
    per = -(-int(total) // int(world))
    lo = min(rank * per, total)
    return lo, min(per, total - lo)


def local_problems(n, seed0=5000, half=495.0, dmin=4.0, dmax=14.0, occ_r2=None, grid=None, max_pieces=64, xlim=None, **mk):
    """SURVEY.md 8c row 5 (BASELINE.json configs[4]) protocol: one PCG64 stream per problem (seed0+i): start ~ U([-half,half]^2), goal at
    distance U[dmin,dmax] in a uniform direction, both yaws ~ U(-pi,pi); accepted when both cells are free, the goal is inside the
    square and the resampled path has at most max_pieces position pieces (UPH_MAX_PIECE_XY).  xlim = (lo, hi): starts drawn with x in
    that interval (the x-slab of a tile owner)."""
    out = []
    for i in range(n):
        rng = np.random.Generator(np.random.PCG64(seed0 + i))
        while True:
            s = np.array([rng.uniform(-half, half) if xlim is None else rng.uniform(xlim[0], xlim[1]), rng.uniform(-half, half), rng.uniform(-math.pi, math.pi)])
            d, th = rng.uniform(dmin, dmax), rng.uniform(-math.pi, math.pi)
            g = np.array([s[0] + d * math.cos(th), s[1] + d * math.sin(th), rng.uniform(-math.pi, math.pi)])
            if abs(g[0]) > half or abs(g[1]) > half:
                continue
            if occ_r2 is not None and grid is not None:
                nx, ny, res, ox, oy = grid
                ok = True
                for p in (s, g):
                    ix, iy = int(math.floor((p[0] - ox) / res)), int(math.floor((p[1] - oy) / res))
                    if ix < 0 or iy < 0 or ix >= nx or iy >= ny or occ_r2[ix * ny + iy]:
                        ok = False
                if not ok:
                    continue
            pr = make_problem(s, g, **mk)
            if pr["inner_xy"].shape[1] + 1 > max_pieces:
                continue
            out.append(pr)
            break
    return out


def hill_problem():
    """launch pose (4.3,-4.3,1.57) (plan_manager/launch/run_hill.launch:7-10) -> goal (-3.5,3.5,2.36)."""
    return make_problem((4.3, -4.3, 1.57), (-3.5, 3.5, 2.36))


def read_pcd(path):
    """PCD v0.7 reader (ASCII header + `DATA binary` / `DATA ascii`), fields x y z only -- what
    pcl::PCDReader::read<pcl::PointXYZ> extracts in uneven_map.cpp:130-131.  Returns (n,3) float32."""
    with open(path, "rb") as f:
        fields, sizes, counts, npts, data = [], [], [], 0, None
        while True:
            line = f.readline()
            if not line:
                raise IOError("bad PCD header: " + path)
            txt = line.decode("ascii", "replace").strip()
            if not txt or txt.startswith("#"):
                continue
            key, *rest = txt.split()
            if key == "FIELDS":
                fields = rest
            elif key == "SIZE":
                sizes = [int(r) for r in rest]
            elif key == "COUNT":
                counts = [int(r) for r in rest]
            elif key == "POINTS":
                npts = int(rest[0])
            elif key == "DATA":
                data = rest[0]
                break
        if not counts:
            counts = [1] * len(fields)
        offs, stride = {}, 0
        for nme, sz, ct in zip(fields, sizes, counts):
            offs[nme] = stride
            stride += sz * ct
        if data == "binary":
            raw = np.frombuffer(f.read(stride * npts), dtype=np.uint8).reshape(npts, stride)
            cols = [raw[:, offs[k]:offs[k] + 4].copy().view(np.float32)[:, 0] for k in ("x", "y", "z")]
            return np.stack(cols, axis=1).astype(np.float32)
        arr = np.loadtxt(f, dtype=np.float32).reshape(npts, -1)
        return np.stack([arr[:, offs[k] // 4] for k in ("x", "y", "z")], axis=1)
