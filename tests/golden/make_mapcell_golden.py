"""Generates tests/golden/mapcells_golden.npz: cells of the SE(2) terrain grid of the reference's desert cloud (fixture
desert_xyz.npz) from an INDEPENDENT numpy restatement of UnevenMap::constructMap + filter (uneven_map/src/uneven_map.cpp:317-417,
5-43): brute-force searches over the whole cloud instead of kd-trees (PCL's result sets are defined by float predicates, evaluated here
in numpy float32), numpy.linalg.eigh instead of a Jacobi sweep / Eigen::EigenSolver.  300 random cells (x, y, yaw), both refinement
iterations, parameters of run_hill.yaml.  The map-build parity of the device and of the oracle otherwise rests on two solvers by the same
author; this fixture is the second derivation on real data.
Run:  python tests/golden/make_mapcell_golden.py   (numpy only; deterministic)."""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PAR = dict(iter_num=2, map_size=10.0, ell=(0.2, 0.1, 0.1), xy_res=0.05, yaw_res=0.1)


def crop_and_voxel(xyz):
    """pcl::CropBox [-10,10]^2 x [-0.01,5] (inclusive, float) then pcl::VoxelGrid leaf 0.01: float centroid per voxel, output ordered by
    voxel index (uneven_map.cpp:133-143)"""
    p = xyz[np.all(np.isfinite(xyz), axis=1)]
    mn, mx = np.array([-10, -10, -0.01], np.float32), np.array([10, 10, 5], np.float32)
    p = p[np.all(p >= mn, axis=1) & np.all(p <= mx, axis=1)]
    inv = np.float32(1.0) / np.float32(0.01)
    lo = np.floor(p.min(axis=0) * inv).astype(np.int64)
    hi = np.floor(p.max(axis=0) * inv).astype(np.int64)
    div = hi - lo + 1
    ijk = (np.floor(p * inv) - lo.astype(np.float32)).astype(np.int64)
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    key, p = key[order], p[order]
    out = []
    start = 0
    for end in list(np.flatnonzero(np.diff(key)) + 1) + [len(key)]:
        blk = p[start:end]
        s = np.zeros(3, np.float32)
        for q in blk:                       # float accumulation in point order, as PCL does
            s = s + q
        out.append(s / np.float32(len(blk)))
        start = end
    return np.array(out, dtype=np.float32)


def fit_cell(cloud, ix, iy, iw):
    n3 = 2.0 * math.pi + 5e-2
    ox = -PAR["map_size"] / 2.0
    cx, cy = (ix + 0.5) * PAR["xy_res"] + ox, (iy + 0.5) * PAR["xy_res"] + ox
    yaw = (iw + 0.5) * PAR["yaw_res"] - n3 / 2.0
    ex, ey, ez = PAR["ell"]
    box_r = max(ex, ey, ez)
    c64 = cloud.astype(np.float64)
    z, sig, zbx, zby, cc = 0.0, 0.0, 0.0, 0.0, 1.0
    counts = []
    for it in range(PAR["iter_num"]):
        xyaw = np.array([math.cos(yaw), math.sin(yaw), 0.0])
        zb = np.array([zbx, zby, cc])
        yb = np.cross(zb, xyaw)
        yb = yb / np.linalg.norm(yb)
        xb = np.cross(yb, zb)
        w = np.array([cx + xb[0] * 0.12, cy + xb[1] * 0.12, z])
        if it == 0:                                                    # nearest neighbour in the xy plane, float metric
            q = w[:2].astype(np.float32)
            d = (cloud[:, 0] - q[0]) ** 2 + (cloud[:, 1] - q[1]) ** 2
            w[2] = float(cloud[int(np.argmin(d)), 2])
        q = w.astype(np.float32)
        dx, dy, dz = cloud[:, 0] - q[0], cloud[:, 1] - q[1], cloud[:, 2] - q[2]
        d = (dx * dx + dy * dy) + dz * dz                               # float32 throughout (L2_Simple<float>)
        cand = c64[d < np.float32(box_r) * np.float32(box_r)]
        sub = cand - w
        inrob = np.stack([sub @ xb / ex, sub @ yb / ey, sub @ zb / ez], axis=1)
        pts = cand[(inrob ** 2).sum(axis=1) < 1.0]
        counts.append(len(pts))
        if len(pts) == 0:
            z, sig, zbx, zby, cc = w[2], 0.0, 0.0, 0.0, 1.0
            continue
        mean = pts.mean(axis=0)
        cov = (pts - mean).T @ (pts - mean) / len(pts)
        D, V = np.linalg.eigh(cov)
        n = V[:, 0] / np.linalg.norm(V[:, 0])
        if n[2] < 0:
            n = -n
        sig = D[0] / D.sum() * 3.0
        if math.isnan(sig):
            sig, n = 1.0, np.array([1.0, 0.0, 0.0])
        z, zbx, zby = mean[2], n[0], n[1]
        cc = math.sqrt(1.0 - zbx * zbx - zby * zby)
    fit_cell.min_points = min(counts)       # (side channel: the smallest fit of the cell's iterations -- fewer than 4 points make the plane ambiguous)
    return np.array([z, sig, zbx, zby]), len(pts)


def main(scene="desert"):
    """scene: desert (mapcells_golden.npz, the round-1 fixture), forest / mountain (mapcells_<scene>_golden.npz): forest.pcd is the cloud whose
    voxel filter merges points (141 068 -> 137 491 leaves with PCL's float leaf arithmetic; 137 490 if the leaf index is formed in double) and
    holds vertical structure (trunks: near-degenerate fits), mountain.pcd the sparse one (50 000 points: ~17 per fit)"""
    xyz = np.load(os.path.join(HERE, "%s_xyz.npz" % scene))["xyz"]
    cloud = crop_and_voxel(xyz)
    print("cloud after crop + voxel:", cloud.shape)
    rng = np.random.default_rng(20240917 if scene == "desert" else (20260927 if scene == "forest" else 20260928))
    idx, cells, npts, nmin = [], [], [], []
    lo, hi = (10, 190) if scene != "mountain" else (30, 170)
    while len(idx) < 300:
        ix, iy, iw = int(rng.integers(lo, hi)), int(rng.integers(lo, hi)), int(rng.integers(64))
        c, k = fit_cell(cloud, ix, iy, iw)
        idx.append((ix, iy, iw)); cells.append(c); npts.append(k); nmin.append(fit_cell.min_points)
    name = "mapcells_golden.npz" if scene == "desert" else "mapcells_%s_golden.npz" % scene
    extra = {} if scene == "desert" else dict(npts_min=np.array(nmin, dtype=np.int32))      # (the desert fixture keeps its round-1 content: no fit below 33 points there)
    np.savez_compressed(os.path.join(HERE, name), idx=np.array(idx, dtype=np.int32), cells=np.array(cells), npts=np.array(npts, dtype=np.int32),
                        cloud_points=np.array(cloud.shape[0]), **extra)
    print("%s: points per fit: min %d median %d max %d" % (name, min(npts), int(np.median(npts)), max(npts)))


if __name__ == "__main__":
    import sys
    main(sys.argv[1] if len(sys.argv) > 1 else "desert")
