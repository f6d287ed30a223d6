"""Host view of the SE(2) terrain grid: the value-only queries of the reference's UnevenMap that stay on the host for the
untouched consumers (kinodynamic A*, RViz / SE(3) path publishing, the post-solve report):

  getTerrain            uneven_map/include/uneven_map/uneven_map.h:154-201
  getTerrainPos         :203-218
  getTerrainVariables   :221-256
  getTerrainSig         :389-396
  isInMap / boundIndex / posToIndex / indexToPos / toAddress   :398-471
  `.map` CSV cache      uneven_map/src/uneven_map.cpp:270-315 (read), :400-412 (write, 6 significant digits)

Works on the host copy of map_buffer (ncell x 4: z, sigma, zb.x, zb.y) that UnevenMap.download() fills from the device, so it
needs no GPU itself (and is unit-tested against the oracle in the CPU tier)."""
import math

import numpy as np


class HostGridView:
    def __init__(self, cells, map_size_x=10.0, map_size_y=10.0, xy_resolution=0.05, yaw_resolution=0.1):
        self.xy_resolution, self.yaw_resolution = float(xy_resolution), float(yaw_resolution)
        self.map_size = np.array([map_size_x, map_size_y, 2.0 * math.pi + 5e-2])            # uneven_map.cpp:96
        self.min_boundary, self.max_boundary = -self.map_size / 2.0, self.map_size / 2.0       # :99-101
        self.map_origin = self.min_boundary.copy()
        self.voxel_num = np.array([math.ceil(map_size_x / xy_resolution), math.ceil(map_size_y / xy_resolution),
                                   math.ceil(self.map_size[2] / yaw_resolution)], dtype=np.int64)   # :108-110
        self.cells = np.ascontiguousarray(cells, dtype=np.float64).reshape(int(self.voxel_num[0]), int(self.voxel_num[1]),
                                                                            int(self.voxel_num[2]), 4)

    # ---- index helpers
    def isInMap(self, pos):
        p = np.asarray(pos, dtype=np.float64)
        return bool(np.all(p >= self.min_boundary + 1e-4) and np.all(p <= self.max_boundary - 1e-4))

    @staticmethod
    def normSO2(yaw):
        while yaw < -math.pi:
            yaw += 2 * math.pi
        while yaw > math.pi:
            yaw -= 2 * math.pi
        return yaw

    def _corners(self, pos):
        x, y, w = float(pos[0]), float(pos[1]), float(pos[2])
        rx, rw = self.xy_resolution, self.yaw_resolution
        wm = self.normSO2(w - 0.5 * rw)
        ix = math.floor((x - 0.5 * rx - self.map_origin[0]) / rx)
        iy = math.floor((y - 0.5 * rx - self.map_origin[1]) / rx)
        iw = math.floor((wm - self.map_origin[2]) / rw)
        cx = (ix + 0.5) * rx + self.map_origin[0]
        cy = (iy + 0.5) * rx + self.map_origin[1]
        cw = (iw + 0.5) * rw + self.map_origin[2]
        d = ((x - cx) / rx, (y - cy) / rx, math.atan2(math.sin(w - cw), math.cos(w - cw)) / rw)
        nx, ny, nyaw = (int(v) for v in self.voxel_num)
        vals = np.zeros((2, 2, 2, 4))
        for a in (0, 1):
            for b in (0, 1):
                for c in (0, 1):
                    vals[a, b, c] = self.cells[min(max(ix + a, 0), nx - 1), min(max(iy + b, 0), ny - 1), (iw + c) % nyaw]
        return d, vals

    def getTerrain(self, pos):
        """-> (z, sigma, zb.x, zb.y); zeros outside the map (uneven_map.h:156-161)"""
        if not self.isInMap(pos):
            return np.zeros(4)
        d, v = self._corners(pos)
        v00 = v[0, 0, 0] * (1 - d[0]) + v[1, 0, 0] * d[0]
        v01 = v[0, 0, 1] * (1 - d[0]) + v[1, 0, 1] * d[0]
        v10 = v[0, 1, 0] * (1 - d[0]) + v[1, 1, 0] * d[0]
        v11 = v[0, 1, 1] * (1 - d[0]) + v[1, 1, 1] * d[0]
        v0 = v00 * (1 - d[1]) + v10 * d[1]
        v1 = v01 * (1 - d[1]) + v11 * d[1]
        return v0 * (1 - d[2]) + v1 * d[2]

    def getTerrainSig(self, pos):
        return float(self.getTerrain(pos)[1])

    def getTerrainVariables(self, pos):
        """invCosVphix, sinPhix, invCosVphiy, sinPhiy, cosXi, invCosXi, sigma"""
        z, sg, zx, zy = self.getTerrain(pos)
        c = math.sqrt(1.0 - zx * zx - zy * zy)
        cy_, sy_ = math.cos(pos[2]), math.sin(pos[2])
        t = cy_ * zx + sy_ * zy
        s = -(-sy_ * zx + cy_ * zy)
        r = math.sqrt(1.0 - t * t)
        return np.array([1.0 / r, -c * t / r, r / c, s / r, c, 1.0 / c, sg])

    def getTerrainPos(self, pos):
        """SE(3) pose on the terrain: rotation R (columns x_b, y_b, z_b) and position p"""
        z, sg, zx, zy = self.getTerrain(pos)
        zb = np.array([zx, zy, math.sqrt(1.0 - zx * zx - zy * zy)])
        xyaw = np.array([math.cos(pos[2]), math.sin(pos[2]), 0.0])
        yb = np.cross(zb, xyaw)
        yb /= np.linalg.norm(yb)
        xb = np.cross(yb, zb)
        return np.column_stack([xb, yb, zb]), np.array([pos[0], pos[1], z])

    # ---- `.map` text cache
    def write_map_file(self, path):
        nx, ny, nyaw = (int(v) for v in self.voxel_num)
        with open(path, "w") as f:
            for x in range(nx):
                for y in range(ny):
                    for w in range(nyaw):
                        z, s, a, b = self.cells[x, y, w]
                        f.write("%d,%d,%d,%.6g,%.6g,%.6g,%.6g\n" % (x, y, w, z, s, a, b))

    @classmethod
    def read_map_file(cls, path, map_size_x=10.0, map_size_y=10.0, xy_resolution=0.05, yaw_resolution=0.1):
        """constructMapInput (uneven_map.cpp:270-315): cells start as RXS2() zeros, lines may come in any order, out-of-range
        indices are dropped"""
        ncell = (math.ceil(map_size_x / xy_resolution) * math.ceil(map_size_y / xy_resolution) *
                 math.ceil((2.0 * math.pi + 5e-2) / yaw_resolution))
        view = cls(np.zeros((ncell, 4)), map_size_x, map_size_y, xy_resolution, yaw_resolution)
        nx, ny, nyaw = (int(v) for v in view.voxel_num)
        # the reference parses the four values with stold and narrows to double (uneven_map.cpp:296-299): two roundings, which a direct
        # string -> double conversion does not always reproduce (1 ulp on ~1e-4 of the values)
        arr = np.loadtxt(path, delimiter=",", dtype=np.longdouble).reshape(-1, 7)
        ix, iy, iw = arr[:, 0].astype(int), arr[:, 1].astype(int), arr[:, 2].astype(int)
        ok = (ix >= 0) & (iy >= 0) & (iw >= 0) & (ix < nx) & (iy < ny) & (iw < nyaw)
        view.cells[ix[ok], iy[ok], iw[ok]] = arr[ok, 3:7].astype(np.float64)
        return view

    # ---- binary side-car of the `.map` cache: header (magic, nx, ny, nyaw) + ncell x 4 float64, bit-exact
    MAGIC = b"UPHMAP01"

    def write_map_binary(self, path):
        with open(path, "wb") as f:
            f.write(self.MAGIC)
            f.write(np.asarray(self.voxel_num, dtype="<i8").tobytes())
            f.write(np.ascontiguousarray(self.cells, dtype="<f8").tobytes())

    @classmethod
    def read_map_binary(cls, path, map_size_x=10.0, map_size_y=10.0, xy_resolution=0.05, yaw_resolution=0.1):
        with open(path, "rb") as f:
            if f.read(8) != cls.MAGIC:
                raise ValueError("%s is not a binary .map side-car" % path)
            dims = np.frombuffer(f.read(24), dtype="<i8")
            cells = np.frombuffer(f.read(), dtype="<f8")
        view = cls(np.zeros((int(np.prod(dims)), 4)), map_size_x, map_size_y, xy_resolution, yaw_resolution)
        if not np.array_equal(dims, view.voxel_num) or cells.size != view.cells.size:
            raise ValueError("%s was written for a %s grid, this map is %s" % (path, dims.tolist(), view.voxel_num.tolist()))
        view.cells[...] = cells.reshape(view.cells.shape)
        return view
