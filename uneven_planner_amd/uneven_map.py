"""Host-side mirror of the reference's UnevenMap (uneven_map/include/uneven_map/uneven_map.h:66-152) backed by the
device-resident grid of libunevenhip.so.  Same member names and argument meaning; Eigen vectors become numpy arrays.

Only `constructMap` moved to the GPU (uph_map_build).  After build()/init() the host copies of map_buffer, c_buffer,
occ_buffer and occ_r2_buffer are filled exactly as the reference's members would be, so the untouched host consumers
(kinodynamic A*, RViz publishing, `.map` caching) read them as before."""
import ctypes as C
import math
import os

import numpy as np

from . import _lib
from .host_map import HostGridView
from .scenes import read_pcd

# plan_manager/params/run_hill.yaml:2-14
HILL_MAP_PARAMS = dict(iter_num=2, map_size_x=10.0, map_size_y=10.0, ellipsoid_x=0.2, ellipsoid_y=0.1,
                       ellipsoid_z=0.1, xy_resolution=0.05, yaw_resolution=0.1, min_cnormal=0.8, max_rho=0.05,
                       gravity=9.81)
# BASELINE.json configs[4] / SURVEY.md 8c row 5: 1 km^2 at 0.25 m x 64 yaw bins = 1.02e9 cells, 16.4 GB as fp32 cells (replicated per GPU);
# fBm H = 0.8, seed 7, amplitude <= 15 m, worst-case slope <= 35 deg, wavelengths 2 .. 512 m; ripples of 0.5 m wavelength and 6 cm
# amplitude inside rough patches 25 .. 75 m across
KM2_MAP_PARAMS = dict(map_size_x=1000.0, map_size_y=1000.0, xy_resolution=0.25)
FBM_PARAMS = dict(seed=7, hurst=0.8, lambda_min=2.0, lambda_max=512.0, amplitude=15.0, max_slope_deg=35.0, n_waves=32, rough_amp=0.06,
                  rough_lambda=0.5, patch_lambda=25.0, rough_threshold=0.62)
DOWNLOAD_LIMIT_CELLS = 1 << 26      # larger grids keep their cells on the device; the host copy holds occ_r2 only


def _fbm(params):
    q = dict(FBM_PARAMS)
    if params:
        q.update(params)
    return _lib.FbmParams(**{k: (int(v) if k in ("seed", "n_waves") else float(v)) for k, v in q.items()})


def fbm_table(params=None):
    """the wave table the device fill uses (uph_fbm_table): dict of arrays a, kx, ky, ph [n_waves], ripples [4,3], envelope [3,3]"""
    fp = _fbm(params)
    t = np.zeros(_lib.FBM_TABLE_DOUBLES)
    _lib.check(_lib.load().uph_fbm_table(C.byref(fp), _dp(t)), "uph_fbm_table")
    w = t[:4 * _lib.FBM_MAX_WAVES].reshape(-1, 4)[:fp.n_waves]
    o = 4 * _lib.FBM_MAX_WAVES
    return dict(a=w[:, 0].copy(), kx=w[:, 1].copy(), ky=w[:, 2].copy(), ph=w[:, 3].copy(), ripples=t[o:o + 12].reshape(4, 3).copy(),
                envelope=t[o + 12:o + 21].reshape(3, 3).copy(), rough_amp=fp.rough_amp, rough_threshold=fp.rough_threshold)


def _dp(a):
    return a.ctypes.data_as(_lib.DP)


# ---- x-slab sharding of constructMap over the ranks of one node ----------------------------------------------------
# The cell array is x-slowest (uneven_map.h:427-435), so the x-slab of a rank is one contiguous block and the exchange is ONE
# all-gather of equally sized blocks (the last block is zero-padded when nx does not divide).  These two helpers are the whole
# rule; UnevenMap.build_sharded runs them around the device kernel with RCCL, the CPU tier runs them around the oracle with gloo.
def slab_bounds(nx, rank, world):
    """(rows per rank, x0, x1): rank r fits x in [x0, x1)"""
    per = -(-int(nx) // int(world))
    x0 = min(rank * per, nx)
    return per, x0, min(x0 + per, nx)


def gather_slabs(slab, nx, row_elems, world, all_gather):
    """slab: this rank's rows as a 1-D torch tensor of per * row_elems elements (zero-padded); all_gather(full, slab) fills a tensor
    of world * per rows; returns the first nx rows = the complete cell array, identical on every rank"""
    import torch
    per = slab.numel() // row_elems
    full = torch.empty(world * per * row_elems, dtype=slab.dtype, device=slab.device)
    all_gather(full, slab)
    return full[:nx * row_elems]


# ---- tiles: x-slab ownership with a halo -------------------------------------------------------------------------
def tile_rows(nx, rank, world, halo_cells):
    """rows [x0, x1) rank r holds: its x-slab (slab_bounds) widened by the halo, cut at the grid border"""
    _, a, b = slab_bounds(nx, rank, world)
    return max(0, a - int(halo_cells)), min(int(nx), b + int(halo_cells))


def owner_of(prob, nx, world, xy_resolution, origin_x):
    """the rank whose x-slab contains the midpoint of the problem's x-extent"""
    xs = np.concatenate([np.asarray(prob["init_xy"])[0, :1], np.asarray(prob["end_xy"])[0, :1], np.asarray(prob["inner_xy"]).reshape(2, -1)[0]])
    ix = int(math.floor((0.5 * (xs.min() + xs.max()) - origin_x) / xy_resolution))
    per = -(-int(nx) // int(world))
    return min(max(ix // per, 0), world - 1)


def route_problems(probs, nx, world, xy_resolution, origin_x):
    """host-side routing of a batch to tile owners: list (per rank) of the indices of the problems it solves"""
    out = [[] for _ in range(world)]
    for i, p in enumerate(probs):
        out[owner_of(p, nx, world, xy_resolution, origin_x)].append(i)
    return out


def km2_map(map_size, rank=0, world=1, device=0, tiled=False, all_gather=None, halo_m=20.0):
    """BASELINE.json configs[4] scene for one rank: analytic fractal terrain in fp32 cells.  Replicated grid (filled in x-slabs with one
    all-gather when `all_gather` is given), or -- tiled, world > 1 -- only this rank's x-slab plus halo, filled locally (no exchange: the
    surface is analytic)."""
    kp = dict(KM2_MAP_PARAMS, map_size_x=float(map_size), map_size_y=float(map_size))
    tile = None
    if tiled and world > 1:
        nx_all = int(math.ceil(float(map_size) / kp["xy_resolution"]))
        tile = tile_rows(nx_all, rank, world, int(round(halo_m / kp["xy_resolution"])))
    m = UnevenMap(kp, device=device, storage="f32", tile=tile)
    if tile is None and all_gather is not None:
        m.fill_fbm_sharded(None, rank, world, all_gather)
    else:
        m.fill_fbm()
    return m


def km2_problems(m, map_size, count, first_seed_offset, rank=0, world=1):
    """`count` local-goal problems (seeds 5000 + first_seed_offset + i) on a km2_map; on a tile they start inside the rank's own x-slab --
    owner routing by construction -- and occupancy is read from the held rows"""
    from . import scenes
    nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
    half = 0.5 * float(map_size) - 5.0
    grid, xlim = (nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1]), None
    if m.tile is not None:
        _, sa, sb = slab_bounds(nx, rank, world)
        xlim = (max(-half, m.map_origin[0] + sa * m.xy_resolution), min(half, m.map_origin[0] + sb * m.xy_resolution))
        grid = (m.rows_held, ny, m.xy_resolution, m.map_origin[0] + m.tile[0] * m.xy_resolution, m.map_origin[1])
    return scenes.local_problems(count, seed0=5000 + first_seed_offset, half=half, occ_r2=m.occ_r2_buffer, grid=grid, xlim=xlim)


class UnevenMap:
    def __init__(self, params=None, device=0, storage="f64", tile=None):
        """storage "f64": cells as the reference's doubles; "f32": four floats per cell (configs[4]; lookups widen to double).
        tile = (x0, x1): hold only the x-rows [x0, x1) of the grid (uph_map_create_tile; grids that do not fit one GPU): voxel_num keeps
        the whole grid's dimensions, the host buffers cover the held rows."""
        assert storage in ("f64", "f32")
        self.storage = storage
        self.tile = None if tile is None else (int(tile[0]), int(tile[1]))
        self.L = _lib.load()
        _lib.require_device()
        q = dict(HILL_MAP_PARAMS)
        if params:
            q.update(params)
        self.params = q
        self._mp = _lib.MapParams(**{k: (int(v) if k == "iter_num" else float(v)) for k, v in q.items()})
        h = C.c_void_p()
        if self.tile is not None:
            _lib.check(self.L.uph_map_create_tile(C.byref(self._mp), int(device), self.tile[0], self.tile[1], int(storage == "f32"), C.byref(h)), "uph_map_create_tile")
        else:
            create = self.L.uph_map_create if storage == "f64" else self.L.uph_map_create_f32
            _lib.check(create(C.byref(self._mp), int(device), C.byref(h)), "uph_map_create")
        self.h = h
        d = (C.c_int32 * 3)()
        _lib.check(self.L.uph_map_dims(self.h, d), "uph_map_dims")
        self.voxel_num = np.array(list(d), dtype=np.int64)
        self.xy_resolution, self.yaw_resolution = q["xy_resolution"], q["yaw_resolution"]
        self.map_size = np.array([q["map_size_x"], q["map_size_y"], 2.0 * math.pi + 5e-2])
        self.min_boundary, self.max_boundary = -self.map_size / 2.0, self.map_size / 2.0
        self.map_origin = self.min_boundary.copy()
        self.rows_held = int(self.voxel_num[0]) if self.tile is None else self.tile[1] - self.tile[0]
        self.ncell = self.rows_held * int(self.voxel_num[1]) * int(self.voxel_num[2])      # cells in memory
        self.map_ready = False
        self.map_buffer = self.c_buffer = self.occ_buffer = self.occ_r2_buffer = None
        self.device = device

    def __del__(self):
        try:
            if self.h:
                self.L.uph_map_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- construction ------------------------------------------------------------------------------------------
    def init(self, pcd_file=None, map_file=None, xyz=None):
        """UnevenMap::init (uneven_map.cpp:73-268), data part: read the cloud, then constructMapInput() (the `.map`
        text cache) if it exists, else constructMap() on the GPU and write the cache."""
        # uph_map_load_cache = constructMapInput: the `.map` CSV is the source of truth (the reference's cache, uneven_map.cpp:270-315); the binary
        # side-car only stands in for it while it is at least as new -- a `.map` regenerated later wins.  No cache: build, then write both
        # (the "to txt" block at the end of constructMap, :400-412, + the bit-exact side-car).  All of it in the C-ABI (csrc/map_io_host.cpp).
        if map_file and self.load_cache(map_file):
            return self
        if xyz is None:
            xyz = read_pcd(pcd_file)
        self.build(xyz)
        if map_file:
            self.save_cache(map_file)
        return self

    def build(self, xyz, x0=0, x1=None, download=True):
        """UnevenMap::constructMap (uneven_map.cpp:317-417) on x-slab [x0, x1) -- crop box + 1 cm voxel filter (uneven_map.cpp:133-143),
        xy bucketing and plane fits on the device."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        if x1 is None:
            x1 = int(self.voxel_num[0])
        _lib.check(self.L.uph_map_build(self.h, xyz.ctypes.data_as(C.POINTER(C.c_float)), xyz.shape[0], int(x0), int(x1)),
                   "uph_map_build")
        if download:
            self.download()
        self.map_ready = True
        return self

    @staticmethod
    def filter_cloud(xyz):
        """CropBox + 1 cm VoxelGrid of UnevenMap::init (uneven_map.cpp:133-143) as uph_map_build applies them; returns the (m,3) float32 cloud"""
        L = _lib.load()
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        fp = C.POINTER(C.c_float)
        n = L.uph_map_filter_cloud(xyz.ctypes.data_as(fp), xyz.shape[0], None, 0)
        if n < 0:
            _lib.check(int(n), "uph_map_filter_cloud")
        out = np.zeros((int(n), 3), dtype=np.float32)
        L.uph_map_filter_cloud(xyz.ctypes.data_as(fp), xyz.shape[0], out.ctypes.data_as(fp), int(n))
        return out

    def build_stats(self):
        ms, ci, cp = C.c_double(0), C.c_int64(0), C.c_int64(0)
        _lib.check(self.L.uph_map_build_stats(self.h, C.byref(ms), C.byref(ci), C.byref(cp)), "uph_map_build_stats")
        st = np.zeros(6)
        _lib.check(self.L.uph_map_build_stages(self.h, _dp(st)), "uph_map_build_stages")
        return dict(kernel_ms=ms.value, cell_iters=ci.value, cloud_points=cp.value,
                    stages_ms=dict(upload=st[0], crop_voxel=st[1], bucket=st[2], kernel=st[3], commit=st[4], call=st[5]))

    def built_cloud(self):
        """the cloud the last build fitted planes to, as the device filtered it (test hook: equals filter_cloud(xyz) bit for bit)"""
        fp = C.POINTER(C.c_float)
        n = self.L.uph_map_built_cloud(self.h, None, 0)
        if n < 0:
            _lib.check(int(n), "uph_map_built_cloud")
        out = np.zeros((int(n), 3), dtype=np.float32)
        self.L.uph_map_built_cloud(self.h, out.ctypes.data_as(fp), int(n))
        return out

    def set_cells(self, rxs2):
        """Fill the grid from host cells (ncell x 4: z, sigma, zb.x, zb.y in the reference's address order)."""
        rxs2 = np.ascontiguousarray(rxs2, dtype=np.float64).reshape(self.ncell, 4)
        _lib.check(self.L.uph_map_set_cells(self.h, _dp(rxs2)), "uph_map_set_cells")
        self.download()
        self.map_ready = True
        return self

    def download(self):
        occ2 = np.zeros(self.rows_held * int(self.voxel_num[1]), dtype=np.int8)
        if self.ncell > DOWNLOAD_LIMIT_CELLS or self.tile is not None:         # km^2-scale grid: cells stay on the device (get_window serves pieces of it)
            _lib.check(self.L.uph_map_get_cells(self.h, None, None, None, occ2.ctypes.data_as(C.c_char_p)), "uph_map_get_cells")
            self.map_buffer = self.c_buffer = self.occ_buffer = self.host = None
            self.occ_r2_buffer = occ2
            return
        cells = np.zeros((self.ncell, 4))
        occ = np.zeros(self.ncell, dtype=np.int8)
        cb = np.zeros(self.ncell) if self.storage == "f64" else None
        _lib.check(self.L.uph_map_get_cells(self.h, _dp(cells), _dp(cb) if cb is not None else None, occ.ctypes.data_as(C.c_char_p),
                                            occ2.ctypes.data_as(C.c_char_p)), "uph_map_get_cells")
        if cb is None:
            cb = np.sqrt(1.0 - cells[:, 2] ** 2 - cells[:, 3] ** 2)
        self.map_buffer, self.c_buffer, self.occ_buffer, self.occ_r2_buffer = cells, cb, occ, occ2
        self.host = HostGridView(cells, self.params["map_size_x"], self.params["map_size_y"], self.xy_resolution, self.yaw_resolution)

    def get_window(self, x0, x1, y0, y1):
        """cells of the xy index window [x0, x1) x [y0, y1) as float64 (x1-x0, y1-y0, nyaw, 4), from the device grid"""
        out = np.zeros((int(x1) - int(x0), int(y1) - int(y0), int(self.voxel_num[2]), 4))
        _lib.check(self.L.uph_map_get_window(self.h, int(x0), int(x1), int(y0), int(y1), _dp(out)), "uph_map_get_window")
        return out

    # ---- analytic fractal terrain (BASELINE.json configs[4]) ---------------------------------------------------------
    def fill_fbm(self, fbm=None, x0=0, x1=None, download=True):
        """fill the x-slab [x0, x1) with the analytic fBm terrain (uph_map_fill_fbm): a constructMap-style plane fit per cell on samples of
        the analytic surface"""
        fp = _fbm(fbm)
        if x1 is None:
            x0, x1 = (0, int(self.voxel_num[0])) if self.tile is None else self.tile
        _lib.check(self.L.uph_map_fill_fbm(self.h, C.byref(fp), int(x0), int(x1)), "uph_map_fill_fbm")
        if download:
            self.download()
        self.map_ready = True
        return self

    def fill_fbm_sharded(self, fbm, rank, world, all_gather):
        """the fill sharded like constructMap: rank r fills its x-slab, one all-gather of the slabs (RCCL), every rank imports the whole grid"""
        return self._sharded(lambda x0, x1: self.fill_fbm(fbm, x0, x1, download=False), rank, world, all_gather)

    def cells_device(self):
        """(device pointer, nbytes) of the AoS cell array, for the host framework's RCCL all-gather of x-slabs."""
        p, nb = C.c_void_p(), C.c_int64(0)
        _lib.check(self.L.uph_map_cells_device(self.h, C.byref(p), C.byref(nb)), "uph_map_cells_device")
        return p.value, nb.value

    def build_sharded(self, xyz, rank, world, all_gather):
        """constructMap sharded over `world` ranks: rank r fits its x-slab (slab_bounds), then ONE all-gather of the
        slabs (RCCL over xGMI when `all_gather` is torch.distributed.all_gather_into_tensor on CUDA tensors) and every rank
        imports the complete cell array.  `all_gather(full_tensor, slab_tensor)` works on torch tensors of the storage type."""
        return self._sharded(lambda x0, x1: self.build(xyz, x0=x0, x1=x1, download=False), rank, world, all_gather)

    def _sharded(self, make_slab, rank, world, all_gather):
        import torch
        nx, ny, nyaw = (int(v) for v in self.voxel_num)
        row = ny * nyaw * 4
        per, x0, x1 = slab_bounds(nx, rank, world)
        dev = torch.device("cuda", self.device)
        slab = torch.zeros(per * row, dtype=torch.float64 if self.storage == "f64" else torch.float32, device=dev)
        if x1 > x0:
            make_slab(x0, x1)
            _lib.check(self.L.uph_map_export_slab_dev(self.h, x0, x1, C.c_void_p(slab.data_ptr())), "uph_map_export_slab_dev")
        torch.cuda.synchronize(dev)
        full = gather_slabs(slab, nx, row, world, all_gather).contiguous()
        torch.cuda.synchronize(dev)
        _lib.check(self.L.uph_map_import_cells_dev(self.h, C.c_void_p(full.data_ptr())), "uph_map_import_cells_dev")
        self.download()
        self.map_ready = True
        return self

    def commit(self):
        _lib.check(self.L.uph_map_commit(self.h), "uph_map_commit")
        self.map_ready = True

    # ---- several GPUs, ONE process: the C-ABI's own sharded build (uph_map_build_multi: host threads + RCCL clique inside the library) ----
    @staticmethod
    def _handles(maps):
        arr = (C.c_void_p * len(maps))(*[m.h for m in maps])
        return arr

    @staticmethod
    def build_multi(maps, xyz, download=True):
        """constructMap over len(maps) devices from this one process: maps[g] fits the x-slab slab_bounds(nx, g, n), ONE ncclAllGather inside
        the library, every map commits.  Returns the stage timings of uph_map_multi_stats."""
        L = maps[0].L
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        _lib.check(L.uph_map_build_multi(UnevenMap._handles(maps), len(maps), xyz.ctypes.data_as(C.POINTER(C.c_float)), xyz.shape[0]), "uph_map_build_multi")
        return UnevenMap._after_multi(maps, download)

    @staticmethod
    def fill_fbm_multi(maps, fbm=None, download=True):
        L = maps[0].L
        fp = _fbm(fbm)
        _lib.check(L.uph_map_fill_fbm_multi(UnevenMap._handles(maps), len(maps), C.byref(fp)), "uph_map_fill_fbm_multi")
        return UnevenMap._after_multi(maps, download)

    @staticmethod
    def _after_multi(maps, download):
        for m in maps:
            if download:
                m.download()
            m.map_ready = True
        v = [C.c_double(0) for _ in range(4)]
        r = C.c_int32(0)
        _lib.check(maps[0].L.uph_map_multi_stats(maps[0].h, *[C.byref(x) for x in v], C.byref(r)), "uph_map_multi_stats")
        return dict(fit_ms=v[0].value, exchange_ms=v[1].value, commit_ms=v[2].value, exchange_device_ms=v[3].value, via_rccl=bool(r.value))

    # ---- `.map` cache through the C-ABI (uph_map_load_cache / uph_map_save_cache) ----------------------------------
    def load_cache(self, map_file):
        """constructMapInput: False when neither `map_file` nor `map_file.bin` can be read; self.cache_source = "bin" / "csv" otherwise"""
        src = C.c_int32(0)
        rc = self.L.uph_map_load_cache(self.h, map_file.encode(), (map_file + ".bin").encode(), C.byref(src))
        if rc == _lib.UPH_ERR_NO_CACHE:      # no readable cache: the caller builds the map.  Anything else (tile map, host memory, HIP) is a failure,
            return False                     # not a reason to rebuild and overwrite the user's cache files
        _lib.check(rc, "uph_map_load_cache")
        self.cache_source = {1: "csv", 2: "bin"}[src.value]
        self.download()
        self.map_ready = True
        return True

    def save_cache(self, map_file, sidecar=True):
        _lib.check(self.L.uph_map_save_cache(self.h, map_file.encode(), (map_file + ".bin").encode() if sidecar else None), "uph_map_save_cache")

    # ---- `.map` text cache, host-mirror form (uneven_map.cpp:270-315, 400-412) ---------------------------------------
    def write_map_file(self, path):
        """CSV `x,y,yaw,z,sigma,zbx,zby`, default ostream precision (6 significant digits) like the reference."""
        self.host.write_map_file(path)

    def write_map_binary(self, path):
        """binary side-car of the `.map` cache: the CSV keeps 6 significant digits (uneven_map.cpp:400-412), so a grid read back
        from it differs from the built one by ~1e-6; this file holds the float64 cells bit for bit"""
        self.host.write_map_binary(path)

    def constructMapInputBinary(self, path):
        view = HostGridView.read_map_binary(path, self.params["map_size_x"], self.params["map_size_y"], self.xy_resolution, self.yaw_resolution)
        self.set_cells(view.cells.reshape(-1, 4))
        return True

    def constructMapInput(self, path):
        view = HostGridView.read_map_file(path, self.params["map_size_x"], self.params["map_size_y"], self.xy_resolution, self.yaw_resolution)
        self.set_cells(view.cells.reshape(-1, 4))
        return True

    # ---- queries used by other packages (host side, uneven_map.h:398-509) ----------------------------------------
    def mapReady(self):
        return self.map_ready

    def getGravity(self):
        return self.params["gravity"]

    def getXYNum(self):
        return int(self.voxel_num[0] * self.voxel_num[1])

    def posToIndex(self, pos):
        return np.array([math.floor((pos[0] - self.map_origin[0]) / self.xy_resolution),
                         math.floor((pos[1] - self.map_origin[1]) / self.xy_resolution),
                         math.floor((pos[2] - self.map_origin[2]) / self.yaw_resolution)], dtype=np.int64)

    def indexToPos(self, idx):
        return np.array([(idx[0] + 0.5) * self.xy_resolution + self.map_origin[0],
                         (idx[1] + 0.5) * self.xy_resolution + self.map_origin[1],
                         (idx[2] + 0.5) * self.yaw_resolution + self.map_origin[2]])

    def toAddress(self, x, y, yaw):
        return int(x) * int(self.voxel_num[1]) * int(self.voxel_num[2]) + int(y) * int(self.voxel_num[2]) + int(yaw)

    def isInMapIdx(self, idx):
        return bool(np.all(np.asarray(idx) >= 0) and np.all(np.asarray(idx) <= self.voxel_num - 1))

    def isOccupancy(self, pos):
        idx = self.posToIndex(pos)
        if not self.isInMapIdx(idx):
            return -1
        return int(self.occ_buffer[self.toAddress(*idx)])

    def isOccupancyXY(self, pxy):
        idx = self.posToIndex([pxy[0], pxy[1], pxy[2] if len(pxy) > 2 else 0.0])     # the reference indexes all three components (uneven_map.h:488-498)
        if not self.isInMapIdx(idx):
            return -1
        return int(self.occ_r2_buffer[int(idx[0]) * int(self.voxel_num[1]) + int(idx[1])])

    # value-only lookups served from the host copy (uneven_map.h:154-256, 389-396), as the A* / RViz consumers use them
    def getTerrain(self, pos):
        return self.host.getTerrain(pos)

    def getTerrainSig(self, pos):
        return self.host.getTerrainSig(pos)

    def getTerrainVariables(self, pos):
        return self.host.getTerrainVariables(pos)

    def getTerrainPos(self, pos):
        return self.host.getTerrainPos(pos)

    def frontend_query(self, pos):
        """Batched front-end cost queries on the device grid: pos (n,3) -> sigma (n,) = getTerrainSig,
        occ (n,) = isOccupancy, occ_xy (n,) = isOccupancyXY (uneven_map.h:389-396, 471-498; -1 outside the map)."""
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        n = pos.shape[0]
        sg, oc, oxy = np.zeros(n), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        ip = lambda a: a.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int32))
        _lib.check(self.L.uph_frontend_query(self.h, _dp(pos), n, _dp(sg), ip(oc), ip(oxy)), "uph_frontend_query")
        return sg, oc, oxy

    def getTerrainPosBatch(self, pos):
        """batched getTerrainPos (uneven_map.h:203-218) on the device grid: pos (n,3) -> R (n,3,3) with columns x_b, y_b, z_b and p (n,3)"""
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        out = np.zeros((pos.shape[0], 12))
        _lib.check(self.L.uph_terrain_pose_query(self.h, _dp(pos), pos.shape[0], _dp(out)), "uph_terrain_pose_query")
        return out[:, :9].reshape(-1, 3, 3).transpose(0, 2, 1).copy(), out[:, 9:].copy()

    def frontend_query_ms(self):
        ms = _lib.C.c_double(0.0)
        _lib.check(self.L.uph_frontend_query_ms(self.h, _lib.C.byref(ms)), "uph_frontend_query_ms")
        return ms.value

    def getAllWithGrad(self, pos):
        """Device twin of UnevenMap::getAllWithGrad (uneven_map.h:318-377).  pos: (n,3) with yaw in [-pi,pi].
        Returns values (n,7) and grads (n,7,3): invCosVphix, sinPhix, invCosVphiy, sinPhiy, cosXi, invCosXi, sigma."""
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        v = np.zeros((pos.shape[0], 7))
        g = np.zeros((pos.shape[0], 7, 3))
        _lib.check(self.L.uph_terrain_query(self.h, _dp(pos), pos.shape[0], _dp(v), _dp(g)), "uph_terrain_query")
        return v, g
