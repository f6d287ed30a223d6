/*
 * uneven_hip.h -- C-ABI of libunevenhip.so, the MI355X (gfx950) back-end that drops in behind the reference's
 * ALMTrajOpt::optimizeSE2Traj and UnevenMap interfaces (ZJU-FAST-Lab/uneven_planner).
 *
 * Plain C: opaque handles, caller-owned arrays, int status returns (0 = ok, <0 = error, text via uph_last_error()).
 * Nothing throws across this boundary.  A context is not thread-safe (one solve at a time, like the reference's
 * in_opt flag, back_end/src/alm_traj_opt.cpp:61,178,275); distinct contexts may be used concurrently.
 *
 * Reference interfaces replaced (paths relative to /root/reference/src/uneven_planner):
 *   uph_optimize_batch      <- ALMTrajOpt::optimizeSE2Traj  back_end/include/back_end/alm_traj_opt.h:92-98,
 *                              back_end/src/alm_traj_opt.cpp:168-278 (B = 1 reproduces one call)
 *   uph_result              <- ALMTrajOpt::getTraj / MINCO_SE2::getTraj / getTrajJerkCost
 *                              alm_traj_opt.h:165-168, back_end/include/utils/se2traj.hpp:682-695,844-855
 *   uph_opt_params          <- ALMTrajOpt public parameter members  alm_traj_opt.h:29-53 (rosparam, alm_traj_opt.cpp:7-29)
 *   uph_map_build           <- UnevenMap::constructMap  uneven_map/src/uneven_map.cpp:317-417 (+ crop/voxel filter :130-144)
 *   uph_map_set_cells       <- UnevenMap::constructMapInput  uneven_map.cpp:270-315 (cells from the .map cache)
 *   uph_map_save_csv / uph_map_load_csv / uph_map_save_cache / uph_map_load_cache
 *                           <- the `.map` cache: written at the end of UnevenMap::constructMap (uneven_map.cpp:400-412), read by constructMapInput (:270-315)
 *   uph_map_get_cells       -> fills UnevenMap::map_buffer / c_buffer / occ_buffer / occ_r2_buffer
 *                              uneven_map/include/uneven_map/uneven_map.h:91-94, occupancy rule uneven_map.cpp:170-179
 *   uph_terrain_query       <- UnevenMap::getAllWithGrad  uneven_map.h:318-377 (device-side twin, exposed for parity tests)
 *   uph_terrain_pose_query  <- UnevenMap::getTerrainPos  uneven_map.h:203-218 (batched)
 *   uph_map_filter_cloud    <- pcl::CropBox + pcl::VoxelGrid as UnevenMap::init applies them  uneven_map.cpp:133-143
 *   uph_frontend_query      <- UnevenMap::getTerrainSig / isOccupancy / isOccupancyXY  uneven_map.h:389-396, 471-498 (batched)
 *   uph_eval_batch          <- innerCallback  alm_traj_opt.cpp:280-347 (one objective+gradient evaluation; test/bench hook)
 *   uph_penalty_batch       <- ALMTrajOpt::calConstrainCostGrad  alm_traj_opt.cpp:663-991 (the penalty kernel alone; test/bench hook)
 *   uph_init_scaling_batch  <- ALMTrajOpt::initScaling  alm_traj_opt.cpp:349-661 (test hook)
 *   uph_report_batch        <- ALMTrajOpt::getMaxVxAxAyCurAttSig alm_traj_opt.h:170-229 + SE2Trajectory::getNonHolError
 *                              se2traj.hpp:551-561
 *   uph_map_build_multi     <- UnevenMap::constructMap (as above) sharded over the GPUs of one node from ONE host process -- the reference
 *                              is a single process (plan_manager/src/manager_node.cpp): x-slabs + one RCCL all-gather (BASELINE.json configs[3])
 *   uph_optimize_batch_multi<- B x ALMTrajOpt::optimizeSE2Traj split over per-device contexts (BASELINE.json configs[2], [4])
 *   uph_kino_plan_batch     <- KinoAstar::plan  front_end/src/kino_astar.cpp:67-236 (one call per query; B = 1 reproduces one call), the caller of
 *                              the map's query interface and the producer of the front-end path PlanManager resamples (plan_manager.cpp:59-60)
 *   uph_kino_params         <- rosparam kino_astar/...  kino_astar.cpp:7-20, values of plan_manager/params/run_hill.yaml:16-30
 */
#ifndef UNEVEN_HIP_H
#define UNEVEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uph_map uph_map;   /* device-resident SE(2) -> R x S2+ terrain grid            */
typedef struct uph_ctx uph_ctx;   /* optimiser context bound to one map, one device, one stream */
typedef struct uph_kino uph_kino; /* front-end search context bound to one map: per-query workspaces (node pools, open heaps) in HBM */

/* ---- status codes */
#define UPH_OK 0
#define UPH_ERR_INVALID (-1)   /* bad argument                                   */
#define UPH_ERR_HIP (-2)       /* a HIP runtime call failed (see uph_last_error) */
#define UPH_ERR_NO_DEVICE (-3) /* no gfx950 device visible                        */
#define UPH_ERR_LIMIT (-4)     /* problem exceeds a compiled limit (UPH_MAX_*)    */
#define UPH_ERR_NO_CACHE (-5)  /* uph_map_load_cache only: neither cache file is readable -- build the map (uneven_map.cpp:166-167); any other code is a real failure */

/* uph_result.ret_code beyond the reference's 0 / 1 / 2: */
#define UPH_RET_STOPPED 3      /* test hook only: the ALM loop was stopped by uph_batch_alm_passes' pass cap                          */
#define UPH_RET_UNSUPPORTED 4  /* the problem lies outside the compiled limits (more than UPH_MAX_PIECE_* pieces; piece_yaw < piece_xy):
                                  not solved, outputs untouched, last_lbfgs_ret = the UPH_ERR_* reason.  The other problems of the
                                  batch are solved normally.  (A goal closer than one piece length -- n_inner_* = 0, a single quintic
                                  piece per block -- IS solved, as the reference does.)                                               */

#define UPH_RET_LEFT_TILE 5    /* tile maps only: the solved path reached the border of the rows the tile holds (lookups were clamped there) */
#define UPH_TILE_MARGIN 2.0    /* [m] a problem is accepted by a tile map when its initial path keeps this distance from the tile's border */

#define UPH_MAX_PIECE_XY 128   /* Nxy  <= 128 (38 m at piece_len 0.3)  */
#define UPH_MAX_PIECE_YAW 256  /* Nyaw <= 256                          */
#define UPH_MAX_MEM 256        /* L-BFGS history length                */
#define UPH_MAX_PAST 8

/* ---- map parameters: rosparam uneven_map/... (uneven_map.cpp:75-88), values of plan_manager/params/run_hill.yaml:2-14 */
typedef struct uph_map_params {
    int32_t iter_num;           /* 2     */
    double map_size_x;          /* 10.0  */
    double map_size_y;          /* 10.0  */
    double ellipsoid_x;         /* 0.2   */
    double ellipsoid_y;         /* 0.1   */
    double ellipsoid_z;         /* 0.1   */
    double xy_resolution;       /* 0.05  */
    double yaw_resolution;      /* 0.1   */
    double min_cnormal;         /* 0.8   */
    double max_rho;             /* 0.05  */
    double gravity;             /* 9.81  */
} uph_map_params;

/* ---- analytic fractal terrain of BASELINE.json configs[4] ("synthetic 1 km^2 fractal terrain ... fp32").  No reference
 *      counterpart: the reference only loads .pcd clouds (uneven_map.cpp:121-128); the cell fit applied to the analytic surface is
 *      constructMap's (uneven_map.cpp:329-391).  Values of tools / bench: seed 7, H 0.8, 2..512 m, 15 m, 35 deg, 32 waves */
#define UPH_FBM_MAX_WAVES 48
#define UPH_FBM_TABLE_DOUBLES (4 * UPH_FBM_MAX_WAVES + 12 + 9)
typedef struct uph_fbm_params {
    uint64_t seed;
    double hurst;               /* amplitude ~ wavelength^hurst                                             */
    double lambda_min, lambda_max;   /* band limits of the fBm sum [m]                                       */
    double amplitude;           /* bound on the sum of the wave amplitudes [m]                               */
    double max_slope_deg;       /* bound on the sum of amplitude x wavenumber (worst-case slope)             */
    int32_t n_waves;            /* <= UPH_FBM_MAX_WAVES                                                      */
    double rough_amp;           /* ripple amplitude inside rough patches [m]                                 */
    double rough_lambda;        /* ripple wavelength [m] (x 0.8..1.2)                                        */
    double patch_lambda;        /* envelope wavelength [m] (x 1..3)                                          */
    double rough_threshold;     /* envelope level above which ripples appear (0..1)                          */
} uph_fbm_params;

/* ---- optimiser parameters: alm_traj_opt.h:29-53, values of run_hill.yaml:32-55 */
typedef struct uph_opt_params {
    double rho_T, rho_ter, max_vel, max_acc_lon, max_acc_lat, max_kap, min_cxi, max_sig;
    int32_t use_scaling;
    double rho, beta, gamma, epsilon_con, max_iter;            /* max_iter is a double in the reference */
    double g_epsilon, min_step, inner_max_iter, delta;
    int32_t mem_size, past, int_K;
} uph_opt_params;

/* ---- PlanManager parameters of the initial-guess stage: rosparam manager/... (plan_manager.cpp:9-13), values of run_hill.yaml:57-62 */
typedef struct uph_manager_params {
    double piece_len;           /* 0.3  */
    double mean_vel;            /* 0.5  */
    double init_time_times;     /* 1.2  */
    double yaw_piece_times;     /* 2.0  */
    double init_sig_vel;        /* 0.05 */
    /* test_mode != 0 selects the OTHER producer of optimizeSE2Traj's arguments in the reference, the back-end's stand-alone test node
     * ALMTrajOpt::rcvWpsCallBack (back_end/src/alm_traj_opt.cpp:73-144): its literals replace the five parameters above (piece_len 0.3, yaw
     * pitch piece_len / 2, end velocities 0.05, total time = length / max_vel * 1.2 with the optimiser's max_vel), each comb emits AT MOST ONE
     * node per path segment (`if`, :122,128, where PlanManager loops with `while`), and every position node also contributes its interpolated
     * yaw to the yaw way-points (:132).  Zero-initialised aggregates ({0.3, 0.5, 1.2, 2.0, 0.05}) keep selecting PlanManager's stage. */
    int32_t test_mode;          /* 0    */
    double test_max_vel;        /* ALMTrajOpt::max_vel, run_hill.yaml:35 (0.5); read in test mode only */
} uph_manager_params;

/* ---- KinoAstar parameters: rosparam kino_astar/... (front_end/src/kino_astar.cpp:7-19), values of run_hill.yaml:16-30 */
typedef struct uph_kino_params {
    double yaw_resolution;      /* 3.15 : lattice yaw bin of the search (NOT the map's yaw_resolution) */
    double lambda_heu;          /* 1.0  */
    double weight_r2;           /* 1.0  */
    double weight_so2;          /* 0.5  */
    double weight_v_change;     /* 0.0  */
    double weight_delta_change; /* 0.0  */
    double weight_sigma;        /* 10.0 */
    double time_interval;       /* 0.3  */
    double collision_interval;  /* 0.06 */
    double oneshot_range;       /* 1.0  */
    double wheel_base;          /* 0.26 */
    double max_steer;           /* 0.5  */
    double max_vel;             /* 0.5  */
} uph_kino_params;

/* uph_kino_plan_batch status per query (the reference prints a message and returns an empty path for 1-4, kino_astar.cpp:86-95, 212-216, 233) */
#define UPH_KINO_OK 0
#define UPH_KINO_START_OCCUPIED 1
#define UPH_KINO_GOAL_OCCUPIED 2
#define UPH_KINO_NO_PATH 3
#define UPH_KINO_POOL_EXHAUSTED 4
#define UPH_KINO_EXPANSION_CAP 5   /* test hook: stopped by max_expand */
#define UPH_KINO_INTERNAL 6

/* ---- one optimizeSE2Traj call.  Matrices are column-major like Eigen::MatrixXd:
 *      init_xy/end_xy = 2x3 {P,V,A columns} -> [Px,Py,Vx,Vy,Ax,Ay]; inner_xy = 2 x n_inner_xy -> [x0,y0,x1,y1,...] */
typedef struct uph_problem {
    int32_t n_inner_xy;         /* piece_xy  - 1 */
    int32_t n_inner_yaw;        /* piece_yaw - 1 */
    double init_xy[6], end_xy[6];
    double init_yaw[3], end_yaw[3];
    const double* inner_xy;
    const double* inner_yaw;
    double total_time;
} uph_problem;

/* ---- result of one solve.  Pointer members are caller-owned (may be NULL to skip);
 *      sizes: x_final[n], n = 2*n_inner_xy + n_inner_yaw + 1 (alm_traj_opt.cpp:188);
 *      c_xy[6*piece_xy*2] row-major (row 6i+k = t^k coefficient of piece i, se2traj.hpp:570), c_yaw[6*piece_yaw];
 *      hx[S], gx[6*S], lambda[S], mu[6*S] in the reference's constraint order (alm_traj_opt.cpp:705-708), S = piece_xy*(int_K+1) */
typedef struct uph_result {
    int32_t ret_code;           /* 0 ok, 1 L-BFGS hard error, 2 ALM hit max_iter (alm_traj_opt.cpp:176,252,267); UPH_RET_* above */
    int32_t alm_iters, lbfgs_iters, evals, last_lbfgs_ret;
    double cost;                /* inner_cost of the last L-BFGS call */
    double jerk_cost;           /* minco_se2.getTrajJerkCost() of the last evaluated trajectory */
    double piece_T_xy, piece_T_yaw;   /* uniform piece durations of the last evaluated trajectory */
    double rho_final;
    double scale_fx;
    double* x_final;
    double* c_xy;
    double* c_yaw;
    double* hx;
    double* gx;
    double* lambda;
    double* mu;
    double* scale_cx;           /* [7*S] */
} uph_result;

/* ---- library */
const char* uph_last_error(void);
int uph_device_count(void);
const char* uph_version(void);
/* hash (16 hex digits) of the sources this library was built from: every .hip / .hpp / .cpp file of csrc and this header, in sorted order (csrc/Makefile) --
 * `uneven_planner_amd._lib.sources_id()` computes the same over the tree, so a run can tell whether the library it loaded is a build of the tree next to it */
const char* uph_build_id(void);

/* ---- initial guess: PlanManager::rcvWpsCallBack between kino_astar->plan and traj_opt.optimizeSE2Traj (plan_manager.cpp:62-132) -- or, with
 *      mp->test_mode, the test node's ALMTrajOpt::rcvWpsCallBack (alm_traj_opt.cpp:73-144) --, for a
 *      batch of front-end paths.  paths = concatenated poses [x, y, yaw]; path b is poses offsets[b] .. offsets[b+1]-1.  Outputs are the
 *      optimizeSE2Traj arguments in uph_problem's layout, problem b at init_xy + 6b, init_yaw + 3b, inner_xy + 2*cap_xy*b ([x0,y0,x1,y1,..]),
 *      inner_yaw + cap_yaw*b; n_inner_* receive the way-point counts.  unwrapped (may be NULL) receives the yaw column after :62-78.
 *      UPH_ERR_LIMIT when a path needs more than cap_xy / cap_yaw way-points (the counts are still written). */
int uph_resample_batch(const uph_manager_params* mp, int32_t B, const double* paths, const int64_t* offsets, int32_t cap_xy, int32_t cap_yaw,
                       double* init_xy, double* end_xy, double* init_yaw, double* end_yaw, double* inner_xy, double* inner_yaw,
                       int32_t* n_inner_xy, int32_t* n_inner_yaw, double* total_time, double* unwrapped);

/* ---- terrain map */
int uph_map_create(const uph_map_params* mp, int device, uph_map** out);
/* the same grid with its cells stored as four floats (16 bytes) instead of four doubles: BASELINE.json configs[4]'s fp32 mode.  Lookups widen
 * to double on load and all arithmetic stays fp64; uph_map_build is refused, the cells come from uph_map_fill_fbm / uph_map_set_cells / import */
int uph_map_create_f32(const uph_map_params* mp, int device, uph_map** out);
int uph_map_storage_bytes(const uph_map* m);                                                    /* 8 or 4: bytes per stored field */
/* a TILE of the grid: only the x-rows [x0, x1) of the whole grid (uph_map_dims keeps reporting the whole grid) are held in memory -- for
 * grids that do not fit one GPU (1 km^2 at 0.05 m is 410 GB).  Index arithmetic is the whole grid's, so every lookup
 * inside the tile is bit-identical to the replicated grid; lookups outside are clamped to the tile.  Problems are routed on the host to the
 * tile that owns them (x-slab + halo); uph_batch_upload rejects (UPH_RET_UNSUPPORTED) a path closer than UPH_TILE_MARGIN to the border and
 * uph_batch_download flags one that ended there (UPH_RET_LEFT_TILE).  Filled by uph_map_fill_fbm / uph_map_set_cells (held rows only);
 * uph_map_get_cells / get_window / occupancy cover the held rows. */
int uph_map_create_tile(const uph_map_params* mp, int device, int32_t x0, int32_t x1, int32_t f32, uph_map** out);
int uph_map_tile(const uph_map* m, int32_t* x0, int32_t* x1);                                  /* the rows held: whole grid -> [0, nx) */
/* fill the x-slab [x0, x1) (x1 <= 0: whole map) with the analytic fractal terrain: one constructMap-style plane fit per cell on a 5 x 3
 * body-frame lattice of surface samples; then commit.  uph_fbm_table returns the wave table a checker needs to restate the surface:
 * [UPH_FBM_MAX_WAVES][a, kx, ky, phase], ripples [4][kx, ky, phase], envelope [3][kx, ky, phase] */
int uph_map_fill_fbm(uph_map* m, const uph_fbm_params* fp, int32_t x0, int32_t x1);
int uph_fbm_table(const uph_fbm_params* fp, double* table);
/* cells of the xy window [x0, x1) x [y0, y1) (all yaw bins) as doubles, whatever the storage: rxs2[(x1-x0)*(y1-y0)*nyaw*4] */
int uph_map_get_window(uph_map* m, int32_t x0, int32_t x1, int32_t y0, int32_t y1, double* rxs2);
void uph_map_destroy(uph_map* m);
int uph_map_dims(const uph_map* m, int32_t dims3[3]);                       /* voxel_num (uneven_map.cpp:108-110) */
/* cells: ncell x 4 {z, sigma, zb.x, zb.y} in the reference's address order (uneven_map.h:427-435); recomputes c and occupancy */
int uph_map_set_cells(uph_map* m, const double* rxs2);
/* any pointer may be NULL.  rxs2: ncell x 4, c: ncell, occ: ncell chars, occ_r2: nx*ny chars */
int uph_map_get_cells(uph_map* m, double* rxs2, double* c, char* occ, char* occ_r2);
/* ---- the `.map` cache: the CSV UnevenMap::constructMap writes at its end (uneven_map.cpp:400-412: one line
 * `x,y,yaw,z,sigma,zb.x,zb.y` per cell in address order, ostream default format = six significant digits) and constructMapInput reads
 * (uneven_map.cpp:270-315: cells not mentioned stay RXS2() zeros, indices through atoi, values through stold narrowed to double, lines with an
 * index outside the grid dropped), plus a binary side-car that holds the cells bit for bit.  Host functions, no device needed:
 * rxs2 = ncell x 4 doubles in address order, dims3 = voxel_num; c (may be NULL) receives c_buffer = sqrt(1 - |zb|^2) of the cells read, 1 for
 * the others (uneven_map.cpp:117-119, 307); n_lines (may be NULL) the number of lines used.  load_*: UPH_ERR_INVALID when the file cannot be
 * opened (the reference then builds the map, uneven_map.cpp:166-167), load_bin UPH_ERR_LIMIT when it was written for another grid. */
int uph_map_save_csv(const char* path, const double* rxs2, const int32_t dims3[3]);
int uph_map_load_csv(const char* path, const int32_t dims3[3], double* rxs2, double* c, int64_t* n_lines);
int uph_map_save_bin(const char* path, const double* rxs2, const int32_t dims3[3]);
int uph_map_load_bin(const char* path, const int32_t dims3[3], double* rxs2);
/* the same against a device map: save = download the cells and write the CSV and / or the side-car (either path may be NULL); load =
 * constructMapInput: the side-car if bin_path names a readable one for this grid that is not older than the CSV (the CSV is the reference's own
 * cache and the source of truth: one regenerated later wins), else the CSV, then uph_map_set_cells (commit: c, occupancy).
 * A named CSV that does not exist means NO cache even when a side-car lies next to it (the reference rebuilds whenever map_file is absent,
 * uneven_map.cpp:166-167, 270-277); the side-car alone is used only with csv_path == NULL.
 * source (may be NULL): 2 side-car, 1 CSV.  UPH_ERR_NO_CACHE when no cache can be read: build the map then (every other error code -- a tile
 * map, a grid that does not fit host memory, a HIP failure -- is a failure, not a reason to rebuild and overwrite the cache). */
int uph_map_save_cache(uph_map* m, const char* csv_path, const char* bin_path);
int uph_map_load_cache(uph_map* m, const char* csv_path, const char* bin_path, int32_t* source);
/* constructMap on the x-slab [x0, x1): crop box + 1 cm voxel filter, xy bucketing and plane fits all on the device (the host uploads the cloud).
 * xyz: n x 3 float32 (what pcl::PCDReader delivers).  Cells outside the slab are untouched.  Blocking. */
int uph_map_build(uph_map* m, const float* xyz, int64_t n, int32_t x0, int32_t x1);
/* ---- several GPUs, one host process.  maps[g] = one map per device (same parameters and storage, whole-grid maps).  Device g produces the
 * x-slab [g per, min(nx, (g + 1) per)), per = ceil(nx / n_gpus), of the cell array -- all devices concurrently, one host thread each --, ONE
 * ncclAllGather over an in-process RCCL clique (ncclCommInitAll; in place on the cell arrays when n_gpus divides nx) leaves the complete array
 * on every device, every map commits.  Results are bit-identical to uph_map_build on one device.  RCCL is bound at first use (dlopen: an RCCL
 * the process already carries, else the system's); n_gpus = 1 needs none.  maps on the SAME device are accepted (single-GPU test
 * configuration: RCCL refuses a device twice, the slabs then move by device-to-device copies).  Blocking. */
int uph_map_build_multi(uph_map* const* maps, int32_t n_gpus, const float* xyz, int64_t n);
/* the host-side decisions of the two *_multi builds for a grid of nx rows over n_gpus devices, without a device (dry run of an 8-GPU node on
 * any box): x0[g], x1[g] = the x-slab device g fits (per = ceil(nx / n_gpus) rows; the last slabs shorter or empty), *in_place = 1 when the
 * all-gather runs in place on the cell arrays (n_gpus divides nx), 0 when it goes through zero-padded staging of n_gpus x per rows */
int uph_multi_slab_plan(int32_t nx, int32_t n_gpus, int32_t* x0, int32_t* x1, int32_t* per, int32_t* in_place);
/* the analytic terrain (uph_map_fill_fbm) sharded the same way; fp32 maps exchange float slabs */
int uph_map_fill_fbm_multi(uph_map* const* maps, int32_t n_gpus, const uph_fbm_params* fp);
/* wall milliseconds of the last *_multi call led by `lead` (= maps[0]): slab fits, slab exchange, commit; HIP-event milliseconds of the
 * all-gather on device 0's stream; via_rccl = 1 if the exchange was an RCCL collective */
int uph_map_multi_stats(uph_map* lead, double* fit_ms, double* exchange_ms, double* commit_ms, double* exchange_device_ms, int32_t* via_rccl);
/* diagnostic: binds RCCL the way the sharded build does, forms the in-process clique of devices 0 .. n_devices-1 and all-gathers a known
 * pattern; info (may be NULL) receives which librccl was bound and its version.  UPH_OK = the collective path works in this process */
int uph_rccl_selftest(int32_t n_devices, char* info, int32_t info_cap);
/* releases the cached RCCL cliques (optional; they live until process exit otherwise) */
void uph_multi_shutdown(void);

/* device pointer / byte size of the AoS cell array (ncell x 4 doubles) so that the host framework can all-gather
 * x-slabs across GPUs (RCCL) in place; call uph_map_commit afterwards to refresh c and occupancy */
int uph_map_cells_device(uph_map* m, void** dptr, int64_t* nbytes);
int uph_map_commit(uph_map* m);
/* sharded build helpers (device pointers owned by the caller, e.g. torch tensors used with torch.distributed / RCCL):
 * export the x-slab [x0,x1) of the AoS cell array (contiguous: x is the slowest index), import a full gathered array + commit */
int uph_map_export_slab_dev(uph_map* m, int32_t x0, int32_t x1, void* dst_dev);
int uph_map_import_cells_dev(uph_map* m, const void* src_dev);
/* device twin of UnevenMap::getAllWithGrad: pos n x 3 (x, y, yaw already wrapped to [-pi,pi]) -> values n x 7, grads n x 21 */
int uph_terrain_query(uph_map* m, const double* pos, int32_t n, double* values7, double* grads21);
/* batched front-end cost queries (what kino_astar.cpp:86,91,179,193 and kino_astar.h:263 ask per expanded state):
 * pos n x 3 (x, y, yaw) -> sigma[n] = UnevenMap::getTerrainSig (uneven_map.h:389-396, zeros outside the map),
 * occ[n] = isOccupancy(pos), occ_xy[n] = isOccupancyXY(pos) (uneven_map.h:471-498; -1 outside).  Any output may be NULL. */
int uph_frontend_query(uph_map* m, const double* pos, int32_t n, double* sigma, int32_t* occ, int32_t* occ_xy);
/* kernel milliseconds of the last uph_frontend_query (HIP events) */
/* batched UnevenMap::getTerrainPos (uneven_map.h:203-218) served from the device grid: pose12[n][12] = rotation matrix column-major
 * (x_b, y_b, z_b), then the position (x, y, interpolated z) */
int uph_terrain_pose_query(uph_map* m, const double* pos, int32_t n, double* pose12);
int uph_frontend_query_ms(uph_map* m, double* kernel_ms);
/* the cloud uph_map_build fits planes to: UnevenMap::init's pcl::CropBox [-10,10]^2 x [-0.01,5] followed by pcl::VoxelGrid with a 1 cm
 * leaf (uneven_map.cpp:133-143), applied to xyz (n points).  out_xyz takes at most cap points (NULL: count only); returns the
 * number of filtered points, < 0 on error.  Host function: no device needed. */
int64_t uph_map_filter_cloud(const float* xyz, int64_t n, float* out_xyz, int64_t cap);
/* last uph_map_build timing: kernel milliseconds (HIP events) and number of cell-iterations processed */
int uph_map_build_stats(uph_map* m, double* kernel_ms, int64_t* cell_iters, int64_t* cloud_points);
/* stages of the last uph_map_build, milliseconds: out6 = cloud upload, crop box + voxel filter, bucketing + LDS sizing (all three on the device: the
 * host only launches), plane-fit kernel (HIP events), commit (c, occupancy), the whole call (wall) */
int uph_map_build_stages(uph_map* m, double* out6);
/* test hook: the cloud the last uph_map_build fitted planes to, as the DEVICE filtered it (must equal uph_map_filter_cloud, the host form, bit for bit);
 * at most cap points into out_xyz (NULL: count only); returns the count, < 0 on error */
int64_t uph_map_built_cloud(uph_map* m, float* out_xyz, int64_t cap);

/* ---- front end: KinoAstar::plan for a batch of queries, one wave64 per query (csrc/kino_search.hip).
 * uph_kino_create = KinoAstar::init + setEnvironment (kino_astar.cpp:5-43, kino_astar.h:170-178): parameters, the Dubins radius wheel_base / tan(max_steer),
 * a node pool of getXYNum() nodes per concurrent query.  slots = number of queries searched concurrently (each owns ~3.7 MB of HBM at 200 x 200 cells);
 * when the device cannot hold that many, as many as fit (uph_kino_slots tells).  slots = 0: automatic -- the workspaces follow the batches that
 * arrive (allocated by the first uph_kino_plan_batch for its B, at least 16; grown when a larger batch comes), up to one per wave slot of the default
 * kernel instantiation (16 per compute unit) and never beyond half of the HBM that was free at creation: a single plan() costs ~60 MB, not 16 GB.
 * UPH_ERR_LIMIT when the grid has more than 2^31 columns or (cells + 1) x yaw bins of the lattice does not fit 32-bit keys, or when not even one
 * workspace fits.  The map must stay alive and must not be rebuilt while a search runs. */
int uph_kino_create(uph_map* m, const uph_kino_params* kp, int32_t slots, uph_kino** out);
void uph_kino_destroy(uph_kino* k);
int uph_kino_slots(const uph_kino* k);           /* workspaces allocated (automatic context before its first search: the upper bound) */
/* experiment knob: waves per SIMD the search kernel is compiled for -- 2, 4 (default), 6 or 8 (register caps 256 / 128 / 80 / 64) */
int uph_kino_set_wps(uph_kino* k, int32_t wps);
/* experiment knob: bit 0 = queries handed out dynamically, longest first (default on); bit 1 = sincosFast instead of the device library's sin / cos (default on) */
int uph_kino_set_flags(uph_kino* k, int32_t flags);
int uph_kino_primitives(const uph_kino* k);   /* motion primitives per expansion produced by the reference's loops (kino_astar.cpp:138-145): 15 */
/* starts / goals [B][3] = (x, y, yaw): plan(start_state, end_state) per query.  paths [B][path_cap][3] receives front_end_path (the poses of the
 * node chain, then the Dubins shot samples, kino_astar.h:273-292), n_path[B] its length (may exceed path_cap: truncated), status[B] UPH_KINO_*;
 * iter_num / use_node_num [B] (may be NULL) the reference's counters.  max_expand > 0 stops a query after that many expansions (test hook).
 * expanded (may be NULL with exp_cap = 0): [B][exp_cap][3] lattice index (ix, iy, iyaw) of every expanded node in order.  Blocking. */
int uph_kino_plan_batch(uph_kino* k, int32_t B, const double* starts, const double* goals, int32_t path_cap, double* paths, int32_t* n_path, int32_t* status,
                        int32_t* iter_num, int32_t* use_node_num, int32_t max_expand, int32_t exp_cap, int32_t* expanded);
/* kernel milliseconds (HIP events) of the last uph_kino_plan_batch */
int uph_kino_stats(uph_kino* k, double* kernel_ms);

/* ---- optimiser */
int uph_ctx_create(uph_map* m, const uph_opt_params* p, uph_ctx** out);
void uph_ctx_destroy(uph_ctx* c);
/* rho is a member that persists across solves in the reference (alm_traj_opt.cpp:16, alm_traj_opt.h:137): every problem of a
 * batch starts from the context's rho.  After a solve of ONE problem the context's rho is that problem's final rho (the reference's
 * behaviour over consecutive calls); a batch of B > 1 independent problems leaves it unchanged (each result carries its rho_final). */
/* lanes of one workgroup that cooperate on ONE trajectory: 64, 128, 256 or 512 (one, two, four or eight wave64); 0 = automatic (default):
 * 128 lanes with up to four workgroups per CU from 2304 problems (throughput), 256 lanes below, 512 lanes up to 256 problems (latency).  Takes effect at the next
 * upload.  Results do not depend on the choice beyond the summation order of the block reductions. */
int uph_ctx_set_lanes(uph_ctx* c, int32_t lanes);
/* experiment knob (takes effect at the next upload): group >= 16 permutes the launch order inside groups of `group` workgroups of similar predicted
 * cost so that the workgroups dispatched to one XCD (workgroup index mod 8) work in one octant of the map (per-XCD L2 locality); 0 = off (default).
 * Placement never changes results. */
int uph_ctx_set_xcd_locality(uph_ctx* c, int32_t group);
/* experiment knob: 2 = register-capped kernel build (two waves per SIMD), 1 = uncapped, 0 = choose from the batch size */
int uph_ctx_set_wps(uph_ctx* c, int32_t wps);
/* BASELINE.json configs[4] "fp32": bits = 32 makes the sample phase of the objective (polynomial evaluation, terrain lookup, penalties and
 * their chain rule: alm_traj_opt.cpp:716-988) compute in fp32 -- MINCO, the gradient reduction, L-BFGS and ALM stay fp64.  The reference is
 * double-only: results are then NOT comparable at 1e-9, only through cost / feasibility statistics (tests/test_gpu_km2.py).  64 = default.
 * Applies to the solve and evaluation kernels of 128 / 256 / 512 lanes (every automatic selection); a forced 64-lane context stays fp64. */
int uph_ctx_set_sample_precision(uph_ctx* c, int32_t bits);
int uph_ctx_set_rho(uph_ctx* c, double rho);
int uph_ctx_get_rho(uph_ctx* c, double* rho);

/* diagnostic: keep the first `cap` entries of each trajectory's cost trace of the next solves (cost after every accepted
 * L-BFGS iteration, -1 at the start of an ALM pass); cap = 0 switches it off.  uph_ctx_get_trace: out[B][cap]. */
int uph_ctx_set_trace(uph_ctx* c, int32_t cap);
int uph_ctx_get_trace(uph_ctx* c, double* out);

/* blocking: upload + solve + download.  B = 1 is one ALMTrajOpt::optimizeSE2Traj call.  A problem outside the compiled limits does not fail
 * the batch: its result carries ret_code UPH_RET_UNSUPPORTED and the others are solved; only a batch without any supported problem
 * returns UPH_ERR_INVALID / UPH_ERR_LIMIT. */
int uph_optimize_batch(uph_ctx* c, int32_t B, const uph_problem* probs, uph_result* results);
/* the same call over several GPUs of this process: ctxs[g] = one context per device (each bound to its device's copy of the map, e.g. from
 * uph_map_build_multi).  The problems are dealt to the contexts in descending predicted cost, round-robin; one host thread per device runs
 * upload -> solve -> download on its share; results[] comes back in the caller's order.  No collective: trajectories are independent.
 * Contexts on the same device are accepted (test configuration).  n_gpus = 1 is uph_optimize_batch. */
int uph_optimize_batch_multi(uph_ctx* const* ctxs, int32_t n_gpus, int32_t B, const uph_problem* probs, uph_result* results);
/* the split uph_optimize_batch_multi makes, without a device: share_of[b] = index of the context problem b is dealt to (descending predicted
 * cost, round-robin); predicted_cost (may be NULL) [B] = the a-priori cost the split and every upload's launch order sort by */
int uph_multi_batch_plan(int32_t n_gpus, int32_t B, const uph_problem* probs, int32_t* share_of, double* predicted_cost);
/* number of problems of the uploaded batch (0: none) */
int uph_batch_count(const uph_ctx* c);
/* after uph_optimize_batch_multi every context holds ITS SHARE of the batch (uph_batch_count problems, in share order): idx[k] = the caller's index
 * of problem k of this context -- what maps the rows of uph_report_batch / uph_batch_cycles of a context back to the caller's problems.  Identity
 * after uph_batch_upload / uph_optimize_batch. */
int uph_batch_origin(const uph_ctx* c, int32_t* idx);
/* split form (inputs resident in HBM before the timed region): upload -> solve (kernel only, blocking) -> download */
int uph_batch_upload(uph_ctx* c, int32_t B, const uph_problem* probs);
int uph_batch_solve(uph_ctx* c);
/* the same solve split into enqueue and completion: uph_batch_solve_async returns after queuing the two kernels on the context's own HIP
 * stream, uph_batch_wait blocks until they are done and collects the statistics (uph_batch_solve = the two back to back).  Contexts are
 * independent, so two of them driven alternately keep the GPU full across batch boundaries (the tail of one launch overlaps the head of
 * the next).  One asynchronous solve per context at a time; upload / download / hooks require a waited context. */
int uph_batch_solve_async(uph_ctx* c);
int uph_batch_wait(uph_ctx* c);
int uph_batch_download(uph_ctx* c, uph_result* results);
/* timing / work counters of the last uph_batch_solve: kernel ms (HIP events on the context's stream), total objective
 * evaluations, total constraint-sample evaluations, total L-BFGS iterations, bytes streamed from the L-BFGS history */
int uph_batch_stats(uph_ctx* c, double* kernel_ms, int64_t* evals, int64_t* sample_evals, int64_t* lbfgs_iters, int64_t* hist_bytes);

/* uph_batch_solve = two launches: reset+initScaling, then the ALM/L-BFGS solve kernel (uph_batch_stats reports the latter);
 * this returns the kernel milliseconds of the former */
int uph_batch_prepare_ms(uph_ctx* c, double* ms);

/* diagnostic: shader-clock cycles per phase of the last uph_batch_solve, out[B][16]:
 * 0 MINCO generate, 1 constraint samples, 2 per-piece scatter, 3 adjoint, 4 L-BFGS two-loop, 5 initScaling, 6 whole solve.
 * All zero in the shipped library: the in-kernel timers are compiled out (they cost 1.7 % of a solve launch); build with -DUPH_CYC=1 for them
 * (tools/build_variants.sh cyc="-DUPH_CYC=1"). */
int uph_batch_cycles(uph_ctx* c, long long* out);

/* test / bench hooks on the uploaded batch (state = duals, scales, rho as currently resident):
 * one innerCallback evaluation at x (packed [sum n]); outputs f[B], grad (packed), and refreshes hx/gx/c on the device.
 * `repeat` >= 1 re-runs the same evaluation that many times inside one launch per trajectory (roofline measurement). */
int uph_eval_batch(uph_ctx* c, const double* x_packed, double* f, double* grad_packed, int32_t repeat);
/* A5 ALONE -- ALMTrajOpt::calConstrainCostGrad (back_end/src/alm_traj_opt.cpp:663-991) and nothing else of innerCallback: the coefficients and piece
 * durations RESIDENT on the device (those of the last uph_eval_batch / solve of this batch), the resident duals, scales, rho and scale_fx in ->
 * cost[B], gdCxy (packed [sum 12 Nxy]: per trajectory the reference's 6 Nxy x 2 block, row 6 i + k, row-major), gdCyaw (packed [sum 6 Nyaw]),
 * gdT2[B][2] = (sum_i gdTxy(i), sum_i gdTyaw(i)) -- the pieces share one duration (alm_traj_opt.h:257-261), only the sums enter the gradient
 * (alm_traj_opt.cpp:341-344) -- out.  store_residuals != 0: hx / gx are written by every call as the reference's function writes them (:835, 846 ...;
 * read them with uph_batch_download).  `repeat` >= 1 calls inside one launch per trajectory: this is SURVEY.md 8d's "penalty kernel" measured as
 * north_star defines it (bench.py roofline.penalty_kernel.frac_a5_only).  Any output pointer may be NULL. */
int uph_penalty_batch(uph_ctx* c, int32_t repeat, int32_t store_residuals, double* cost, double* gdcxy_packed, double* gdcyaw_packed, double* gdT2);
int uph_init_scaling_batch(uph_ctx* c);
/* diagnostic: measures the workgroup primitives (barrier, reductions, dependent global load, ...) on the uploaded batch;
 * per-trajectory results via uph_batch_cycles (a -DUPH_CYC=1 build) */
int uph_microbench_batch(uph_ctx* c, int32_t reps);
/* overwrite resident duals / scales (packed in the reference's order; any pointer may be NULL) */
int uph_batch_set_state(uph_ctx* c, const double* lambda, const double* mu, const double* scale_cx, const double* scale_fx, const double* rho);
/* ---- test hooks of the teacher-forced late-state tests (tests/test_gpu_forced.py): drive the device state machine from a state the
 * CPU oracle dumped, one L-BFGS iteration / one ALM pass at a time */
/* overwrite the resident x (packed like x_final) */
int uph_batch_set_x(uph_ctx* c, const double* x_packed);
/* the ALM loop of optimizeSE2Traj (alm_traj_opt.cpp:234-271) from the RESIDENT x, duals, scales and rho -- no reset, no initScaling --
 * for at most max_passes passes (0 = until it ends); a solve stopped by the cap reports ret_code 3 */
int uph_batch_alm_passes(uph_ctx* c, int32_t max_passes);
/* L-BFGS state at the top of the iteration loop (lbfgs.hpp:555): g, d packed like x; pf [B][UPH_MAX_PAST]; the history as the
 * reference holds it (lm_s / lm_y column j of trajectory b at off_b + j*n, off_b = mem * sum of the n before b; lm_ys [B][mem]);
 * scal5 [B][5] = step, fx, k, end, bound */
int uph_batch_set_lbfgs_state(uph_ctx* c, const double* g, const double* d, const double* pf, const double* lm_s, const double* lm_y, const double* lm_ys, const double* scal5);
/* continue from that state for at most `budget` iterations (< 0: until the loop ends); finish_pass: then react as the ALM loop does */
int uph_batch_lbfgs_resume(uph_ctx* c, int32_t budget, int32_t finish_pass);
/* state afterwards; scal8 [B][8] = step, fx, k, end, bound, L-BFGS code (999 = budget ran out), accepted, converged */
int uph_batch_get_lbfgs_state(uph_ctx* c, double* g, double* d, double* pf, double* lm_s, double* lm_y, double* lm_ys, double* scal8);
/* post-solve feasibility report per trajectory: out[B][7] = max vx, ax, ay, cur, att(-cos xi), sigma, non-holonomic error; B = uph_batch_count */
int uph_report_batch(uph_ctx* c, double* out7);

#ifdef __cplusplus
}
#endif
#endif /* UNEVEN_HIP_H */
