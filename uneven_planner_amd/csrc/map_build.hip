// libunevenhip.so -- map half: device-resident SE(2) -> R x S2+ terrain grid and its construction.
//
// Reference (paths under /root/reference/src/uneven_planner/uneven_map):
//   uph_map_build   <- UnevenMap::init data part src/uneven_map.cpp:130-162 (crop box, 1 cm voxel filter; host side)
//                      + UnevenMap::constructMap :317-417 + UnevenMap::filter :5-43 (device kernel below)
//   uph_map_commit  <- occupancy rule :170-179, c_buffer :385,390
//   uph_map_set_cells <- constructMapInput :270-315
//
// Kernel design.  One wave64 per (x,y) column; lane = yaw bin.  Every query of the column -- all yaw bins, both
// refinement iterations -- lies within 0.12 m (probe offset, :342) of the cell centre and searches a radius of
// max(ellipsoid) = 0.2 m (:319,363), so the wave stages the points within 0.12 + 0.2 (+ margin) of the centre ONCE
// from the xy-bucketed cloud into LDS (coalesced contiguous bucket rows, order-preserving ballot compaction) and serves
// all 2 x nyaw plane fits from there with broadcast LDS reads.  PCL's kd-trees are not needed: the radius search's
// result set is defined by FLANN's float predicate (L2_Simple<float>: ((dx*dx)+dy*dy)+dz*dz < r*r), which is evaluated
// literally, followed by the reference's fp64 ellipsoid test (:366-377).  The 3x3 covariance eigen-problem
// (Eigen::EigenSolver in the reference) is solved with cyclic Jacobi in registers.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/uneven_hip.h"
#include "uph_internal.hpp"
#include "terrain_dev.hpp"
#include "rccl_dyn.hpp"

using namespace uph;

struct uph_map {
    int device = 0;
    uph_map_params mp;
    GridDev g;
    size_t ncell = 0;
    double* d_cells = nullptr;   // ncell x 4 doubles {z, sigma, zbx, zby}: the array every lookup gathers and RCCL all-gathers across GPUs
    float* d_cells32 = nullptr;  // fp32 storage mode (uph_map_create_f32): ncell x 4 floats instead of d_cells
    double* d_c = nullptr;       // c_buffer (fp64 storage only)
    char* d_occ = nullptr;       // ncell
    char* d_occ2 = nullptr;      // nx * ny
    void* scratch[4] = {nullptr, nullptr, nullptr, nullptr};      // query scratch (uphMapScratch)
    size_t scratch_cap[4] = {0, 0, 0, 0};
    double last_build_ms = 0.0, last_query_ms = 0.0;
    int64_t last_cell_iters = 0, last_cloud = 0;
    // build scratch (grow-only, reused by every build: no allocation per call once warm) and the events of the kernel timing
    void* bscr[16] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t bscr_cap[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t bev0 = nullptr, bev1 = nullptr;
    double stage_ms[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // last build: cloud upload, crop + voxel filter, bucketing, plane-fit kernel (HIP events), commit, total wall
    int64_t last_raw = 0;
    double multi_ms[4] = {0.0, 0.0, 0.0, 0.0};      // last uph_map_*_multi led by this map: slab fits (wall), slab exchange (wall), commit (wall), exchange (HIP events, device 0)
    int multi_rccl = 0;                             // ... and whether the exchange went through RCCL (1) or device-to-device copies (0)
};

UphDevTmp::~UphDevTmp() { if (p) hipFree(p); }
UphEventTmp::~UphEventTmp() { if (e) hipEventDestroy((hipEvent_t)e); }

int uphMapDevice(const uph_map* m) { return m->device; }
void* uphMapScratch(uph_map* m, int slot, size_t bytes) {
    if (bytes <= m->scratch_cap[slot]) return m->scratch[slot];
    if (m->scratch[slot]) hipFree(m->scratch[slot]);
    m->scratch[slot] = nullptr; m->scratch_cap[slot] = 0;
    const size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&m->scratch[slot], want) != hipSuccess) { setError("query scratch: hipMalloc failed"); return nullptr; }
    m->scratch_cap[slot] = want;
    return m->scratch[slot];
}
GridDev uphMapGrid(const uph_map* m) { return m->g; }
void uphMapOcc(const uph_map* m, const char** occ, const char** occ_r2) { *occ = m->d_occ; *occ_r2 = m->d_occ2; }

#define HIPCHK(call)                                                                               \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            setError(std::string(#call) + ": " + hipGetErrorString(_e));                           \
            return UPH_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

// ------------------------------------------------------------------------------------------------ device code
struct CloudDev {
    const float4* pts;       // sorted by bucket; .w carries the point's index in the filtered cloud (as int bits)
    const int* bstart;       // [bnx*bny + 1]
    float bx0, by0, bsize;
    int bnx, bny;
    int npts;
};

__device__ __forceinline__ void jacobiEig3(double a00, double a01, double a02, double a11, double a12, double a22, double D[3], double V[3][3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-40 * diag || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                const double apq = A[p][q];
                if (apq != 0.0) {
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 3; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
#pragma unroll
                    for (int k = 0; k < 3; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
#pragma unroll
                    for (int k = 0; k < 3; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
                }
            }
    }
    D[0] = A[0][0]; D[1] = A[1][1]; D[2] = A[2][2];
}

// global nearest neighbour in the xy plane over the bucket grid (fallback when the staged disc cannot prove the answer)
__device__ int nearest2DGlobal(const CloudDev& cd, float qx, float qy, float* zout) {
    const int cx = (int)floorf((qx - cd.bx0) / cd.bsize), cy = (int)floorf((qy - cd.by0) / cd.bsize);
    int best = -1; float bestd = 3.0e38f, bestz = 0.f;
    const int maxr = max(cd.bnx, cd.bny) + max(max(abs(cx), abs(cy)), 1) + 1;
    for (int r = 0; r <= maxr; r++) {
        if (best >= 0) { const float lim = (float)(r - 1) * cd.bsize; if (lim > 0 && lim * lim > bestd) break; }
        for (int ix = cx - r; ix <= cx + r; ix++) {
            if (ix < 0 || ix >= cd.bnx) continue;
            for (int iy = cy - r; iy <= cy + r; iy++) {
                if (iy < 0 || iy >= cd.bny) continue;
                if (max(abs(ix - cx), abs(iy - cy)) != r) continue;
                const int b = ix * cd.bny + iy;
                for (int t = cd.bstart[b]; t < cd.bstart[b + 1]; t++) {
                    const float4 p = cd.pts[t];
                    const float dx = p.x - qx, dy = p.y - qy;
                    const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                    const int idx = __float_as_int(p.w);
                    if (d < bestd || (d == bestd && idx < best)) { bestd = d; best = idx; bestz = p.z; }
                }
            }
        }
    }
    *zout = bestz;
    return best;
}

// The disc of points a column stages: every point of the xy-bucketed cloud within Rst = 0.12 + max(ellipsoid) + 1 mm of the cell centre, visited bucket row by
// bucket row in cloud order.  ONE definition for the two kernels that must agree on it -- uph_map_build_kernel (stages the disc into LDS) and uph_disc_cap_kernel
// (counts the largest disc so that the host can size that LDS window): sink(keep, p, ballot) is called once per 64-point batch by the whole wave, with this lane's
// verdict and point and the wave's ballot of verdicts.
struct DiscWalk {
    float Rst, fcx, fcy;
    int bxa, bxb, bya, byb;
};
__device__ __forceinline__ DiscWalk discOf(const GridDev& g, const CloudDev& cd, int x, int y, double ell_x, double ell_y, double ell_z) {
    DiscWalk w;
    const double ccx = (x + 0.5) * g.xy_res + g.origin[0];           // indexToPos, uneven_map.h:419-425
    const double ccy = (y + 0.5) * g.xy_res + g.origin[1];
    const double box_r = fmax(fmax(ell_x, ell_y), ell_z);            // uneven_map.cpp:319
    w.Rst = (float)(0.12 + box_r) + 1.0e-3f;                         // staging radius around the cell centre
    w.fcx = (float)ccx; w.fcy = (float)ccy;
    w.bxa = max(0, (int)floorf((w.fcx - w.Rst - cd.bx0) / cd.bsize)); w.bxb = min(cd.bnx - 1, (int)floorf((w.fcx + w.Rst - cd.bx0) / cd.bsize));
    w.bya = max(0, (int)floorf((w.fcy - w.Rst - cd.by0) / cd.bsize)); w.byb = min(cd.bny - 1, (int)floorf((w.fcy + w.Rst - cd.by0) / cd.bsize));
    return w;
}
template <class SINK>
__device__ __forceinline__ void walkDisc(const DiscWalk& w, const CloudDev& cd, int lane, SINK sink) {
    for (int bx = w.bxa; bx <= w.bxb; bx++) {
        if (w.bya > w.byb) break;
        const int t0 = cd.bstart[bx * cd.bny + w.bya], t1 = cd.bstart[bx * cd.bny + w.byb + 1];
        for (int base = t0; base < t1; base += 64) {
            const int t = base + lane;
            bool keep = false;
            float4 p = make_float4(0, 0, 0, 0);
            if (t < t1) {
                p = cd.pts[t];
                const float dx = p.x - w.fcx, dy = p.y - w.fcy;
                keep = (dx * dx + dy * dy) <= w.Rst * w.Rst;
            }
            sink(keep, p, __ballot(keep));
        }
    }
}

// one wave64 per (x,y) column of the slab [x0, x1)
__global__ __launch_bounds__(64) void uph_map_build_kernel(GridDev g, CloudDev cd, double* __restrict__ cells, int x0, int x1, int iter_num,
                                                           double ell_x, double ell_y, double ell_z, int lds_cap, int* __restrict__ overflow) {
    extern __shared__ float4 spts[];
    const int col = blockIdx.x;
    const int x = x0 + col / g.ny, y = col % g.ny;
    if (x >= x1) return;
    const int lane = threadIdx.x;
    const double ccx = (x + 0.5) * g.xy_res + g.origin[0];           // indexToPos, uneven_map.h:419-425
    const double ccy = (y + 0.5) * g.xy_res + g.origin[1];
    const double box_r = fmax(fmax(ell_x, ell_y), ell_z);            // uneven_map.cpp:319
    // ---- stage the neighbourhood into LDS (order preserving): the disc of discOf / walkDisc, the definition uph_disc_cap_kernel sized the window with
    const DiscWalk dw = discOf(g, cd, x, y, ell_x, ell_y, ell_z);
    const float Rst = dw.Rst, fcx = dw.fcx, fcy = dw.fcy;
    int count = 0;
    walkDisc(dw, cd, lane, [&](bool keep, const float4& p, unsigned long long mask) {
        const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep && pos < lds_cap) spts[pos] = p;
        count += __popcll(mask);
    });
    if (count > lds_cap) {                     // the host sized the staging area for the largest window: report, never fit silently on a truncated disc
        if (lane == 0) atomicExch(overflow, 1);
        count = lds_cap;
    }
    __syncthreads();
    const int npts = count;
    const double einv0 = 1.0 / ell_x, einv1 = 1.0 / ell_y, einv2 = 1.0 / ell_z;
    const float r2f = (float)box_r * (float)box_r;
    for (int yaw = lane; yaw < g.nyaw; yaw += 64) {
        const size_t addr = ((size_t)(x - g.x_off) * g.ny + y) * g.nyaw + yaw;
        // constructMap starts every cell from a fresh RXS2() with c_buffer = 1 (uneven_map.cpp:117-119, 329-331), whatever an earlier
        // build, set_cells or import left in the slab
        double cz = 0.0, csig = 0.0, czbx = 0.0, czby = 0.0, cc = 1.0;
        const double yawc = (yaw + 0.5) * g.yaw_res + g.origin[2];
        const double cyw = cos(yawc), syw = sin(yawc);
        for (int iter = 0; iter < iter_num; iter++) {
            // body frame from the current normal (:333-340)
            const double zb0 = czbx, zb1 = czby, zb2 = cc;
            double yb0 = zb1 * 0.0 - zb2 * syw, yb1 = zb2 * cyw - zb0 * 0.0, yb2 = zb0 * syw - zb1 * cyw;
            const double ybn = sqrt(yb0 * yb0 + yb1 * yb1 + yb2 * yb2);
            yb0 /= ybn; yb1 /= ybn; yb2 /= ybn;
            const double xb0 = yb1 * zb2 - yb2 * zb1, xb1 = yb2 * zb0 - yb0 * zb2, xb2 = yb0 * zb1 - yb1 * zb0;
            double wx = ccx, wy = ccy, wz = cz;                       // :341-342
            wx += xb0 * 0.12;
            wy += xb1 * 0.12;
            if (iter == 0) {                                          // :346-355 nearest neighbour in the xy plane (float metric)
                const float qx = (float)wx, qy = (float)wy;
                int best = -1; float bestd = 3.0e38f, bestz = 0.f;
                for (int t = 0; t < npts; t++) {
                    const float4 p = spts[t];
                    const float dx = p.x - qx, dy = p.y - qy;
                    const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                    const int idx = __float_as_int(p.w);
                    if (d < bestd || (d == bestd && idx < best)) { bestd = d; best = idx; bestz = p.z; }
                }
                // the staged disc proves the answer only if the best distance fits inside it
                const float dq = sqrtf((qx - fcx) * (qx - fcx) + (qy - fcy) * (qy - fcy));
                const float slack = Rst - dq - 1.0e-4f;
                if (best < 0 || slack <= 0.f || bestd > slack * slack) best = nearest2DGlobal(cd, qx, qy, &bestz);
                if (best >= 0) wz = (double)bestz;
            }
            // radius search (float predicate) + ellipsoid test (fp64), two passes: mean, then covariance (:5-20, :363-377)
            const float qx = (float)wx, qy = (float)wy, qz = (float)wz;
            double sx = 0, sy = 0, sz = 0;
            int cnt = 0;
            for (int t = 0; t < npts; t++) {
                const float4 p = spts[t];
                const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                if (!(d < r2f)) continue;
                const double s0 = (double)p.x - wx, s1 = (double)p.y - wy, s2 = (double)p.z - wz;
                const double e0 = einv0 * (xb0 * s0 + xb1 * s1 + xb2 * s2);
                const double e1 = einv1 * (yb0 * s0 + yb1 * s1 + yb2 * s2);
                const double e2 = einv2 * (zb0 * s0 + zb1 * s1 + zb2 * s2);
                if (e0 * e0 + e1 * e1 + e2 * e2 < 1.0) { sx += (double)p.x; sy += (double)p.y; sz += (double)p.z; cnt++; }
            }
            if (cnt == 0) {                                           // :379-386
                cz = wz; csig = 0.0; czbx = 0.0; czby = 0.0; cc = 1.0;
                continue;
            }
            const double mx = sx / (double)cnt, my = sy / (double)cnt, mz = sz / (double)cnt;
            double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
            for (int t = 0; t < npts; t++) {
                const float4 p = spts[t];
                const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                if (!(d < r2f)) continue;
                const double s0 = (double)p.x - wx, s1 = (double)p.y - wy, s2 = (double)p.z - wz;
                const double e0 = einv0 * (xb0 * s0 + xb1 * s1 + xb2 * s2);
                const double e1 = einv1 * (yb0 * s0 + yb1 * s1 + yb2 * s2);
                const double e2 = einv2 * (zb0 * s0 + zb1 * s1 + zb2 * s2);
                if (e0 * e0 + e1 * e1 + e2 * e2 < 1.0) {
                    const double v0 = (double)p.x - mx, v1 = (double)p.y - my, v2 = (double)p.z - mz;
                    c00 += v0 * v0; c01 += v0 * v1; c02 += v0 * v2; c11 += v1 * v1; c12 += v1 * v2; c22 += v2 * v2;
                }
            }
            const double inv = 1.0 / (double)cnt;
            double D[3], V[3][3];
            jacobiEig3(c00 * inv, c01 * inv, c02 * inv, c11 * inv, c12 * inv, c22 * inv, D, V);
            int im = 0;                                               // D.minCoeff (:25-26)
            if (D[1] < D[im]) im = 1;
            if (D[2] < D[im]) im = 2;
            double n0 = V[0][im], n1 = V[1][im], n2 = V[2][im];
            const double nn = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
            n0 /= nn; n1 /= nn; n2 /= nn;
            if (n2 < 0.0) { n0 = -n0; n1 = -n1; n2 = -n2; }
            double sig = D[im] / (D[0] + D[1] + D[2]) * 3.0;          // :31
            if (isnan(sig)) { sig = 1.0; n0 = 1.0; n1 = 0.0; n2 = 0.0; }   // :32-36
            cz = mz; csig = sig; czbx = n0; czby = n1;
            cc = sqrt(1.0 - czbx * czbx - czby * czby);
        }
        cells[addr * 4 + 0] = cz; cells[addr * 4 + 1] = csig; cells[addr * 4 + 2] = czbx; cells[addr * 4 + 3] = czby;
    }
}

// cells -> c_buffer + occupancy (uneven_map.cpp:170-179, 385, 390).  One thread per (x,y) column.
__global__ void uph_map_commit_kernel(int nx, int ny, int nyaw, const double* __restrict__ cells, const float* __restrict__ cells32, double* __restrict__ cbuf,
                                      char* __restrict__ occ, char* __restrict__ occ2, double min_cnormal, double max_rho) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= nx * ny) return;
    char any = 0;
    for (int w = 0; w < nyaw; w++) {
        const size_t a = (size_t)col * nyaw + w;
        double sg, zx, zy;
        if (cells32) { sg = (double)cells32[a * 4 + 1]; zx = (double)cells32[a * 4 + 2]; zy = (double)cells32[a * 4 + 3]; }
        else { sg = cells[a * 4 + 1]; zx = cells[a * 4 + 2]; zy = cells[a * 4 + 3]; }
        const double c = sqrt(1.0 - zx * zx - zy * zy);
        if (cbuf) cbuf[a] = c;
        const char o = (c < min_cnormal || sg > max_rho) ? 1 : 0;
        occ[a] = o;
        any |= o;
    }
    occ2[col] = any;
}

// ------------------------------------------------------------------------------------------------ analytic fractal terrain (BASELINE.json configs[4])
// Height field of the synthetic 1 km^2 scene: a spectral fBm sum of plane waves (amplitude ~ wavelength^H, wavelengths log-spaced
// between lambda_min and lambda_max, random direction and phase from the seed) plus short-wavelength ripples confined to "rough"
// patches by a smooth envelope, so that sigma reaches the occupancy / max_sig thresholds somewhere.  No point cloud exists for it:
// the cell fit below is UnevenMap::constructMap (uneven_map.cpp:329-391, filter :5-43) with the radius search replaced by a fixed
// body-frame lattice of surface samples inside the robot ellipsoid's footprint.
struct FbmDev {
    int nw;
    double a[UPH_FBM_MAX_WAVES], kx[UPH_FBM_MAX_WAVES], ky[UPH_FBM_MAX_WAVES], ph[UPH_FBM_MAX_WAVES];
    double rq[4][3];          // ripple waves: kx, ky, phase
    double ev[3][3];          // envelope waves
    double rough_amp, rough_thr;
};

__device__ __forceinline__ double fbmHeight(const FbmDev& f, double x, double y) {
    double h = 0.0;
    for (int i = 0; i < f.nw; i++) h += f.a[i] * cos(f.kx[i] * x + f.ky[i] * y + f.ph[i]);
    double e = 0.0;
#pragma unroll
    for (int m = 0; m < 3; m++) e += cos(f.ev[m][0] * x + f.ev[m][1] * y + f.ev[m][2]);
    e = 0.5 + e * (0.5 / 3.0);
    double E = (e - f.rough_thr) / (1.0 - f.rough_thr);
    E = fmin(fmax(E, 0.0), 1.0);
    if (E > 0.0) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) r += cos(f.rq[j][0] * x + f.rq[j][1] * y + f.rq[j][2]);
        h += f.rough_amp * E * E * 0.25 * r;
    }
    return h;
}

// lattice of the fit in units of the ellipsoid's x / y semi-axes (all 15 nodes inside the unit disc)
__constant__ double UPH_FBM_LU[5] = {-0.75, -0.375, 0.0, 0.375, 0.75};
__constant__ double UPH_FBM_LV[3] = {-0.6, 0.0, 0.6};

// one wave64 per (x,y) column of the slab [x0, x1), lane = yaw bin (+64 per trip)
template <typename CellT>
__global__ __launch_bounds__(64) void uph_map_fbm_kernel(GridDev g, FbmDev f, CellT* __restrict__ cells, int x0, int x1, int iter_num, double ell_x, double ell_y) {
    const int col = blockIdx.x;
    const int x = x0 + col / g.ny, y = col % g.ny;
    if (x >= x1) return;
    const double ccx = (x + 0.5) * g.xy_res + g.origin[0];           // indexToPos, uneven_map.h:419-425
    const double ccy = (y + 0.5) * g.xy_res + g.origin[1];
    for (int yaw = threadIdx.x; yaw < g.nyaw; yaw += 64) {
        const size_t addr = ((size_t)(x - g.x_off) * g.ny + y) * g.nyaw + yaw;
        double cz = 0.0, csig = 0.0, czbx = 0.0, czby = 0.0, cc = 1.0;      // fresh RXS2(), c = 1 (uneven_map.cpp:117-119)
        const double yawc = (yaw + 0.5) * g.yaw_res + g.origin[2];
        const double cyw = cos(yawc), syw = sin(yawc);
        for (int iter = 0; iter < iter_num; iter++) {
            const double zb0 = czbx, zb1 = czby, zb2 = cc;                   // body frame from the current normal (:333-340)
            double yb0 = zb1 * 0.0 - zb2 * syw, yb1 = zb2 * cyw - zb0 * 0.0, yb2 = zb0 * syw - zb1 * cyw;
            const double ybn = sqrt(yb0 * yb0 + yb1 * yb1 + yb2 * yb2);
            yb0 /= ybn; yb1 /= ybn; yb2 /= ybn;
            const double xb0 = yb1 * zb2 - yb2 * zb1, xb1 = yb2 * zb0 - yb0 * zb2;
            const double wx = ccx + xb0 * 0.12, wy = ccy + xb1 * 0.12;        // probe point (:341-342)
            double px[15], py[15], pz[15];
            double sx = 0, sy = 0, sz = 0;
#pragma unroll
            for (int a = 0; a < 5; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const double u = UPH_FBM_LU[a] * ell_x, v = UPH_FBM_LV[b] * ell_y;
                    const int t = a * 3 + b;
                    px[t] = wx + u * xb0 + v * yb0;
                    py[t] = wy + u * xb1 + v * yb1;
                    pz[t] = fbmHeight(f, px[t], py[t]);
                    sx += px[t]; sy += py[t]; sz += pz[t];
                }
            const double mx = sx / 15.0, my = sy / 15.0, mz = sz / 15.0;
            double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
#pragma unroll
            for (int t = 0; t < 15; t++) {
                const double v0 = px[t] - mx, v1 = py[t] - my, v2 = pz[t] - mz;
                c00 += v0 * v0; c01 += v0 * v1; c02 += v0 * v2; c11 += v1 * v1; c12 += v1 * v2; c22 += v2 * v2;
            }
            const double inv = 1.0 / 15.0;
            double D[3], V[3][3];
            jacobiEig3(c00 * inv, c01 * inv, c02 * inv, c11 * inv, c12 * inv, c22 * inv, D, V);
            int im = 0;                                               // D.minCoeff (:25-26)
            if (D[1] < D[im]) im = 1;
            if (D[2] < D[im]) im = 2;
            double n0 = V[0][im], n1 = V[1][im], n2 = V[2][im];
            const double nn = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
            n0 /= nn; n1 /= nn; n2 /= nn;
            if (n2 < 0.0) { n0 = -n0; n1 = -n1; n2 = -n2; }
            double sig = D[im] / (D[0] + D[1] + D[2]) * 3.0;          // :31
            if (isnan(sig)) { sig = 1.0; n0 = 1.0; n1 = 0.0; n2 = 0.0; }
            cz = mz; csig = sig; czbx = n0; czby = n1;
            cc = sqrt(1.0 - czbx * czbx - czby * czby);
        }
        cells[addr * 4 + 0] = (CellT)cz; cells[addr * 4 + 1] = (CellT)csig; cells[addr * 4 + 2] = (CellT)czbx; cells[addr * 4 + 3] = (CellT)czby;
    }
}

// storage conversions for the host-facing copies of an fp32 map
__global__ void uph_cvt_f64_f32(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
__global__ void uph_cvt_f32_f64(const float* __restrict__ src, double* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}
// xy window [x0, x1) x [y0, y1), all yaw bins, as doubles whatever the storage
__global__ void uph_window_kernel(int ny, int nyaw, const double* __restrict__ cells, const float* __restrict__ cells32, int x0, int y0, int wx, int wy, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)nyaw * 4;
    if (i >= (size_t)wx * wy * per) return;
    const size_t colw = i / per, r = i % per;
    const int ix = x0 + (int)(colw / wy), iy = y0 + (int)(colw % wy);
    const size_t a = ((size_t)ix * ny + iy) * per + r;
    out[i] = cells32 ? (double)cells32[a] : cells[a];
}


// ------------------------------------------------------------------------------------------------ cloud preparation on the device
// UnevenMap::init's pcl::CropBox + pcl::VoxelGrid (uneven_map.cpp:133-143) and the xy bucketing of the plane-fit kernel's input, without the
// host: the same float predicates and integer keys as cropAndVoxel() below (which stays as the host form behind uph_map_filter_cloud and as
// the checker of this path: tests/test_gpu_map.py compares the two bit for bit), stable radix sorts (hipcub) for "ordered by leaf index" /
// "ordered by bucket", sequential float sums per leaf in input order.
__device__ __forceinline__ unsigned fOrd(float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }      // order-preserving float -> uint
__host__ __device__ inline float fOrdInv(unsigned u) { const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &b, 4); return f; }

__global__ void uph_crop_flag_kernel(const float* __restrict__ xyz, int n, int* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    const bool fin = isfinite(px) && isfinite(py) && isfinite(pz);
    flag[i] = (fin && !(px < -10.0f || py < -10.0f || pz < -0.01f || px > 10.0f || py > 10.0f || pz > 5.0f)) ? 1 : 0;
}
__global__ void uph_compact_kernel(const float* __restrict__ xyz, int n, const int* __restrict__ flag, const int* __restrict__ pos, float* __restrict__ cx, float* __restrict__ cy,
                                   float* __restrict__ cz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const int o = pos[i];
    cx[o] = xyz[3 * (size_t)i]; cy[o] = xyz[3 * (size_t)i + 1]; cz[o] = xyz[3 * (size_t)i + 2];
}
// mm[0..2] = min of x, y, z; mm[3..5] = max (order-preserving uint encoding; initialised to 0xffffffff / 0)
__global__ void uph_minmax3_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n, unsigned* __restrict__ mm) {
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned a = fOrd(x[i]), b = fOrd(y[i]), c = fOrd(z[i]);
        lo[0] = min(lo[0], a); hi[0] = max(hi[0], a); lo[1] = min(lo[1], b); hi[1] = max(hi[1], b); lo[2] = min(lo[2], c); hi[2] = max(hi[2], c);
    }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off >= 1; off >>= 1) { lo[k] = min(lo[k], (unsigned)__shfl_xor((int)lo[k], off)); hi[k] = max(hi[k], (unsigned)__shfl_xor((int)hi[k], off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&mm[k], lo[k]); atomicMax(&mm[3 + k], hi[k]); }
    }
}
// leaf key of pcl::VoxelGrid: ijk = floor(coord * inv_leaf) - min_b in float, key = i + j * div0 + k * div0 * div1
__global__ void uph_voxel_key_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n, float inv, float mb0, float mb1, float mb2, int mul1,
                                     int mul2, int* __restrict__ key, int* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int i0 = (int)(floorf(x[i] * inv) - mb0), i1 = (int)(floorf(y[i] * inv) - mb1), i2 = (int)(floorf(z[i] * inv) - mb2);
    key[i] = i0 + i1 * mul1 + i2 * mul2;
    idx[i] = i;
}
__global__ void uph_head_kernel(const int* __restrict__ skey, int n, int* __restrict__ head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
}
// one thread per leaf: float centroid of its points, summed in input order (the sort is stable)
__global__ void uph_centroid_kernel(const int* __restrict__ skey, const int* __restrict__ sidx, const int* __restrict__ head, const int* __restrict__ vid, int n, const float* __restrict__ x,
                                    const float* __restrict__ y, const float* __restrict__ z, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int t = i;
    for (; t < n && skey[t] == skey[i]; t++) { const int q = sidx[t]; sx += x[q]; sy += y[q]; sz += z[q]; }
    const float cnt = (float)(t - i);
    const int o = vid[i];
    ox[o] = sx / cnt; oy[o] = sy / cnt; oz[o] = sz / cnt;
}
__global__ void uph_bucket_key_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, float bx0, float by0, float bsize, int bnx, int bny, int* __restrict__ key,
                                      int* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ix = min(bnx - 1, (int)((x[i] - bx0) / bsize)), iy = min(bny - 1, (int)((y[i] - by0) / bsize));
    key[i] = ix * bny + iy;
    idx[i] = i;
}
__global__ void uph_pack_pts_kernel(const int* __restrict__ sidx, int n, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, float4* __restrict__ pts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int q = sidx[i];
    pts[i] = make_float4(x[q], y[q], z[q], __int_as_float(q));
}
// bstart[b] = first sorted position whose bucket key is >= b (b = 0 .. nb)
__global__ void uph_bstart_kernel(const int* __restrict__ skey, int n, int nb, int* __restrict__ bstart) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (skey[mid] < b) lo = mid + 1; else hi = mid; }
    bstart[b] = lo;
}
// LDS capacity of the plane-fit kernel's staging area = the largest point count of any disc it stages: one wave per (x, y) column of the slab counts the
// points within the staging radius of its cell centre with the kernel's own predicate (discOf / walkDisc above: one definition for both kernels).  (Until round 5 the bound
// was the largest count of a 9 x 9 window of 0.16 m buckets -- several times the disc of 0.32 m radius: on the reference's forest cloud, where trunks and
// canopy stack thousands of points over one column, that bound passed the LDS limit although the largest disc holds 4 500 points = 72 KB.)
__global__ __launch_bounds__(64) void uph_disc_cap_kernel(GridDev g, CloudDev cd, int x0, int x1, double ell_x, double ell_y, double ell_z, int* __restrict__ cap) {
    const int col = blockIdx.x;
    const int x = x0 + col / g.ny, y = col % g.ny;
    if (x >= x1) return;
    const int lane = threadIdx.x;
    int count = 0;
    walkDisc(discOf(g, cd, x, y, ell_x, ell_y, ell_z), cd, lane, [&](bool, const float4&, unsigned long long mask) { count += __popcll(mask); });
    if (lane == 0) atomicMax(cap, count);
}

// ------------------------------------------------------------------------------------------------ host side
namespace {

struct HostCloud { std::vector<float> x, y, z; size_t size() const { return x.size(); } };

// pcl::CropBox (inclusive, float) :133-137 followed by pcl::VoxelGrid with a 1 cm leaf :139-143
// (leaf index = floor(coord * inv_leaf) - min_b, float centroid per leaf, output ordered by leaf index)
HostCloud cropAndVoxel(const float* xyz, int64_t n) {
    HostCloud in;
    const float mn[3] = {-10.0f, -10.0f, -0.01f}, mx[3] = {10.0f, 10.0f, 5.0f};
    for (int64_t i = 0; i < n; i++) {
        const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        if (!std::isfinite(px) || !std::isfinite(py) || !std::isfinite(pz)) continue;
        if (px < mn[0] || py < mn[1] || pz < mn[2] || px > mx[0] || py > mx[1] || pz > mx[2]) continue;
        in.x.push_back(px); in.y.push_back(py); in.z.push_back(pz);
    }
    if (in.size() == 0) return in;
    const float inv = 1.0f / 0.01f;
    float lo[3] = {in.x[0], in.y[0], in.z[0]}, hi[3] = {in.x[0], in.y[0], in.z[0]};
    for (size_t i = 0; i < in.size(); i++) {
        lo[0] = std::min(lo[0], in.x[i]); hi[0] = std::max(hi[0], in.x[i]);
        lo[1] = std::min(lo[1], in.y[i]); hi[1] = std::max(hi[1], in.y[i]);
        lo[2] = std::min(lo[2], in.z[i]); hi[2] = std::max(hi[2], in.z[i]);
    }
    const int64_t dx = (int64_t)((hi[0] - lo[0]) * inv) + 1, dy = (int64_t)((hi[1] - lo[1]) * inv) + 1, dz = (int64_t)((hi[2] - lo[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) return in;    // PCL: leaf too small for the extent -> cloud passed through unfiltered
    int minb[3], divb[3];
    for (int k = 0; k < 3; k++) {
        minb[k] = (int)std::floor(lo[k] * inv);
        divb[k] = (int)std::floor(hi[k] * inv) - minb[k] + 1;
    }
    const int mul[3] = {1, divb[0], divb[0] * divb[1]};
    std::vector<std::pair<int, int>> key(in.size());
    for (size_t i = 0; i < in.size(); i++) {
        const int i0 = (int)(std::floor(in.x[i] * inv) - (float)minb[0]);
        const int i1 = (int)(std::floor(in.y[i] * inv) - (float)minb[1]);
        const int i2 = (int)(std::floor(in.z[i] * inv) - (float)minb[2]);
        key[i] = {i0 * mul[0] + i1 * mul[1] + i2 * mul[2], (int)i};
    }
    std::stable_sort(key.begin(), key.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
    HostCloud out;
    size_t a = 0;
    while (a < key.size()) {
        size_t b = a + 1;
        while (b < key.size() && key[b].first == key[a].first) b++;
        float sx = 0, sy = 0, sz = 0;
        for (size_t t = a; t < b; t++) { sx += in.x[key[t].second]; sy += in.y[key[t].second]; sz += in.z[key[t].second]; }
        const float cnt = (float)(b - a);
        out.x.push_back(sx / cnt); out.y.push_back(sy / cnt); out.z.push_back(sz / cnt);
        a = b;
    }
    return out;
}

int commitMap(uph_map* m) {
    const GridDev& g = m->g;
    const int ncol = g.nx_hold * g.ny;
    hipLaunchKernelGGL(uph_map_commit_kernel, dim3((ncol + 255) / 256), dim3(256), 0, 0, g.nx_hold, g.ny, g.nyaw, m->d_cells, m->d_cells32, m->d_c, m->d_occ,
                       m->d_occ2, m->mp.min_cnormal, m->mp.max_rho);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return UPH_OK;
}

}  // namespace

// One query per lane.  getTerrainSig = sigma of the value-only trilinear lookup (zeros outside the map, uneven_map.h:154-200,
// 389-396); isOccupancy / isOccupancyXY = cell lookups through posToIndex + isInMap(idx) (-1 outside, :411-417, 455-498).
__global__ void uph_frontend_kernel(GridDev g, const char* __restrict__ occ, const char* __restrict__ occ2, const double* __restrict__ pos, int n,
                                    double* __restrict__ sigma, int* __restrict__ occ_out, int* __restrict__ occxy_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = pos[3 * i], y = pos[3 * i + 1], w = pos[3 * i + 2];
    Corners c;
    locate(g, x, y, w, c);
    double tv[4];
    terrainValues(g, c, tv);
    sigma[i] = tv[0];
    const int ix = (int)floor((x - g.origin[0]) * g.xy_inv), iy = (int)floor((y - g.origin[1]) * g.xy_inv), iw = (int)floor((w - g.origin[2]) * g.yaw_inv);
    const int ixh = ix - g.x_off;           // (a tile answers -1 outside the rows it holds)
    const bool in = ix >= 0 && iy >= 0 && iw >= 0 && ix <= g.nx - 1 && iy <= g.ny - 1 && iw <= g.nyaw - 1 && ixh >= 0 && ixh <= g.nx_hold - 1;
    occ_out[i] = in ? (int)occ[((size_t)ixh * g.ny + iy) * g.nyaw + iw] : -1;
    occxy_out[i] = in ? (int)occ2[(size_t)ixh * g.ny + iy] : -1;
}

// UnevenMap::getTerrainPos (uneven_map.h:203-218): SE(3) pose on the terrain, one query per lane.  out[12] = R column-major
// (x_b, y_b, z_b) then p
__global__ void uph_pose_kernel(GridDev g, const double* __restrict__ pos, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = pos[3 * i], y = pos[3 * i + 1], w = pos[3 * i + 2];
    Corners c;
    locate(g, x, y, w, c);
    double tv[4];
    terrainValues(g, c, tv);
    const double z = tv[3], zx = tv[1], zy = tv[2];
    const double zz = sqrt(1.0 - zx * zx - zy * zy);                 // RXS2::getC
    const double cw = cos(w), sw = sin(w);
    double y0 = zy * 0.0 - zz * sw, y1 = zz * cw - zx * 0.0, y2 = zx * sw - zy * cw;      // zb x xyaw
    const double yn = sqrt(y0 * y0 + y1 * y1 + y2 * y2);
    y0 /= yn; y1 /= yn; y2 /= yn;
    const double x0 = y1 * zz - y2 * zy, x1 = y2 * zx - y0 * zz, x2 = y0 * zy - y1 * zx; // yb x zb
    double* o = out + 12 * (size_t)i;
    o[0] = x0; o[1] = x1; o[2] = x2; o[3] = y0; o[4] = y1; o[5] = y2; o[6] = zx; o[7] = zy; o[8] = zz;
    o[9] = x; o[10] = y; o[11] = z;
}

extern "C" {

static int createMap(const uph_map_params* mp, int device, uph_map** out, bool f32, int tx0 = 0, int tx1 = -1) {
    if (!mp || !out) { setError("uph_map_create: null argument"); return UPH_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { setError("uph_map_create: no HIP device visible"); return UPH_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { setError("uph_map_create: bad device index"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    uph_map* m = new uph_map();
    m->device = device;
    m->mp = *mp;
    GridDev& g = m->g;
    const double PI = 3.14159265358979323846;
    const double size[3] = {mp->map_size_x, mp->map_size_y, 2.0 * PI + 5e-2};       // uneven_map.cpp:96
    g.xy_res = mp->xy_resolution; g.yaw_res = mp->yaw_resolution;
    g.xy_inv = 1.0 / g.xy_res; g.yaw_inv = 1.0 / g.yaw_res;                          // :104-105
    for (int i = 0; i < 3; i++) { g.minb[i] = -size[i] / 2.0; g.maxb[i] = size[i] / 2.0; g.origin[i] = g.minb[i]; }   // :99-101
    finishGrid(g);
    g.nx = (int)std::ceil(size[0] / g.xy_res); g.ny = (int)std::ceil(size[1] / g.xy_res); g.nyaw = (int)std::ceil(size[2] / g.yaw_res);   // :108-110
    g.gravity = mp->gravity;
    if (tx1 < 0) tx1 = g.nx;
    if (tx0 < 0 || tx1 > g.nx || tx0 >= tx1) { delete m; setError("uph_map_create_tile: bad x-range"); return UPH_ERR_INVALID; }
    g.x_off = tx0; g.nx_hold = tx1 - tx0;
    m->ncell = (size_t)g.nx_hold * g.ny * g.nyaw;
    if (m->ncell >= ((size_t)1 << 31)) { setError("uph_map_create: more than 2^31 cells held by one device (the lookups index cells with 32 bits): use tiles"); delete m; return UPH_ERR_LIMIT; }
    g.cells = nullptr; g.cells32 = nullptr;
    int r = UPH_OK;
    auto alloc = [&](void** p, size_t bytes) { if (r == UPH_OK && hipMalloc(p, bytes) != hipSuccess) { setError("uph_map_create: hipMalloc of the grid failed"); r = UPH_ERR_HIP; } };
    if (f32) {
        alloc((void**)&m->d_cells32, m->ncell * 4 * sizeof(float));
        if (r == UPH_OK && hipMemset(m->d_cells32, 0, m->ncell * 4 * sizeof(float)) != hipSuccess) r = UPH_ERR_HIP;
        g.cells32 = m->d_cells32;
    } else {
        alloc((void**)&m->d_cells, m->ncell * 4 * sizeof(double));
        alloc((void**)&m->d_c, m->ncell * sizeof(double));
        if (r == UPH_OK && hipMemset(m->d_cells, 0, m->ncell * 4 * sizeof(double)) != hipSuccess) r = UPH_ERR_HIP;   // map_buffer = RXS2() zeros (:118)
        g.cells = m->d_cells;
    }
    alloc((void**)&m->d_occ, m->ncell);
    alloc((void**)&m->d_occ2, (size_t)g.nx_hold * g.ny);
    if (r == UPH_OK) r = commitMap(m);
    if (r != UPH_OK) { uph_map_destroy(m); return r; }
    *out = m;
    return UPH_OK;
}

int uph_map_create(const uph_map_params* mp, int device, uph_map** out) { return createMap(mp, device, out, false); }
int uph_map_create_f32(const uph_map_params* mp, int device, uph_map** out) { return createMap(mp, device, out, true); }
int uph_map_create_tile(const uph_map_params* mp, int device, int32_t x0, int32_t x1, int32_t f32, uph_map** out) { return createMap(mp, device, out, f32 != 0, x0, x1); }
int uph_map_tile(const uph_map* m, int32_t* x0, int32_t* x1) {
    if (!m || !x0 || !x1) return UPH_ERR_INVALID;
    *x0 = m->g.x_off; *x1 = m->g.x_off + m->g.nx_hold;
    return UPH_OK;
}
int uph_map_storage_bytes(const uph_map* m) { return !m ? UPH_ERR_INVALID : (m->d_cells32 ? 4 : 8); }

void uph_map_destroy(uph_map* m) {
    if (!m) return;
    hipSetDevice(m->device);
    hipFree(m->d_cells); hipFree(m->d_cells32); hipFree(m->d_c); hipFree(m->d_occ); hipFree(m->d_occ2);
    for (int k = 0; k < 4; k++) hipFree(m->scratch[k]);
    for (int k = 0; k < 16; k++) hipFree(m->bscr[k]);
    if (m->bev0) hipEventDestroy(m->bev0);
    if (m->bev1) hipEventDestroy(m->bev1);
    delete m;
}

int uph_map_dims(const uph_map* m, int32_t dims3[3]) {
    if (!m || !dims3) return UPH_ERR_INVALID;
    dims3[0] = m->g.nx; dims3[1] = m->g.ny; dims3[2] = m->g.nyaw;
    return UPH_OK;
}

// host <-> fp32 storage in chunks through a bounded fp64 staging buffer
static int copyCellsF32(uph_map* m, const double* from_host, double* to_host) {
    const size_t total = m->ncell * 4, chunk = std::min<size_t>(total, (size_t)64 << 20);
    UphDevTmp st;
    HIPCHK(hipMalloc(&st.p, chunk * sizeof(double)));
    for (size_t o = 0; o < total; o += chunk) {
        const size_t n = std::min(chunk, total - o);
        if (from_host) {
            HIPCHK(hipMemcpy(st.p, from_host + o, n * sizeof(double), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(uph_cvt_f64_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, st.as<double>(), m->d_cells32 + o, n);
        } else {
            hipLaunchKernelGGL(uph_cvt_f32_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, m->d_cells32 + o, st.as<double>(), n);
            HIPCHK(hipMemcpy(to_host + o, st.p, n * sizeof(double), hipMemcpyDeviceToHost));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
    }
    return UPH_OK;
}

int uph_map_set_cells(uph_map* m, const double* rxs2) {
    if (!m || !rxs2) { setError("uph_map_set_cells: null argument"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(m->device));
    if (m->d_cells32) { const int r = copyCellsF32(m, rxs2, nullptr); if (r != UPH_OK) return r; }
    else HIPCHK(hipMemcpy(m->d_cells, rxs2, m->ncell * 4 * sizeof(double), hipMemcpyHostToDevice));
    return commitMap(m);
}

int uph_map_get_cells(uph_map* m, double* rxs2, double* c, char* occ, char* occ_r2) {
    if (!m) return UPH_ERR_INVALID;
    HIPCHK(hipSetDevice(m->device));
    if (c && !m->d_c) { setError("uph_map_get_cells: c_buffer is not kept in fp32 storage mode"); return UPH_ERR_INVALID; }
    if (rxs2) {
        if (m->d_cells32) { const int r = copyCellsF32(m, nullptr, rxs2); if (r != UPH_OK) return r; }
        else HIPCHK(hipMemcpy(rxs2, m->d_cells, m->ncell * 4 * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (c) HIPCHK(hipMemcpy(c, m->d_c, m->ncell * sizeof(double), hipMemcpyDeviceToHost));
    if (occ) HIPCHK(hipMemcpy(occ, m->d_occ, m->ncell, hipMemcpyDeviceToHost));
    if (occ_r2) HIPCHK(hipMemcpy(occ_r2, m->d_occ2, (size_t)m->g.nx_hold * m->g.ny, hipMemcpyDeviceToHost));
    return UPH_OK;
}

// ---- the `.map` cache against a device map (uneven_map.cpp:166-167, 270-315, 400-412; file formats: map_io_host.cpp)
// save: the cells of the (whole-grid, fp64 or fp32) map into the CSV the reference's constructMapInput reads and / or the bit-exact side-car
int uph_map_save_cache(uph_map* m, const char* csv_path, const char* bin_path) {
    if (!m || (!csv_path && !bin_path)) { setError("uph_map_save_cache: bad arguments"); return UPH_ERR_INVALID; }
    if (m->g.nx_hold != m->g.nx) { setError("uph_map_save_cache: a tile holds only part of the grid"); return UPH_ERR_INVALID; }
    std::vector<double> cells;
    try { cells.resize((size_t)m->ncell * 4); } catch (...) { setError("uph_map_save_cache: the grid does not fit host memory"); return UPH_ERR_LIMIT; }
    int r = uph_map_get_cells(m, cells.data(), nullptr, nullptr, nullptr);
    if (r != UPH_OK) return r;
    const int32_t d[3] = {m->g.nx, m->g.ny, m->g.nyaw};
    if (csv_path && (r = uph_map_save_csv(csv_path, cells.data(), d)) != UPH_OK) return r;
    if (bin_path && (r = uph_map_save_bin(bin_path, cells.data(), d)) != UPH_OK) return r;      // (after the CSV: uph_map_load_cache trusts a side-car that is not older than it)
    return UPH_OK;
}
// load = constructMapInput: the side-car when bin_path names a readable one for this grid (bit-exact), else the CSV (six digits); the cells
// go to the device and the map commits (c, occupancy).  source (may be NULL): 2 = side-car, 1 = CSV.  UPH_ERR_INVALID when neither file can
// be opened (UPH_ERR_NO_CACHE) -- the caller then builds the map, as `if (!constructMapInput()) constructMap()` does.
int uph_map_load_cache(uph_map* m, const char* csv_path, const char* bin_path, int32_t* source) {
    if (!m || (!csv_path && !bin_path)) { setError("uph_map_load_cache: bad arguments"); return UPH_ERR_INVALID; }
    if (m->g.nx_hold != m->g.nx) { setError("uph_map_load_cache: a tile holds only part of the grid"); return UPH_ERR_INVALID; }
    std::vector<double> cells;
    try { cells.resize((size_t)m->ncell * 4); } catch (...) { setError("uph_map_load_cache: the grid does not fit host memory"); return UPH_ERR_LIMIT; }
    const int32_t d[3] = {m->g.nx, m->g.ny, m->g.nyaw};
    int src = 0;
    // the CSV is the source of truth (the reference's own cache): the side-car stands in for it only while it is at least as new -- a `.map`
    // regenerated later (by the reference, from another cloud or other ellipsoid parameters) wins over a stale side-car
    // A named CSV that does not exist means NO cache, whatever side-car lies next to it: the reference rebuilds whenever map_file is absent
    // (uneven_map.cpp:166-167, 270-277), so deleting `hill.map` to force a rebuild -- or pointing map_file at another cloud's name -- must not
    // resurrect a stale `hill.map.bin`.  The side-car alone is loaded only when the caller names no CSV at all.
    bool bin_fresh = bin_path != nullptr;
    if (bin_path && csv_path) {
        struct stat sb, sc;
        if (stat(csv_path, &sc) != 0) bin_fresh = false;
        else if (stat(bin_path, &sb) == 0 &&
                 (sb.st_mtim.tv_sec < sc.st_mtim.tv_sec || (sb.st_mtim.tv_sec == sc.st_mtim.tv_sec && sb.st_mtim.tv_nsec < sc.st_mtim.tv_nsec))) bin_fresh = false;
    }
    if (bin_fresh && uph_map_load_bin(bin_path, d, cells.data()) == UPH_OK) src = 2;
    if (!src && csv_path && uph_map_load_csv(csv_path, d, cells.data(), nullptr, nullptr) == UPH_OK) src = 1;
    if (!src) { setError(std::string("uph_map_load_cache: no readable cache (") + uph_last_error() + ")"); return UPH_ERR_NO_CACHE; }
    if (source) *source = src;
    return uph_map_set_cells(m, cells.data());
}

int uph_map_get_window(uph_map* m, int32_t x0, int32_t x1, int32_t y0, int32_t y1, double* rxs2) {
    if (!m || !rxs2 || x0 < m->g.x_off || y0 < 0 || x1 > m->g.x_off + m->g.nx_hold || y1 > m->g.ny || x0 >= x1 || y0 >= y1) { setError("uph_map_get_window: bad arguments (x is a global row index inside the held rows)"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(m->device));
    const size_t n = (size_t)(x1 - x0) * (y1 - y0) * m->g.nyaw * 4;
    UphDevTmp t;
    HIPCHK(hipMalloc(&t.p, n * sizeof(double)));
    hipLaunchKernelGGL(uph_window_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, m->g.ny, m->g.nyaw, m->d_cells, m->d_cells32, x0 - m->g.x_off, y0, x1 - x0, y1 - y0, t.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(rxs2, t.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return UPH_OK;
}

static size_t cellBytes(const uph_map* m) { return 4 * (m->d_cells32 ? sizeof(float) : sizeof(double)); }
static char* cellBase(uph_map* m) { return m->d_cells32 ? (char*)m->d_cells32 : (char*)m->d_cells; }

int uph_map_cells_device(uph_map* m, void** dptr, int64_t* nbytes) {
    if (!m || !dptr || !nbytes) return UPH_ERR_INVALID;
    *dptr = cellBase(m);
    *nbytes = (int64_t)(m->ncell * cellBytes(m));
    return UPH_OK;
}

// device-to-device slab traffic for the sharded build: export this rank's x-slab into a caller buffer (e.g. a torch tensor
// that RCCL all-gathers), import the gathered full cell array.  Element type = the map's storage (double, or float in fp32 mode)
int uph_map_export_slab_dev(uph_map* m, int32_t x0, int32_t x1, void* dst_dev) {
    if (!m || !dst_dev || x0 < m->g.x_off || x1 > m->g.x_off + m->g.nx_hold || x0 >= x1) { setError("uph_map_export_slab_dev: bad arguments"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(m->device));
    const size_t per_x = (size_t)m->g.ny * m->g.nyaw * cellBytes(m);
    HIPCHK(hipMemcpy(dst_dev, cellBase(m) + (size_t)(x0 - m->g.x_off) * per_x, (size_t)(x1 - x0) * per_x, hipMemcpyDeviceToDevice));
    return UPH_OK;
}
int uph_map_import_cells_dev(uph_map* m, const void* src_dev) {
    if (!m || !src_dev) { setError("uph_map_import_cells_dev: bad arguments"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipMemcpy(cellBase(m), src_dev, m->ncell * cellBytes(m), hipMemcpyDeviceToDevice));
    return commitMap(m);
}

int uph_map_commit(uph_map* m) {
    if (!m) return UPH_ERR_INVALID;
    HIPCHK(hipSetDevice(m->device));
    return commitMap(m);
}

// wave table of the analytic terrain from the seed (splitmix64 stream, three uniforms per wave: wavelength jitter, direction, phase)
static void buildFbm(const uph_fbm_params& fp, FbmDev& f) {
    uint64_t st = fp.seed;
    auto next = [&]() {
        st += 0x9E3779B97F4A7C15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    };
    const double TWO_PI = 6.28318530717958647692;
    f.nw = fp.n_waves;
    double sa = 0.0, sk = 0.0;
    for (int i = 0; i < f.nw; i++) {
        const double u1 = next(), u2 = next(), u3 = next();
        const double lam = fp.lambda_min * std::pow(fp.lambda_max / fp.lambda_min, ((double)i + u1) / (double)f.nw);
        const double k = TWO_PI / lam, th = TWO_PI * u2;
        f.kx[i] = k * std::cos(th); f.ky[i] = k * std::sin(th); f.ph[i] = TWO_PI * u3;
        f.a[i] = std::pow(lam, fp.hurst);
        sa += f.a[i]; sk += f.a[i] * k;
    }
    const double PI = 3.14159265358979323846;
    const double c = std::min(fp.amplitude / sa, std::tan(fp.max_slope_deg * PI / 180.0) / sk);    // both bounds hold in the worst case (all waves in phase)
    for (int i = 0; i < f.nw; i++) f.a[i] *= c;
    for (int i = f.nw; i < UPH_FBM_MAX_WAVES; i++) { f.a[i] = 0.0; f.kx[i] = 0.0; f.ky[i] = 0.0; f.ph[i] = 0.0; }
    for (int j = 0; j < 4; j++) {
        const double u1 = next(), u2 = next(), u3 = next();
        const double k = TWO_PI / (fp.rough_lambda * (0.8 + 0.4 * u1)), th = TWO_PI * u2;
        f.rq[j][0] = k * std::cos(th); f.rq[j][1] = k * std::sin(th); f.rq[j][2] = TWO_PI * u3;
    }
    for (int j = 0; j < 3; j++) {
        const double u1 = next(), u2 = next(), u3 = next();
        const double k = TWO_PI / (fp.patch_lambda * (1.0 + 2.0 * u1)), th = TWO_PI * u2;
        f.ev[j][0] = k * std::cos(th); f.ev[j][1] = k * std::sin(th); f.ev[j][2] = TWO_PI * u3;
    }
    f.rough_amp = fp.rough_amp; f.rough_thr = fp.rough_threshold;
}
static bool checkFbm(const uph_fbm_params* fp) {
    return fp && fp->n_waves >= 1 && fp->n_waves <= UPH_FBM_MAX_WAVES && fp->lambda_min > 0 && fp->lambda_max >= fp->lambda_min && fp->rough_lambda > 0 &&
           fp->patch_lambda > 0 && fp->rough_threshold < 1.0 && fp->max_slope_deg > 0 && fp->max_slope_deg < 90 && fp->amplitude > 0;
}

int uph_fbm_table(const uph_fbm_params* fp, double* table) {
    if (!checkFbm(fp) || !table) { setError("uph_fbm_table: bad arguments"); return UPH_ERR_INVALID; }
    FbmDev f;
    buildFbm(*fp, f);
    double* t = table;
    for (int i = 0; i < UPH_FBM_MAX_WAVES; i++) { *t++ = f.a[i]; *t++ = f.kx[i]; *t++ = f.ky[i]; *t++ = f.ph[i]; }
    for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) *t++ = f.rq[j][k];
    for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) *t++ = f.ev[j][k];
    return UPH_OK;
}

static int fillSlab(uph_map* m, const uph_fbm_params* fp, int32_t x0, int32_t x1, bool commit);
int uph_map_fill_fbm(uph_map* m, const uph_fbm_params* fp, int32_t x0, int32_t x1) { return fillSlab(m, fp, x0, x1, true); }
static int fillSlab(uph_map* m, const uph_fbm_params* fp, int32_t x0, int32_t x1, bool commit) {
    if (!m || !checkFbm(fp)) { setError("uph_map_fill_fbm: bad arguments"); return UPH_ERR_INVALID; }
    if (x1 <= 0) { x0 = m->g.x_off; x1 = m->g.x_off + m->g.nx_hold; }
    if (x0 < m->g.x_off || x1 > m->g.x_off + m->g.nx_hold || x0 >= x1) { setError("uph_map_fill_fbm: slab outside the rows this map holds"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(m->device));
    FbmDev f;
    buildFbm(*fp, f);
    UphEventTmp e0, e1;
    HIPCHK(hipEventCreate((hipEvent_t*)&e0.e)); HIPCHK(hipEventCreate((hipEvent_t*)&e1.e));
    const size_t ncol = (size_t)(x1 - x0) * m->g.ny;
    if (ncol > (size_t)INT32_MAX) { setError("uph_map_fill_fbm: slab has more than 2^31 columns"); return UPH_ERR_LIMIT; }
    HIPCHK(hipEventRecord((hipEvent_t)e0.e, 0));
    if (m->d_cells32) hipLaunchKernelGGL(uph_map_fbm_kernel<float>, dim3((unsigned)ncol), dim3(64), 0, 0, m->g, f, m->d_cells32, (int)x0, (int)x1, (int)m->mp.iter_num, m->mp.ellipsoid_x, m->mp.ellipsoid_y);
    else hipLaunchKernelGGL(uph_map_fbm_kernel<double>, dim3((unsigned)ncol), dim3(64), 0, 0, m->g, f, m->d_cells, (int)x0, (int)x1, (int)m->mp.iter_num, m->mp.ellipsoid_x, m->mp.ellipsoid_y);
    HIPCHK(hipEventRecord((hipEvent_t)e1.e, 0));
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, (hipEvent_t)e0.e, (hipEvent_t)e1.e));
    m->last_build_ms = ms;
    m->last_cell_iters = (int64_t)(x1 - x0) * m->g.ny * m->g.nyaw * m->mp.iter_num;
    m->last_cloud = 0;
    return commit ? commitMap(m) : UPH_OK;
}

int uph_map_build_stats(uph_map* m, double* kernel_ms, int64_t* cell_iters, int64_t* cloud_points) {
    if (!m) return UPH_ERR_INVALID;
    if (kernel_ms) *kernel_ms = m->last_build_ms;
    if (cell_iters) *cell_iters = m->last_cell_iters;
    if (cloud_points) *cloud_points = m->last_cloud;
    return UPH_OK;
}

static int buildSlab(uph_map* m, const float* xyz, int64_t n, int32_t x0, int32_t x1, bool commit);
int uph_map_build(uph_map* m, const float* xyz, int64_t n, int32_t x0, int32_t x1) { return buildSlab(m, xyz, n, x0, x1, true); }
static int buildSlab(uph_map* m, const float* xyz, int64_t n, int32_t x0, int32_t x1, bool commit) {
    if (m && m->d_cells32) { setError("uph_map_build: the plane fit writes fp64 cells; create the map with uph_map_create"); return UPH_ERR_INVALID; }
    if (!m || !xyz || n <= 0) { setError("uph_map_build: bad arguments"); return UPH_ERR_INVALID; }
    if (n > (int64_t)INT32_MAX) { setError("uph_map_build: more than 2^31 points"); return UPH_ERR_LIMIT; }
    const GridDev& g = m->g;
    if (x0 < g.x_off || x1 > g.x_off + g.nx_hold || x0 >= x1) { setError("uph_map_build: bad x-slab"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(m->device));
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_mark = t_begin;
    auto lap = [&]() { const auto now = std::chrono::steady_clock::now(); const double ms = std::chrono::duration<double, std::milli>(now - t_mark).count(); t_mark = now; return ms; };
    // grow-only scratch of the map handle: slot -> bytes
    auto scr = [&](int slot, size_t bytes) -> void* {
        if (bytes <= m->bscr_cap[slot]) return m->bscr[slot];
        if (m->bscr[slot]) hipFree(m->bscr[slot]);
        m->bscr[slot] = nullptr; m->bscr_cap[slot] = 0;
        const size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&m->bscr[slot], want) != hipSuccess) { setError("uph_map_build: hipMalloc of build scratch failed"); return nullptr; }
        m->bscr_cap[slot] = want;
        return m->bscr[slot];
    };
    const int N = (int)n;
    const unsigned nb_ = (unsigned)((N + 255) / 256);
    // slots: 0 raw xyz | 1 flag / head | 2 scan | 3,4,5 cropped x y z | 6 key | 7 idx | 8 sorted key | 9 sorted idx | 10,11,12 filtered x y z | 13 pts + bstart | 14 cub temp | 15 small
    float* d_raw = (float*)scr(0, sizeof(float) * 3 * (size_t)N);
    int* d_flag = (int*)scr(1, sizeof(int) * (size_t)N);
    int* d_scan = (int*)scr(2, sizeof(int) * (size_t)N);
    float *d_cx = (float*)scr(3, 4 * (size_t)N), *d_cy = (float*)scr(4, 4 * (size_t)N), *d_cz = (float*)scr(5, 4 * (size_t)N);
    int *d_key = (int*)scr(6, 4 * (size_t)N), *d_idx = (int*)scr(7, 4 * (size_t)N), *d_skey = (int*)scr(8, 4 * (size_t)N), *d_sidx = (int*)scr(9, 4 * (size_t)N);
    float *d_fx = (float*)scr(10, 4 * (size_t)N), *d_fy = (float*)scr(11, 4 * (size_t)N), *d_fz = (float*)scr(12, 4 * (size_t)N);
    unsigned* d_small = (unsigned*)scr(15, 64);
    if (!d_raw || !d_flag || !d_scan || !d_cx || !d_cy || !d_cz || !d_key || !d_idx || !d_skey || !d_sidx || !d_fx || !d_fy || !d_fz || !d_small) return UPH_ERR_HIP;
    size_t tb_scan = 0, tb_sort = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, tb_scan, d_flag, d_scan, N, 0);
    hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, d_key, d_skey, d_idx, d_sidx, N, 0, 32, 0);
    size_t tb = std::max(tb_scan, tb_sort);
    void* d_tmp = scr(14, tb);
    if (!d_tmp) return UPH_ERR_HIP;
    HIPCHK(hipMemcpy(d_raw, xyz, sizeof(float) * 3 * (size_t)N, hipMemcpyHostToDevice));
    m->stage_ms[0] = lap();
    // ---- pcl::CropBox (:133-137): inclusive float box, finite points, order kept
    hipLaunchKernelGGL(uph_crop_flag_kernel, dim3(nb_), dim3(256), 0, 0, d_raw, N, d_flag);
    hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, d_flag, d_scan, N, 0);
    hipLaunchKernelGGL(uph_compact_kernel, dim3(nb_), dim3(256), 0, 0, d_raw, N, d_flag, d_scan, d_cx, d_cy, d_cz);
    int last_flag = 0, last_pos = 0;
    HIPCHK(hipMemcpy(&last_flag, d_flag + (N - 1), 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&last_pos, d_scan + (N - 1), 4, hipMemcpyDeviceToHost));
    const int nc = last_pos + last_flag;                       // points inside the crop box
    if (nc == 0) { setError("uph_map_build: no points inside the crop box"); return UPH_ERR_INVALID; }
    // ---- pcl::VoxelGrid, 1 cm leaf (:139-143): extent -> min_b / div_b on the host (six floats cross), keys + stable sort + per-leaf centroid on the device
    unsigned mm[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HIPCHK(hipMemcpy(d_small, mm, sizeof(mm), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(uph_minmax3_kernel, dim3(std::min(1024, (nc + 255) / 256)), dim3(256), 0, 0, d_cx, d_cy, d_cz, nc, d_small);
    HIPCHK(hipMemcpy(mm, d_small, sizeof(mm), hipMemcpyDeviceToHost));
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) { lo[k] = fOrdInv(mm[k]); hi[k] = fOrdInv(mm[3 + k]); }
    const float inv = 1.0f / 0.01f;
    const int64_t dx = (int64_t)((hi[0] - lo[0]) * inv) + 1, dy = (int64_t)((hi[1] - lo[1]) * inv) + 1, dz = (int64_t)((hi[2] - lo[2]) * inv) + 1;
    int np = 0;
    const float *fx = d_cx, *fy = d_cy, *fz = d_cz;            // the filtered cloud (device)
    if (dx * dy * dz > (int64_t)INT32_MAX) {
        np = nc;                                               // PCL: leaf too small for the extent -> the cropped cloud passes through unfiltered
    } else {
        int minb[3], divb[3];
        for (int k = 0; k < 3; k++) { minb[k] = (int)std::floor(lo[k] * inv); divb[k] = (int)std::floor(hi[k] * inv) - minb[k] + 1; }
        const unsigned nbc = (unsigned)((nc + 255) / 256);
        hipLaunchKernelGGL(uph_voxel_key_kernel, dim3(nbc), dim3(256), 0, 0, d_cx, d_cy, d_cz, nc, inv, (float)minb[0], (float)minb[1], (float)minb[2], divb[0], divb[0] * divb[1], d_key, d_idx);
        hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_key, d_skey, d_idx, d_sidx, nc, 0, 32, 0);
        hipLaunchKernelGGL(uph_head_kernel, dim3(nbc), dim3(256), 0, 0, d_skey, nc, d_flag);
        hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, d_flag, d_scan, nc, 0);
        hipLaunchKernelGGL(uph_centroid_kernel, dim3(nbc), dim3(256), 0, 0, d_skey, d_sidx, d_flag, d_scan, nc, d_cx, d_cy, d_cz, d_fx, d_fy, d_fz);
        HIPCHK(hipMemcpy(&last_flag, d_flag + (nc - 1), 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&last_pos, d_scan + (nc - 1), 4, hipMemcpyDeviceToHost));
        np = last_pos + last_flag;
        fx = d_fx; fy = d_fy; fz = d_fz;
    }
    HIPCHK(hipGetLastError());
    m->stage_ms[1] = lap();
    // ---- xy buckets of 0.16 m: a query disc of 0.32 m around a cell centre touches at most 5 x 5 buckets
    const float bsize = 0.16f;
    unsigned mm2[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HIPCHK(hipMemcpy(d_small, mm2, sizeof(mm2), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(uph_minmax3_kernel, dim3(std::min(1024, (np + 255) / 256)), dim3(256), 0, 0, fx, fy, fz, np, d_small);
    HIPCHK(hipMemcpy(mm2, d_small, sizeof(mm2), hipMemcpyDeviceToHost));
    const float bx0 = fOrdInv(mm2[0]), by0 = fOrdInv(mm2[1]), bx1 = fOrdInv(mm2[3]), by1 = fOrdInv(mm2[4]);
    const int bnx = (int)((bx1 - bx0) / bsize) + 1, bny = (int)((by1 - by0) / bsize) + 1;
    const size_t nbuck = (size_t)bnx * bny;
    char* d_pb = (char*)scr(13, sizeof(float4) * (size_t)np + sizeof(int) * (nbuck + 1) + 64);
    if (!d_pb) return UPH_ERR_HIP;
    float4* d_pts = (float4*)d_pb;
    int* d_bstart = (int*)(d_pb + sizeof(float4) * (size_t)np);
    const unsigned nbp = (unsigned)((np + 255) / 256);
    hipLaunchKernelGGL(uph_bucket_key_kernel, dim3(nbp), dim3(256), 0, 0, fx, fy, np, bx0, by0, bsize, bnx, bny, d_key, d_idx);
    int key_bits = 1;
    while (((size_t)1 << key_bits) < nbuck + 1 && key_bits < 31) key_bits++;
    hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_key, d_skey, d_idx, d_sidx, np, 0, key_bits, 0);      // stable: filtered-cloud order inside a bucket
    hipLaunchKernelGGL(uph_pack_pts_kernel, dim3(nbp), dim3(256), 0, 0, d_sidx, np, fx, fy, fz, d_pts);
    hipLaunchKernelGGL(uph_bstart_kernel, dim3((unsigned)((nbuck + 1 + 255) / 256)), dim3(256), 0, 0, d_skey, np, (int)nbuck, d_bstart);
    // LDS capacity of the staging area: the largest point count of any disc the kernel stages (uph_disc_cap_kernel: the kernel's own predicate over
    // the columns of this slab) -- exact, so a cloud fits whenever its densest disc does (9 600 points = 150 KB)
    CloudDev cd;
    cd.pts = d_pts; cd.bstart = d_bstart; cd.bx0 = bx0; cd.by0 = by0; cd.bsize = bsize; cd.bnx = bnx; cd.bny = bny; cd.npts = np;
    const int ncol = (x1 - x0) * g.ny;
    int cap = 64;
    HIPCHK(hipMemcpy(d_small + 8, &cap, 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(uph_disc_cap_kernel, dim3(ncol), dim3(64), 0, 0, g, cd, (int)x0, (int)x1, m->mp.ellipsoid_x, m->mp.ellipsoid_y, m->mp.ellipsoid_z, (int*)(d_small + 8));
    int ovf0 = 0;
    HIPCHK(hipMemcpy(d_small + 9, &ovf0, 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(&cap, d_small + 8, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipGetLastError());
    m->stage_ms[2] = lap();
    const size_t lds_bytes = (size_t)cap * sizeof(float4);
    if (lds_bytes > 150 * 1024) { setError("uph_map_build: cloud too dense for the LDS staging window (" + std::to_string(cap) + " points within the staging radius of one cell centre; 9600 fit)"); return UPH_ERR_LIMIT; }
    HIPCHK(hipFuncSetAttribute((const void*)uph_map_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (!m->bev0) { HIPCHK(hipEventCreate(&m->bev0)); HIPCHK(hipEventCreate(&m->bev1)); }
    HIPCHK(hipEventRecord(m->bev0, 0));
    hipLaunchKernelGGL(uph_map_build_kernel, dim3(ncol), dim3(64), lds_bytes, 0, g, cd, m->d_cells, (int)x0, (int)x1, (int)m->mp.iter_num, m->mp.ellipsoid_x,
                       m->mp.ellipsoid_y, m->mp.ellipsoid_z, cap, (int*)(d_small + 9));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(m->bev1, 0));
    int ovf = 0;
    HIPCHK(hipMemcpy(&ovf, d_small + 9, sizeof(int), hipMemcpyDeviceToHost));      // (synchronises with the kernel)
    if (ovf) { setError("uph_map_build: a staged neighbourhood exceeded the LDS window (internal sizing error)"); return UPH_ERR_LIMIT; }
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, m->bev0, m->bev1));
    m->last_build_ms = ms;
    m->stage_ms[3] = ms; (void)lap();
    m->last_cell_iters = (int64_t)ncol * g.nyaw * m->mp.iter_num;
    m->last_cloud = (int64_t)np;
    m->last_raw = n;
    const int rc = commit ? commitMap(m) : UPH_OK;
    m->stage_ms[4] = lap();
    m->stage_ms[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return rc;
}

// stages of the last uph_map_build on this map, milliseconds: [0] cloud upload, [1] crop box + voxel filter (device), [2] bucketing + LDS sizing (device),
// [3] plane-fit kernel (HIP events), [4] commit (c, occupancy), [5] the whole call (wall)
int uph_map_build_stages(uph_map* m, double* out6) {
    if (!m || !out6) return UPH_ERR_INVALID;
    for (int k = 0; k < 6; k++) out6[k] = m->stage_ms[k];
    return UPH_OK;
}
// the cloud the last uph_map_build fitted planes to, as the device filtered it (crop box + voxel grid): at most cap points into out_xyz, returns
// the count.  Diagnostic / test hook: must equal uph_map_filter_cloud (the host form) bit for bit.
int64_t uph_map_built_cloud(uph_map* m, float* out_xyz, int64_t cap) {
    if (!m || m->last_cloud <= 0 || !m->bscr[13]) { setError("uph_map_built_cloud: no build on this map"); return UPH_ERR_INVALID; }
    if (hipSetDevice(m->device) != hipSuccess) return UPH_ERR_HIP;
    const int64_t np = m->last_cloud;
    if (out_xyz) {
        std::vector<float4> pts((size_t)np);
        if (hipMemcpy(pts.data(), m->bscr[13], sizeof(float4) * (size_t)np, hipMemcpyDeviceToHost) != hipSuccess) { setError("uph_map_built_cloud: hipMemcpy failed"); return UPH_ERR_HIP; }
        for (const float4& p : pts) {                         // bucket order -> filtered-cloud order through the index carried in .w
            int idx; std::memcpy(&idx, &p.w, 4);
            if (idx >= 0 && idx < cap) { out_xyz[3 * (size_t)idx] = p.x; out_xyz[3 * (size_t)idx + 1] = p.y; out_xyz[3 * (size_t)idx + 2] = p.z; }
        }
    }
    return np;
}


// ---- one grid over several GPUs of ONE process (SURVEY.md 8b "Environment", 8e row 2; BASELINE.json configs[3]) ------------------------
// The reference is a single process (plan_manager/src/manager_node.cpp, ros::spin()), so the sharded build has to be reachable from one
// host thread: the caller hands over one map per device, device g produces the x-slab [g per, (g+1) per) of the cell array (x is the
// slowest index, uneven_map.h:427-435: a slab is one contiguous block) on its own host thread, ONE ncclAllGather over an in-process RCCL
// clique (ncclCommInitAll) leaves the whole array on every device, every map commits (c, occupancy).  When nx is a multiple of the device
// count the gather runs IN PLACE on the cell arrays (send = the slab where it lies); otherwise through zero-padded staging.
// Cliques are cached per device list for the life of the process (creating one costs ~1 s); uph_multi_shutdown releases them.
}  // extern "C"
namespace {
struct Clique {
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
};
std::mutex g_clique_mu;
std::vector<Clique*> g_cliques;

Clique* getClique(const std::vector<int>& devs, const RcclApi* api, std::string& why) {
    std::lock_guard<std::mutex> lk(g_clique_mu);
    for (Clique* q : g_cliques) if (q->devs == devs) return q;
    Clique* q = new Clique();
    q->devs = devs;
    q->comms.assign(devs.size(), nullptr);
    const ncclResult_t r = api->CommInitAll(q->comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) { why = std::string("ncclCommInitAll: ") + api->GetErrorString(r); delete q; return nullptr; }
    q->streams.assign(devs.size(), nullptr);
    for (size_t g = 0; g < devs.size(); g++) {
        if (hipSetDevice(devs[g]) != hipSuccess || hipStreamCreateWithFlags(&q->streams[g], hipStreamNonBlocking) != hipSuccess) {
            why = "hipStreamCreate for the RCCL clique failed";
            for (size_t h = 0; h < devs.size(); h++) {      // give back what was made
                hipSetDevice(devs[h]);
                if (q->streams[h]) hipStreamDestroy(q->streams[h]);
                if (q->comms[h]) api->CommDestroy(q->comms[h]);
            }
            delete q;
            return nullptr;
        }
    }
    g_cliques.push_back(q);
    return q;
}

// slab rule shared with the host mirrors (uneven_planner_amd/uneven_map.py slab_bounds): per = ceil(nx / n)
inline void slabOf(int nx, int g, int n, int& per, int& x0, int& x1) { per = (nx + n - 1) / n; x0 = std::min(g * per, nx); x1 = std::min(x0 + per, nx); }
// the all-gather runs IN PLACE on the cell arrays (send buffer = the slab where it lies) when every slab is full; otherwise through staging of
// n x per zero-padded rows
inline bool slabsInPlace(int nx, int n) { return nx % n == 0; }

int checkMultiMaps(uph_map* const* maps, int n, const char* who, bool need_f64) {
    if (!maps || n < 1) { setError(std::string(who) + ": bad arguments"); return UPH_ERR_INVALID; }
    for (int g = 0; g < n; g++) {
        const uph_map* m = maps[g];
        if (!m) { setError(std::string(who) + ": null map"); return UPH_ERR_INVALID; }
        for (int h = 0; h < g; h++) if (maps[h] == m) { setError(std::string(who) + ": the same map twice"); return UPH_ERR_INVALID; }
        if (m->g.nx_hold != m->g.nx) { setError(std::string(who) + ": tile maps hold their own rows only (nothing to gather)"); return UPH_ERR_INVALID; }
        if (need_f64 && m->d_cells32) { setError(std::string(who) + ": the plane fit writes fp64 cells; create the maps with uph_map_create"); return UPH_ERR_INVALID; }
        const uph_map_params &a = m->mp, &b = maps[0]->mp;          // field by field: the struct has padding after iter_num, memcmp would compare it
        const bool same = a.iter_num == b.iter_num && a.map_size_x == b.map_size_x && a.map_size_y == b.map_size_y && a.ellipsoid_x == b.ellipsoid_x &&
                          a.ellipsoid_y == b.ellipsoid_y && a.ellipsoid_z == b.ellipsoid_z && a.xy_resolution == b.xy_resolution && a.yaw_resolution == b.yaw_resolution &&
                          a.min_cnormal == b.min_cnormal && a.max_rho == b.max_rho && a.gravity == b.gravity;
        if (m->g.nx != maps[0]->g.nx || m->g.ny != maps[0]->g.ny || m->g.nyaw != maps[0]->g.nyaw || (m->d_cells32 != nullptr) != (maps[0]->d_cells32 != nullptr) || !same) {
            setError(std::string(who) + ": the maps differ in parameters or storage"); return UPH_ERR_INVALID;
        }
    }
    return UPH_OK;
}

// the *_multi entry points visit every device; the caller's current HIP device is left as it was found (a PyTorch host relies on it)
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

double wallMs(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }

// every map holds its own slab; afterwards every map holds all slabs
int exchangeSlabs(uph_map* const* maps, int n, const char* who) {
    uph_map* lead = maps[0];
    const int nx = lead->g.nx;
    const size_t row = (size_t)lead->g.ny * lead->g.nyaw * cellBytes(lead);
    int per, a0, a1;
    slabOf(nx, 0, n, per, a0, a1);
    lead->multi_ms[3] = 0.0;
    if (n == 1) { lead->multi_rccl = 0; return UPH_OK; }
    std::vector<int> devs(n);
    bool distinct = true;
    for (int g = 0; g < n; g++) { devs[g] = maps[g]->device; for (int h = 0; h < g; h++) distinct &= devs[h] != devs[g]; }
    if (!distinct) {
        // several maps on one device: RCCL refuses a clique with a device twice.  This is the single-GPU test configuration (a "world" of
        // slabs played on one device); the exchange is plain device-to-device copies, the slab rule and everything around it is the same.
        lead->multi_rccl = 0;
        for (int g = 0; g < n; g++) {
            int x0, x1;
            slabOf(nx, g, n, per, x0, x1);
            if (x1 <= x0) continue;
            for (int h = 0; h < n; h++) {
                if (h == g) continue;
                HIPCHK(hipMemcpyPeer(cellBase(maps[h]) + (size_t)x0 * row, maps[h]->device, cellBase(maps[g]) + (size_t)x0 * row, maps[g]->device, (size_t)(x1 - x0) * row));
            }
        }
        return UPH_OK;
    }
    std::string why;
    const RcclApi* api = rcclApi(why);
    if (!api) { setError(std::string(who) + ": " + why); return UPH_ERR_HIP; }
    Clique* q = getClique(devs, api, why);
    if (!q) { setError(std::string(who) + ": " + why); return UPH_ERR_HIP; }
    lead->multi_rccl = 1;
    const bool inplace = slabsInPlace(nx, n);
    const size_t slab_bytes = (size_t)per * row;
    const ncclDataType_t dt = lead->d_cells32 ? ncclFloat : ncclDouble;
    const size_t count = slab_bytes / (lead->d_cells32 ? sizeof(float) : sizeof(double));
    std::vector<UphDevTmp> stage(n);
    if (!inplace) {
        for (int g = 0; g < n; g++) {
            int x0, x1;
            slabOf(nx, g, n, per, x0, x1);
            HIPCHK(hipSetDevice(devs[g]));
            HIPCHK(hipMalloc(&stage[g].p, slab_bytes * n));
            HIPCHK(hipMemsetAsync((char*)stage[g].p + (size_t)g * slab_bytes, 0, slab_bytes, q->streams[g]));
            if (x1 > x0) HIPCHK(hipMemcpyAsync((char*)stage[g].p + (size_t)g * slab_bytes, cellBase(maps[g]) + (size_t)x0 * row, (size_t)(x1 - x0) * row, hipMemcpyDeviceToDevice, q->streams[g]));
        }
    }
    UphEventTmp e0, e1;
    HIPCHK(hipSetDevice(devs[0]));
    HIPCHK(hipEventCreate((hipEvent_t*)&e0.e)); HIPCHK(hipEventCreate((hipEvent_t*)&e1.e));
    HIPCHK(hipEventRecord((hipEvent_t)e0.e, q->streams[0]));
    ncclResult_t r = api->GroupStart();
    for (int g = 0; g < n && r == ncclSuccess; g++) {
        char* base = inplace ? cellBase(maps[g]) : (char*)stage[g].p;
        r = api->AllGather(base + (size_t)g * slab_bytes, base, count, dt, q->comms[g], q->streams[g]);
    }
    const ncclResult_t re = api->GroupEnd();
    if (r == ncclSuccess) r = re;
    if (r != ncclSuccess) { setError(std::string(who) + ": ncclAllGather: " + api->GetErrorString(r)); return UPH_ERR_HIP; }
    HIPCHK(hipSetDevice(devs[0]));
    HIPCHK(hipEventRecord((hipEvent_t)e1.e, q->streams[0]));
    for (int g = 0; g < n; g++) {
        HIPCHK(hipSetDevice(devs[g]));
        if (!inplace) HIPCHK(hipMemcpyAsync(cellBase(maps[g]), stage[g].p, (size_t)nx * row, hipMemcpyDeviceToDevice, q->streams[g]));
        HIPCHK(hipStreamSynchronize(q->streams[g]));
    }
    float ms = 0.f;
    HIPCHK(hipSetDevice(devs[0]));
    HIPCHK(hipEventElapsedTime(&ms, (hipEvent_t)e0.e, (hipEvent_t)e1.e));
    lead->multi_ms[3] = ms;
    for (int g = 0; g < n; g++) if (stage[g].p) { hipSetDevice(devs[g]); hipFree(stage[g].p); stage[g].p = nullptr; }
    return UPH_OK;
}

// fits on all devices at once (one host thread each), exchange, commit
template <class Fit>
int multiBuild(uph_map* const* maps, int n, const char* who, Fit fit) {
    uph_map* lead = maps[0];
    const int nx = lead->g.nx;
    std::vector<int> rc(n, UPH_OK);
    std::vector<std::string> err(n);
    auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> th;
        for (int g = 0; g < n; g++) {
            int per, x0, x1;
            slabOf(nx, g, n, per, x0, x1);
            if (x1 <= x0) continue;
            auto work = [&, g, x0, x1]() { rc[g] = fit(maps[g], x0, x1); if (rc[g] != UPH_OK) err[g] = uph_last_error(); };
            try { th.emplace_back(work); }
            catch (...) { work(); }       // no thread to be had: this slab is fitted here (nothing throws across the ABI)
        }
        for (auto& t : th) t.join();
    }
    for (int g = 0; g < n; g++) if (rc[g] != UPH_OK) { setError(std::string(who) + ": slab " + std::to_string(g) + ": " + err[g]); return rc[g]; }
    lead->multi_ms[0] = wallMs(t0);
    t0 = std::chrono::steady_clock::now();
    const int r = exchangeSlabs(maps, n, who);
    if (r != UPH_OK) return r;
    lead->multi_ms[1] = wallMs(t0);
    t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> th;
        for (int g = 0; g < n; g++) {
            auto work = [&, g]() { rc[g] = uph_map_commit(maps[g]); if (rc[g] != UPH_OK) err[g] = uph_last_error(); };
            try { th.emplace_back(work); }
            catch (...) { work(); }
        }
        for (auto& t : th) t.join();
    }
    for (int g = 0; g < n; g++) if (rc[g] != UPH_OK) { setError(std::string(who) + ": commit " + std::to_string(g) + ": " + err[g]); return rc[g]; }
    lead->multi_ms[2] = wallMs(t0);
    return UPH_OK;
}
}  // namespace
extern "C" {

// the host-side decisions of the *_multi builds, without a device: which x-slab each of n_gpus devices fits and how the slabs are exchanged
int uph_multi_slab_plan(int32_t nx, int32_t n_gpus, int32_t* x0, int32_t* x1, int32_t* per_out, int32_t* in_place) {
    if (nx < 1 || n_gpus < 1 || !x0 || !x1) { setError("uph_multi_slab_plan: bad arguments"); return UPH_ERR_INVALID; }
    int per = 0;
    for (int g = 0; g < n_gpus; g++) { int a, b; slabOf(nx, g, n_gpus, per, a, b); x0[g] = a; x1[g] = b; }
    if (per_out) *per_out = per;
    if (in_place) *in_place = n_gpus > 1 && slabsInPlace(nx, n_gpus) ? 1 : 0;
    return UPH_OK;
}

int uph_map_build_multi(uph_map* const* maps, int32_t n_gpus, const float* xyz, int64_t n) {
    DeviceRestore keep;
    int r = checkMultiMaps(maps, n_gpus, "uph_map_build_multi", true);
    if (r != UPH_OK) return r;
    if (!xyz || n <= 0) { setError("uph_map_build_multi: bad arguments"); return UPH_ERR_INVALID; }
    return multiBuild(maps, n_gpus, "uph_map_build_multi", [&](uph_map* m, int x0, int x1) { return buildSlab(m, xyz, n, x0, x1, false); });
}

int uph_map_fill_fbm_multi(uph_map* const* maps, int32_t n_gpus, const uph_fbm_params* fp) {
    DeviceRestore keep;
    int r = checkMultiMaps(maps, n_gpus, "uph_map_fill_fbm_multi", false);
    if (r != UPH_OK) return r;
    if (!checkFbm(fp)) { setError("uph_map_fill_fbm_multi: bad arguments"); return UPH_ERR_INVALID; }
    return multiBuild(maps, n_gpus, "uph_map_fill_fbm_multi", [&](uph_map* m, int x0, int x1) { return fillSlab(m, fp, x0, x1, false); });
}

int uph_map_multi_stats(uph_map* lead, double* fit_ms, double* exchange_ms, double* commit_ms, double* exchange_device_ms, int32_t* via_rccl) {
    if (!lead) return UPH_ERR_INVALID;
    if (fit_ms) *fit_ms = lead->multi_ms[0];
    if (exchange_ms) *exchange_ms = lead->multi_ms[1];
    if (commit_ms) *commit_ms = lead->multi_ms[2];
    if (exchange_device_ms) *exchange_device_ms = lead->multi_ms[3];
    if (via_rccl) *via_rccl = lead->multi_rccl;
    return UPH_OK;
}

// diagnostic: binds RCCL as the sharded build would, forms the clique of devices 0 .. n_devices-1 and all-gathers a known pattern (out of
// place, 4096 doubles per device).  0 = every device received every block; the bound library and RCCL version go to `info`.
int uph_rccl_selftest(int32_t n_devices, char* info, int32_t info_cap) {
    DeviceRestore keep;
    int have = 0;
    if (n_devices < 1 || hipGetDeviceCount(&have) != hipSuccess || have < n_devices) { setError("uph_rccl_selftest: not that many devices"); return UPH_ERR_INVALID; }
    std::string why;
    const RcclApi* api = rcclApi(why);
    if (!api) { setError("uph_rccl_selftest: " + why); return UPH_ERR_HIP; }
    std::vector<int> devs(n_devices);
    for (int g = 0; g < n_devices; g++) devs[g] = g;
    Clique* q = getClique(devs, api, why);
    if (!q) { setError("uph_rccl_selftest: " + why); return UPH_ERR_HIP; }
    int ver = 0;
    if (api->GetVersion) api->GetVersion(&ver);
    if (info && info_cap > 0) std::snprintf(info, (size_t)info_cap, "%s, RCCL version code %d, clique of %d", api->origin.c_str(), ver, n_devices);
    const size_t cnt = 4096;
    std::vector<UphDevTmp> snd(n_devices), rcv(n_devices);
    std::vector<double> host(cnt);
    for (int g = 0; g < n_devices; g++) {
        HIPCHK(hipSetDevice(g));
        HIPCHK(hipMalloc(&snd[g].p, cnt * 8)); HIPCHK(hipMalloc(&rcv[g].p, cnt * 8 * n_devices));
        for (size_t i = 0; i < cnt; i++) host[i] = 1000.0 * g + (double)i;
        HIPCHK(hipMemcpy(snd[g].p, host.data(), cnt * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(rcv[g].p, 0, cnt * 8 * n_devices));
    }
    ncclResult_t r = api->GroupStart();
    for (int g = 0; g < n_devices && r == ncclSuccess; g++) r = api->AllGather(snd[g].p, rcv[g].p, cnt, ncclDouble, q->comms[g], q->streams[g]);
    const ncclResult_t re = api->GroupEnd();
    if (r == ncclSuccess) r = re;
    if (r != ncclSuccess) { setError(std::string("uph_rccl_selftest: ncclAllGather: ") + api->GetErrorString(r)); return UPH_ERR_HIP; }
    std::vector<double> back(cnt * n_devices);
    for (int g = 0; g < n_devices; g++) {
        HIPCHK(hipSetDevice(g));
        HIPCHK(hipStreamSynchronize(q->streams[g]));
        HIPCHK(hipMemcpy(back.data(), rcv[g].p, cnt * 8 * n_devices, hipMemcpyDeviceToHost));
        for (int h = 0; h < n_devices; h++)
            for (size_t i = 0; i < cnt; i++)
                if (back[h * cnt + i] != 1000.0 * h + (double)i) { setError("uph_rccl_selftest: wrong data after the all-gather"); return UPH_ERR_HIP; }
    }
    for (int g = 0; g < n_devices; g++) { hipSetDevice(g); hipFree(snd[g].p); snd[g].p = nullptr; hipFree(rcv[g].p); rcv[g].p = nullptr; }
    return UPH_OK;
}

void uph_multi_shutdown(void) {
    DeviceRestore keep;
    std::lock_guard<std::mutex> lk(g_clique_mu);
    std::string why;
    const RcclApi* api = rcclApi(why);
    for (Clique* q : g_cliques) {
        for (size_t g = 0; g < q->devs.size(); g++) {
            hipSetDevice(q->devs[g]);
            if (q->streams[g]) hipStreamDestroy(q->streams[g]);
            if (api && q->comms[g]) api->CommDestroy(q->comms[g]);
        }
        delete q;
    }
    g_cliques.clear();
}

/* the cloud the map is built from: UnevenMap::init's CropBox [-10,10]^2 x [-0.01,5] + VoxelGrid 1 cm (uneven_map.cpp:133-143) applied to
 * xyz (n points); out_xyz receives at most cap points (may be NULL to query the count); returns the number of filtered points or < 0 */
int64_t uph_map_filter_cloud(const float* xyz, int64_t n, float* out_xyz, int64_t cap) {
    if (!xyz || n <= 0) { setError("uph_map_filter_cloud: bad arguments"); return UPH_ERR_INVALID; }
    const HostCloud cl = cropAndVoxel(xyz, n);
    const int64_t np = (int64_t)cl.size();
    if (out_xyz) for (int64_t i = 0; i < np && i < cap; i++) { out_xyz[3 * i] = cl.x[i]; out_xyz[3 * i + 1] = cl.y[i]; out_xyz[3 * i + 2] = cl.z[i]; }
    return np;
}


/* batched front-end cost queries: what the kinodynamic A* asks the map for at every expanded state */
int uph_frontend_query(uph_map* m, const double* pos, int32_t n, double* sigma, int32_t* occ, int32_t* occ_xy) {
    if (!m || !pos || n <= 0 || (!sigma && !occ && !occ_xy)) { setError("uph_frontend_query: bad arguments"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(m->device));
    UphPtr tp, ts, t1, t2;
    tp.p = uphMapScratch(m, 0, 8 * 3 * (size_t)n); ts.p = uphMapScratch(m, 1, 8 * (size_t)n);
    t1.p = uphMapScratch(m, 2, 4 * (size_t)n); t2.p = uphMapScratch(m, 3, 4 * (size_t)n);
    if (!tp.p || !ts.p || !t1.p || !t2.p) return UPH_ERR_HIP;
    HIPCHK(hipMemcpy(tp.p, pos, 8 * 3 * (size_t)n, hipMemcpyHostToDevice));
    UphEventTmp e0, e1;
    HIPCHK(hipEventCreate((hipEvent_t*)&e0.e)); HIPCHK(hipEventCreate((hipEvent_t*)&e1.e));
    HIPCHK(hipEventRecord((hipEvent_t)e0.e, 0));
    hipLaunchKernelGGL(uph_frontend_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, m->g, m->d_occ, m->d_occ2, tp.as<double>(), n, ts.as<double>(), t1.as<int>(), t2.as<int>());
    HIPCHK(hipEventRecord((hipEvent_t)e1.e, 0));
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, (hipEvent_t)e0.e, (hipEvent_t)e1.e));
    m->last_query_ms = ms;
    if (sigma) HIPCHK(hipMemcpy(sigma, ts.p, 8 * (size_t)n, hipMemcpyDeviceToHost));
    if (occ) HIPCHK(hipMemcpy(occ, t1.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    if (occ_xy) HIPCHK(hipMemcpy(occ_xy, t2.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    return UPH_OK;
}
/* batched UnevenMap::getTerrainPos (uneven_map.h:203-218) on the device grid: pose12[n][12] = rotation (column-major: x_b, y_b, z_b), position */
int uph_terrain_pose_query(uph_map* m, const double* pos, int32_t n, double* pose12) {
    if (!m || !pos || n <= 0 || !pose12) { setError("uph_terrain_pose_query: bad arguments"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(m->device));
    UphPtr tp, to;
    tp.p = uphMapScratch(m, 0, 8 * 3 * (size_t)n); to.p = uphMapScratch(m, 1, 8 * 12 * (size_t)n);
    if (!tp.p || !to.p) return UPH_ERR_HIP;
    HIPCHK(hipMemcpy(tp.p, pos, 8 * 3 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(uph_pose_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, m->g, tp.as<double>(), n, to.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(pose12, to.p, 8 * 12 * (size_t)n, hipMemcpyDeviceToHost));
    return UPH_OK;
}
int uph_frontend_query_ms(uph_map* m, double* kernel_ms) { if (!m || !kernel_ms) return UPH_ERR_INVALID; *kernel_ms = m->last_query_ms; return UPH_OK; }

}  // extern "C"
