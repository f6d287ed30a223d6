// A/B of the terrain-lookup memory path of the penalty kernel (north_star: "terrain grid read coalesced from HBM and staged through
// LDS tiles").  Three ways to fetch the 8 trilinear corners x (sigma, zb.x, zb.y) of every constraint sample, on the same grid
// (200 x 200 x 64 cells) and the same samples (trajectory pieces of 0.3 m with 17 samples each, random position / heading /
// curvature, i.e. ~2 cm between samples against 5 cm cells):
//   planes : three field planes, 24 scattered 8-byte loads per sample                       (round-1 form, terrain_dev.hpp terrainBase)
//   cells  : array of 32-byte cells {z, sigma, zb.x, zb.y}, 16 x 16-byte loads per sample   (shipped form, terrainBaseCells)
//   tile   : one wave per piece stages the piece's bounding box of cells (x, y, yaw) into LDS with coalesced row loads, then every
//            sample interpolates from LDS
// Prints ns per sample and a checksum per variant (the three must agree).  usage: terrain_gather [pieces]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
constexpr int NX = 200, NY = 200, NW = 64, K1 = 17;
constexpr double RES = 0.05, WRES = 0.1, OX = -5.0, OY = -5.0, OW = -3.1665926535897932;

struct Loc { int ix, iy, iw; double dx, dy, dw; };
__device__ __forceinline__ Loc locate(double x, double y, double w) {
    Loc l;
    const double xm = x - 0.5 * RES, ym = y - 0.5 * RES, wm = w - 0.5 * WRES;
    l.ix = (int)floor((xm - OX) / RES); l.iy = (int)floor((ym - OY) / RES); l.iw = (int)floor((wm - OW) / WRES);
    l.dx = (x - ((l.ix + 0.5) * RES + OX)) / RES; l.dy = (y - ((l.iy + 0.5) * RES + OY)) / RES; l.dw = (w - ((l.iw + 0.5) * WRES + OW)) / WRES;
    l.ix = min(max(l.ix, 0), NX - 2); l.iy = min(max(l.iy, 0), NY - 2); l.iw = min(max(l.iw, 0), NW - 2);
    return l;
}
__device__ __forceinline__ double tri(const double v[2][2][2], const Loc& l) {
    const double a = (v[0][0][0] * (1 - l.dx) + v[1][0][0] * l.dx) * (1 - l.dy) + (v[0][1][0] * (1 - l.dx) + v[1][1][0] * l.dx) * l.dy;
    const double b = (v[0][0][1] * (1 - l.dx) + v[1][0][1] * l.dx) * (1 - l.dy) + (v[0][1][1] * (1 - l.dx) + v[1][1][1] * l.dx) * l.dy;
    return a * (1 - l.dw) + b * l.dw;
}
__global__ void k_planes(const double* __restrict__ pl, const double* __restrict__ pos, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Loc l = locate(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    double acc = 0.0;
    for (int f = 0; f < 3; f++) {
        const double* p = pl + (size_t)f * NX * NY * NW;
        double v[2][2][2];
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int c = 0; c < 2; c++) v[a][b][c] = p[((size_t)(l.ix + a) * NY + l.iy + b) * NW + l.iw + c];
        acc += tri(v, l);
    }
    out[i] = acc;
}
__global__ void k_cells(const double2* __restrict__ ce, const double* __restrict__ pos, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Loc l = locate(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    double v[3][2][2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int c = 0; c < 2; c++) {
        const double2* p = ce + 2 * (((size_t)(l.ix + a) * NY + l.iy + b) * NW + l.iw + c);
        const double2 lo = p[0], hi = p[1];
        v[0][a][b][c] = lo.y; v[1][a][b][c] = hi.x; v[2][a][b][c] = hi.y;
    }
    out[i] = tri(v[0], l) + tri(v[1], l) + tri(v[2], l);
}
// one wave64 per piece (17 samples in lanes 0..16): bounding box of the piece's cells -> LDS tile [bx][by][bw][3], rows loaded coalesced
constexpr int TX = 9, TY = 9, TW = 9;
__global__ __launch_bounds__(64) void k_tile(const double2* __restrict__ ce, const double* __restrict__ pos, int npieces, double* __restrict__ out, int* __restrict__ overflow) {
    __shared__ double tile[TX * TY * TW * 3];
    const int piece = blockIdx.x, lane = threadIdx.x;
    const int i = piece * K1 + (lane < K1 ? lane : K1 - 1);
    const Loc l = locate(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    int x0 = l.ix, x1 = l.ix + 1, y0 = l.iy, y1 = l.iy + 1, w0 = l.iw, w1 = l.iw + 1;
    for (int off = 32; off >= 1; off >>= 1) {
        x0 = min(x0, __shfl_xor(x0, off)); x1 = max(x1, __shfl_xor(x1, off)); y0 = min(y0, __shfl_xor(y0, off)); y1 = max(y1, __shfl_xor(y1, off));
        w0 = min(w0, __shfl_xor(w0, off)); w1 = max(w1, __shfl_xor(w1, off));
    }
    const int sx = x1 - x0 + 1, sy = y1 - y0 + 1, sw = w1 - w0 + 1;
    if (sx > TX || sy > TY || sw > TW) { if (lane == 0) atomicAdd(overflow, 1); if (lane < K1) out[i] = 0.0; return; }
    const int ncell = sx * sy * sw;
    for (int t = lane; t < ncell; t += 64) {                      // yaw fastest: consecutive lanes read consecutive cells of a (x, y) column
        const int cw = t % sw, cy = (t / sw) % sy, cx = t / (sw * sy);
        const double2* p = ce + 2 * (((size_t)(x0 + cx) * NY + y0 + cy) * NW + w0 + cw);
        const double2 lo = p[0], hi = p[1];
        double* d = tile + (size_t)t * 3;
        d[0] = lo.y; d[1] = hi.x; d[2] = hi.y;
    }
    __syncthreads();
    if (lane >= K1) return;
    double v[3][2][2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int c = 0; c < 2; c++) {
        const double* d = tile + (size_t)(((l.ix + a - x0) * sy + (l.iy + b - y0)) * sw + (l.iw + c - w0)) * 3;
        v[0][a][b][c] = d[0]; v[1][a][b][c] = d[1]; v[2][a][b][c] = d[2];
    }
    out[i] = tri(v[0], l) + tri(v[1], l) + tri(v[2], l);
}

int main(int argc, char** argv) {
    const int npieces = argc > 1 ? std::atoi(argv[1]) : 400000;
    const int n = npieces * K1;
    const size_t ncell = (size_t)NX * NY * NW;
    std::vector<double> planes(3 * ncell), cells(4 * ncell), pos(3 * (size_t)n);
    unsigned long long s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
    for (size_t c = 0; c < ncell; c++) { const double a = rnd(), b = rnd() - 0.5, d = rnd() - 0.5; planes[c] = a; planes[ncell + c] = b; planes[2 * ncell + c] = d; cells[4 * c] = 0.0; cells[4 * c + 1] = a; cells[4 * c + 2] = b; cells[4 * c + 3] = d; }
    for (int p = 0; p < npieces; p++) {                            // a 0.3 m arc: start, heading, curvature up to 2.1 1/m (max_kap)
        double x = -4.0 + 8.0 * rnd(), y = -4.0 + 8.0 * rnd(), h = -2.8 + 5.6 * rnd();
        const double kap = (rnd() - 0.5) * 4.2, ds = 0.3 / 16.0;
        for (int j = 0; j < K1; j++) { pos[3 * ((size_t)p * K1 + j)] = x; pos[3 * ((size_t)p * K1 + j) + 1] = y; pos[3 * ((size_t)p * K1 + j) + 2] = h; x += ds * std::cos(h); y += ds * std::sin(h); h += kap * ds; if (h > 2.9) h = 2.9; if (h < -2.9) h = -2.9; }
    }
    double *dpl, *dce, *dpos, *dout; int* dovf;
    CHK(hipMalloc((void**)&dpl, planes.size() * 8)); CHK(hipMalloc((void**)&dce, cells.size() * 8)); CHK(hipMalloc((void**)&dpos, pos.size() * 8));
    CHK(hipMalloc((void**)&dout, (size_t)n * 8)); CHK(hipMalloc((void**)&dovf, 4));
    CHK(hipMemcpy(dpl, planes.data(), planes.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dce, cells.data(), cells.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dpos, pos.data(), pos.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemset(dovf, 0, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    std::vector<double> out(n);
    for (int var = 0; var < 3; var++) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CHK(hipEventRecord(e0, 0));
            if (var == 0) hipLaunchKernelGGL(k_planes, dim3((n + 255) / 256), dim3(256), 0, 0, dpl, dpos, n, dout);
            else if (var == 1) hipLaunchKernelGGL(k_cells, dim3((n + 255) / 256), dim3(256), 0, 0, (const double2*)dce, dpos, n, dout);
            else hipLaunchKernelGGL(k_tile, dim3(npieces), dim3(64), 0, 0, (const double2*)dce, dpos, npieces, dout, dovf);
            CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        CHK(hipMemcpy(out.data(), dout, (size_t)n * 8, hipMemcpyDeviceToHost));
        double cs = 0; for (double v : out) cs += v;
        int ovf = 0; CHK(hipMemcpy(&ovf, dovf, 4, hipMemcpyDeviceToHost));
        std::printf("%-6s %8.3f ms  %6.3f ns/sample  %7.1f G corner-values/s  checksum %.9e%s\n", var == 0 ? "planes" : (var == 1 ? "cells" : "tile"), best, best * 1e6 / n,
                    24.0 * n / best / 1e6, cs, var == 2 && ovf ? "  (tile overflow on some pieces!)" : "");
    }
    return 0;
}
