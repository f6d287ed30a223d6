// Device terrain lookup: trilinear SE(2) interpolation of (sigma, zb) with analytic gradients and the derived
// attitude terms.  Twin of UnevenMap::getTerrainWithGradI / getAllWithGrad / getTerrain / getTerrainVariables
// (uneven_map/include/uneven_map/uneven_map.h:154-201, 221-256, 258-315, 318-377), index helpers :398-454.
// Reads the array of cells of GridDev (the build output itself): {z, sigma, zb.x, zb.y} per cell, yaw fastest.
#pragma once
#include "uph_common.hpp"


namespace uph {

template <class R>
struct CornersT {
    R dx, dy, dyaw;            // fractional offsets diff[0..2]
    uint32_t a[2][2];          // cell index of the (x,y) corner at yaw bin 0 (held rows; < 2^31, checked when the map is created)
    int w0, w1;                // the two yaw bins
    bool inmap;
};
typedef CornersT<double> Corners;

// R = double (the reference's arithmetic) or f32r (fp32 sample mode): the same code, see uph_common.hpp
template <class R>
UPH_HD bool isInMap(const GridDev& g, R x, R y, R yaw) {     // uneven_map.h:437-454
    if (x < g.lo[0] || y < g.lo[1] || yaw < g.lo[2]) return false;      // lo = minb + 1e-4, hi = maxb - 1e-4
    if (x > g.hi[0] || y > g.hi[1] || yaw > g.hi[2]) return false;
    return true;
}

UPH_HD int clampIdx(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }      // boundIndex for x / y (uneven_map.h:398-409); one v_med3_i32 on the device

template <class R>
UPH_HD void locate(const GridDev& g, R x, R y, R yaw, CornersT<R>& c) {
    c.inmap = isInMap<R>(g, x, y, yaw);
    if (!c.inmap) return;
    // uneven_map.h:268-284
    R xm = x - g.half_xy, ym = y - g.half_xy;
    R wm = normSO2(R(yaw - g.half_yaw));
    int ix = (int)floor((xm - g.origin[0]) * g.xy_inv);
    int iy = (int)floor((ym - g.origin[1]) * g.xy_inv);
    int iw = (int)floor((wm - g.origin[2]) * g.yaw_inv);
    R cx = (ix + 0.5) * g.xy_res + g.origin[0];
    R cy = (iy + 0.5) * g.xy_res + g.origin[1];
    R cw = (iw + 0.5) * g.yaw_res + g.origin[2];
    c.dx = (x - cx) * g.xy_inv;
    c.dy = (y - cy) * g.xy_inv;
    // reference: atan2(sin(d), cos(d)) * yaw_inv  (:284).  d lies within one wrap of a yaw cell, so the exact range
    // reduction d - 2*pi*rint(d/(2*pi)) gives the same angle to ~1e-17 without three transcendentals.  The quotient only feeds rint():
    // d times the rounded reciprocal picks the same integer (|d| stays below one cell + one wrap, nowhere near a half-integer quotient)
    // and spares the twelve-instruction IEEE division sequence per sample.
    const double TWO_PI = 6.28318530717958647692, INV_TWO_PI = 0.15915494309189533577;
    R d = yaw - cw;
    d = d - TWO_PI * rint(d * INV_TWO_PI);
    c.dyaw = d * g.yaw_inv;
    // boundIndex :398-409: clamp x,y; wrap yaw modulo nyaw
    // (a trajectory's local frame counts cells from its own corner: ix_off / iy_off lead back to the grid's index; zero in the map's frame)
    const int gx = ix + g.ix_off, gy = iy + g.iy_off;
    const int x0 = clampIdx(gx, g.nx - 1), x1 = clampIdx(gx + 1, g.nx - 1);
    const int y0 = clampIdx(gy, g.ny - 1), y1 = clampIdx(gy + 1, g.ny - 1);
    // yaw passed isInMap and wm lies in [-pi, pi], so iw lies in [-1, nyaw]: ONE conditional wrap each way is boundIndex's modulo for
    // iw and iw + 1 alike
    int w0 = iw, w1 = iw + 1;
    w0 = w0 > g.nyaw - 1 ? w0 - g.nyaw : (w0 < 0 ? w0 + g.nyaw : w0);
    w1 = w1 > g.nyaw - 1 ? w1 - g.nyaw : (w1 < 0 ? w1 + g.nyaw : w1);
    c.w0 = w0; c.w1 = w1;
    // rows relative to the held part of the grid (a tile holds the x-rows [x_off, x_off + nx_hold) of the global grid: same index
    // arithmetic as the whole grid, so a lookup inside the tile is bit-identical).  Rows outside are clamped to the tile -- memory
    // safe; the host keeps trajectories inside their tile (uph_batch_upload / download checks).
    const int xa = clampIdx(x0 - g.x_off, g.nx_hold - 1), xb = clampIdx(x1 - g.x_off, g.nx_hold - 1);
    // 32-bit cell indices (the held cells number < 2^31): the 64-bit arithmetic happens once per load, as the byte offset
    const uint32_t ra = (uint32_t)xa * (uint32_t)g.ny, rb = (uint32_t)xb * (uint32_t)g.ny, nw = (uint32_t)g.nyaw;
    c.a[0][0] = (ra + (uint32_t)y0) * nw;
    c.a[0][1] = (ra + (uint32_t)y1) * nw;
    c.a[1][0] = (rb + (uint32_t)y0) * nw;
    c.a[1][1] = (rb + (uint32_t)y1) * nw;
}

// One cell of the grid: {z, sigma, zb.x, zb.y} in the reference's RXS2 order (uneven_map.h:36-64, 427-435), stored either as four
// doubles (32 bytes, two 16-byte loads; bit-faithful to the reference's map_buffer) or -- GridDev::cells32, BASELINE.json configs[4] --
// as four floats (16 bytes, one load) widened to double on load: the arithmetic is fp64 either way.
// WITH_Z = false requests sigma and zb only -- an 8-byte and a 16-byte load instead of two 16-byte loads.  Not for the eight bytes: a register pair that a
// load fills and nobody reads is free for reuse as far as the compiler is concerned, and its next writer has to wait for the load to land first
// (s_waitcnt before the overwrite).  In the sample code that was the address temporary of the NEXT cell: the first cell's round trip to the grid ran to
// completion before the other fifteen loads of the gather were even issued.
template <bool F32, class R, bool WITH_Z = true>
UPH_HD void loadCell(const GridDev& g, uint32_t idx, R f[4]) {           // f = {sigma, zb.x, zb.y, z}
#if defined(__HIP_DEVICE_COMPILE__)
    typedef double dbl2_t __attribute__((ext_vector_type(2)));
    typedef float flt4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) dbl2_t* cellp;
    typedef const __attribute__((address_space(1))) flt4_t* cellp32;
    typedef const __attribute__((address_space(1))) double* wordp;
#else
    struct dbl2_t { double x, y; };
    struct flt4_t { float x, y, z, w; };
    typedef const dbl2_t* cellp;
    typedef const flt4_t* cellp32;
    typedef const double* wordp;
#endif
    if (F32) {
        const flt4_t v = ((cellp32)g.cells32)[idx];
        f[3] = toReal<R>(v.x); f[0] = toReal<R>(v.y); f[1] = toReal<R>(v.z); f[2] = toReal<R>(v.w);
    } else if (WITH_Z) {
        const cellp p = (cellp)(g.cells + 4 * (size_t)idx);
        const dbl2_t lo = p[0], hi = p[1];                  // (z, sigma), (zb.x, zb.y)
        f[3] = lo.x; f[0] = lo.y; f[1] = hi.x; f[2] = hi.y;
    } else {
        const double* p = g.cells + 4 * (size_t)idx;
        const double sg = ((wordp)p)[1];
        const dbl2_t hi = ((cellp)p)[1];
        f[3] = R(0.0); f[0] = sg; f[1] = hi.x; f[2] = hi.y;
    }
}

// Trilinear values val[k] (k = 0 sigma, 1 zb.x, 2 zb.y, 3 z when WITH_Z) and, when GRAD, the gradients grd[k][x, y, yaw] of the first
// three at a located point: the operations and their order of uneven_map.h:297-311 (values alone: :192-198, the same expressions).
// One yaw slice at a time -- the bilinear values and the x / y partial sums of a slice need only that slice's four cells, the two
// slices meet in the last lerp -- so at most four cells are live at once (eight corners = 16 x 16-byte loads in the fp64 form, 8 in fp32).
template <bool GRAD, bool WITH_Z, class R, class CELLS>
UPH_HD void interpCellsWith(const GridDev& g, R dx, R dy, R dw, CELLS cells, R val[4], R grd[3][3]) {      // cells(w, f): the four cells f[x][y][field] of yaw slice w
    constexpr int NF = WITH_Z ? 4 : 3;
    R v0[NF], v1[NF], gy0[3], gy1[3], gx[3];
#pragma unroll
    for (int w = 0; w < 2; w++) {
        R f[2][2][4];                                  // (requesting all eight cells before the first use was measured: no gain under load, -0.7 %)
        cells(w, f);
#pragma unroll
        for (int k = 0; k < NF; k++) {
            const R vy0 = f[0][0][k] * (1 - dx) + f[1][0][k] * dx;       // v00 / v01 of uneven_map.h:297-300
            const R vy1 = f[0][1][k] * (1 - dx) + f[1][1][k] * dx;       // v10 / v11
            const R vv = vy0 * (1 - dy) + vy1 * dy;
            if (w == 0) v0[k] = vv; else v1[k] = vv;
            if (GRAD && k < 3) {
                const R gxa = f[1][0][k] - f[0][0][k], gxb = f[1][1][k] - f[0][1][k];
                if (w == 0) { gy0[k] = vy1 - vy0; gx[k] = (1 - dw) * (1 - dy) * gxa; gx[k] += (1 - dw) * dy * gxb; }       // :305-308, same summation order
                else { gy1[k] = vy1 - vy0; gx[k] += dw * (1 - dy) * gxa; gx[k] += dw * dy * gxb; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NF; k++) {
        val[k] = v0[k] * (1 - dw) + v1[k] * dw;
        if (GRAD && k < 3) {
            grd[k][2] = (v1[k] - v0[k]) * g.yaw_inv;
            grd[k][1] = (gy0[k] * (1 - dw) + gy1[k] * dw) * g.xy_inv;
            grd[k][0] = gx[k] * g.xy_inv;
        }
    }
}
template <bool F32, bool GRAD, bool WITH_Z, class R>
UPH_HD void interpCells(const GridDev& g, const CornersT<R>& c, R val[4], R grd[3][3]) {
    interpCellsWith<GRAD, WITH_Z, R>(g, c.dx, c.dy, c.dyaw, [&](int w, R f[2][2][4]) {
        const int wi = w == 0 ? c.w0 : c.w1;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) loadCell<F32, R, WITH_Z>(g, c.a[a][b] + (uint32_t)wi, f[a][b]);
    }, val, grd);
}

// Base quantities for the fused penalty kernel: interpolated (sigma, zb.x, zb.y) and their gradients w.r.t. (x, y, yaw).
template <class R>
UPH_HD void terrainBase(const GridDev& g, R x, R y, R yaw, R& sg, R& zx, R& zy, R gs[3], R gzx[3], R gzy[3]) {
    CornersT<R> c;
    locate<R>(g, x, y, yaw, c);
    sg = R(0.0); zx = R(0.0); zy = R(0.0);
#pragma unroll
    for (int k = 0; k < 3; k++) { gs[k] = R(0.0); gzx[k] = R(0.0); gzy[k] = R(0.0); }
    if (!c.inmap) return;                                   // out of map: zeros (uneven_map.h:260-265)
    R val[4], grd[3][3];
    if (g.cells32) interpCells<true, true, false, R>(g, c, val, grd);
    else interpCells<false, true, false, R>(g, c, val, grd);
    sg = val[0]; zx = val[1]; zy = val[2];
#pragma unroll
    for (int k = 0; k < 3; k++) { gs[k] = grd[0][k]; gzx[k] = grd[1][k]; gzy[k] = grd[2][k]; }
}

// value-only lookup (uneven_map.h:154-201): val = {sigma, zb.x, zb.y, z}, zeros outside the map
UPH_HD void terrainValues(const GridDev& g, const Corners& c, double val[4]) {
    val[0] = val[1] = val[2] = val[3] = 0.0;
    if (!c.inmap) return;
    double grd[3][3];
    if (g.cells32) interpCells<true, false, true, double>(g, c, val, grd);
    else interpCells<false, false, true, double>(g, c, val, grd);
}

// values / gradient rows: 0 invCosVphix, 1 sinPhix, 2 invCosVphiy, 3 sinPhiy, 4 cosXi, 5 invCosXi, 6 sigma
// (cyaw, syaw) = cos/sin of the WRAPPED yaw (uneven_map.h:329-330)
UPH_HD void terrainAllWithGrad(const GridDev& g, double x, double y, double yaw, double cyaw, double syaw,
                               double values[7], double grads[7][3]) {
    double sg, zx, zy, gs[3], gzx[3], gzy[3];
    terrainBase(g, x, y, yaw, sg, zx, zy, gs, gzx, gzy);
    double cc = sqrt(1.0 - zx * zx - zy * zy);                         // RXS2::getC :46
    double gc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) gc[k] = -(gzx[k] * zx + gzy[k] * zy) / cc;   // :312
    double inv_c = 1.0 / cc;
    double t = cyaw * zx + syaw * zy;                                  // :333-337
    double s = -(-syaw * zx + cyaw * zy);
    double sqrt_1_t2 = sqrt(1.0 - t * t);
    double inv_sqrt_1_t2 = 1.0 / sqrt_1_t2;
    double inv_sqrt_1_t2_3 = inv_sqrt_1_t2 * inv_sqrt_1_t2 * inv_sqrt_1_t2;
    double dt[3], ds[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {                                      // :338-339
        dt[k] = gzx[k] * cyaw + gzy[k] * syaw;
        ds[k] = -(gzx[k] * (-syaw) + gzy[k] * cyaw);
    }
    dt[2] -= s;                                                        // :340-341
    ds[2] += t;
    values[0] = inv_sqrt_1_t2;                                         // :343-348
    values[1] = -cc * t * inv_sqrt_1_t2;
    values[2] = sqrt_1_t2 * inv_c;
    values[3] = s * inv_sqrt_1_t2;
    values[4] = cc;
    values[5] = inv_c;
    values[6] = sg;
#pragma unroll
    for (int k = 0; k < 3; k++) {                                      // :350-355, :374
        grads[0][k] = t * inv_sqrt_1_t2_3 * dt[k];
        grads[1][k] = -(t * inv_sqrt_1_t2 * gc[k] + inv_sqrt_1_t2_3 * cc * dt[k]);
        grads[2][k] = -inv_c * (t * inv_sqrt_1_t2 * dt[k] + sqrt_1_t2 * inv_c * gc[k]);
        grads[3][k] = inv_sqrt_1_t2 * ds[k] + t * inv_sqrt_1_t2_3 * s * dt[k];
        grads[4][k] = gc[k];
        grads[5][k] = -inv_c * inv_c * gc[k];
        grads[6][k] = gs[k];
    }
}

// value-only variant: getTerrain + getTerrainVariables (uneven_map.h:154-201, 221-256).  zout = interpolated z
UPH_HD void terrainVariables(const GridDev& g, double x, double y, double yaw, double cyaw, double syaw, double values[7], double* zout) {
    Corners c;
    locate(g, x, y, yaw, c);
    double val[4];
    terrainValues(g, c, val);
    const double sg = val[0], zx = val[1], zy = val[2], zz = val[3];
    double cc = sqrt(1.0 - zx * zx - zy * zy);
    double inv_c = 1.0 / cc;
    double t = cyaw * zx + syaw * zy;
    double s = -(-syaw * zx + cyaw * zy);
    double sqrt_1_t2 = sqrt(1.0 - t * t);
    double inv_sqrt_1_t2 = 1.0 / sqrt_1_t2;
    values[0] = inv_sqrt_1_t2;
    values[1] = -cc * t * inv_sqrt_1_t2;
    values[2] = sqrt_1_t2 * inv_c;
    values[3] = s * inv_sqrt_1_t2;
    values[4] = cc;
    values[5] = inv_c;
    values[6] = sg;
    if (zout) *zout = zz;
}

}  // namespace uph
