// Device terrain lookup: trilinear SE(2) interpolation of (sigma, zb) with analytic gradients and the derived
// attitude terms.  Twin of UnevenMap::getTerrainWithGradI / getAllWithGrad / getTerrain / getTerrainVariables
// (uneven_map/include/uneven_map/uneven_map.h:154-201, 221-256, 258-315, 318-377), index helpers :398-454.
// Reads the SoA planes of GridDev: for each (x,y) corner the two yaw neighbours are adjacent in memory (yaw fastest),
// so a corner pair is one 16-byte access unless the yaw index wraps (64-bin period, uneven_map.h:403-406).
#pragma once
#include "uph_common.hpp"

namespace uph {

struct Corners {
    double dx, dy, dyaw;       // fractional offsets diff[0..2]
    int64_t a[2][2];           // address of (x,y) corner at yaw index w0
    int w0, w1;                // the two yaw bins
    bool inmap;
};

UPH_HD bool isInMap(const GridDev& g, double x, double y, double yaw) {     // uneven_map.h:437-454
    if (x < g.lo[0] || y < g.lo[1] || yaw < g.lo[2]) return false;      // lo = minb + 1e-4, hi = maxb - 1e-4
    if (x > g.hi[0] || y > g.hi[1] || yaw > g.hi[2]) return false;
    return true;
}

UPH_HD void locate(const GridDev& g, double x, double y, double yaw, Corners& c) {
    c.inmap = isInMap(g, x, y, yaw);
    if (!c.inmap) return;
    // uneven_map.h:268-284
    double xm = x - g.half_xy, ym = y - g.half_xy;
    double wm = normSO2(yaw - g.half_yaw);
    int ix = (int)floor((xm - g.origin[0]) * g.xy_inv);
    int iy = (int)floor((ym - g.origin[1]) * g.xy_inv);
    int iw = (int)floor((wm - g.origin[2]) * g.yaw_inv);
    double cx = (ix + 0.5) * g.xy_res + g.origin[0];
    double cy = (iy + 0.5) * g.xy_res + g.origin[1];
    double cw = (iw + 0.5) * g.yaw_res + g.origin[2];
    c.dx = (x - cx) * g.xy_inv;
    c.dy = (y - cy) * g.xy_inv;
    // reference: atan2(sin(d), cos(d)) * yaw_inv  (:284).  d lies within one wrap of a yaw cell, so the exact range
    // reduction d - 2*pi*rint(d/(2*pi)) gives the same angle to ~1e-17 without three transcendentals.
    const double TWO_PI = 6.28318530717958647692;
    double d = yaw - cw;
    d = d - TWO_PI * rint(d / TWO_PI);
    c.dyaw = d * g.yaw_inv;
    // boundIndex :398-409: clamp x,y; wrap yaw modulo nyaw
    int x0 = ix < 0 ? 0 : (ix > g.nx - 1 ? g.nx - 1 : ix);
    int x1 = ix + 1 < 0 ? 0 : (ix + 1 > g.nx - 1 ? g.nx - 1 : ix + 1);
    int y0 = iy < 0 ? 0 : (iy > g.ny - 1 ? g.ny - 1 : iy);
    int y1 = iy + 1 < 0 ? 0 : (iy + 1 > g.ny - 1 ? g.ny - 1 : iy + 1);
    int w0 = iw, w1 = iw + 1;
    // yaw passed isInMap, so iw lies in [-1, nyaw]: two conditional wraps each way cover boundIndex's modulo with margin
#pragma unroll
    for (int it = 0; it < 2; it++) {
        w0 = w0 > g.nyaw - 1 ? w0 - g.nyaw : (w0 < 0 ? w0 + g.nyaw : w0);
        w1 = w1 > g.nyaw - 1 ? w1 - g.nyaw : (w1 < 0 ? w1 + g.nyaw : w1);
    }
    c.w0 = w0; c.w1 = w1;
    c.a[0][0] = ((int64_t)x0 * g.ny + y0) * g.nyaw;
    c.a[0][1] = ((int64_t)x0 * g.ny + y1) * g.nyaw;
    c.a[1][0] = ((int64_t)x1 * g.ny + y0) * g.nyaw;
    c.a[1][1] = ((int64_t)x1 * g.ny + y1) * g.nyaw;
}

// trilinear value and gradient of one field plane, operation order of uneven_map.h:297-311
UPH_HD void interpField(const double* __restrict__ f, const Corners& c, double& val, double& gx, double& gy, double& gw,
                        double xy_inv, double yaw_inv) {
    double v[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            v[a][b][0] = f[c.a[a][b] + c.w0];
            v[a][b][1] = f[c.a[a][b] + c.w1];
        }
    const double dx = c.dx, dy = c.dy, dw = c.dyaw;
    double v00 = v[0][0][0] * (1 - dx) + v[1][0][0] * dx;
    double v01 = v[0][0][1] * (1 - dx) + v[1][0][1] * dx;
    double v10 = v[0][1][0] * (1 - dx) + v[1][1][0] * dx;
    double v11 = v[0][1][1] * (1 - dx) + v[1][1][1] * dx;
    double v0 = v00 * (1 - dy) + v10 * dy;
    double v1 = v01 * (1 - dy) + v11 * dy;
    val = v0 * (1 - dw) + v1 * dw;
    gw = (v1 - v0) * yaw_inv;
    gy = ((v10 - v00) * (1 - dw) + (v11 - v01) * dw) * xy_inv;
    double g0 = (1 - dw) * (1 - dy) * (v[1][0][0] - v[0][0][0]);
    g0 += (1 - dw) * dy * (v[1][1][0] - v[0][1][0]);
    g0 += dw * (1 - dy) * (v[1][0][1] - v[0][0][1]);
    g0 += dw * dy * (v[1][1][1] - v[0][1][1]);
    gx = g0 * xy_inv;
}

UPH_HD double interpValue(const double* __restrict__ f, const Corners& c) {       // uneven_map.h:192-198
    double v00 = f[c.a[0][0] + c.w0] * (1 - c.dx) + f[c.a[1][0] + c.w0] * c.dx;
    double v01 = f[c.a[0][0] + c.w1] * (1 - c.dx) + f[c.a[1][0] + c.w1] * c.dx;
    double v10 = f[c.a[0][1] + c.w0] * (1 - c.dx) + f[c.a[1][1] + c.w0] * c.dx;
    double v11 = f[c.a[0][1] + c.w1] * (1 - c.dx) + f[c.a[1][1] + c.w1] * c.dx;
    double v0 = v00 * (1 - c.dy) + v10 * c.dy;
    double v1 = v01 * (1 - c.dy) + v11 * c.dy;
    return v0 * (1 - c.dyaw) + v1 * c.dyaw;
}

// values / gradient rows: 0 invCosVphix, 1 sinPhix, 2 invCosVphiy, 3 sinPhiy, 4 cosXi, 5 invCosXi, 6 sigma
// (cyaw, syaw) = cos/sin of the WRAPPED yaw (uneven_map.h:329-330)
UPH_HD void terrainAllWithGrad(const GridDev& g, double x, double y, double yaw, double cyaw, double syaw,
                               double values[7], double grads[7][3]) {
    Corners c;
    locate(g, x, y, yaw, c);
    double sg = 0, zx = 0, zy = 0;
    double gs[3] = {0, 0, 0}, gzx[3] = {0, 0, 0}, gzy[3] = {0, 0, 0};
    if (c.inmap) {
        interpField(g.sigma, c, sg, gs[0], gs[1], gs[2], g.xy_inv, g.yaw_inv);
        interpField(g.zbx, c, zx, gzx[0], gzx[1], gzx[2], g.xy_inv, g.yaw_inv);
        interpField(g.zby, c, zy, gzy[0], gzy[1], gzy[2], g.xy_inv, g.yaw_inv);
    }
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

// Base quantities for the fused penalty kernel: interpolated (sigma, zb.x, zb.y) and their gradients w.r.t. (x, y, yaw).
UPH_HD void terrainBase(const GridDev& g, double x, double y, double yaw, double& sg, double& zx, double& zy, double gs[3], double gzx[3], double gzy[3]) {
    Corners c;
    locate(g, x, y, yaw, c);
    sg = zx = zy = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) { gs[k] = 0.0; gzx[k] = 0.0; gzy[k] = 0.0; }
    if (!c.inmap) return;                                   // out of map: zeros (uneven_map.h:260-265)
    // one field plane at a time (8 gathers each): issuing all 24 at once was measured slower -- the 48 extra live registers
    // spill in the register-capped build
    const double xi = g.xy_inv, wi = g.yaw_inv;
    double val[3], grd[3][3];
    interpField(g.sigma, c, val[0], grd[0][0], grd[0][1], grd[0][2], xi, wi);
    interpField(g.zbx, c, val[1], grd[1][0], grd[1][1], grd[1][2], xi, wi);
    interpField(g.zby, c, val[2], grd[2][0], grd[2][1], grd[2][2], xi, wi);
    sg = val[0]; zx = val[1]; zy = val[2];
#pragma unroll
    for (int k = 0; k < 3; k++) { gs[k] = grd[0][k]; gzx[k] = grd[1][k]; gzy[k] = grd[2][k]; }
}

// The same base quantities from the array-of-cells form of the grid (GridDev::cells: {z, sigma, zb.x, zb.y} per cell, the reference's
// own RXS2 order, uneven_map.h:36-64, 427-435): a cell is 32 bytes, so one corner is two 16-byte loads and the eight corners of a
// sample are 16 loads on 8 addresses (24 8-byte loads on 24 addresses in the three-plane form).  One yaw slice at a time: the
// bilinear values and the x / y partial sums of a slice need only that slice's four cells; the two slices meet in the last lerp.
// Same operations and order as interpField (uneven_map.h:297-311).
UPH_HD void terrainBaseCells(const GridDev& g, double x, double y, double yaw, double& sg, double& zx, double& zy, double gs[3], double gzx[3], double gzy[3]) {
    Corners c;
    locate(g, x, y, yaw, c);
    sg = zx = zy = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) { gs[k] = 0.0; gzx[k] = 0.0; gzy[k] = 0.0; }
    if (!c.inmap) return;                                   // out of map: zeros (uneven_map.h:260-265)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef double dbl2_t __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) dbl2_t* cellp;
#else
    struct dbl2_t { double x, y; };
    typedef const dbl2_t* cellp;
#endif
    const double dx = c.dx, dy = c.dy, dw = c.dyaw;
    double v0[3], v1[3], gy0[3], gy1[3], gx0[3], gx1[3];      // per yaw slice: bilinear value, (v1x - v0x) blends for d/dy and d/dx
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const int wi = w == 0 ? c.w0 : c.w1;
        double f[2][2][3];                                  // [a][b][sigma, zb.x, zb.y]
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const cellp p = (cellp)(g.cells + 4 * (c.a[a][b] + wi));
                const dbl2_t lo = p[0], hi = p[1];          // (z, sigma), (zb.x, zb.y)
                f[a][b][0] = lo.y; f[a][b][1] = hi.x; f[a][b][2] = hi.y;
            }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double vy0 = f[0][0][k] * (1 - dx) + f[1][0][k] * dx;       // v00 / v01 of uneven_map.h:297-300
            const double vy1 = f[0][1][k] * (1 - dx) + f[1][1][k] * dx;       // v10 / v11
            const double vv = vy0 * (1 - dy) + vy1 * dy;
            const double gyv = vy1 - vy0;
            const double gxa = f[1][0][k] - f[0][0][k], gxb = f[1][1][k] - f[0][1][k];
            if (w == 0) { v0[k] = vv; gy0[k] = gyv; gx0[k] = (1 - dw) * (1 - dy) * gxa; gx0[k] += (1 - dw) * dy * gxb; }
            else { v1[k] = vv; gy1[k] = gyv; gx1[k] = dw * (1 - dy) * gxa; gx1[k] += dw * dy * gxb; }
        }
    }
    const double xi = g.xy_inv, wi_ = g.yaw_inv;
    double val[3], grd[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        val[k] = v0[k] * (1 - dw) + v1[k] * dw;
        grd[k][2] = (v1[k] - v0[k]) * wi_;
        grd[k][1] = (gy0[k] * (1 - dw) + gy1[k] * dw) * xi;
        grd[k][0] = (gx0[k] + gx1[k]) * xi;
    }
    sg = val[0]; zx = val[1]; zy = val[2];
#pragma unroll
    for (int k = 0; k < 3; k++) { gs[k] = grd[0][k]; gzx[k] = grd[1][k]; gzy[k] = grd[2][k]; }
}

// value-only variant: getTerrain + getTerrainVariables (uneven_map.h:154-201, 221-256).  zout = interpolated z
UPH_HD void terrainVariables(const GridDev& g, double x, double y, double yaw, double cyaw, double syaw, double values[7], double* zout) {
    Corners c;
    locate(g, x, y, yaw, c);
    double sg = 0, zx = 0, zy = 0, zz = 0;
    if (c.inmap) {
        sg = interpValue(g.sigma, c);
        zx = interpValue(g.zbx, c);
        zy = interpValue(g.zby, c);
        if (zout) zz = interpValue(g.z, c);
    }
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
