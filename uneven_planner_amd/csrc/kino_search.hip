// libunevenhip.so -- front-end half: the kinodynamic best-first search of KinoAstar::plan for a BATCH of start / goal queries.
//
// Reference (paths under /root/reference/src/uneven_planner/front_end):
//   uph_kino_plan_batch <- KinoAstar::plan  src/kino_astar.cpp:67-236 (one call per query), with
//                          stateTransit / stateToIndex / normalizeAngle / getHeu   include/front_end/kino_astar.h:180-240
//                          asignShotTraj (ompl::base::DubinsStateSpace::distance / interpolate, OMPL 1.4.2)   kino_astar.h:242-271
//                          retrievePath   kino_astar.h:273-292
//                          the map queries isInMap / isOccupancy / isOccupancyXY / getTerrainSig   uneven_map.h:389-396, 437-500
//   uph_kino_create     <- KinoAstar::init (parameters, kino_astar.cpp:7-33) + setEnvironment (node pool = getXYNum nodes, kino_astar.h:170-178)
//
// One search is a strictly sequential best-first loop, so the batch axis is the parallel one: ONE wave64 per query, many waves per CU, every
// wave with its own workspace in HBM (node pool, open heap, lattice table).  Inside a query the wave parallelises what one expansion offers:
// the <= 64 motion primitives of a node (successor state, map test, collision samples, trilinear sigma lookup, cost) go one per lane; a heap
// insertion fetches all ancestors of the new leaf at once and resolves the sift-up with one ballot; the Dubins shot checks its samples 64 at a
// time.  What stays serial is what the reference's result depends on: the order in which the primitives of one expansion meet the lattice
// table, and the sift-down of a pop.
//
// The open set is the reference's std::priority_queue of node pointers compared through their CURRENT f_score (kino_astar.h:49-57, 107);
// the reference lowers an open node's f_score in place without re-heapifying (kino_astar.cpp:218-229), so the array can violate the heap
// property and the pop order is whatever libstdc++'s __push_heap / __adjust_heap (bits/stl_heap.h) make of it.  Those two routines are
// restated here operation for operation -- same comparisons on the same keys in the same order -- so that the expansion sequence, ties
// included, is the reference's.  Each heap entry carries a copy of its node's f_score, kept current through a node -> heap position table
// when the node is relaxed: the comparisons see exactly the keys the reference's pointer dereferences would.
//
// Platform behaviour of the reference made explicit (oracle/kino_astar.hpp header): a v = 0 primitive with steer != 0 produces the state
// (NaN, NaN, yaw); isInMap(NaN) is true, floor(NaN) converts to INT_MIN (x86-64 cvttsd2si), getTerrainSig is NaN, and a node with key
// (INT_MIN, INT_MIN, yaw index) and g = f = NaN enters the open set.  kFloorToInt below is that conversion; NaN keys get their own table rows.
//
// Arithmetic: fp64, no contraction (the reference is built without FMA: back_end / front_end CMakeLists have no -march), operation order of the
// cited lines.  tan(steer) of the primitives and their collision sample times are formed on the host by the reference's own loops.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uneven_hip.h"
#include "terrain_dev.hpp"
#include "uph_internal.hpp"

using namespace uph;

#pragma clang fp contract(off)

#define KHIPCHK(call)                                                                              \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            setError(std::string(#call) + ": " + hipGetErrorString(_e));                           \
            return UPH_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

namespace {

constexpr int K_MAX_INPUTS = 64;        // motion primitives per expansion (one per lane); the reference's loops give 3 x 5 = 15
constexpr int K_MAX_TSAMP = 8;          // collision samples per primitive (arc / collision_interval; 2 with run_hill.yaml)
constexpr int K_CLOSE = 'a', K_OPEN = 'b';

struct KinoDev {
    double yaw_inv, lambda_heu, w_r2, w_so2, w_vch, w_dch, w_sigma, time_interval, coll_interval, oneshot_range, wheel_base, rho, tie_breaker;
    int n_inputs, nyawk, allocate_num, nxy;
    int spread;                                    // 1: <= 16 primitives with <= 3 collision samples each -- end state and samples of a primitive go to lanes p, 16 + p, 32 + p, 48 + p
    double in_v[K_MAX_INPUTS], in_steer[K_MAX_INPUTS];
    int in_nt[K_MAX_INPUTS];
    // stateTransit's per-(primitive, duration) constants, formed on the host by the reference's own operations (kino_astar.h:221-223):
    // s = v T, y = s tan(delta) / wheel_base, r = s / y;  index 0 = the end state (T = time_interval), k + 1 = collision sample k
    double in_s[K_MAX_INPUTS][K_MAX_TSAMP + 1], in_y[K_MAX_INPUTS][K_MAX_TSAMP + 1], in_r[K_MAX_INPUTS][K_MAX_TSAMP + 1];
};

struct __attribute__((aligned(64))) KNode {       // PathNode, kino_astar.h:34-46 (one 64-byte line)
    double sx, sy, syaw;
    double g, f;
    double in_v, in_steer;
    int parent;                                    // pool index, -1 = NULL
    int flag;                                      // K_CLOSE / K_OPEN
};
struct __attribute__((aligned(16))) KHeap { double f; int id; int pad; };

struct KinoWork {                                  // per-slot workspaces, slot s at base + s * stride
    KNode* nodes;                                  // [allocate_num]
    KHeap* heap;                                   // [allocate_num + 1]
    int* pos;                                      // [allocate_num]  node -> heap position
    int* key;                                      // [allocate_num]  node -> table key (for the expansion log)
    int* table;                                    // [nxy * nyawk + nyawk]  key -> node, -1 empty; the last nyawk rows are the NaN-state keys
    size_t table_len;
};

struct KinoIO {
    const double* starts; const double* goals;     // [B][3]
    double* paths; int path_cap;                   // [B][path_cap][3]
    int* n_path; int* status; int* iter_num; int* use_node_num;      // [B]
    int* expanded; int exp_cap;                    // [B][exp_cap][3] or null
    int max_expand;
    const int* order; int* next;                   // launch order of the queries [B], shared cursor
};

__device__ __forceinline__ int kuni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double kuni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double klane(double v, int l) {      // l wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ int klane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
template <int R> __device__ __forceinline__ int kRowRor(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + R, 0xf, 0xf, false); }      // row_ror:R inside each row of 16 lanes

__device__ __forceinline__ int kFloorToInt(double v) {          // (int)floor(v) as x86-64 converts it
    const double f = floor(v);
    if (!(f >= -2147483648.0 && f <= 2147483647.0)) return INT_MIN;
    return (int)f;
}
__device__ __forceinline__ double kNormalizeAngle(double angle) {      // kino_astar.h:193-204
    double a = angle;                          // (bounded like normSO2 in uph_common.hpp: an absurd angle cannot hang a wave; the search's angles stay within a few turns)
    for (int it = 0; it < 4096 && a > 3.14159265358979323846; it++) a -= 6.283185307179586;
    for (int it = 0; it < 4096 && a < -3.14159265358979323846; it++) a += 6.283185307179586;
    return a;
}
// kino_astar.h:218-240 with s, y = s tan(delta) / L and r = s / y handed in (constants of the primitive and the duration, formed on the host by the
// same operations) and sin / cos of the node's own heading (sw0, cw0: the same for every
// primitive and collision sample of an expansion) computed once; sincosFast: fdlibm kernels, < 0.8 ulp (uph_common.hpp)
template <bool FAST>
__device__ __forceinline__ void kSinCos(double x, double& sn, double& cs) {
    if (FAST) sincosFast(x, sn, cs);
    else { sn = sin(x); cs = cos(x); }
}
template <bool FAST>
__device__ __forceinline__ void kStateTransit(double x0, double y0, double w0, double sw0, double cw0, double delta, double s, double y, double r, double& x1, double& y1, double& w1) {
    if (fabs(delta) > 1e-4) {
        double sw1, cw1;
        kSinCos<FAST>(w0 + y, sw1, cw1);
        x1 = x0 + r * (sw1 - sw0);
        y1 = y0 - r * (cw1 - cw0);
        w1 = kNormalizeAngle(w0 + y);
    } else {
        x1 = x0 + s * cw0;
        y1 = y0 + s * sw0;
        w1 = w0;
    }
}
// UnevenMap::isOccupancyXY (uneven_map.h:490-500): 1 occupied, 0 free, -1 outside
__device__ __forceinline__ int kOccXY(const GridDev& g, const char* __restrict__ occ2, double x, double y, double w) {
    const int ix = kFloorToInt((x - g.origin[0]) * g.xy_inv), iy = kFloorToInt((y - g.origin[1]) * g.xy_inv), iw = kFloorToInt((w - g.origin[2]) * g.yaw_inv);
    if (ix < 0 || iy < 0 || iw < 0 || ix > g.nx - 1 || iy > g.ny - 1 || iw > g.nyaw - 1) return -1;
    return (int)occ2[(size_t)ix * g.ny + iy];
}
__device__ __forceinline__ int kOcc(const GridDev& g, const char* __restrict__ occ, double x, double y, double w) {      // isOccupancy(pos), :473-488
    const int ix = kFloorToInt((x - g.origin[0]) * g.xy_inv), iy = kFloorToInt((y - g.origin[1]) * g.xy_inv), iw = kFloorToInt((w - g.origin[2]) * g.yaw_inv);
    if (ix < 0 || iy < 0 || iw < 0 || ix > g.nx - 1 || iy > g.ny - 1 || iw > g.nyaw - 1) return -1;
    return (int)occ[((size_t)ix * g.ny + iy) * g.nyaw + iw];
}
__device__ __forceinline__ double kTerrainSig(const GridDev& g, double x, double y, double w) {      // getTerrainSig, uneven_map.h:389-396
    if (isnan(x) || isnan(y)) return __longlong_as_double(0x7ff8000000000000ll);      // isInMap(NaN) is true, the trilinear weights are NaN
    Corners c;
    locate(g, x, y, w, c);
    double tv[4];
    terrainValues(g, c, tv);
    return tv[0];
}
// lattice key of a state (stateToIndex, kino_astar.h:187-191): -1 = no table row (cannot occur for a state that passed isInMap)
__device__ __forceinline__ int kKey(const GridDev& g, const KinoDev& P, double x, double y, double w, int idx3[3]) {
    const int ix = kFloorToInt((x - g.origin[0]) * g.xy_inv), iy = kFloorToInt((y - g.origin[1]) * g.xy_inv);
    const int iw = kFloorToInt((kNormalizeAngle(w) + 3.14159265358979323846) * P.yaw_inv);
    idx3[0] = ix; idx3[1] = iy; idx3[2] = iw;
    if (iw < 0 || iw >= P.nyawk) return -1;
    if (ix == INT_MIN && iy == INT_MIN) return P.nxy * P.nyawk + iw;
    if (ix < 0 || iy < 0 || ix >= g.nx || iy >= g.ny) return -1;
    return (ix * g.ny + iy) * P.nyawk + iw;
}

// ---- Dubins (OMPL 1.4.2 DubinsStateSpace.cpp; formulas as restated in oracle/kino_astar.hpp, which cites them)
__device__ __forceinline__ double kMod2pi(double x) {
    const double twopi = 2.0 * 3.14159265358979323846;
    if (x < 0 && x > -1e-7) return 0;
    double xm = x - twopi * floor(x / twopi);
    if (twopi - xm < .5 * 1e-6) xm = 0.;
    return xm;
}
struct KDubins { int type; double len[3]; };
__device__ void kDubinsShortest(double d, double alpha, double beta, KDubins& best) {
    const double twopi = 2.0 * 3.14159265358979323846, ZERO = -1e-7, DMAX = 1.7976931348623157e308;
    if (d < 1e-6 && fabs(alpha - beta) < 1e-6) { best.type = 0; best.len[0] = 0; best.len[1] = d; best.len[2] = 0; return; }
    const double ca = cos(alpha), sa = sin(alpha), cb = cos(beta), sb = sin(beta);
    double cand[6][3];
#pragma unroll
    for (int k = 0; k < 6; k++) { cand[k][0] = 0.0; cand[k][1] = DMAX; cand[k][2] = 0.0; }
    {   // LSL
        const double tmp = 2. + d * d - 2. * (ca * cb + sa * sb - d * (sa - sb));
        if (tmp >= ZERO) { const double theta = atan2(cb - ca, d + sa - sb); cand[0][0] = kMod2pi(-alpha + theta); cand[0][1] = sqrt(fmax(tmp, 0.)); cand[0][2] = kMod2pi(beta - theta); }
    }
    {   // RSR
        const double tmp = 2. + d * d - 2. * (ca * cb + sa * sb - d * (sb - sa));
        if (tmp >= ZERO) { const double theta = atan2(ca - cb, d - sa + sb); cand[1][0] = kMod2pi(alpha - theta); cand[1][1] = sqrt(fmax(tmp, 0.)); cand[1][2] = kMod2pi(-beta + theta); }
    }
    {   // RSL
        const double tmp = d * d - 2. + 2. * (ca * cb + sa * sb - d * (sa + sb));
        if (tmp >= ZERO) { const double p = sqrt(fmax(tmp, 0.)); const double theta = atan2(ca + cb, d - sa - sb) - atan2(2., p); cand[2][0] = kMod2pi(alpha - theta); cand[2][1] = p; cand[2][2] = kMod2pi(beta - theta); }
    }
    {   // LSR
        const double tmp = -2. + d * d + 2. * (ca * cb + sa * sb + d * (sa + sb));
        if (tmp >= ZERO) { const double p = sqrt(fmax(tmp, 0.)); const double theta = atan2(-ca - cb, d + sa + sb) - atan2(-2., p); cand[3][0] = kMod2pi(-alpha + theta); cand[3][1] = p; cand[3][2] = kMod2pi(-beta + theta); }
    }
    {   // RLR
        const double tmp = .125 * (6. - d * d + 2. * (ca * cb + sa * sb + d * (sa - sb)));
        if (fabs(tmp) < 1.) { const double p = twopi - acos(tmp); const double theta = atan2(ca - cb, d - sa + sb); const double t = kMod2pi(alpha - theta + .5 * p); cand[4][0] = t; cand[4][1] = p; cand[4][2] = kMod2pi(alpha - beta - t + p); }
    }
    {   // LRL
        const double tmp = .125 * (6. - d * d + 2. * (ca * cb + sa * sb - d * (sa - sb)));
        if (fabs(tmp) < 1.) { const double p = twopi - acos(tmp); const double theta = atan2(-ca + cb, d + sa - sb); const double t = kMod2pi(-alpha + theta + .5 * p); cand[5][0] = t; cand[5][1] = p; cand[5][2] = kMod2pi(beta - alpha - t + p); }
    }
    // DubinsStateSpace.cpp dubins(): LSL first, a later word wins only if strictly shorter
    int bt = 0;
    double minLength = cand[0][0] + cand[0][1] + cand[0][2];
#pragma unroll
    for (int k = 1; k < 6; k++) { const double len = cand[k][0] + cand[k][1] + cand[k][2]; if (len < minLength) { minLength = len; bt = k; } }
    best.type = bt;
#pragma unroll
    for (int k = 0; k < 6; k++) if (k == bt) { best.len[0] = cand[k][0]; best.len[1] = cand[k][1]; best.len[2] = cand[k][2]; }
}
__device__ void kDubinsBetween(const KinoDev& P, const double s1[3], const double s2[3], KDubins& path) {
    const double dx = s2[0] - s1[0], dy = s2[1] - s1[1], d = sqrt(dx * dx + dy * dy) / P.rho, th = atan2(dy, dx);
    kDubinsShortest(d, kMod2pi(s1[2] - th), kMod2pi(s2[2] - th), path);
}
// segment kinds of the six words: 0 left, 1 straight, 2 right
__device__ __forceinline__ int kSegKind(int type, int i) {
    const int tab[6][3] = {{0, 1, 0}, {2, 1, 2}, {2, 1, 0}, {0, 1, 2}, {2, 0, 2}, {0, 2, 0}};
    int r = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) if (a == type && b == i) r = tab[a][b];
    return r;
}
__device__ void kDubinsInterpolate(const KinoDev& P, const double from[3], const double to[3], const KDubins& path, double t, double out[3]) {
    if (t >= 1.) { out[0] = to[0]; out[1] = to[1]; out[2] = to[2]; return; }
    if (t <= 0.) { out[0] = from[0]; out[1] = from[1]; out[2] = from[2]; return; }
    double sx = 0., sy = 0., syaw = from[2];
    double seg = t * (path.len[0] + path.len[1] + path.len[2]);
    for (int i = 0; i < 3 && seg > 0; ++i) {
        const double v = fmin(seg, path.len[i]);
        const double phi = syaw;
        seg -= v;
        const int kind = kSegKind(path.type, i);
        if (kind == 0) { sx = sx + sin(phi + v) - sin(phi); sy = sy - cos(phi + v) + cos(phi); syaw = phi + v; }
        else if (kind == 2) { sx = sx - sin(phi - v) + sin(phi); sy = sy + cos(phi - v) - cos(phi); syaw = phi - v; }
        else { sx = sx + v * cos(phi); sy = sy + v * sin(phi); }
    }
    out[0] = sx * P.rho + from[0];
    out[1] = sy * P.rho + from[1];
    double w = fmod(syaw, 2.0 * 3.14159265358979323846);      // SO2StateSpace::enforceBounds
    if (w < -3.14159265358979323846) w += 2.0 * 3.14159265358979323846;
    else if (w >= 3.14159265358979323846) w -= 2.0 * 3.14159265358979323846;
    out[2] = w;
}

// ---- open heap: libstdc++ __push_heap / __adjust_heap with comp(a, b) = a.f > b.f (NodeComparator, kino_astar.h:49-57)
// The first TOPN positions (the top levels of the tree, which every pop walks and every push may reach) live in the wave's LDS, the rest in the
// slot's HBM workspace: a pop of a 10 000-entry heap then has ~4 dependent HBM round trips instead of ~13.
template <int TOPN>
struct KHeapRef {
    KHeap* top;      // LDS, positions [0, TOPN)
    KHeap* glob;     // HBM, positions [TOPN, ...)  (indexed by position)
    __device__ __forceinline__ KHeap get(int i) const { return i < TOPN ? top[i] : glob[i]; }
    __device__ __forceinline__ void set(int i, const KHeap& e) const { if (i < TOPN) top[i] = e; else glob[i] = e; }
    __device__ __forceinline__ void setF(int i, double f) const { if (i < TOPN) top[i].f = f; else glob[i].f = f; }
};
// push: the new leaf sits at position n; every ancestor is fetched at once (lane d holds the ancestor d levels up), the sift-up stops at the
// first ancestor that is NOT greater than the value (NaN compares false: stops), the ancestors below move down one level each.
template <int TOPN>
__device__ __forceinline__ void kHeapPush(const KHeapRef<TOPN>& heap, int* pos, int n, int id, double f, int lane) {
    const int hole = n;
    const int depth = 31 - __builtin_clz((unsigned)hole + 1u);      // number of ancestors = level of the leaf
    KHeap anc; anc.f = 0.0; anc.id = -1; anc.pad = 0;
    int apos = -1, cpos = -1;                        // position of this lane's ancestor and of the child on the path below it
    if (lane >= 1 && lane <= depth) {
        apos = ((hole + 1) >> lane) - 1;
        cpos = ((hole + 1) >> (lane - 1)) - 1;
        anc = heap.get(apos);
    }
    const unsigned long long stop = __ballot(lane >= 1 && lane <= depth && !(anc.f > f));
    const int D = stop ? (int)__builtin_ctzll(stop) : depth + 1;      // first level that does not move
    if (lane >= 1 && lane < D) { heap.set(cpos, anc); pos[anc.id] = cpos; }
    if (lane == 0) {
        const int fin = ((hole + 1) >> (D - 1)) - 1;
        KHeap e; e.f = f; e.id = id; e.pad = 0;
        heap.set(fin, e); pos[id] = fin;
    }
}
// pop of the root: value = last entry, __adjust_heap(first, 0, len = n - 1, value).  The sift-down always runs to a leaf (smaller-f child,
// the right one when neither compares greater), then the value sifts up along that path.  With E_k the entry the path met at level k
// (k = 1..L, positions p_k) the final layout is: heap[p_{k-1}] = E_k for k <= h, heap[p_h] = value, levels below h untouched, where h walks up
// from L while E_h.f > value.f.  Lane k keeps (E_k, p_k, p_{k-1}); one dependent load pair per level.
template <int TOPN>
__device__ __forceinline__ void kHeapPop(const KHeapRef<TOPN>& heap, int* pos, int n, int lane) {
    if (n <= 1) return;
    const int len = n - 1;
    KHeap value = heap.get(len);
    value.f = kuni(value.f); value.id = kuni(value.id);
    KHeap mine; mine.f = 0.0; mine.id = -1; mine.pad = 0;      // lane k: E_k
    int my_p = -1, my_pp = -1;                                   // p_k, p_{k-1}
    int hole = 0, second = 0, L = 0;
    while (second < (len - 1) / 2 && L < 60) {
        second = 2 * (second + 1);
        const KHeap r = heap.get(second), l = heap.get(second - 1);      // (uniform addresses: one transaction each)
        const double rf = kuni(r.f), lf = kuni(l.f);
        const bool takeLeft = rf > lf;                           // comp(first + second, first + (second - 1))
        if (takeLeft) second--;
        L++;
        if (lane == L) { mine.f = takeLeft ? lf : rf; mine.id = kuni(takeLeft ? l.id : r.id); my_p = second; my_pp = hole; }
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        const KHeap l = heap.get(second - 1);
        L++;
        if (lane == L) { mine.f = kuni(l.f); mine.id = kuni(l.id); my_p = second - 1; my_pp = hole; }
        hole = second - 1;
    }
    // sift-up of value from level L: h = L; while (h > 0 && E_h.f > value.f) h--
    const unsigned long long keep = __ballot(lane >= 1 && lane <= L && !(mine.f > value.f));      // levels that stop the walk
    int h = 0;
    if (keep) h = 63 - (int)__builtin_clzll(keep);               // highest such level <= L
    // entries at levels <= h move up one level; levels > h stay where they were (they were never written)
    if (lane >= 1 && lane <= h) { heap.set(my_pp, mine); pos[mine.id] = my_pp; }
    const int ph = h == 0 ? 0 : klane(my_p, h);
    if (lane == 0) { heap.set(ph, value); pos[value.id] = ph; }
}

// ------------------------------------------------------------------------------------------------ the search kernel: one wave64 per query
#ifndef UPH_KINO_WPS
#define UPH_KINO_WPS 4            // default waves per SIMD of the search kernel (register cap 512 / WPS): the search is latency-bound, residency is what scales it
#endif
// WPS = waves per SIMD the instantiation is compiled for (uph_kino_set_wps picks one: 2 = 256 registers, 4 = 128, 6 = 80, 8 = 64);
// FAST = sincosFast (fdlibm kernels, uph_common.hpp) instead of the device library's sin / cos
// -DUPH_KINO_PROF (tools/kino_phase_probe.py): shader-clock cycles per section of the expansion loop, left in the last two rows of the query's
// path output (a profiling build only: those rows are then not poses)
#ifdef UPH_KINO_PROF
#define KPROF(k) do { const long long t_ = (long long)__builtin_readcyclecounter(); kprof[k] += t_ - kprof_t; kprof_t = t_; } while (0)
#else
#define KPROF(k) do { } while (0)
#endif
template <int WPS, bool FAST>
__global__ __launch_bounds__(64, WPS) void uph_kino_kernel(GridDev g, const char* __restrict__ occ, const char* __restrict__ occ2, const KinoDev* __restrict__ Pp, KinoWork W, size_t node_stride,
                                                      size_t heap_stride, size_t table_stride, KinoIO io, int B) {
    const int lane = threadIdx.x;
    const KinoDev& P = *Pp;
    KNode* nodes = W.nodes + (size_t)blockIdx.x * node_stride;
    // gfx950 only (like the whole library: Makefile ARCH): the residency these sizes aim at -- 16 one-wave blocks per CU in the default WPS = 4 form, which is
    // what the automatic workspace count of uph_kino_create provides -- needs the 160 KB of LDS per CU of this part (16 x 8 KB); a part with 64 KB would hold half
    // as many.  The WPS = 6 / 8 forms are experiment knobs: they run with the same workspaces (at most 16 resident queries per CU), not with more.
    constexpr int TOPN = WPS <= 2 ? 1023 : (WPS <= 4 ? 511 : 255);      // 16 / 8 / 4 KB of LDS per wave: 8 / 16 / 24-32 waves per CU
    __shared__ KHeap heap_top[TOPN];
    const KHeapRef<TOPN> heap = {heap_top, W.heap + (size_t)blockIdx.x * heap_stride};
    int* pos = W.pos + (size_t)blockIdx.x * node_stride;
    int* nkey = W.key + (size_t)blockIdx.x * node_stride;
    int* table = W.table + (size_t)blockIdx.x * table_stride;
    // This lane's primitive and its role, the same for every expansion of every query: with <= 16 primitives of <= 3 collision samples
    // (run_hill.yaml: 15 of 2) the end state and the samples of primitive p -- independent evaluations of stateTransit -- go to lanes p, 16 + p,
    // 32 + p, 48 + p and are formed in ONE pass, a ballot brings the samples' verdicts to the primitive's lane.  (The reference's loop stops at
    // the first occupied sample and uses only "was one occupied": the same predicate.)  Otherwise the primitive's lane walks its samples itself.
    const int plane = P.spread ? (lane & 15) : lane;             // this lane's primitive
    const int sgrp = P.spread ? (lane >> 4) : 0;                 // 0: end state, k: collision sample k - 1
    const bool pvalid = plane < P.n_inputs;
    bool l_samp = false;
    double l_iv = 0.0, l_is = 0.0, l_s = 0.0, l_y = 0.0, l_r = 0.0;      // the primitive's (v, delta) and stateTransit's constants for this lane's duration
    if (pvalid) {
        l_samp = P.spread && sgrp > 0 && sgrp - 1 < P.in_nt[plane];
        const int ti = l_samp ? sgrp : 0;
        l_iv = P.in_v[plane]; l_is = P.in_steer[plane];
        l_s = P.in_s[plane][ti]; l_y = P.in_y[plane][ti]; l_r = P.in_r[plane][ti];
    }
    // queries are handed out dynamically, longest (by straight-line distance) first: io.order is that order, io.next the shared cursor
    // (io.next == nullptr: static assignment, query blockIdx.x + k gridDim.x of the order)
    for (int turn = 0; turn <= B; turn++) {                       // (a wave can take at most B queries: the bound is a guard)
        int qi = blockIdx.x + turn * (int)gridDim.x;
        if (io.next != nullptr) {
            int t = 0;
            if (lane == 0) t = atomicAdd(io.next, 1);
            qi = kuni(t);
        }
        if (qi >= B) break;
        const int q = io.order[qi];
        const double sx0 = io.starts[3 * q], sy0 = io.starts[3 * q + 1], sw0 = io.starts[3 * q + 2];
        const double gx = io.goals[3 * q], gy = io.goals[3 * q + 1], gw = io.goals[3 * q + 2];
        int status = -1, iter_num = 0, use_node_num = 0, n = 0, n_path = 0;
#ifdef UPH_KINO_PROF
        long long kprof[6] = {0, 0, 0, 0, 0, 0}, kprof_t = (long long)__builtin_readcyclecounter();
#endif
        if (kOcc(g, occ, sx0, sy0, sw0) == 1) status = 1;                                  // kino_astar.cpp:86-90
        else if (kOccXY(g, occ2, gx, gy, gw) == 1) status = 2;                             // :91-95
        if (status < 0) {
            {                                                                              // expanded_nodes.clear(): 16-byte stores (the row stride is a multiple of 16 ints)
                int4* t4 = (int4*)table;
                const int4 m1 = make_int4(-1, -1, -1, -1);
                for (size_t t = lane; t < (W.table_len + 3) / 4; t += 64) t4[t] = m1;
            }
            // :97-109
            const double w0 = kNormalizeAngle(sw0);
            int id3[3];
            const int key0 = kKey(g, P, sx0, sy0, w0, id3);
            const double dx = sx0 - gx, dy = sy0 - gy;
            const double f0 = P.lambda_heu * (P.tie_breaker * sqrt(dx * dx + dy * dy));
            if (lane == 0) {
                KNode nd; nd.sx = sx0; nd.sy = sy0; nd.syaw = w0; nd.g = 0.0; nd.f = f0; nd.in_v = 0.0; nd.in_steer = 0.0; nd.parent = -1; nd.flag = K_OPEN;
                nodes[0] = nd;
                nkey[0] = key0;
                if (key0 >= 0) table[key0] = 0;
            }
            kHeapPush(heap, pos, 0, 0, f0, lane);
            n = 1; use_node_num = 1;
        }
        while (status < 0) {
            if (n == 0) { status = 3; break; }                                             // :111, :233
            if (iter_num > 2 * P.allocate_num) { status = 6; break; }                      // (cannot happen: every pop consumes a push and pushes stop at allocate_num -- a guard, not a rule)
            KPROF(5);
            const int cur = kuni(heap_top[0].id);
            const KNode cn = nodes[cur];
            const double cx = kuni(cn.sx), cy = kuni(cn.sy), cw = kuni(cn.syaw), cg = kuni(cn.g), civ = kuni(cn.in_v), cis = kuni(cn.in_steer);
            {
                const double dx = cx - gx, dy = cy - gy;
                if (sqrt(dx * dx + dy * dy) < P.oneshot_range) {                           // :115-127, asignShotTraj kino_astar.h:242-271
                    const double from[3] = {cx, cy, cw}, to[3] = {gx, gy, gw};
                    KDubins path;
                    kDubinsBetween(P, from, to, path);
                    const double len = P.rho * (path.len[0] + path.len[1] + path.len[2]);
                    int M = 0;
                    for (double l = 0.0; l <= len && M < (1 << 20); l += P.coll_interval) M++;      // `for (l = 0; l <= len; l += collision_interval)`
                    bool blocked = false;
                    {
                        double l = 0.0; int k = 0;
                        for (int s = lane; s < M; s += 64) {
                            for (; k < s; k++) l += P.coll_interval;                       // the running sum the reference forms
                            double sp[3];
                            kDubinsInterpolate(P, from, to, path, l / len, sp);
                            if (kOccXY(g, occ2, sp[0], sp[1], sp[2]) == 1) blocked = true;
                        }
                    }
                    if (!__ballot(blocked) && M > 0) {
                        // retrievePath (kino_astar.h:273-292): root .. cur, then the shot samples
                        int depth = 0;
                        if (lane == 0) { for (int v = cur; v >= 0 && depth < P.allocate_num; v = nodes[v].parent) depth++; }
                        depth = kuni(depth);
                        n_path = depth + M;
                        double* out = io.paths + (size_t)q * io.path_cap * 3;
                        if (lane == 0) {
                            int at = depth - 1;
                            for (int v = cur; v >= 0 && at >= 0; v = nodes[v].parent, at--) {
                                if (at < io.path_cap) { out[3 * at] = nodes[v].sx; out[3 * at + 1] = nodes[v].sy; out[3 * at + 2] = nodes[v].syaw; }
                            }
                        }
                        double l = 0.0; int k = 0;
                        for (int s = lane; s < M; s += 64) {
                            for (; k < s; k++) l += P.coll_interval;
                            double sp[3];
                            kDubinsInterpolate(P, from, to, path, l / len, sp);
                            if (depth + s < io.path_cap) { out[3 * (depth + s)] = sp[0]; out[3 * (depth + s) + 1] = sp[1]; out[3 * (depth + s) + 2] = sp[2]; }
                        }
                        status = 0;
                        break;
                    }
                }
            }
            // ---- the primitives of this node, one per lane (:147-195), in two stages around the pop.  Nothing they read is written by the pop
            // (heap and pos only), so the loads of stage 0 -- table rows, occupancy of the collision samples -- are in flight while the pop
            // walks the heap; stage 1 (the nodes found in the table, the terrain lookup) follows it.
            KPROF(0);                                                                      // 0: top of the heap, node fetch, one-shot test
            double csw, ccw;
            kSinCos<FAST>(cw, csw, ccw);
            bool act = false;
            double px = 0.0, py = 0.0, pw = 0.0, tg = 0.0, tf = 0.0, iv = 0.0, is = 0.0;
            int key = -1, pre = -1, pre_flag = 0;
            double pre_g = 0.0;
            iv = l_iv; is = l_is;
            int occ_samp = 0;                                        // spread form: this lane's collision sample (1 = occupied)
            bool inmap = false;
            if (pvalid) kStateTransit<FAST>(cx, cy, cw, csw, ccw, is, l_s, l_y, l_r, px, py, pw);
            if (l_samp) occ_samp = kOccXY(g, occ2, px, py, pw);
            if (pvalid && sgrp == 0 && isInMap<double>(g, px, py, pw)) {                   // :154-158
                int id3[3];
                inmap = true;
                key = kKey(g, P, px, py, pw, id3);
                pre = key >= 0 ? table[key] : -1;                                          // :163-164
            }
            KPROF(1);                                                                      // 1: stage 0 (state transit, keys; loads issued)
            kHeapPop(heap, pos, n, lane);                                                  // :129-131
            KPROF(2);                                                                      // 2: pop
            n--;
            if (lane == 0) nodes[cur].flag = K_CLOSE;
            if (io.expanded && iter_num < io.exp_cap && lane == 0) {
                const int kk = nkey[cur];
                int* e = io.expanded + ((size_t)q * io.exp_cap + iter_num) * 3;
                const bool nanrow = kk >= P.nxy * P.nyawk;
                const int cell = nanrow ? 0 : kk / P.nyawk;
                e[0] = nanrow ? INT_MIN : cell / g.ny; e[1] = nanrow ? INT_MIN : cell % g.ny; e[2] = nanrow ? kk - P.nxy * P.nyawk : kk % P.nyawk;
            }
            iter_num++;
            if (io.max_expand > 0 && iter_num >= io.max_expand) { status = 5; break; }
            const unsigned long long occ_any = __ballot(occ_samp == 1);     // bit 16 k + p: sample k - 1 of primitive p is occupied (spread form)
            if (inmap) {
                if (pre >= 0) { pre_flag = pre == cur ? K_CLOSE : nodes[pre].flag; pre_g = nodes[pre].g; }      // (cur was closed above)
                bool closed = pre >= 0 && pre_flag == K_CLOSE;                             // :166-169
                bool blocked = false;
                if (P.spread) blocked = (((occ_any >> (16 + plane)) | (occ_any >> (32 + plane)) | (occ_any >> (48 + plane))) & 1ull) != 0;
                else {
                    const int nt = P.in_nt[lane];
                    for (int s = 0; s < nt && !closed; s++) {                              // :171-185
                        double xt, yt, wt;
                        kStateTransit<FAST>(cx, cy, cw, csw, ccw, is, P.in_s[lane][s + 1], P.in_y[lane][s + 1], P.in_r[lane][s + 1], xt, yt, wt);
                        if (kOccXY(g, occ2, xt, yt, wt) == 1) { blocked = true; break; }
                    }
                }
                if (!closed && !blocked) {
                    const double arc = iv * P.time_interval;
                    double t = 0.0;                                                        // :187-195
                    t += P.w_r2 * arc;
                    t += P.w_so2 * fabs(is) * arc;
                    t += P.w_vch * fabs(iv - civ);
                    t += P.w_dch * fabs(is - cis);
                    t += P.w_sigma * kTerrainSig(g, px, py, pw);
                    t += cg;
                    tg = t;
                    const double dx = px - gx, dy = py - gy;
                    tf = tg + P.lambda_heu * (P.tie_breaker * sqrt(dx * dx + dy * dy));
                    act = true;
                }
            }
            // ---- the primitives meet the table in their order (:197-229).  What each one finds there depends only on the EARLIER primitives of this
            // expansion with the same key: the first of them creates the node unless the table held one, every later one relaxes it iff its g is
            // below the running g.  Every lane replays that for its own key (one pass over the active lanes), so the actions -- NEW (with the pool
            // index the sequential order would hand out), RELAX, nothing -- are known at once, the node records are written in parallel (the last
            // writer of a key only), and the heap operations, whose order matters, follow in primitive order: push for NEW, the in-place key
            // change for RELAX (no re-heapify, as the reference).
            KPROF(3);                                                                      // 3: stage 1 (table nodes, terrain, cost)
            unsigned long long todo = __ballot(act);
            if (__ballot(act && key < 0)) { status = 6; break; }
            bool exists = pre >= 0;
            double gcur = pre_g;
            int first_new = -1;
            if (P.spread) {
                // the primitives sit in lanes 0..15, one DPP row: fifteen row rotations show every lane every other one.  Of the EARLIER active
                // lanes with its key a lane needs the first (it creates the node when the table had none) and the running g after all of them:
                // g0 = the table node's g or the creator's, then g = g_i wherever g_i < g -- i.e. NaN if g0 is NaN, else the smallest of g0 and
                // the members' non-NaN g.  Both are order-free (a minimum of lane indices, a NaN-ignoring minimum), so the rotation order is too.
                const int actlane = act ? lane : -1;
                const int tg_hi = __double2hiint(tg), tg_lo = __double2loint(tg);
                int first_idx = 64, nearlier = 0;
                double m_all = __longlong_as_double(0x7ff0000000000000ll);
#define UPH_KSTEP(R)                                                                                                             \
                {                                                                                                                \
                    const int ol = kRowRor<R>(actlane), ok = kRowRor<R>(key);                                                    \
                    const double og = __hiloint2double(kRowRor<R>(tg_hi), kRowRor<R>(tg_lo));                                    \
                    if (act && ol >= 0 && ol < lane && ok == key) {                                                              \
                        nearlier++;                                                                                              \
                        if (ol < first_idx) first_idx = ol;                                                                      \
                        if (og < m_all) m_all = og;                                                                              \
                    }                                                                                                            \
                }
                UPH_KSTEP(1) UPH_KSTEP(2) UPH_KSTEP(3) UPH_KSTEP(4) UPH_KSTEP(5) UPH_KSTEP(6) UPH_KSTEP(7) UPH_KSTEP(8)
                UPH_KSTEP(9) UPH_KSTEP(10) UPH_KSTEP(11) UPH_KSTEP(12) UPH_KSTEP(13) UPH_KSTEP(14) UPH_KSTEP(15)
#undef UPH_KSTEP
                const double gfirst = __shfl(tg, first_idx & 63);
                if (pre >= 0) gcur = isnan(pre_g) ? pre_g : (m_all < pre_g ? m_all : pre_g);
                else if (nearlier > 0) { exists = true; first_new = first_idx; gcur = isnan(gfirst) ? gfirst : (m_all < gfirst ? m_all : gfirst); }
            } else {
                for (unsigned long long m = todo; m; m &= m - 1) {
                    const int i = (int)__builtin_ctzll(m);
                    const int ki = klane(key, i);
                    const double gi = klane(tg, i);
                    if (act && ki == key && i < lane) {
                        if (!exists) { exists = true; gcur = gi; first_new = i; }
                        else if (gi < gcur) gcur = gi;
                    }
                }
            }
            KPROF(4);                                                                      // 4: replay of the table order
            const bool is_new = act && !exists, is_relax = act && exists && tg < gcur;
            const unsigned long long newmask = __ballot(is_new), wmask = __ballot(is_new || is_relax);
            const int n_new = __popcll(newmask);
            if (use_node_num + n_new < P.allocate_num) {
                const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
                int mypn = pre;
                if (is_new) mypn = use_node_num + __popcll(newmask & below);
                else if (first_new >= 0) mypn = use_node_num + __popcll(newmask & ((1ull << first_new) - 1ull));
                if (is_new) { nkey[mypn] = key; table[key] = mypn; }
                const bool iswr = is_new || is_relax;
                for (unsigned long long m = wmask; m; m &= m - 1) {
                    const int i = (int)__builtin_ctzll(m);
                    const int pn = klane(mypn, i), ki = klane(key, i);
                    const double ifs = klane(tf, i);
                    // the node record: the LAST writer of a key leaves its state there (an earlier one's would be overwritten at once)
                    const bool later = __ballot(iswr && key == ki && lane > i) != 0;
                    if (!later && lane == i) {
                        KNode nd;
                        nd.sx = px; nd.sy = py; nd.syaw = pw; nd.g = tg; nd.f = tf; nd.in_v = iv; nd.in_steer = is; nd.parent = cur; nd.flag = K_OPEN;
                        nodes[mypn] = nd;
                    }
                    if ((newmask >> i) & 1ull) { kHeapPush(heap, pos, n, pn, ifs, lane); n++; }
                    else if (lane == 0) heap.setF(pos[pn], ifs);
                }
                use_node_num += n_new;
                continue;
            }
            // the pool runs out inside this expansion (:212-216): one primitive at a time, as the reference, up to the one that takes the last node
            unsigned long long done = 0;
            while (todo) {
                const int i = (int)__builtin_ctzll(todo);
                todo &= todo - 1;
                const int ki = klane(key, i);
                if (ki < 0) { status = 6; break; }
                int pn = klane(pre, i), pf = klane(pre_flag, i);
                double pg = klane(pre_g, i);
                if (__ballot(((done >> lane) & 1ull) && key == ki)) {
                    pn = kuni(table[ki]);
                    if (pn >= 0) { const KNode t = nodes[pn]; pf = kuni(t.flag); pg = kuni(t.g); }
                }
                const double ig = klane(tg, i), ifs = klane(tf, i);
                if (pn >= 0 && pf == K_CLOSE) continue;
                KNode nd;                                                                  // (wave-uniform: the successor of primitive i)
                nd.sx = klane(px, i); nd.sy = klane(py, i); nd.syaw = klane(pw, i); nd.g = ig; nd.f = ifs; nd.in_v = klane(iv, i); nd.in_steer = klane(is, i);
                nd.parent = cur; nd.flag = K_OPEN;
                if (pn < 0) {                                                              // :197-217
                    pn = use_node_num;
                    if (lane == 0) { nodes[pn] = nd; nkey[pn] = ki; table[ki] = pn; }
                    kHeapPush(heap, pos, n, pn, ifs, lane);
                    n++;
                    use_node_num++;
                    done |= 1ull << i;
                    if (use_node_num == P.allocate_num) { status = 4; break; }             // :212-216
                } else if (pf == K_OPEN) {                                                 // :218-229
                    if (ig < pg) {
                        if (lane == 0) {
                            nodes[pn] = nd;
                            heap.setF(pos[pn], ifs);                                       // the key the comparisons see from now on; no re-heapify (as the reference)
                        }
                        done |= 1ull << i;
                    }
                }
            }
        }
#ifdef UPH_KINO_PROF
        if (lane == 0 && io.path_cap >= 2) {
            double* pr = io.paths + ((size_t)q * io.path_cap + io.path_cap - 2) * 3;
            for (int k = 0; k < 6; k++) pr[k] = (double)kprof[k];
        }
#endif
        if (lane == 0) { io.status[q] = status; io.n_path[q] = n_path; io.iter_num[q] = iter_num; io.use_node_num[q] = use_node_num; }
    }
}

}  // namespace

struct uph_kino {
    uph_map* map = nullptr;
    int device = 0;
    KinoDev P;
    int slots = 0, wps = UPH_KINO_WPS, flags = 3;      // flags: bit 0 = dynamic query hand-out, bit 1 = sincosFast
    // slots = workspaces allocated now.  auto_slots (uph_kino_create with slots = 0): the workspaces follow the batch sizes that actually arrive --
    // allocated at the first uph_kino_plan_batch, grown when a larger batch comes -- up to slots_cap = what the device can run at once, clamped to
    // half of the HBM that was free at creation (one workspace is ~4 MB at 200 x 200 cells: a single plan() must not reserve 16 GB)
    int slots_cap = 0;
    bool auto_slots = false;
    size_t slot_bytes = 0;
    size_t node_stride = 0, heap_stride = 0, table_stride = 0;
    void *d_P = nullptr, *d_nodes = nullptr, *d_heap = nullptr, *d_pos = nullptr, *d_key = nullptr, *d_table = nullptr;
    void *d_io[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t io_cap[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double last_ms = 0.0;
};

// (re)allocate the per-query workspaces for `want` concurrent queries; when the device cannot hold them, for half as many, and so on
static void kinoFreeWork(uph_kino* k) {
    hipFree(k->d_nodes); hipFree(k->d_heap); hipFree(k->d_pos); hipFree(k->d_key); hipFree(k->d_table);
    k->d_nodes = k->d_heap = k->d_pos = k->d_key = k->d_table = nullptr;
    k->slots = 0;
}
static bool kinoTryAlloc(const uph_kino* k, int n, void* w[5]) {
    for (int q = 0; q < 5; q++) w[q] = nullptr;
    const bool ok = hipMalloc(&w[0], sizeof(KNode) * k->node_stride * n) == hipSuccess && hipMalloc(&w[1], sizeof(KHeap) * k->heap_stride * n) == hipSuccess &&
                    hipMalloc(&w[2], sizeof(int) * k->node_stride * n) == hipSuccess && hipMalloc(&w[3], sizeof(int) * k->node_stride * n) == hipSuccess &&
                    hipMalloc(&w[4], sizeof(int) * k->table_stride * n) == hipSuccess;
    if (ok) return true;
    (void)hipGetLastError();             // (out of memory is sticky until read)
    for (int q = 0; q < 5; q++) { if (w[q]) hipFree(w[q]); w[q] = nullptr; }
    return false;
}
static int kinoAllocWork(uph_kino* k, int want) {
    void* w[5];
    // a growing context first tries the new workspaces NEXT to the old ones: a failed growth then keeps the capacity it had (the searches of the call run
    // in more rounds) instead of leaving the context without any
    if (k->slots > 0 && want > k->slots) {
        if (kinoTryAlloc(k, want, w)) {
            kinoFreeWork(k);
            k->d_nodes = w[0]; k->d_heap = w[1]; k->d_pos = w[2]; k->d_key = w[3]; k->d_table = w[4];
            k->slots = want;
            return UPH_OK;
        }
        // both sets do not fit together: release the old one and take what fits (below)
    }
    kinoFreeWork(k);
    for (int n = std::max(1, want); n >= 1; n = n == 1 ? 0 : n / 2) {
        if (kinoTryAlloc(k, n, w)) { k->d_nodes = w[0]; k->d_heap = w[1]; k->d_pos = w[2]; k->d_key = w[3]; k->d_table = w[4]; k->slots = n; return UPH_OK; }
    }
    setError("uph_kino: hipMalloc of one search workspace (" + std::to_string(k->slot_bytes >> 20) + " MiB) failed");
    return UPH_ERR_HIP;
}

extern "C" {

int uph_kino_create(uph_map* m, const uph_kino_params* kp, int32_t slots, uph_kino** out) {
    if (!m || !kp || !out || slots < 0) { setError("uph_kino_create: bad arguments"); return UPH_ERR_INVALID; }
    const GridDev g = uphMapGrid(m);
    if (g.nx_hold != g.nx) { setError("uph_kino_create: tile maps are not searched (the search needs the whole grid's occupancy)"); return UPH_ERR_INVALID; }
    if (!(kp->yaw_resolution > 0) || !(kp->time_interval > 0) || !(kp->collision_interval > 0) || !(kp->max_vel > 0) || !(kp->max_steer > 0) || !(kp->wheel_base > 0)) {
        setError("uph_kino_create: parameters must be positive"); return UPH_ERR_INVALID;
    }
    KHIPCHK(hipSetDevice(uphMapDevice(m)));
    uph_kino* k = new uph_kino();
    k->map = m; k->device = uphMapDevice(m);
    KinoDev& P = k->P;
    std::memset(&P, 0, sizeof(P));
    P.yaw_inv = 1.0 / kp->yaw_resolution;                                   // kino_astar.cpp:31
    P.lambda_heu = kp->lambda_heu; P.w_r2 = kp->weight_r2; P.w_so2 = kp->weight_so2; P.w_vch = kp->weight_v_change; P.w_dch = kp->weight_delta_change; P.w_sigma = kp->weight_sigma;
    P.time_interval = kp->time_interval; P.coll_interval = kp->collision_interval; P.oneshot_range = kp->oneshot_range; P.wheel_base = kp->wheel_base;
    P.rho = kp->wheel_base / std::tan(kp->max_steer);                       // :33
    P.tie_breaker = 1.0 + 1.0 / 10000;                                      // kino_astar.h:127
    if ((int64_t)g.nx * g.ny > (int64_t)INT32_MAX - 1) { delete k; setError("uph_kino_create: the grid has more than 2^31 columns (node indices and lattice keys are 32-bit)"); return UPH_ERR_LIMIT; }
    P.nxy = g.nx * g.ny;
    P.allocate_num = g.nx * g.ny;                                           // setEnvironment: getXYNum nodes
    {
        const double top = std::floor((3.14159265358979323846 + 3.14159265358979323846) * P.yaw_inv);
        if (!(top >= 0 && top < 4096)) { delete k; setError("uph_kino_create: yaw_resolution gives more than 4096 yaw bins"); return UPH_ERR_LIMIT; }
        P.nyawk = (int)top + 1;
        // the lattice key (ix * ny + iy) * nyawk + iyaw and the table of (nxy + 1) * nyawk entries are 32-bit
        if (((int64_t)P.nxy + 1) * (int64_t)P.nyawk > (int64_t)INT32_MAX) { delete k; setError("uph_kino_create: (cells + 1) x yaw bins of the search lattice exceeds 2^31 (coarser kino_astar/yaw_resolution or a smaller grid)"); return UPH_ERR_LIMIT; }
    }
    // the primitives and their collision sample times, by the reference's own loops (kino_astar.cpp:138-145, 173-175)
    int ni = 0;
    for (double v = 0; v <= kp->max_vel + 1e-3; v += 0.5 * kp->max_vel)
        for (double steer = -kp->max_steer; steer <= kp->max_steer + 1e-3; steer += 0.5 * kp->max_steer) {
            if (ni >= K_MAX_INPUTS) { delete k; setError("uph_kino_create: more than 64 motion primitives"); return UPH_ERR_LIMIT; }
            P.in_v[ni] = v; P.in_steer[ni] = steer;
            const double tand = std::tan(steer);
            double tsamp[K_MAX_TSAMP];
            const double arc = v * kp->time_interval;
            const double temp_ct = kp->collision_interval / arc * kp->time_interval;
            int nt = 0;
            for (double t = temp_ct; t <= kp->time_interval + 1e-3; t += temp_ct) {
                if (nt >= K_MAX_TSAMP) { delete k; setError("uph_kino_create: more than 8 collision samples per primitive (collision_interval too fine)"); return UPH_ERR_LIMIT; }
                tsamp[nt++] = t;
            }
            P.in_nt[ni] = nt;
            for (int q = 0; q <= nt; q++) {
                const double T = q == 0 ? kp->time_interval : tsamp[q - 1];
                const double s_ = v * T;
                const double y_ = s_ * tand / kp->wheel_base;
                P.in_s[ni][q] = s_; P.in_y[ni][q] = y_; P.in_r[ni][q] = s_ / y_;
            }
            ni++;
        }
    P.n_inputs = ni;
    {
        int mx = 0;
        for (int i = 0; i < ni; i++) mx = std::max(mx, P.in_nt[i]);
        P.spread = (ni <= 16 && mx <= 3) ? 1 : 0;
    }
    k->node_stride = (size_t)P.allocate_num;
    k->heap_stride = (size_t)P.allocate_num + 1;
    k->table_stride = (((size_t)P.nxy + 1) * P.nyawk + 15) & ~(size_t)15;
    k->slot_bytes = sizeof(KNode) * k->node_stride + sizeof(KHeap) * k->heap_stride + 2 * sizeof(int) * k->node_stride + sizeof(int) * k->table_stride;
    if (hipMalloc(&k->d_P, sizeof(KinoDev)) != hipSuccess) { setError("uph_kino_create: hipMalloc(params) failed"); uph_kino_destroy(k); return UPH_ERR_HIP; }
    if (slots == 0) {
        hipDeviceProp_t prop;
        KHIPCHK(hipGetDeviceProperties(&prop, k->device));
        size_t free_b = 0, total_b = 0;
        KHIPCHK(hipMemGetInfo(&free_b, &total_b));
        const int64_t hw = (int64_t)prop.multiProcessorCount * 4 * UPH_KINO_WPS;        // one workspace per wave slot of the default instantiation
        const int64_t fit = (int64_t)((free_b / 2) / k->slot_bytes);
        if (fit < 1) { setError("uph_kino_create: one search workspace (" + std::to_string(k->slot_bytes >> 20) + " MiB for this grid) does not fit the free device memory"); uph_kino_destroy(k); return UPH_ERR_LIMIT; }
        k->slots_cap = (int)std::min(hw, fit);
        k->auto_slots = true;
        k->slots = 0;                        // allocated by the first uph_kino_plan_batch, for the batch that arrives
    } else {
        k->slots_cap = slots;
        const int r = kinoAllocWork(k, slots);      // (fewer than asked for when the device cannot hold them: uph_kino_slots tells)
        if (r != UPH_OK) { uph_kino_destroy(k); return r; }
    }
    if (hipMemcpy(k->d_P, &P, sizeof(KinoDev), hipMemcpyHostToDevice) != hipSuccess) { setError("uph_kino_create: hipMemcpy failed"); uph_kino_destroy(k); return UPH_ERR_HIP; }
    if (hipEventCreate(&k->e0) != hipSuccess || hipEventCreate(&k->e1) != hipSuccess) { setError("uph_kino_create: hipEventCreate failed"); uph_kino_destroy(k); return UPH_ERR_HIP; }
    *out = k;
    return UPH_OK;
}

void uph_kino_destroy(uph_kino* k) {
    if (!k) return;
    hipSetDevice(k->device);
    hipFree(k->d_P);
    kinoFreeWork(k);
    for (int i = 0; i < 10; i++) hipFree(k->d_io[i]);
    if (k->e0) hipEventDestroy(k->e0);
    if (k->e1) hipEventDestroy(k->e1);
    delete k;
}

// workspaces allocated; an automatic context (created with slots = 0) that has not searched yet reports its upper bound
int uph_kino_slots(const uph_kino* k) { return k ? ((k->auto_slots && k->slots == 0) ? k->slots_cap : k->slots) : UPH_ERR_INVALID; }
// experiment knob: which instantiation of the search kernel runs -- 2, 4, 6 or 8 waves per SIMD (register caps 256 / 128 / 80 / 64)
// experiment knob: bit 0 = queries handed out dynamically (longest first) instead of statically, bit 1 = sincosFast instead of the device library's sin / cos
int uph_kino_set_flags(uph_kino* k, int32_t flags) { if (!k || flags < 0 || flags > 3) return UPH_ERR_INVALID; k->flags = flags; return UPH_OK; }
int uph_kino_set_wps(uph_kino* k, int32_t wps) {
    if (!k || (wps != 2 && wps != 4 && wps != 6 && wps != 8)) { setError("uph_kino_set_wps: 2, 4, 6 or 8"); return UPH_ERR_INVALID; }
    k->wps = wps;
    return UPH_OK;
}
int uph_kino_primitives(const uph_kino* k) { return k ? k->P.n_inputs : UPH_ERR_INVALID; }

int uph_kino_plan_batch(uph_kino* k, int32_t B, const double* starts, const double* goals, int32_t path_cap, double* paths, int32_t* n_path, int32_t* status,
                        int32_t* iter_num, int32_t* use_node_num, int32_t max_expand, int32_t exp_cap, int32_t* expanded) {
    if (!k || B <= 0 || !starts || !goals || path_cap < 0 || (path_cap > 0 && !paths) || !n_path || !status || exp_cap < 0 || (exp_cap > 0 && !expanded)) {
        setError("uph_kino_plan_batch: bad arguments"); return UPH_ERR_INVALID;
    }
    KHIPCHK(hipSetDevice(k->device));
    if (k->auto_slots && k->slots < std::min((int)B, k->slots_cap)) {
        // automatic workspaces follow the batch, GEOMETRICALLY: first call, or a larger batch than any before -> at least twice the previous capacity, so a
        // context fed slowly growing batches (17, 18, 19 ... queries) reallocates its ~4 MB-per-slot arrays a few times over its life, not at every call
        const int want = std::min(k->slots_cap, std::max(std::max((int)B, 16), 2 * k->slots));
        const int r = kinoAllocWork(k, want);
        if (r != UPH_OK) return r;
        if (k->slots < want) k->slots_cap = k->slots;      // the device could not hold more: stop asking
    }
    const size_t need[10] = {sizeof(double) * 3 * (size_t)B, sizeof(double) * 3 * (size_t)B, sizeof(double) * 3 * (size_t)B * (size_t)std::max(1, path_cap), sizeof(int) * (size_t)B,
                             sizeof(int) * (size_t)B, sizeof(int) * (size_t)B, sizeof(int) * (size_t)B, sizeof(int) * 3 * (size_t)B * (size_t)std::max(1, exp_cap), sizeof(int) * (size_t)B, sizeof(int)};
    for (int i = 0; i < 10; i++) {
        if (need[i] <= k->io_cap[i]) continue;
        if (k->d_io[i]) hipFree(k->d_io[i]);
        k->d_io[i] = nullptr; k->io_cap[i] = 0;
        const size_t want = need[i] + need[i] / 4 + 256;
        if (hipMalloc(&k->d_io[i], want) != hipSuccess) { setError("uph_kino_plan_batch: hipMalloc failed"); return UPH_ERR_HIP; }
        k->io_cap[i] = want;
    }
    KHIPCHK(hipMemcpy(k->d_io[0], starts, need[0], hipMemcpyHostToDevice));
    KHIPCHK(hipMemcpy(k->d_io[1], goals, need[1], hipMemcpyHostToDevice));
    {   // longest searches first (the expansion count grows with the start-goal distance), handed out from a shared cursor: the launch ends with short ones
        std::vector<int> order((size_t)B);
        std::vector<double> dist((size_t)B);
        for (int b = 0; b < B; b++) { order[b] = b; const double dx = goals[3 * b] - starts[3 * b], dy = goals[3 * b + 1] - starts[3 * b + 1]; dist[b] = dx * dx + dy * dy; }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return dist[a] > dist[b2]; });
        const int zero = 0;
        KHIPCHK(hipMemcpy(k->d_io[8], order.data(), need[8], hipMemcpyHostToDevice));
        KHIPCHK(hipMemcpy(k->d_io[9], &zero, sizeof(int), hipMemcpyHostToDevice));
    }
    KinoIO io;
    io.starts = (const double*)k->d_io[0]; io.goals = (const double*)k->d_io[1];
    io.paths = (double*)k->d_io[2]; io.path_cap = path_cap;
    io.n_path = (int*)k->d_io[3]; io.status = (int*)k->d_io[4]; io.iter_num = (int*)k->d_io[5]; io.use_node_num = (int*)k->d_io[6];
    io.expanded = exp_cap > 0 ? (int*)k->d_io[7] : nullptr; io.exp_cap = exp_cap;
    io.max_expand = max_expand;
    io.order = (const int*)k->d_io[8]; io.next = (k->flags & 1) ? (int*)k->d_io[9] : nullptr;
    KinoWork W;
    W.nodes = (KNode*)k->d_nodes; W.heap = (KHeap*)k->d_heap; W.pos = (int*)k->d_pos; W.key = (int*)k->d_key; W.table = (int*)k->d_table;
    W.table_len = ((size_t)k->P.nxy + 1) * k->P.nyawk;
    const char *occ = nullptr, *occ2 = nullptr;
    uphMapOcc(k->map, &occ, &occ2);
    const int grid = std::min((int)B, k->slots);
    if (grid < 1) { setError("uph_kino_plan_batch: no workspace"); return UPH_ERR_HIP; }
    KHIPCHK(hipEventRecord(k->e0, 0));
#define UPH_KINO_LAUNCH2(WPS_, F_) hipLaunchKernelGGL((uph_kino_kernel<WPS_, F_>), dim3(grid), dim3(64), 0, 0, uphMapGrid(k->map), occ, occ2, (const KinoDev*)k->d_P, W, k->node_stride, k->heap_stride, k->table_stride, io, (int)B)
#define UPH_KINO_LAUNCH(WPS_) do { if (k->flags & 2) UPH_KINO_LAUNCH2(WPS_, true); else UPH_KINO_LAUNCH2(WPS_, false); } while (0)
    if (k->wps == 2) UPH_KINO_LAUNCH(2);
    else if (k->wps == 6) UPH_KINO_LAUNCH(6);
    else if (k->wps == 8) UPH_KINO_LAUNCH(8);
    else UPH_KINO_LAUNCH(4);
#undef UPH_KINO_LAUNCH2
#undef UPH_KINO_LAUNCH
    KHIPCHK(hipGetLastError());
    KHIPCHK(hipEventRecord(k->e1, 0));
    KHIPCHK(hipDeviceSynchronize());
    float ms = 0.f;
    KHIPCHK(hipEventElapsedTime(&ms, k->e0, k->e1));
    k->last_ms = ms;
    if (path_cap > 0) KHIPCHK(hipMemcpy(paths, k->d_io[2], sizeof(double) * 3 * (size_t)B * path_cap, hipMemcpyDeviceToHost));
    KHIPCHK(hipMemcpy(n_path, k->d_io[3], need[3], hipMemcpyDeviceToHost));
    KHIPCHK(hipMemcpy(status, k->d_io[4], need[4], hipMemcpyDeviceToHost));
    if (iter_num) KHIPCHK(hipMemcpy(iter_num, k->d_io[5], need[5], hipMemcpyDeviceToHost));
    if (use_node_num) KHIPCHK(hipMemcpy(use_node_num, k->d_io[6], need[6], hipMemcpyDeviceToHost));
    if (exp_cap > 0) KHIPCHK(hipMemcpy(expanded, k->d_io[7], sizeof(int) * 3 * (size_t)B * exp_cap, hipMemcpyDeviceToHost));
    return UPH_OK;
}

int uph_kino_stats(uph_kino* k, double* kernel_ms) { if (!k || !kernel_ms) return UPH_ERR_INVALID; *kernel_ms = k->last_ms; return UPH_OK; }

}  // extern "C"
