// ORACLE -- TEST INFRASTRUCTURE ONLY (see banded.hpp header).  PARITY UNPINNED -- doubly so for the Dubins one-shot: OMPL is not in this
// image, so ompl::base::DubinsStateSpace is restated from its published algorithm (OMPL 1.4.2 = ROS Noetic's libompl-dev,
// src/ompl/base/spaces/src/DubinsStateSpace.cpp; Shkel & Lumelsky 2001 classification, the six word formulas below).
//
// Restates front_end (paths under /root/reference/src/uneven_planner/front_end):
//   KinoAstar::plan            src/kino_astar.cpp:67-236      best-first search over motion primitives, sigma-weighted cost :187-195
//   stateTransit               include/front_end/kino_astar.h:218-240
//   stateToIndex / yawToIndex  kino_astar.h:180-191
//   normalizeAngle, getHeu     kino_astar.h:193-216
//   asignShotTraj              kino_astar.h:242-271   (ompl::base::DubinsStateSpace::distance / interpolate)
//   retrievePath               kino_astar.h:273-292
//   the open set               std::priority_queue<PathNodePtr, std::vector<PathNodePtr>, NodeComparator> (kino_astar.h:49-57, 107): a binary heap
//                              of node POINTERS compared through their CURRENT f_score -- the reference lowers an open node's f_score in
//                              place (kino_astar.cpp:218-229) without re-heapifying, so the heap property may be violated; libstdc++'s
//                              push_heap / pop_heap (bits/stl_heap.h __push_heap, __adjust_heap) are restated literally below so that the
//                              pop order, ties and these violations included, is THE reference's and not "some valid best-first order".
//
// Behaviour of the reference that is undefined C++ but deterministic on the platform it runs on (x86-64, gcc), made explicit here:
//   * v = 0 primitives with a non-zero steer (kino_astar.cpp:138-145; four of the fifteen) divide 0 / 0 in stateTransit (:228): the successor
//     state is (NaN, NaN, yaw).  UnevenMap::isInMap(NaN) is true (every comparison is false, uneven_map.h:437-454), floor(NaN) converted to
//     int is 0x80000000 (cvttsd2si's "integer indefinite"), getTerrainSig returns NaN, so a node with index (INT_MIN, INT_MIN, yaw index)
//     and g = f = NaN enters the open set -- at most one per yaw index.  NaN compares false both ways in NodeComparator; where such a node
//     travels in the heap follows from the literal heap code.  floorToInt() below is that conversion.
#pragma once
#include <climits>
#include <cmath>
#include <cstdint>
#include <vector>

#include "terrain.hpp"

namespace orc {

struct KinoParams {                     // rosparam kino_astar/... (kino_astar.cpp:7-20), values of run_hill.yaml:16-30
    double yaw_resolution = 3.15, lambda_heu = 1.0, weight_r2 = 1.0, weight_so2 = 0.5, weight_v_change = 0.0, weight_delta_change = 0.0,
           weight_sigma = 10.0, time_interval = 0.3, collision_interval = 0.06, oneshot_range = 1.0, wheel_base = 0.26, max_steer = 0.5, max_vel = 0.5;
};

inline int floorToInt(double v) {       // (int)floor(v) as x86-64 converts it: NaN and out-of-range give INT_MIN
    const double f = std::floor(v);
    if (!(f >= -2147483648.0 && f <= 2147483647.0)) return INT_MIN;
    return (int)f;
}

// ---------------------------------------------------------------------------------------------------------------- Dubins (OMPL 1.4.2)
namespace dubins {
constexpr double twopi = 2.0 * M_PI;
constexpr double DUBINS_EPS = 1e-6, DUBINS_ZERO = -1e-7;
enum Seg { LEFT = 0, STRAIGHT = 1, RIGHT = 2 };
// dubinsPathType[6][3]: LSL, RSR, RSL, LSR, RLR, LRL
static const int pathType[6][3] = {{LEFT, STRAIGHT, LEFT}, {RIGHT, STRAIGHT, RIGHT}, {RIGHT, STRAIGHT, LEFT}, {LEFT, STRAIGHT, RIGHT}, {RIGHT, LEFT, RIGHT}, {LEFT, RIGHT, LEFT}};
struct Path {
    int type = 0;
    double len[3] = {0.0, 1.7976931348623157e308, 0.0};      // DubinsPath(type = LSL, t = 0, p = max double, q = 0)
    double length() const { return len[0] + len[1] + len[2]; }
};
inline double mod2pi(double x) {
    if (x < 0 && x > DUBINS_ZERO) return 0;
    double xm = x - twopi * std::floor(x / twopi);
    if (twopi - xm < .5 * DUBINS_EPS) xm = 0.;
    return xm;
}
inline Path mk(int type, double t, double p, double q) { Path r; r.type = type; r.len[0] = t; r.len[1] = p; r.len[2] = q; return r; }
inline Path LSL(double d, double alpha, double beta) {
    double ca = std::cos(alpha), sa = std::sin(alpha), cb = std::cos(beta), sb = std::sin(beta);
    double tmp = 2. + d * d - 2. * (ca * cb + sa * sb - d * (sa - sb));
    if (tmp >= DUBINS_ZERO) {
        double theta = std::atan2(cb - ca, d + sa - sb);
        double t = mod2pi(-alpha + theta);
        double p = std::sqrt(std::max(tmp, 0.));
        double q = mod2pi(beta - theta);
        return mk(0, t, p, q);
    }
    return Path();
}
inline Path RSR(double d, double alpha, double beta) {
    double ca = std::cos(alpha), sa = std::sin(alpha), cb = std::cos(beta), sb = std::sin(beta);
    double tmp = 2. + d * d - 2. * (ca * cb + sa * sb - d * (sb - sa));
    if (tmp >= DUBINS_ZERO) {
        double theta = std::atan2(ca - cb, d - sa + sb);
        double t = mod2pi(alpha - theta);
        double p = std::sqrt(std::max(tmp, 0.));
        double q = mod2pi(-beta + theta);
        return mk(1, t, p, q);
    }
    return Path();
}
inline Path RSL(double d, double alpha, double beta) {
    double ca = std::cos(alpha), sa = std::sin(alpha), cb = std::cos(beta), sb = std::sin(beta);
    double tmp = d * d - 2. + 2. * (ca * cb + sa * sb - d * (sa + sb));
    if (tmp >= DUBINS_ZERO) {
        double p = std::sqrt(std::max(tmp, 0.));
        double theta = std::atan2(ca + cb, d - sa - sb) - std::atan2(2., p);
        double t = mod2pi(alpha - theta);
        double q = mod2pi(beta - theta);
        return mk(2, t, p, q);
    }
    return Path();
}
inline Path LSR(double d, double alpha, double beta) {
    double ca = std::cos(alpha), sa = std::sin(alpha), cb = std::cos(beta), sb = std::sin(beta);
    double tmp = -2. + d * d + 2. * (ca * cb + sa * sb + d * (sa + sb));
    if (tmp >= DUBINS_ZERO) {
        double p = std::sqrt(std::max(tmp, 0.));
        double theta = std::atan2(-ca - cb, d + sa + sb) - std::atan2(-2., p);
        double t = mod2pi(-alpha + theta);
        double q = mod2pi(-beta + theta);
        return mk(3, t, p, q);
    }
    return Path();
}
inline Path RLR(double d, double alpha, double beta) {
    double ca = std::cos(alpha), sa = std::sin(alpha), cb = std::cos(beta), sb = std::sin(beta);
    double tmp = .125 * (6. - d * d + 2. * (ca * cb + sa * sb + d * (sa - sb)));
    if (std::fabs(tmp) < 1.) {
        double p = twopi - std::acos(tmp);
        double theta = std::atan2(ca - cb, d - sa + sb);
        double t = mod2pi(alpha - theta + .5 * p);
        double q = mod2pi(alpha - beta - t + p);
        return mk(4, t, p, q);
    }
    return Path();
}
inline Path LRL(double d, double alpha, double beta) {
    double ca = std::cos(alpha), sa = std::sin(alpha), cb = std::cos(beta), sb = std::sin(beta);
    double tmp = .125 * (6. - d * d + 2. * (ca * cb + sa * sb - d * (sa - sb)));
    if (std::fabs(tmp) < 1.) {
        double p = twopi - std::acos(tmp);
        double theta = std::atan2(-ca + cb, d + sa - sb);
        double t = mod2pi(-alpha + theta + .5 * p);
        double q = mod2pi(beta - alpha - t + p);
        return mk(5, t, p, q);
    }
    return Path();
}
inline Path shortest(double d, double alpha, double beta) {
    if (d < DUBINS_EPS && std::fabs(alpha - beta) < DUBINS_EPS) return mk(0, 0, d, 0);
    Path path = LSL(d, alpha, beta), tmp = RSR(d, alpha, beta);
    double len, minLength = path.length();
    if ((len = tmp.length()) < minLength) { minLength = len; path = tmp; }
    tmp = RSL(d, alpha, beta);
    if ((len = tmp.length()) < minLength) { minLength = len; path = tmp; }
    tmp = LSR(d, alpha, beta);
    if ((len = tmp.length()) < minLength) { minLength = len; path = tmp; }
    tmp = RLR(d, alpha, beta);
    if ((len = tmp.length()) < minLength) { minLength = len; path = tmp; }
    tmp = LRL(d, alpha, beta);
    if ((len = tmp.length()) < minLength) path = tmp;
    return path;
}
// DubinsStateSpace::dubins(state1, state2)
inline Path between(const double s1[3], const double s2[3], double rho) {
    double x1 = s1[0], y1 = s1[1], th1 = s1[2];
    double x2 = s2[0], y2 = s2[1], th2 = s2[2];
    double dx = x2 - x1, dy = y2 - y1, d = std::sqrt(dx * dx + dy * dy) / rho, th = std::atan2(dy, dx);
    double alpha = mod2pi(th1 - th), beta = mod2pi(th2 - th);
    return shortest(d, alpha, beta);
}
// DubinsStateSpace::distance (isSymmetric_ = false, the reference's default construction kino_astar.cpp:33)
inline double distance(const double s1[3], const double s2[3], double rho) { return rho * between(s1, s2, rho).length(); }
// DubinsStateSpace::interpolate(from, to, t, state)
inline void interpolate(const double from[3], const double to[3], double t, double rho, double out[3]) {
    if (t >= 1.) { out[0] = to[0]; out[1] = to[1]; out[2] = to[2]; return; }
    if (t <= 0.) { out[0] = from[0]; out[1] = from[1]; out[2] = from[2]; return; }
    const Path path = between(from, to, rho);
    double sx = 0., sy = 0., syaw = from[2];
    double seg = t * path.length(), phi, v;
    for (unsigned int i = 0; i < 3 && seg > 0; ++i) {
        v = std::min(seg, path.len[i]);
        phi = syaw;
        seg -= v;
        switch (pathType[path.type][i]) {
            case LEFT:
                sx = sx + std::sin(phi + v) - std::sin(phi); sy = sy - std::cos(phi + v) + std::cos(phi);
                syaw = phi + v;
                break;
            case RIGHT:
                sx = sx - std::sin(phi - v) + std::sin(phi); sy = sy + std::cos(phi - v) - std::cos(phi);
                syaw = phi - v;
                break;
            case STRAIGHT:
                sx = sx + v * std::cos(phi); sy = sy + v * std::sin(phi);
                break;
        }
    }
    out[0] = sx * rho + from[0];
    out[1] = sy * rho + from[1];
    // SO2StateSpace::enforceBounds
    double w = std::fmod(syaw, 2.0 * M_PI);
    if (w < -M_PI) w += 2.0 * M_PI;
    else if (w >= M_PI) w -= 2.0 * M_PI;
    out[2] = w;
}
}  // namespace dubins

// ---------------------------------------------------------------------------------------------------------------- the search
struct KinoNode {                       // PathNode, kino_astar.h:34-46
    int index[3];
    double state[3];
    double input[2];
    double g_score, f_score;
    char node_state;                    // 'a' CLOSE, 'b' OPEN, 'c' NOT_EXPAND
    int parent;                         // pool index, -1 = NULL
};

struct KinoResult {
    int status = 0;                     // 0 path found, 1 start not free, 2 goal not free, 3 open set ran empty, 4 node pool exhausted (:212-216)
    int iter_num = 0, use_node_num = 0, n_shot = 0;
    std::vector<double> path;           // front_end_path: poses [x, y, yaw]
    std::vector<int> expanded;          // pool index of every node popped and expanded, in order (:129-131)
    std::vector<int> expanded_index;    // ... and its (ix, iy, iyaw)
};

struct KinoAstar {
    const Grid* map = nullptr;
    KinoParams P;
    double yaw_resolution_inv = 1.0, tie_breaker = 1.0 + 1.0 / 10000;       // kino_astar.h:127
    double dubins_rho = 1.0;
    int allocate_num = 0;
    std::vector<KinoNode> pool;
    std::vector<int> heap;              // the priority_queue's underlying vector (pool indices)

    void init(const Grid* g, const KinoParams& p) {
        map = g; P = p;
        yaw_resolution_inv = 1.0 / P.yaw_resolution;                        // kino_astar.cpp:31
        dubins_rho = P.wheel_base / std::tan(P.max_steer);                  // :33
        allocate_num = g->voxel_num[0] * g->voxel_num[1];                   // setEnvironment, kino_astar.h:170-178 (getXYNum)
        pool.assign(allocate_num, KinoNode());
    }
    static double normalizeAngle(double angle) {                            // kino_astar.h:193-204
        double nor_angle = angle;
        while (nor_angle > M_PI) nor_angle -= 6.283185307179586;
        while (nor_angle < -M_PI) nor_angle += 6.283185307179586;
        return nor_angle;
    }
    void stateToIndex(const double state[3], int idx[3]) const {            // kino_astar.h:187-191 (posToIndex uneven_map.h:411-417)
        idx[0] = floorToInt((state[0] - map->map_origin[0]) * map->xy_resolution_inv);
        idx[1] = floorToInt((state[1] - map->map_origin[1]) * map->xy_resolution_inv);
        idx[2] = floorToInt((normalizeAngle(state[2]) + M_PI) * yaw_resolution_inv);
    }
    double getHeu(const double x1[3], const double x2[3]) const {           // kino_astar.h:213-216 (Eigen norm of a 2-vector: sqrt(dx*dx + dy*dy))
        const double dx = x1[0] - x2[0], dy = x1[1] - x2[1];
        return tie_breaker * std::sqrt(dx * dx + dy * dy);
    }
    void stateTransit(const double state0[3], double state1[3], const double ctrl[2], double T) const {      // kino_astar.h:218-240
        double v = ctrl[0];
        double delta = ctrl[1];
        double s = v * T;
        double y = s * std::tan(delta) / P.wheel_base;
        if (std::fabs(delta) > 1e-4) {
            double r = s / y;
            state1[0] = state0[0] + r * (std::sin(state0[2] + y) - std::sin(state0[2]));
            state1[1] = state0[1] - r * (std::cos(state0[2] + y) - std::cos(state0[2]));
            state1[2] = state0[2] + y;
            state1[2] = normalizeAngle(state1[2]);
        } else {
            state1[0] = state0[0] + s * std::cos(state0[2]);
            state1[1] = state0[1] + s * std::sin(state0[2]);
            state1[2] = state0[2];
        }
    }
    // UnevenMap::isOccupancy / isOccupancyXY with the conversion made explicit (uneven_map.h:473-500)
    int isOccupancy(const double pos[3]) const {
        int id[3] = {floorToInt((pos[0] - map->map_origin[0]) * map->xy_resolution_inv), floorToInt((pos[1] - map->map_origin[1]) * map->xy_resolution_inv),
                     floorToInt((pos[2] - map->map_origin[2]) * map->yaw_resolution_inv)};
        if (!map->isInMapIdx(id)) return -1;
        return (int)map->occ_buffer[map->toAddress(id[0], id[1], id[2])];
    }
    int isOccupancyXY(const double pos[3]) const {
        int id[3] = {floorToInt((pos[0] - map->map_origin[0]) * map->xy_resolution_inv), floorToInt((pos[1] - map->map_origin[1]) * map->xy_resolution_inv),
                     floorToInt((pos[2] - map->map_origin[2]) * map->yaw_resolution_inv)};
        if (!map->isInMapIdx(id)) return -1;
        return (int)map->occ_r2_buffer[(size_t)id[0] * map->voxel_num[1] + id[1]];
    }
    double getTerrainSig(const double pos[3]) const {                       // uneven_map.h:389-396
        if (std::isnan(pos[0]) || std::isnan(pos[1])) return std::nan("");  // isInMap(NaN) is true and the trilinear weights are NaN (header)
        RXS2 v;
        map->getTerrain(pos, v);
        return v.sigma;
    }

    // ---- libstdc++ bits/stl_heap.h, comp(a, b) = pool[a].f_score > pool[b].f_score (NodeComparator)
    bool comp(int a, int b) const { return pool[a].f_score > pool[b].f_score; }
    void pushHeap(int value) {                                              // vector::push_back + std::push_heap -> __push_heap(first, len - 1, 0, value)
        heap.push_back(value);
        long hole = (long)heap.size() - 1, top = 0;
        long parent = (hole - 1) / 2;
        while (hole > top && comp(heap[parent], value)) {
            heap[hole] = heap[parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        heap[hole] = value;
    }
    void popHeap() {                                                        // std::pop_heap + vector::pop_back
        if (heap.size() > 1) {
            const long last = (long)heap.size() - 1;
            const int value = heap[last];                                   // __pop_heap: value = *result; *result = *first; __adjust_heap(first, 0, last - first, value)
            heap[last] = heap[0];
            const long len = last;
            long hole = 0, top = 0, second = 0;
            while (second < (len - 1) / 2) {
                second = 2 * (second + 1);
                if (comp(heap[second], heap[second - 1])) second--;
                heap[hole] = heap[second];
                hole = second;
            }
            if ((len & 1) == 0 && second == (len - 2) / 2) {
                second = 2 * (second + 1);
                heap[hole] = heap[second - 1];
                hole = second - 1;
            }
            long parent = (hole - 1) / 2;                                   // __push_heap(first, hole, top, value)
            while (hole > top && comp(heap[parent], value)) {
                heap[hole] = heap[parent];
                hole = parent;
                parent = (hole - 1) / 2;
            }
            heap[hole] = value;
        }
        heap.pop_back();
    }

    // asignShotTraj, kino_astar.h:242-271; returns the shot path (empty: blocked)
    std::vector<double> shot(const double state1[3], const double state2[3]) const {
        std::vector<double> sp;
        const double len = dubins::distance(state1, state2, dubins_rho);
        for (double l = 0.0; l <= len; l += P.collision_interval) {
            double s[3];
            dubins::interpolate(state1, state2, l / len, dubins_rho, s);
            sp.push_back(s[0]); sp.push_back(s[1]); sp.push_back(s[2]);
        }
        for (size_t i = 0; i < sp.size(); i += 3)
            if (isOccupancyXY(&sp[i]) == 1) { sp.clear(); break; }
        return sp;
    }

    KinoResult plan(const double start_state[3], const double end_state[3], int max_expand = 0) {      // kino_astar.cpp:67-236
        KinoResult R;
        int use_node_num = 0, iter_num = 0;
        heap.clear();
        // expanded_nodes: NodeHashTable over (ix, iy, iyaw) -- a dense table over the lattice plus the NaN-state keys (INT_MIN, INT_MIN, iyaw)
        const int nxy = map->voxel_num[0] * map->voxel_num[1];
        const int nyawk = floorToInt((M_PI + M_PI) * yaw_resolution_inv) + 1;
        std::vector<int> table((size_t)nxy * nyawk, -1), nan_table(nyawk, -1);
        auto slot = [&](const int id[3]) -> int* {
            if (id[2] < 0 || id[2] >= nyawk) return nullptr;
            if (id[0] == INT_MIN && id[1] == INT_MIN) return &nan_table[id[2]];
            if (id[0] < 0 || id[1] < 0 || id[0] >= map->voxel_num[0] || id[1] >= map->voxel_num[1]) return nullptr;
            return &table[((size_t)id[0] * map->voxel_num[1] + id[1]) * nyawk + id[2]];
        };
        const double end_pt[3] = {end_state[0], end_state[1], end_state[2]};
        if (isOccupancy(start_state) == 1) { R.status = 1; return R; }                          // :86-90
        if (isOccupancyXY(end_state) == 1) { R.status = 2; return R; }                          // :91-95
        {
            KinoNode& c = pool[0];                                                                // :97-109
            c.parent = -1;
            c.state[0] = start_state[0]; c.state[1] = start_state[1]; c.state[2] = normalizeAngle(start_state[2]);
            stateToIndex(c.state, c.index);
            c.g_score = 0.0;
            c.input[0] = 0.0; c.input[1] = 0.0;
            c.f_score = P.lambda_heu * getHeu(c.state, end_pt);
            c.node_state = 'b';
            pushHeap(0);
            use_node_num += 1;
            int* s = slot(c.index);
            if (s) *s = 0;
        }
        while (!heap.empty()) {                                                                   // :111
            const int cur = heap[0];
            {
                const double dx = pool[cur].state[0] - end_pt[0], dy = pool[cur].state[1] - end_pt[1];
                if (std::sqrt(dx * dx + dy * dy) < P.oneshot_range) {                             // :115-127
                    std::vector<double> sp = shot(pool[cur].state, end_state);
                    if (!sp.empty()) {
                        // retrievePath, kino_astar.h:273-292
                        std::vector<int> chain;
                        for (int n = cur; n >= 0; n = pool[n].parent) chain.push_back(n);
                        for (size_t i = chain.size(); i-- > 0;) for (int k = 0; k < 3; k++) R.path.push_back(pool[chain[i]].state[k]);
                        R.path.insert(R.path.end(), sp.begin(), sp.end());
                        R.n_shot = (int)sp.size() / 3;
                        R.status = 0; R.iter_num = iter_num; R.use_node_num = use_node_num;
                        return R;
                    }
                }
            }
            popHeap();                                                                            // :129-131
            pool[cur].node_state = 'a';
            iter_num += 1;
            R.expanded.push_back(cur);
            for (int k = 0; k < 3; k++) R.expanded_index.push_back(pool[cur].index[k]);
            if (max_expand > 0 && iter_num >= max_expand) { R.status = 5; R.iter_num = iter_num; R.use_node_num = use_node_num; return R; }      // (checker's own cap)
            const double cur_state[3] = {pool[cur].state[0], pool[cur].state[1], pool[cur].state[2]};
            std::vector<double> inputs;                                                           // :138-145
            for (double v = 0; v <= P.max_vel + 1e-3; v += 0.5 * P.max_vel)
                for (double steer = -P.max_steer; steer <= P.max_steer + 1e-3; steer += 0.5 * P.max_steer) { inputs.push_back(v); inputs.push_back(steer); }
            for (size_t i = 0; i < inputs.size() / 2; i++) {                                      // :147-230
                const double input[2] = {inputs[2 * i], inputs[2 * i + 1]};
                double pro_state[3];
                stateTransit(cur_state, pro_state, input, P.time_interval);
                if (!map->isInMap(pro_state)) continue;                                           // :154-158
                int pro_id[3];
                stateToIndex(pro_state, pro_id);
                int* ps = slot(pro_id);
                int pro_node = ps ? *ps : -1;                                                     // (a key outside the table cannot have been inserted)
                if (pro_node >= 0 && pool[pro_node].node_state == 'a') continue;                  // :166-169
                double xt[3];
                int occ = 0;
                double arc = input[0] * P.time_interval;
                double temp_ct = P.collision_interval / arc * P.time_interval;
                for (double t = temp_ct; t <= P.time_interval + 1e-3; t += temp_ct) {             // :175-183
                    stateTransit(cur_state, xt, input, t);
                    occ = isOccupancyXY(xt);
                    if (occ == 1) break;
                }
                if (occ == 1) continue;
                double tmp_g_score = 0.0, tmp_f_score = 0.0;                                      // :187-195
                tmp_g_score += P.weight_r2 * arc;
                tmp_g_score += P.weight_so2 * std::fabs(input[1]) * arc;
                tmp_g_score += P.weight_v_change * std::fabs(input[0] - pool[cur].input[0]);
                tmp_g_score += P.weight_delta_change * std::fabs(input[1] - pool[cur].input[1]);
                tmp_g_score += P.weight_sigma * getTerrainSig(pro_state);
                tmp_g_score += pool[cur].g_score;
                tmp_f_score = tmp_g_score + P.lambda_heu * getHeu(pro_state, end_pt);
                if (pro_node < 0) {                                                               // :197-217
                    if (!ps) { R.status = 6; return R; }                                          // a key the dense table cannot hold: never seen; reported, not guessed
                    pro_node = use_node_num;
                    KinoNode& nd = pool[pro_node];
                    for (int k = 0; k < 3; k++) { nd.index[k] = pro_id[k]; nd.state[k] = pro_state[k]; }
                    nd.f_score = tmp_f_score; nd.g_score = tmp_g_score;
                    nd.input[0] = input[0]; nd.input[1] = input[1];
                    nd.parent = cur;
                    nd.node_state = 'b';
                    pushHeap(pro_node);
                    *ps = pro_node;
                    use_node_num++;
                    if (use_node_num == allocate_num) { R.status = 4; R.iter_num = iter_num; R.use_node_num = use_node_num; return R; }
                } else if (pool[pro_node].node_state == 'b') {                                    // :218-229
                    if (tmp_g_score < pool[pro_node].g_score) {
                        KinoNode& nd = pool[pro_node];
                        for (int k = 0; k < 3; k++) { nd.index[k] = pro_id[k]; nd.state[k] = pro_state[k]; }
                        nd.f_score = tmp_f_score; nd.g_score = tmp_g_score;
                        nd.input[0] = input[0]; nd.input[1] = input[1];
                        nd.parent = cur;
                    }
                }
            }
        }
        R.status = 3; R.iter_num = iter_num; R.use_node_num = use_node_num;                      // :233
        return R;
    }
};

}  // namespace orc
