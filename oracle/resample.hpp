// ORACLE (test infrastructure, never shipped): restatement of the initial-guess stage of PlanManager::rcvWpsCallBack,
// /root/reference/src/uneven_planner/plan_manager/src/plan_manager.cpp:62-132, in the reference's own order: a first pass that
// unwraps the yaw column in place (:62-78), then the boundary states (:87-95), then the segment walk that collects the way-points in
// vectors (:97-121), then the total time (:122).  PARITY UNPINNED: the reference holds no test or fixture for this stage and cannot be
// built here (ROS / Eigen / OMPL); tests/test_resample_cpu.py checks hand-derived known answers and that this restatement, the
// product's streaming C++ form (uph_resample_batch) and its numpy mirror agree bit for bit.
#pragma once
#include <cmath>
#include <vector>

namespace orc {

struct ManagerParams { double piece_len = 0.3, mean_vel = 0.5, init_time_times = 1.2, yaw_piece_times = 2.0, init_sig_vel = 0.05; };   // run_hill.yaml:57-62

struct Resampled {
    double init_xy[6], end_xy[6], init_yaw[3], end_yaw[3];      // 2x3 matrices column-major {P, V, A}
    std::vector<double> inner_xy;                               // x0, y0, x1, y1, ...
    std::vector<double> inner_yaw;
    std::vector<double> yaw_unwrapped;
    double total_time;
};

inline Resampled resamplePath(std::vector<double> path /* 3 per pose, by value: unwrapped in place */, const ManagerParams& mp) {
    const size_t M = path.size() / 3;
    Resampled r;
    // smooth yaw  :62-78
    double dyaw;
    for (size_t i = 0; i + 1 < M; i++) {
        dyaw = path[3 * (i + 1) + 2] - path[3 * i + 2];
        while (dyaw >= M_PI / 2) {
            path[3 * (i + 1) + 2] -= M_PI * 2;
            dyaw = path[3 * (i + 1) + 2] - path[3 * i + 2];
        }
        while (dyaw <= -M_PI / 2) {
            path[3 * (i + 1) + 2] += M_PI * 2;
            dyaw = path[3 * (i + 1) + 2] - path[3 * i + 2];
        }
    }
    r.yaw_unwrapped.resize(M);
    for (size_t i = 0; i < M; i++) r.yaw_unwrapped[i] = path[3 * i + 2];
    // init solution  :87-95
    for (int k = 0; k < 6; k++) { r.init_xy[k] = 0.0; r.end_xy[k] = 0.0; }
    r.init_xy[0] = path[0]; r.init_xy[1] = path[1];
    r.end_xy[0] = path[3 * (M - 1)]; r.end_xy[1] = path[3 * (M - 1) + 1];
    r.init_yaw[0] = path[2]; r.init_yaw[1] = 0.0; r.init_yaw[2] = 0.0;
    r.end_yaw[0] = path[3 * (M - 1) + 2]; r.end_yaw[1] = 0.0; r.end_yaw[2] = 0.0;
    r.init_xy[2] = mp.init_sig_vel * std::cos(r.init_yaw[0]); r.init_xy[3] = mp.init_sig_vel * std::sin(r.init_yaw[0]);
    r.end_xy[2] = mp.init_sig_vel * std::cos(r.end_yaw[0]); r.end_xy[3] = mp.init_sig_vel * std::sin(r.end_yaw[0]);
    // way-points  :97-121
    double temp_len_yaw = 0.0, temp_len_pos = 0.0, total_len = 0.0;
    const double piece_len_yaw = mp.piece_len / mp.yaw_piece_times;
    for (size_t k = 0; k + 1 < M; k++) {
        const double d0 = path[3 * (k + 1)] - path[3 * k], d1 = path[3 * (k + 1) + 1] - path[3 * k + 1], d2 = path[3 * (k + 1) + 2] - path[3 * k + 2];
        const double temp_seg = std::sqrt(d0 * d0 + d1 * d1);
        temp_len_yaw += temp_seg;
        temp_len_pos += temp_seg;
        total_len += temp_seg;
        while (temp_len_yaw > piece_len_yaw) {
            const double temp_yaw = path[3 * k + 2] + (1.0 - (temp_len_yaw - piece_len_yaw) / temp_seg) * d2;
            r.inner_yaw.push_back(temp_yaw);
            temp_len_yaw -= piece_len_yaw;
        }
        while (temp_len_pos > mp.piece_len) {
            const double w = 1.0 - (temp_len_pos - mp.piece_len) / temp_seg;
            r.inner_xy.push_back(path[3 * k] + w * d0);
            r.inner_xy.push_back(path[3 * k + 1] + w * d1);
            temp_len_pos -= mp.piece_len;
        }
    }
    r.total_time = total_len / mp.mean_vel * mp.init_time_times;      // :122
    return r;
}

// The back-end's stand-alone test node: ALMTrajOpt::rcvWpsCallBack, /root/reference/src/uneven_planner/back_end/src/alm_traj_opt.cpp:73-144,
// statement by statement: the same yaw smoothing (:73-88), boundary states with the literal 0.05 (:101-107), the walk with piece_len = 0.3,
// piece_len_yaw = piece_len / 2.0 (:117-118), `if` in both combs (:122, 128) and temp_node.z() appended to the yaw nodes (:132), total time
// from the optimiser's max_vel and the literal 1.2 (:137).
inline Resampled resamplePathTest(std::vector<double> path, double max_vel) {
    const size_t M = path.size() / 3;
    Resampled r;
    double dyaw;
    for (size_t i = 0; i + 1 < M; i++) {
        dyaw = path[3 * (i + 1) + 2] - path[3 * i + 2];
        while (dyaw >= M_PI / 2) {
            path[3 * (i + 1) + 2] -= M_PI * 2;
            dyaw = path[3 * (i + 1) + 2] - path[3 * i + 2];
        }
        while (dyaw <= -M_PI / 2) {
            path[3 * (i + 1) + 2] += M_PI * 2;
            dyaw = path[3 * (i + 1) + 2] - path[3 * i + 2];
        }
    }
    r.yaw_unwrapped.resize(M);
    for (size_t i = 0; i < M; i++) r.yaw_unwrapped[i] = path[3 * i + 2];
    for (int k = 0; k < 6; k++) { r.init_xy[k] = 0.0; r.end_xy[k] = 0.0; }
    r.init_xy[0] = path[0]; r.init_xy[1] = path[1];
    r.end_xy[0] = path[3 * (M - 1)]; r.end_xy[1] = path[3 * (M - 1) + 1];
    r.init_yaw[0] = path[2]; r.init_yaw[1] = 0.0; r.init_yaw[2] = 0.0;
    r.end_yaw[0] = path[3 * (M - 1) + 2]; r.end_yaw[1] = 0.0; r.end_yaw[2] = 0.0;
    r.init_xy[2] = 0.05 * std::cos(r.init_yaw[0]); r.init_xy[3] = 0.05 * std::sin(r.init_yaw[0]);
    r.end_xy[2] = 0.05 * std::cos(r.end_yaw[0]); r.end_xy[3] = 0.05 * std::sin(r.end_yaw[0]);
    double temp_len_yaw = 0.0, temp_len_pos = 0.0, total_len = 0.0;
    const double piece_len = 0.3;
    const double piece_len_yaw = piece_len / 2.0;
    for (size_t k = 0; k + 1 < M; k++) {
        const double d0 = path[3 * (k + 1)] - path[3 * k], d1 = path[3 * (k + 1) + 1] - path[3 * k + 1], d2 = path[3 * (k + 1) + 2] - path[3 * k + 2];
        const double temp_seg = std::sqrt(d0 * d0 + d1 * d1);
        temp_len_yaw += temp_seg;
        temp_len_pos += temp_seg;
        total_len += temp_seg;
        if (temp_len_yaw > piece_len_yaw) {
            const double temp_yaw = path[3 * k + 2] + (1.0 - (temp_len_yaw - piece_len_yaw) / temp_seg) * d2;
            r.inner_yaw.push_back(temp_yaw);
            temp_len_yaw -= piece_len_yaw;
        }
        if (temp_len_pos > piece_len) {
            const double w = 1.0 - (temp_len_pos - piece_len) / temp_seg;
            r.inner_xy.push_back(path[3 * k] + w * d0);
            r.inner_xy.push_back(path[3 * k + 1] + w * d1);
            r.inner_yaw.push_back(path[3 * k + 2] + w * d2);
            temp_len_pos -= piece_len;
        }
    }
    r.total_time = total_len / max_vel * 1.2;
    return r;
}

}  // namespace orc
