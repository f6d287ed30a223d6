// libunevenhip.so -- initial-guess producer of the optimiser boundary, batched over paths.
//
// Reference: PlanManager::rcvWpsCallBack after the front-end returns, plan_manager/src/plan_manager.cpp:62-132 -- yaw unwrapping
// (:62-78), boundary states with init_sig_vel along the end headings (:87-95), arc-length insertion of the position / yaw way-points
// by running `while` loops over the path segments (:97-121), total time (:122).  The output arrays are the argument list of
// ALMTrajOpt::optimizeSE2Traj (alm_traj_opt.h:92-98) in the layout uph_problem takes.  Host code: a path has a few hundred poses and
// the walk is sequential; what matters is that a batch of B front-end results becomes the packed arrays of one upload without a
// per-problem round trip through the caller's language.
//
// mp->test_mode: the same stage as the back-end's own test node runs it, ALMTrajOpt::rcvWpsCallBack back_end/src/alm_traj_opt.cpp:73-144 --
// literals instead of the manager parameters (:106-107, 117-118, 137), `if` instead of `while` in both combs (:122, 128: at most one node
// per comb and segment, the remainder carries over), and the position comb also appends its node's interpolated yaw to the yaw way-points
// (:132), after the yaw comb's node of the same segment.
#include <cmath>
#include <cstdint>
#include <string>

#include "../../include/uneven_hip.h"

namespace uph {
void setError(const std::string& s);      // unevenhip.hip
}

namespace {

struct Walker {          // one arc-length comb: emits a node every `pitch` metres of accumulated path length
    double pitch, carried = 0.0;
    // advances by one segment of length seg; calls emit(fraction along the segment) for every tooth passed
    template <class F> void advance(double seg, F emit) {
        carried += seg;
        while (carried > pitch) {                       // plan_manager.cpp:107, :113
            emit(1.0 - (carried - pitch) / seg);
            carried -= pitch;
        }
    }
    // the test node's form: one tooth per segment at most (alm_traj_opt.cpp:122, 128); what is left stays carried
    template <class F> void advanceOnce(double seg, F emit) {
        carried += seg;
        if (carried > pitch) {
            emit(1.0 - (carried - pitch) / seg);
            carried -= pitch;
        }
    }
};

}  // namespace

extern "C" int uph_resample_batch(const uph_manager_params* mp, int32_t B, const double* paths, const int64_t* offsets, int32_t cap_xy, int32_t cap_yaw,
                                  double* init_xy, double* end_xy, double* init_yaw, double* end_yaw, double* inner_xy, double* inner_yaw,
                                  int32_t* n_inner_xy, int32_t* n_inner_yaw, double* total_time, double* unwrapped) {
    if (!mp || B <= 0 || !paths || !offsets || !init_xy || !end_xy || !init_yaw || !end_yaw || !inner_xy || !inner_yaw || !n_inner_xy || !n_inner_yaw || !total_time ||
        cap_xy < 0 || cap_yaw < 0 || (!mp->test_mode && (!(mp->piece_len > 0.0) || !(mp->yaw_piece_times > 0.0) || !(mp->mean_vel > 0.0))) ||
        (mp->test_mode && !(mp->test_max_vel > 0.0))) {
        uph::setError("uph_resample_batch: bad arguments");
        return UPH_ERR_INVALID;
    }
    const double PI = 3.14159265358979323846;      // M_PI
    int status = UPH_OK;
    for (int32_t b = 0; b < B; b++) {
        const int64_t o = offsets[b], m = offsets[b + 1] - offsets[b];
        if (m < 2) { uph::setError("uph_resample_batch: a path needs at least two poses"); return UPH_ERR_INVALID; }
        const double* p = paths + 3 * o;
        double* ixy = init_xy + 6 * (size_t)b; double* exy = end_xy + 6 * (size_t)b;
        double* iyw = init_yaw + 3 * (size_t)b; double* eyw = end_yaw + 3 * (size_t)b;
        double* oxy = inner_xy + 2 * (size_t)cap_xy * b; double* oyw = inner_yaw + (size_t)cap_yaw * b;
        const bool tm = mp->test_mode != 0;
        const double piece_len = tm ? 0.3 : mp->piece_len;                                        // alm_traj_opt.cpp:117
        const double sig_vel = tm ? 0.05 : mp->init_sig_vel;                                      // alm_traj_opt.cpp:106-107
        Walker pos{piece_len}, yaw{tm ? piece_len / 2.0 : piece_len / mp->yaw_piece_times};      // plan_manager.cpp:100 / alm_traj_opt.cpp:118
        int32_t nxy = 0, nyw = 0;
        double len = 0.0;
        // the unwrapped yaw of pose i+1 depends on the unwrapped yaw of pose i (:62-78); carried along the walk instead of a first pass
        double ya = p[2];
        if (unwrapped) unwrapped[o] = ya;
        const double y_first = ya;
        for (int64_t k = 0; k + 1 < m; k++) {
            double yb = p[3 * (k + 1) + 2];
            while (yb - ya >= PI / 2) yb -= PI * 2;
            while (yb - ya <= -PI / 2) yb += PI * 2;
            if (unwrapped) unwrapped[o + k + 1] = yb;
            const double ax = p[3 * k], ay = p[3 * k + 1];
            const double dx = p[3 * (k + 1)] - ax, dy = p[3 * (k + 1) + 1] - ay, dw = yb - ya;
            const double seg = std::sqrt(dx * dx + dy * dy);                                      // .head(2).norm() (:103)
            len += seg;
            if (!tm) {
                yaw.advance(seg, [&](double t) { if (nyw < cap_yaw) oyw[nyw] = ya + t * dw; nyw++; });                               // :109-110
                pos.advance(seg, [&](double t) { if (nxy < cap_xy) { oxy[2 * nxy] = ax + t * dx; oxy[2 * nxy + 1] = ay + t * dy; } nxy++; });   // :115-116
            } else {
                yaw.advanceOnce(seg, [&](double t) { if (nyw < cap_yaw) oyw[nyw] = ya + t * dw; nyw++; });                           // alm_traj_opt.cpp:122-127
                pos.advanceOnce(seg, [&](double t) {                                                                                 // :128-134
                    if (nxy < cap_xy) { oxy[2 * nxy] = ax + t * dx; oxy[2 * nxy + 1] = ay + t * dy; }
                    nxy++;
                    if (nyw < cap_yaw) oyw[nyw] = ya + t * dw;                                  // temp_node.z() joins the yaw way-points (:132)
                    nyw++;
                });
            }
            ya = yb;
        }
        ixy[0] = p[0]; ixy[1] = p[1]; exy[0] = p[3 * (m - 1)]; exy[1] = p[3 * (m - 1) + 1];      // :87-90, column-major 2x3 {P, V, A}
        iyw[0] = y_first; iyw[1] = 0.0; iyw[2] = 0.0; eyw[0] = ya; eyw[1] = 0.0; eyw[2] = 0.0;    // :91-92
        ixy[2] = sig_vel * std::cos(iyw[0]); ixy[3] = sig_vel * std::sin(iyw[0]);                 // :94
        exy[2] = sig_vel * std::cos(eyw[0]); exy[3] = sig_vel * std::sin(eyw[0]);                 // :95
        ixy[4] = ixy[5] = exy[4] = exy[5] = 0.0;
        total_time[b] = tm ? len / mp->test_max_vel * 1.2 : len / mp->mean_vel * mp->init_time_times;      // alm_traj_opt.cpp:137 / plan_manager.cpp:122
        n_inner_xy[b] = nxy; n_inner_yaw[b] = nyw;
        if (nxy > cap_xy || nyw > cap_yaw) status = UPH_ERR_LIMIT;         // counts are still reported, so the caller can size a second call
    }
    if (status == UPH_ERR_LIMIT) uph::setError("uph_resample_batch: a path produced more way-points than the caller's capacity");
    return status;
}
