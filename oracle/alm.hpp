// ORACLE -- TEST INFRASTRUCTURE ONLY (see banded.hpp header).  PARITY UNPINNED.
//
// Restates back_end/src/alm_traj_opt.cpp and back_end/include/back_end/alm_traj_opt.h:
//   optimizeSE2Traj      cpp:168-278   set-up + PHR augmented-Lagrangian outer loop
//   innerCallback        cpp:280-347   L-BFGS objective f(x), grad f(x)
//   initScaling          cpp:349-661   objective / per-constraint gradient scaling at x0
//   calConstrainCostGrad cpp:663-991   constraint sampling, penalty + gradient accumulation
//   earlyExit            cpp:993-1017  progress callback: cancel when k > 1e3
//   updateDualVars / judgeConvergence / getAugmentedCost/Grad   h:132-163
//   expC2 / logC2 / getTtoTauGrad / calTfromTau                 h:232-261
//   getMaxVxAxAyCurAttSig h:170-229, SE2Trajectory::getNonHolError se2traj.hpp:551-561 (post-solve report)
// Quirks Q1-Q9 of SURVEY.md 8(a) are kept literally (see comments tagged Qn).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include "lbfgs.hpp"
#include "minco.hpp"
#include "terrain.hpp"

namespace orc {

constexpr double delta_sigl = 0.01;          // alm_traj_opt.h:16-19
constexpr double cur_scale = 10.0;
constexpr double sig_scale = 1000.0;
constexpr double scale_trick_jerk = 1000.0;

struct AlmParams {                           // alm_traj_opt.h:29-53, values = plan_manager/params/run_hill.yaml:32-55
    double rho_T = 100000.0, rho_ter = 10.0, max_vel = 0.5, max_acc_lon = 5.0, max_acc_lat = 10.0;
    double max_kap = 2.1, min_cxi = 0.8, max_sig = 0.05;
    bool use_scaling = true;
    double rho = 1.0, beta = 1000.0, gamma = 1.0, epsilon_con = 0.001, max_iter = 10.0;
    double g_epsilon = 1.0e-3, min_step = 1.0e-32, inner_max_iter = 10000.0, delta = 1.0e-4;
    int mem_size = 256, past = 3, int_K = 16;
};

struct AlmStats { int alm_iters = 0, lbfgs_iters = 0, evals = 0, last_lbfgs_ret = 0; double inner_cost = 0; };

struct AlmTrajOpt {
    AlmParams p;
    double rho;                               // member; NOT reset between solves (Q7)
    int piece_xy = 0, piece_yaw = 0, dim_T = 1;
    double equal_num = 0, non_equal_num = 0, scale_fx = 1.0;
    Vec lambda, mu, hx, gx, scale_cx;
    double init_xy[6], end_xy[6], init_yaw[3], end_yaw[3];   // xy: column-major 2x3 (Eigen MatrixXd)
    MincoSE2 minco;
    const Grid* map = nullptr;
    AlmStats stats;
    std::vector<double> trace;     // (test aid) fx after every accepted L-BFGS iteration, -1 marks an ALM pass boundary
    bool flat_debug = false;                  // the commented "debug" block cpp:787-803 (flat terrain)
    // ---- TEST AIDS for the teacher-forced late-state tests (no effect on the algorithm)
    struct PassRec {                          // one ALM pass: what went in, what came out
        Vec x_in, lambda_in, mu_in, x_out, lambda_out, mu_out, hx, gx;
        double rho_in = 0, rho_out = 0, cost = 0;
        int ret = 0, k = 0, converged = 0;
    };
    bool record_passes = false;
    std::vector<PassRec> passes;
    std::vector<LbfgsIterLog> iter_log;       // every completed L-BFGS iteration of the last solve ...
    std::vector<int> pass_log_start;          // ... and where each ALM pass starts in it
    int snap_pass = -1, snap_k = -1;          // capture the L-BFGS state at the top of iteration snap_k of ALM pass snap_pass
    bool snap_valid = false;
    LbfgsState snap_state;

    explicit AlmTrajOpt(const AlmParams& pp = AlmParams()) : p(pp), rho(pp.rho) {}

    // ---- alm_traj_opt.h:232-261
    static double expC2(double tau) { return tau > 0.0 ? ((0.5 * tau + 1.0) * tau + 1.0) : 1.0 / ((0.5 * tau - 1.0) * tau + 1.0); }
    static double logC2(double T) { return T > 1.0 ? (std::sqrt(2.0 * T - 1.0) - 1.0) : (1.0 - std::sqrt(2.0 / T - 1.0)); }
    static double getTtoTauGrad(double tau) {
        if (tau > 0) return tau + 1.0;
        double denSqrt = (0.5 * tau - 1.0) * tau + 1.0;
        return (1.0 - tau) / (denSqrt * denSqrt);
    }
    static void calTfromTau(double tau, Vec& T) { std::fill(T.begin(), T.end(), expC2(tau) / (double)T.size()); }
    // ---- alm_traj_opt.h:153-163
    double getAugmentedCost(double h_or_g, double lambda_or_mu) const { return h_or_g * (lambda_or_mu + 0.5 * rho * h_or_g); }
    double getAugmentedGrad(double h_or_g, double lambda_or_mu) const { return rho * h_or_g + lambda_or_mu; }

    void updateDualVars() {                   // h:132-138
        for (size_t i = 0; i < lambda.size(); i++) lambda[i] += rho * hx[i];
        for (int i = 0; i < non_equal_num; i++) mu[i] = std::max(mu[i] + rho * gx[i], 0.0);
        rho = std::min((1 + p.gamma) * rho, p.beta);
    }
    bool judgeConvergence(double* res_h = nullptr, double* res_g = nullptr) const {   // h:140-151 (Q5: sees updated mu, rho)
        double rh = 0, rg = 0;
        for (double v : hx) rh = std::max(rh, std::fabs(v));
        for (size_t i = 0; i < gx.size(); i++) rg = std::max(rg, std::fabs(std::max(gx[i], -mu[i] / rho)));
        if (res_h) *res_h = rh;
        if (res_g) *res_g = rg;
        return std::max(rh, rg) < p.epsilon_con;
    }

    void generateFromX(const Vec& x) {
        const double tau = x[0];
        Vec Txy(piece_xy), Tyaw(piece_yaw);
        calTfromTau(tau, Txy);
        calTfromTau(tau, Tyaw);
        double hxy[6], txy[6];
        for (int d = 0; d < 2; d++)
            for (int k = 0; k < 3; k++) { hxy[d * 3 + k] = init_xy[k * 2 + d]; txy[d * 3 + k] = end_xy[k * 2 + d]; }
        minco.pos.generate(x.data() + dim_T, Txy.data(), hxy, txy);
        minco.yaw.generate(x.data() + dim_T + 2 * (piece_xy - 1), Tyaw.data(), init_yaw, end_yaw);
    }

    // ---- innerCallback cpp:280-347
    double innerCallback(const Vec& x, Vec& grad) {
        const double tau = x[0];
        generateFromX(x);                                                         // :293-299
        Vec gdCxy_jerk, gdTxy_jerk, gdCyaw_jerk, gdTyaw_jerk;
        minco.pos.calJerkGradCT(gdCxy_jerk, gdTxy_jerk);                          // :307
        minco.yaw.calJerkGradCT(gdCyaw_jerk, gdTyaw_jerk);
        double jerk_cost = minco.getTrajJerkCost() * scale_fx;                    // :308-310
        if (p.use_scaling) jerk_cost *= scale_trick_jerk;
        double constrain_cost = 0.0;
        Vec gdCxy_c, gdTxy_c, gdCyaw_c, gdTyaw_c;
        calConstrainCostGrad(constrain_cost, gdCxy_c, gdTxy_c, gdCyaw_c, gdTyaw_c);   // :318
        if (p.use_scaling) {                                                      // :322-328
            for (double& v : gdCxy_jerk) v *= scale_trick_jerk;
            for (double& v : gdTxy_jerk) v *= scale_trick_jerk;
            for (double& v : gdCyaw_jerk) v *= scale_trick_jerk;
            for (double& v : gdTyaw_jerk) v *= scale_trick_jerk;
        }
        Vec gdCxy(gdCxy_c.size()), gdTxy(gdTxy_c.size()), gdCyaw(gdCyaw_c.size()), gdTyaw(gdTyaw_c.size());   // :329-332
        for (size_t i = 0; i < gdCxy.size(); i++) gdCxy[i] = gdCxy_jerk[i] * scale_fx + gdCxy_c[i];
        for (size_t i = 0; i < gdTxy.size(); i++) gdTxy[i] = gdTxy_jerk[i] * scale_fx + gdTxy_c[i];
        for (size_t i = 0; i < gdCyaw.size(); i++) gdCyaw[i] = gdCyaw_jerk[i] * scale_fx + gdCyaw_c[i];
        for (size_t i = 0; i < gdTyaw.size(); i++) gdTyaw[i] = gdTyaw_jerk[i] * scale_fx + gdTyaw_c[i];
        Vec gPxy, gPyaw;
        minco.pos.calGradCTtoQT(gdCxy, gdTxy, gPxy);                              // :335
        minco.yaw.calGradCTtoQT(gdCyaw, gdTyaw, gPyaw);
        for (size_t i = 0; i < gPxy.size(); i++) grad[dim_T + i] = gPxy[i];       // :336-337
        for (size_t i = 0; i < gPyaw.size(); i++) grad[dim_T + 2 * (piece_xy - 1) + i] = gPyaw[i];
        double tau_cost = p.rho_T * expC2(tau) * scale_fx;                        // :340-344
        double sx = 0, sy = 0;
#if ORACLE_EIGEN_REDUX
        sx = eigen_redux_linear((int)gdTxy.size(), [&](int i) { return gdTxy[i]; });      // gdTxy.sum(), alm_traj_opt.cpp:342-343
        sy = eigen_redux_linear((int)gdTyaw.size(), [&](int i) { return gdTyaw[i]; });
#else
        for (double v : gdTxy) sx += v;
        for (double v : gdTyaw) sy += v;
#endif
        double grad_Tsum = p.rho_T * scale_fx + sx / piece_xy + sy / piece_yaw;
        grad[0] = grad_Tsum * getTtoTauGrad(tau);
        last_jerk_cost_term = jerk_cost; last_constrain_cost = constrain_cost; last_tau_cost = tau_cost;
        return jerk_cost + constrain_cost + tau_cost;                             // :346
    }
    double last_jerk_cost_term = 0, last_constrain_cost = 0, last_tau_cost = 0;

    // per-sample kinematics + terrain shared by calConstrainCostGrad and initScaling (cpp:733-817 == cpp:439-505)
    struct Sample {
        double beta0_xy[6], beta1_xy[6], beta2_xy[6], beta3_xy[6];
        double beta0_yaw[6], beta1_yaw[6], beta2_yaw[6];
        double pos[2], vel[2], acc[2], jer[2];
        double yaw, dyaw, d2yaw, cyaw, syaw, v_norm, xb[2], yb[2], lon_acc, lat_acc;
        double tv[7], tg[7][3];
        double vx, wz, ax, ay, curv_snorm;
        int yaw_idx;
    };
    void sample(int i, double s1, double base_time, double gravity, Sample& S) const {
        const double* cxy = &minco.pos.c[(size_t)i * 6 * 2];
        double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;           // :734-741
        double b0[6] = {1.0, s1, s2, s3, s4, s5};
        double b1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
        double b2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
        double b3[6] = {0.0, 0.0, 0.0, 6.0, 24.0 * s1, 60.0 * s2};
        for (int k = 0; k < 6; k++) { S.beta0_xy[k] = b0[k]; S.beta1_xy[k] = b1[k]; S.beta2_xy[k] = b2[k]; S.beta3_xy[k] = b3[k]; }
        for (int d = 0; d < 2; d++) {                                             // :742-745
            double a = 0, b = 0, c = 0, e = 0;
            for (int k = 0; k < 6; k++) { a += cxy[k * 2 + d] * b0[k]; b += cxy[k * 2 + d] * b1[k]; c += cxy[k * 2 + d] * b2[k]; e += cxy[k * 2 + d] * b3[k]; }
            S.pos[d] = a; S.vel[d] = b; S.acc[d] = c; S.jer[d] = e;
        }
        double now_time = s1 + base_time;                                         // :748-753 (Q2: indexed with the xy piece index)
        const double Tyaw_i = minco.yaw.T1[i];
        int yaw_idx = int((now_time) / Tyaw_i);
        if (yaw_idx >= piece_yaw) yaw_idx = piece_yaw - 1;
        S.yaw_idx = yaw_idx;
        const double* cy = &minco.yaw.c[(size_t)yaw_idx * 6];
        double u1 = now_time - yaw_idx * Tyaw_i;
        double u2 = u1 * u1, u3 = u2 * u1, u4 = u2 * u2, u5 = u4 * u1;
        double y0[6] = {1.0, u1, u2, u3, u4, u5};
        double y1[6] = {0.0, 1.0, 2.0 * u1, 3.0 * u2, 4.0 * u3, 5.0 * u4};
        double y2[6] = {0.0, 0.0, 2.0, 6.0 * u1, 12.0 * u2, 20.0 * u3};
        for (int k = 0; k < 6; k++) { S.beta0_yaw[k] = y0[k]; S.beta1_yaw[k] = y1[k]; S.beta2_yaw[k] = y2[k]; }
        double yaw = 0, dyaw = 0, d2yaw = 0;                                      // :762-764
        for (int k = 0; k < 6; k++) { yaw += cy[k] * y0[k]; dyaw += cy[k] * y1[k]; d2yaw += cy[k] * y2[k]; }
        S.yaw = yaw; S.dyaw = dyaw; S.d2yaw = d2yaw;
        double se2[3] = {S.pos[0], S.pos[1], yaw};                                // :767-775
        normSO2(se2[2]);
        S.syaw = std::sin(yaw);
        S.cyaw = std::cos(yaw);
        S.v_norm = std::sqrt(S.vel[0] * S.vel[0] + S.vel[1] * S.vel[1]);
        S.xb[0] = S.cyaw; S.xb[1] = S.syaw;
        S.yb[0] = -S.syaw; S.yb[1] = S.cyaw;
        S.lon_acc = S.acc[0] * S.xb[0] + S.acc[1] * S.xb[1];
        S.lat_acc = S.acc[0] * S.yb[0] + S.acc[1] * S.yb[1];
        map->getAllWithGrad(se2, S.tv, S.tg);                                     // :778
        if (flat_debug) {                                                         // :787-803
            S.tv[0] = 1.0; S.tv[1] = 0.0; S.tv[2] = 1.0; S.tv[3] = 0.0; S.tv[4] = 1.0; S.tv[5] = 1.0; S.tv[6] = 0.0;
            for (int a = 0; a < 7; a++) for (int b = 0; b < 3; b++) S.tg[a][b] = 0.0;
        }
        S.vx = S.v_norm * S.tv[0];                                                // :813-817
        S.wz = dyaw * S.tv[5];
        S.ax = S.lon_acc * S.tv[0] + gravity * S.tv[1];
        S.ay = S.lat_acc * S.tv[2] + gravity * S.tv[3];
        S.curv_snorm = S.wz * S.wz / (S.vx * S.vx + delta_sigl);
    }

    // ---- calConstrainCostGrad cpp:663-991
    void calConstrainCostGrad(double& cost, Vec& gdCxy, Vec& gdTxy, Vec& gdCyaw, Vec& gdTyaw) {
        cost = 0.0;
        gdCxy.assign((size_t)6 * piece_xy * 2, 0.0);
        gdTxy.assign(piece_xy, 0.0);
        gdCyaw.assign((size_t)6 * piece_yaw, 0.0);
        gdTyaw.assign(piece_yaw, 0.0);
        const double gravity = map->gravity;
        const int int_K = p.int_K;
        const double max_vel = p.max_vel, max_acc_lon = p.max_acc_lon, max_acc_lat = p.max_acc_lat, max_kap = p.max_kap;
        int equal_idx = 0, non_equal_idx = 0, constrain_idx = 0;
        double base_time = 0.0;
        Sample S;
        for (int i = 0; i < piece_xy; i++) {
            double step = minco.pos.T1[i] / int_K;                                // :713
            double s1 = 0.0;
            for (int j = 0; j <= int_K; j++) {
                double alpha = 1.0 / int_K * j;                                   // :718
                double grad_p[2] = {0, 0}, grad_v[2] = {0, 0}, grad_a[2] = {0, 0};
                double grad_yaw = 0.0, grad_dyaw = 0.0, grad_d2yaw = 0.0, grad_vx2 = 0.0, grad_wz = 0.0, grad_ax = 0.0, grad_ay = 0.0;
                double grad_se2[3] = {0, 0, 0};
                double aug_grad = 0.0;
                sample(i, s1, base_time, gravity, S);
                const double inv_cos_vphix = S.tv[0], inv_cos_vphiy = S.tv[2], cos_xi = S.tv[4], inv_cos_xi = S.tv[5], sigma = S.tv[6];
                const double* g_icvx = S.tg[0]; const double* g_spx = S.tg[1]; const double* g_icvy = S.tg[2];
                const double* g_spy = S.tg[3]; const double* g_cxi = S.tg[4]; const double* g_icxi = S.tg[5]; const double* g_sig = S.tg[6];
                const double vx = S.vx, wz = S.wz, ax = S.ax, ay = S.ay, curv_snorm = S.curv_snorm;

                // user-defined cost: surface variation                           :819-827
                double omega;
                if (j == 0 || j == int_K) omega = 0.5 * p.rho_ter * step * scale_fx;
                else omega = p.rho_ter * step * scale_fx;
                double user_cost = omega * sigma * sigma;
                cost += user_cost;
                for (int k = 0; k < 3; k++) grad_se2[k] += omega * g_sig[k] * sigma * 2.0;
                gdTxy[i] += user_cost / int_K;                                    // Q3

                // non-holonomic                                                  :829-838
                double nonh_lambda = lambda[equal_idx];
                double nhy[2] = {S.syaw, -S.cyaw};
                hx[equal_idx] = (S.vel[0] * nhy[0] + S.vel[1] * nhy[1]) * scale_cx[constrain_idx];
                cost += getAugmentedCost(hx[equal_idx], nonh_lambda);
                double nonh_grad = getAugmentedGrad(hx[equal_idx], nonh_lambda) * scale_cx[constrain_idx];
                grad_v[0] += nonh_grad * nhy[0]; grad_v[1] += nonh_grad * nhy[1];
                grad_yaw += nonh_grad * (S.vel[0] * S.xb[0] + S.vel[1] * S.xb[1]);
                equal_idx++; constrain_idx++;

                // longitude velocity                                             :840-854
                double v_mu = mu[non_equal_idx];
                gx[non_equal_idx] = (vx * vx - max_vel * max_vel) * scale_cx[constrain_idx];
                if (rho * gx[non_equal_idx] + v_mu > 0) {
                    cost += getAugmentedCost(gx[non_equal_idx], v_mu);
                    aug_grad = getAugmentedGrad(gx[non_equal_idx], v_mu) * scale_cx[constrain_idx];
                    grad_vx2 += aug_grad;
                } else {
                    cost += -0.5 * v_mu * v_mu / rho;
                }
                non_equal_idx++; constrain_idx++;

                // longitude acceleration                                         :856-870
                double lona_mu = mu[non_equal_idx];
                gx[non_equal_idx] = (ax * ax - max_acc_lon * max_acc_lon) * scale_cx[constrain_idx];
                if (rho * gx[non_equal_idx] + lona_mu > 0) {
                    cost += getAugmentedCost(gx[non_equal_idx], lona_mu);
                    aug_grad = getAugmentedGrad(gx[non_equal_idx], lona_mu) * scale_cx[constrain_idx];
                    grad_ax += aug_grad * 2.0 * ax;
                } else {
                    cost += -0.5 * lona_mu * lona_mu / rho;
                }
                non_equal_idx++; constrain_idx++;

                // latitude acceleration                                          :872-886
                double lata_mu = mu[non_equal_idx];
                gx[non_equal_idx] = (ay * ay - max_acc_lat * max_acc_lat) * scale_cx[constrain_idx];
                if (rho * gx[non_equal_idx] + lata_mu > 0) {
                    cost += getAugmentedCost(gx[non_equal_idx], lata_mu);
                    aug_grad = getAugmentedGrad(gx[non_equal_idx], lata_mu) * scale_cx[constrain_idx];
                    grad_ay += aug_grad * 2.0 * ay;
                } else {
                    cost += -0.5 * lata_mu * lata_mu / rho;
                }
                non_equal_idx++; constrain_idx++;

                // curvature                                                      :888-910 (Q6)
                double curv_mu = mu[non_equal_idx];
                if (p.use_scaling) gx[non_equal_idx] = (curv_snorm - max_kap * max_kap) * scale_cx[constrain_idx];
                else gx[non_equal_idx] = (curv_snorm - max_kap * max_kap) * cur_scale;
                if (rho * gx[non_equal_idx] + curv_mu > 0) {
                    double denominator = 1.0 / (vx * vx + delta_sigl);
                    cost += getAugmentedCost(gx[non_equal_idx], curv_mu);
                    if (p.use_scaling) aug_grad = getAugmentedGrad(gx[non_equal_idx], curv_mu) * scale_cx[constrain_idx];
                    else aug_grad = getAugmentedGrad(gx[non_equal_idx], curv_mu) * cur_scale;
                    grad_wz += aug_grad * denominator * 2.0 * wz;
                    grad_vx2 -= aug_grad * curv_snorm * denominator;
                } else {
                    cost += -0.5 * curv_mu * curv_mu / rho;
                }
                non_equal_idx++; constrain_idx++;

                // attitude                                                       :912-925
                double att_mu = mu[non_equal_idx];
                gx[non_equal_idx] = (p.min_cxi - cos_xi) * scale_cx[constrain_idx];
                if (rho * gx[non_equal_idx] + att_mu > 0) {
                    cost += getAugmentedCost(gx[non_equal_idx], att_mu);
                    double ag = getAugmentedGrad(gx[non_equal_idx], att_mu);
                    for (int k = 0; k < 3; k++) grad_se2[k] -= ag * g_cxi[k] * scale_cx[constrain_idx];
                } else {
                    cost += -0.5 * att_mu * att_mu / rho;
                }
                non_equal_idx++; constrain_idx++;

                // surface variation                                              :927-946 (Q6)
                double sig_mu = mu[non_equal_idx];
                if (p.use_scaling) gx[non_equal_idx] = (sigma - p.max_sig) * scale_cx[constrain_idx];
                else gx[non_equal_idx] = (sigma - p.max_sig) * sig_scale;
                if (rho * gx[non_equal_idx] + sig_mu > 0) {
                    cost += getAugmentedCost(gx[non_equal_idx], sig_mu);
                    double ag = getAugmentedGrad(gx[non_equal_idx], sig_mu);
                    if (p.use_scaling) for (int k = 0; k < 3; k++) grad_se2[k] += ag * g_sig[k] * scale_cx[constrain_idx];
                    else for (int k = 0; k < 3; k++) grad_se2[k] += ag * g_sig[k] * sig_scale;
                } else {
                    cost += -0.5 * sig_mu * sig_mu / rho;
                }
                non_equal_idx++; constrain_idx++;

                // process with vx, wz, ax                                        :948-964
                for (int d = 0; d < 2; d++) grad_v[d] += grad_vx2 * inv_cos_vphix * inv_cos_vphix * 2.0 * S.vel[d];
                for (int k = 0; k < 3; k++) grad_se2[k] += grad_vx2 * S.v_norm * S.v_norm * 2.0 * inv_cos_vphix * g_icvx[k];
                grad_dyaw += grad_wz * inv_cos_xi;
                for (int k = 0; k < 3; k++) grad_se2[k] += grad_wz * S.dyaw * g_icxi[k];
                for (int d = 0; d < 2; d++) grad_a[d] += grad_ax * inv_cos_vphix * S.xb[d];
                grad_yaw += grad_ax * inv_cos_vphix * S.lat_acc;
                for (int k = 0; k < 3; k++) grad_se2[k] += grad_ax * (gravity * g_spx[k] + g_icvx[k] * S.lon_acc);
                for (int d = 0; d < 2; d++) grad_a[d] += grad_ay * inv_cos_vphiy * S.yb[d];
                grad_yaw -= grad_ay * inv_cos_vphiy * S.lon_acc;
                for (int k = 0; k < 3; k++) grad_se2[k] += grad_ay * (gravity * g_spy[k] + g_icvy[k] * S.lat_acc);
                grad_p[0] += grad_se2[0]; grad_p[1] += grad_se2[1];
                grad_yaw += grad_se2[2];

                // add all grad into C,T                                          :966-985
                for (int k = 0; k < 6; k++)
                    for (int d = 0; d < 2; d++)
                        gdCxy[(size_t)(i * 6 + k) * 2 + d] += (S.beta0_xy[k] * grad_p[d] + S.beta1_xy[k] * grad_v[d] + S.beta2_xy[k] * grad_a[d]);
                gdTxy[i] += ((grad_p[0] * S.vel[0] + grad_p[1] * S.vel[1]) + (grad_v[0] * S.acc[0] + grad_v[1] * S.acc[1]) +
                             (grad_a[0] * S.jer[0] + grad_a[1] * S.jer[1])) * alpha;
                for (int k = 0; k < 6; k++)
                    gdCyaw[(size_t)S.yaw_idx * 6 + k] += (S.beta0_yaw[k] * grad_yaw + S.beta1_yaw[k] * grad_dyaw + S.beta2_yaw[k] * grad_d2yaw);   // Q8
                gdTyaw[S.yaw_idx] += -(grad_yaw * S.dyaw + grad_dyaw * S.d2yaw) * S.yaw_idx;
                gdTxy[i] += (grad_yaw * S.dyaw + grad_dyaw * S.d2yaw) * (alpha + i);

                s1 += step;                                                       // :987 (Q2)
            }
            base_time += minco.pos.T1[i];                                         // :989
        }
    }

    // ---- initScaling cpp:349-661
    void initScaling(const Vec& x0) {
        const double tau = x0[0];
        generateFromX(x0);                                                        // :356-363
        Vec gdCxy_fx, gdTxy_fx, gdCyaw_fx, gdTyaw_fx;
        minco.pos.calJerkGradCT(gdCxy_fx, gdTxy_fx);                              // :370
        minco.yaw.calJerkGradCT(gdCyaw_fx, gdTyaw_fx);
        const int ncon = (int)(equal_num + non_equal_num);
        const double gravity = map->gravity;
        const int int_K = p.int_K;
        const size_t nCxy = (size_t)6 * piece_xy * 2, nCyaw = (size_t)6 * piece_yaw;
        // per-constraint gradient (sparse: one xy block, one yaw block, one gdTxy entry, one gdTyaw entry).
        // The reference allocates dense arrays per constraint (:372-383); they are zero outside these blocks.
        struct Con { int i, yaw_idx; double cxy[12]; double txy; double cyaw[6]; double tyaw; };
        std::vector<Con> cons(ncon);
        int constrain_idx = 0;
        double base_time = 0.0;
        Sample S;
        for (int i = 0; i < piece_xy; i++) {
            double step = minco.pos.T1[i] / int_K;
            double s1 = 0.0;
            for (int j = 0; j <= int_K; j++) {
                double alpha = 1.0 / int_K * j;
                sample(i, s1, base_time, gravity, S);
                const double inv_cos_vphix = S.tv[0], inv_cos_vphiy = S.tv[2], inv_cos_xi = S.tv[5], sigma = S.tv[6];
                const double* g_icvx = S.tg[0]; const double* g_spx = S.tg[1]; const double* g_icvy = S.tg[2];
                const double* g_spy = S.tg[3]; const double* g_cxi = S.tg[4]; const double* g_icxi = S.tg[5]; const double* g_sig = S.tg[6];
                const int yaw_idx = S.yaw_idx;
                double grad_p[2], grad_v[2], grad_a[2], grad_se2[3], grad_yaw, grad_dyaw;

                // user-defined cost: surface variation (objective part)          :507-519
                double omega;
                if (j == 0 || j == int_K) omega = 0.5 * p.rho_ter * step;
                else omega = p.rho_ter * step;
                double user_cost = omega * sigma * sigma;
                for (int k = 0; k < 3; k++) grad_se2[k] = omega * g_sig[k] * sigma * 2.0;
                gdTxy_fx[i] += user_cost / int_K;
                for (int k = 0; k < 6; k++)
                    for (int d = 0; d < 2; d++) gdCxy_fx[(size_t)(i * 6 + k) * 2 + d] += S.beta0_xy[k] * grad_se2[d];
                gdTxy_fx[i] += (grad_se2[0] * S.vel[0] + grad_se2[1] * S.vel[1]) * alpha;
                for (int k = 0; k < 6; k++) gdCyaw_fx[(size_t)yaw_idx * 6 + k] += (S.beta0_yaw[k] * grad_se2[2]);
                gdTyaw_fx[yaw_idx] += -(grad_se2[2] * S.dyaw) * yaw_idx;
                gdTxy_fx[i] += (grad_se2[2] * S.dyaw) * (alpha + i);

                auto put = [&](bool use_p, bool use_v, bool use_a, bool use_dyaw) {
                    Con& c = cons[constrain_idx];
                    c.i = i; c.yaw_idx = yaw_idx; c.txy = 0; c.tyaw = 0;
                    for (int k = 0; k < 6; k++)
                        for (int d = 0; d < 2; d++) {
                            double v = 0.0;
                            // operand order as written in the reference for each constraint (:524,:537,:553,:588,:604)
                            if (use_p && use_v) v = S.beta0_xy[k] * grad_p[d] + S.beta1_xy[k] * grad_v[d];
                            else if (use_p && use_a) v = S.beta0_xy[k] * grad_p[d] + S.beta2_xy[k] * grad_a[d];
                            else if (use_p) v = S.beta0_xy[k] * grad_p[d];
                            else if (use_v) v = S.beta1_xy[k] * grad_v[d];
                            c.cxy[k * 2 + d] = v;
                        }
                    if (use_p && use_v) c.txy += ((grad_p[0] * S.vel[0] + grad_p[1] * S.vel[1]) + (grad_v[0] * S.acc[0] + grad_v[1] * S.acc[1])) * alpha;
                    else if (use_p && use_a) c.txy += ((grad_p[0] * S.vel[0] + grad_p[1] * S.vel[1]) + (grad_a[0] * S.jer[0] + grad_a[1] * S.jer[1])) * alpha;
                    else if (use_p) c.txy += (grad_p[0] * S.vel[0] + grad_p[1] * S.vel[1]) * alpha;
                    else if (use_v) c.txy += (grad_v[0] * S.acc[0] + grad_v[1] * S.acc[1]) * alpha;
                    if (use_dyaw) {
                        for (int k = 0; k < 6; k++) c.cyaw[k] = (S.beta0_yaw[k] * grad_yaw + S.beta1_yaw[k] * grad_dyaw);
                        c.tyaw += -(grad_yaw * S.dyaw + grad_dyaw * S.d2yaw) * yaw_idx;
                        c.txy += (grad_yaw * S.dyaw + grad_dyaw * S.d2yaw) * (alpha + i);
                    } else {
                        for (int k = 0; k < 6; k++) c.cyaw[k] = S.beta0_yaw[k] * grad_yaw;
                        c.tyaw += -(grad_yaw * S.dyaw) * yaw_idx;
                        c.txy += (grad_yaw * S.dyaw) * (alpha + i);
                    }
                    constrain_idx++;
                };

                // non-holonomic                                                  :521-529
                grad_v[0] = S.syaw; grad_v[1] = -S.cyaw;
                grad_yaw = S.vel[0] * S.xb[0] + S.vel[1] * S.xb[1];
                put(false, true, false, false);

                // longitude velocity                                             :531-544
                double grad_vx2 = 1.0;
                for (int d = 0; d < 2; d++) grad_v[d] = grad_vx2 * inv_cos_vphix * inv_cos_vphix * 2.0 * S.vel[d];
                for (int k = 0; k < 3; k++) grad_se2[k] = grad_vx2 * S.v_norm * S.v_norm * 2.0 * inv_cos_vphix * g_icvx[k];
                grad_p[0] = grad_se2[0]; grad_p[1] = grad_se2[1];
                grad_yaw = grad_se2[2];
                put(true, true, false, false);

                // longitude acceleration                                         :546-560
                double grad_ax = 2.0 * S.ax;
                for (int d = 0; d < 2; d++) grad_a[d] = grad_ax * inv_cos_vphix * S.xb[d];
                grad_yaw = grad_ax * inv_cos_vphix * S.lat_acc;
                for (int k = 0; k < 3; k++) grad_se2[k] = grad_ax * (gravity * g_spx[k] + g_icvx[k] * S.lon_acc);
                grad_p[0] = grad_se2[0]; grad_p[1] = grad_se2[1];
                grad_yaw += grad_se2[2];
                put(true, false, true, false);

                // latitude acceleration                                          :562-576
                double grad_ay = 2.0 * S.ay;
                for (int d = 0; d < 2; d++) grad_a[d] = grad_ay * inv_cos_vphiy * S.yb[d];
                grad_yaw = -grad_ay * inv_cos_vphiy * S.lon_acc;
                for (int k = 0; k < 3; k++) grad_se2[k] = grad_ay * (gravity * g_spy[k] + g_icvy[k] * S.lat_acc);
                grad_p[0] = grad_se2[0]; grad_p[1] = grad_se2[1];
                grad_yaw += grad_se2[2];
                put(true, false, true, false);

                // curvature                                                      :578-598
                double denominator = 1.0 / (S.vx * S.vx + delta_sigl);
                double grad_wz = denominator * 2.0 * S.wz;
                grad_vx2 = -S.curv_snorm * denominator;
                grad_dyaw = grad_wz * inv_cos_xi;
                for (int k = 0; k < 3; k++) grad_se2[k] = grad_wz * S.dyaw * g_icxi[k];
                for (int d = 0; d < 2; d++) grad_v[d] = grad_vx2 * inv_cos_vphix * inv_cos_vphix * 2.0 * S.vel[d];
                for (int k = 0; k < 3; k++) grad_se2[k] += grad_vx2 * S.v_norm * S.v_norm * 2.0 * inv_cos_vphix * g_icvx[k];
                grad_p[0] = grad_se2[0]; grad_p[1] = grad_se2[1];
                grad_yaw = grad_se2[2];
                put(true, true, false, true);

                // attitude                                                       :600-609
                for (int k = 0; k < 3; k++) grad_se2[k] = -g_cxi[k];
                grad_p[0] = grad_se2[0]; grad_p[1] = grad_se2[1];
                grad_yaw = grad_se2[2];
                put(true, false, false, false);

                // surface variation                                              :611-620
                for (int k = 0; k < 3; k++) grad_se2[k] = g_sig[k];
                grad_p[0] = grad_se2[0]; grad_p[1] = grad_se2[1];
                grad_yaw = grad_se2[2];
                put(true, false, false, false);

                s1 += step;
            }
            base_time += minco.pos.T1[i];
        }

        Vec gdPxy_fx, gdPyaw_fx;                                                  // :627-636
        minco.pos.calGradCTtoQT(gdCxy_fx, gdTxy_fx, gdPxy_fx);
        minco.yaw.calGradCTtoQT(gdCyaw_fx, gdTyaw_fx, gdPyaw_fx);
        double sx = 0, sy = 0;
        for (double v : gdTxy_fx) sx += v;
        for (double v : gdTyaw_fx) sy += v;
        double grad_Tsum_fx = p.rho_T + sx / piece_xy + sy / piece_yaw;
        double gdTau_fx = grad_Tsum_fx * getTtoTauGrad(tau);
        auto absmax = [](const Vec& v) { double m = 0; for (double a : v) m = std::max(m, std::fabs(a)); return m; };
        scale_fx = 1.0 / std::max(1.0, std::max(std::max(absmax(gdPxy_fx), absmax(gdPyaw_fx)), std::fabs(gdTau_fx)));   // :651-652

        Vec gdCxy(nCxy), gdTxy(piece_xy), gdCyaw(nCyaw), gdTyaw(piece_yaw), gPxy, gPyaw;
        for (int ci = 0; ci < ncon; ci++) {                                       // :637-660
            const Con& c = cons[ci];
            std::fill(gdCxy.begin(), gdCxy.end(), 0.0);
            std::fill(gdTxy.begin(), gdTxy.end(), 0.0);
            std::fill(gdCyaw.begin(), gdCyaw.end(), 0.0);
            std::fill(gdTyaw.begin(), gdTyaw.end(), 0.0);
            for (int k = 0; k < 12; k++) gdCxy[(size_t)c.i * 12 + k] = c.cxy[k];
            gdTxy[c.i] = c.txy;
            for (int k = 0; k < 6; k++) gdCyaw[(size_t)c.yaw_idx * 6 + k] = c.cyaw[k];
            gdTyaw[c.yaw_idx] = c.tyaw;
            minco.pos.calGradCTtoQT(gdCxy, gdTxy, gPxy);
            minco.yaw.calGradCTtoQT(gdCyaw, gdTyaw, gPyaw);
            double tx = 0, ty = 0;
            for (double v : gdTxy) tx += v;
            for (double v : gdTyaw) ty += v;
            double gdTau = (tx / piece_xy + ty / piece_yaw) * getTtoTauGrad(tau);
            scale_cx[ci] = 1.0 / std::max(1.0, std::max(std::max(absmax(gPxy), absmax(gPyaw)), std::fabs(gdTau)));
        }
    }

    // ---- optimizeSE2Traj cpp:168-278.  initStateXY/endStateXY: column-major 2x3; innerPtsXY: column-major 2 x (Nxy-1)
    int optimizeSE2Traj(const double* initStateXY, const double* endStateXY, const double* innerPtsXY, int n_inner_xy,
                        const double* initYaw, const double* endYaw, const double* innerPtsYaw, int n_inner_yaw, double totalTime,
                        Vec* x_out = nullptr) {
        int ret_code = 0;
        stats = AlmStats();
        trace.clear();
        piece_xy = n_inner_xy + 1;                                                // :180-186
        piece_yaw = n_inner_yaw + 1;
        minco.reset(piece_xy, piece_yaw);
        for (int k = 0; k < 6; k++) { init_xy[k] = initStateXY[k]; end_xy[k] = endStateXY[k]; }
        for (int k = 0; k < 3; k++) { init_yaw[k] = initYaw[k]; end_yaw[k] = endYaw[k]; }
        int variable_num = 2 * (piece_xy - 1) + (piece_yaw - 1) + 1;              // :188
        equal_num = piece_xy * (p.int_K + 1);                                     // :190-203
        non_equal_num = piece_xy * (p.int_K + 1) * 6;
        hx.assign((size_t)equal_num, 0.0);
        lambda.assign((size_t)equal_num, 0.0);
        gx.assign((size_t)non_equal_num, 0.0);
        mu.assign((size_t)non_equal_num, 0.0);
        scale_fx = 1.0;
        scale_cx.assign((size_t)(equal_num + non_equal_num), 1.0);
        Vec x(variable_num);                                                      // :206-216
        dim_T = 1;
        x[0] = logC2(totalTime);
        for (int i = 0; i < 2 * (piece_xy - 1); i++) x[dim_T + i] = innerPtsXY[i];
        for (int i = 0; i < piece_yaw - 1; i++) x[dim_T + 2 * (piece_xy - 1) + i] = innerPtsYaw[i];
        LbfgsParam lp;                                                            // :219-225
        lp.mem_size = p.mem_size;
        lp.past = p.past;
        lp.g_epsilon = p.g_epsilon;
        lp.min_step = p.min_step;
        lp.delta = p.delta;
        lp.max_iterations = (int)p.inner_max_iter;
        double inner_cost = 0;
        int iter = 0;                                                             // :230-232
        if (p.use_scaling) initScaling(x);
        EvalFn eval = [this](const Vec& xx, Vec& gg) { return innerCallback(xx, gg); };
        ProgressFn prog = [this](const Vec&, const Vec&, double fx, double, int k, int) { trace.push_back(fx); return (int)(k > 1e3); };   // earlyExit :1016
        passes.clear(); iter_log.clear(); pass_log_start.clear(); snap_valid = false;
        int pass = 0;
        while (true) {                                                            // :234-271
            LbfgsStats ls;
            trace.push_back(-1.0);
            PassRec pr;
            if (record_passes) { pr.x_in = x; pr.lambda_in = lambda; pr.mu_in = mu; pr.rho_in = rho; }
            pass_log_start.push_back((int)iter_log.size());
            LbfgsState snap;
            const bool want = pass == snap_pass && !snap_valid;
            int result = lbfgs_optimize(x, inner_cost, eval, prog, lp, &ls, want ? &snap : nullptr, want ? snap_k : -1, &iter_log);
            if (want && !snap.x.empty()) { snap_state = snap; snap_valid = true; }
            stats.lbfgs_iters += ls.iters;
            stats.evals += ls.evals;
            stats.last_lbfgs_ret = result;
            bool stop = false;
            if (result == LBFGS_CONVERGENCE || result == LBFGS_CANCELED || result == LBFGS_STOP || result == LBFGSERR_MAXIMUMITERATION) {
            } else if (result == LBFGSERR_MAXIMUMLINESEARCH) {
            } else {
                ret_code = 1;
                stop = true;
            }
            bool conv = false;
            if (!stop) {
                updateDualVars();                                                 // :257
                conv = judgeConvergence();                                        // :259
            }
            if (record_passes) {
                pr.x_out = x; pr.lambda_out = lambda; pr.mu_out = mu; pr.rho_out = rho; pr.hx = hx; pr.gx = gx;
                pr.cost = inner_cost; pr.ret = result; pr.k = ls.iters; pr.converged = conv ? 1 : 0;
                passes.push_back(pr);
            }
            pass++;
            if (stop) break;
            if (conv) break;
            if (++iter > p.max_iter) { ret_code = 2; break; }                     // :265
        }
        stats.alm_iters = iter;
        stats.inner_cost = inner_cost;
        if (x_out) *x_out = x;
        return ret_code;
    }

    // ---- TEST AIDS: continue the L-BFGS iteration loop from a given state for at most `budget` iterations, with the object's current
    // duals / scales / rho (set_state); and ONE complete ALM pass (lbfgs_optimize + updateDualVars + judgeConvergence) from x.
    LbfgsParam lbfgsParams() const {
        LbfgsParam lp;                                                            // :219-225
        lp.mem_size = p.mem_size; lp.past = p.past; lp.g_epsilon = p.g_epsilon; lp.min_step = p.min_step; lp.delta = p.delta;
        lp.max_iterations = (int)p.inner_max_iter;
        return lp;
    }
    int lbfgsResume(LbfgsState& s, int budget) {
        EvalFn eval = [this](const Vec& xx, Vec& gg) { return innerCallback(xx, gg); };
        ProgressFn prog = [](const Vec&, const Vec&, double, double, int k, int) { return (int)(k > 1e3); };
        LbfgsStats ls;
        return lbfgs_loop(s, eval, prog, lbfgsParams(), &ls, budget, nullptr, -1, nullptr);
    }
    // returns the L-BFGS code; *accepted = the ALM accepted it (dual update + convergence test ran), *converged = judgeConvergence()
    int almPass(Vec& x, double& cost, int& k, int& accepted, int& converged) {
        EvalFn eval = [this](const Vec& xx, Vec& gg) { return innerCallback(xx, gg); };
        ProgressFn prog = [](const Vec&, const Vec&, double, double, int kk, int) { return (int)(kk > 1e3); };
        LbfgsStats ls;
        const int result = lbfgs_optimize(x, cost, eval, prog, lbfgsParams(), &ls);
        k = ls.iters;
        accepted = (result == LBFGS_CONVERGENCE || result == LBFGS_CANCELED || result == LBFGS_STOP || result == LBFGSERR_MAXIMUMITERATION ||
                    result == LBFGSERR_MAXIMUMLINESEARCH) ? 1 : 0;
        converged = 0;
        if (accepted) { updateDualVars(); converged = judgeConvergence() ? 1 : 0; }
        return result;
    }

    // ---- post-solve report: getMaxVxAxAyCurAttSig h:170-229 + getNonHolError se2traj.hpp:551-561.
    // Evaluates the trajectory of the LAST objective evaluation (Q1) through Piece::getValue (se2traj.hpp:106-150)
    // and PolyTrajectory::locatePieceIdx (:343-361).  out[7] = max_vx, max_ax, max_ay, max_cur, max_att, max_sig, nonhol_err
    static int locate(const Vec& T, double& t) {
        int N = (int)T.size(), idx;
        double dur;
        for (idx = 0; idx < N && t > (dur = T[idx]); idx++) t -= dur;
        if (idx == N) { idx--; t += T[idx]; }
        return idx;
    }
    static void polyval(const MinJerk& m, double t, double* v, double* dv, double* ddv) {
        int idx = locate(m.T1, t);
        for (int d = 0; d < m.D; d++) {
            // highest order first (se2traj.hpp:106-150): value += tn * coeff(order i), tn *= t
            double val = 0, tn = 1.0;
            for (int k = 0; k <= 5; k++) { val += tn * m.C(6 * idx + k, d); tn *= t; }
            double dval = 0; tn = 1.0; int n = 1;
            for (int k = 1; k <= 5; k++) { dval += n * tn * m.C(6 * idx + k, d); tn *= t; n++; }
            double ddval = 0; tn = 1.0; int mm = 1; n = 2;
            for (int k = 2; k <= 5; k++) { ddval += mm * n * tn * m.C(6 * idx + k, d); tn *= t; mm++; n++; }
            if (v) v[d] = val;
            if (dv) dv[d] = dval;
            if (ddv) ddv[d] = ddval;
        }
    }
    void report(double out[7]) const {
        double max_ax = 0, max_ay = 0, max_vx = 0, max_cur = 0, max_att = -1.0, max_sig = 0, err = 0;
        double durx = 0, dury = 0;
        for (double t : minco.pos.T1) durx += t;
        for (double t : minco.yaw.T1) dury += t;
        const double total = std::min(durx, dury);
        const double gravity = map->gravity;
        for (double t = 0.0; t < total; t += 0.01) {
            double pxy[2], vxy[2], axy[2], yaw, dyaw;
            polyval(minco.pos, t, pxy, vxy, axy);
            polyval(minco.yaw, t, &yaw, &dyaw, nullptr);
            double se2[3] = {pxy[0], pxy[1], yaw};
            normSO2(se2[2]);
            double tv[7];
            map->getTerrainVariables(se2, tv);
            double vnorm = std::sqrt(vxy[0] * vxy[0] + vxy[1] * vxy[1]);
            double lon = axy[0] * std::cos(yaw) + axy[1] * std::sin(yaw);
            double lat = -axy[0] * std::sin(yaw) + axy[1] * std::cos(yaw);
            double vx = vnorm * tv[0];
            double ax = lon * tv[0] + gravity * tv[1];
            double ay = lat * tv[2] + gravity * tv[3];
            double wz = dyaw * tv[5];
            double cur = wz / std::sqrt(vx * vx + delta_sigl);
            double att = -1.0 / tv[5];
            if (std::fabs(max_ax) < std::fabs(ax)) max_ax = ax;
            if (std::fabs(max_ay) < std::fabs(ay)) max_ay = ay;
            if (std::fabs(max_vx) < std::fabs(vx)) max_vx = vx;
            if (std::fabs(max_cur) < std::fabs(cur)) max_cur = cur;
            if (max_att < att) max_att = att;
            if (max_sig < tv[6]) max_sig = tv[6];
            err += std::fabs(vxy[0] * std::sin(yaw) + vxy[1] * (-std::cos(yaw)));
        }
        out[0] = max_vx; out[1] = max_ax; out[2] = max_ay; out[3] = max_cur; out[4] = max_att; out[5] = max_sig; out[6] = err;
    }
};

}  // namespace orc
