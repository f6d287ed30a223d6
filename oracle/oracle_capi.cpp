// ORACLE -- TEST INFRASTRUCTURE ONLY (see banded.hpp header).  PARITY UNPINNED.
// Flat C entry points so tests / smoke() / bench.py's cpu_baseline leg can drive the CPU restatement via ctypes.
#include <chrono>
#include <cstring>
#include <memory>
#include "alm.hpp"
#include "resample.hpp"
#include "mapbuild.hpp"
#include "kino_astar.hpp"
#include <queue>
#include <random>

using namespace orc;

namespace {
AlmParams paramsFrom(const double* p) {
    // order == uph_opt_params in include/uneven_hip.h (all as doubles)
    AlmParams a;
    a.rho_T = p[0]; a.rho_ter = p[1]; a.max_vel = p[2]; a.max_acc_lon = p[3]; a.max_acc_lat = p[4];
    a.max_kap = p[5]; a.min_cxi = p[6]; a.max_sig = p[7]; a.use_scaling = p[8] != 0.0;
    a.rho = p[9]; a.beta = p[10]; a.gamma = p[11]; a.epsilon_con = p[12]; a.max_iter = p[13];
    a.g_epsilon = p[14]; a.min_step = p[15]; a.inner_max_iter = p[16]; a.delta = p[17];
    a.mem_size = (int)p[18]; a.past = (int)p[19]; a.int_K = (int)p[20];
    return a;
}
}  // namespace

extern "C" {

// ---------------- grid / terrain
void* orc_grid_create(double size_x, double size_y, double xy_res, double yaw_res, double gravity) {
    Grid* g = new Grid();
    g->init(size_x, size_y, xy_res, yaw_res);
    g->gravity = gravity;
    return g;
}
void orc_grid_destroy(void* h) { delete (Grid*)h; }
void orc_grid_dims(void* h, int* dims3) { Grid* g = (Grid*)h; for (int i = 0; i < 3; i++) dims3[i] = g->voxel_num[i]; }
// cells: ncell x 4 {z, sigma, zbx, zby}, reference address order (x slowest, yaw fastest)
void orc_grid_set_cells(void* h, const double* cells) {
    Grid* g = (Grid*)h;
    for (size_t i = 0; i < g->map_buffer.size(); i++) {
        g->map_buffer[i] = RXS2(cells[i * 4], cells[i * 4 + 1], cells[i * 4 + 2], cells[i * 4 + 3]);
        g->c_buffer[i] = g->map_buffer[i].getC();
    }
}
void orc_grid_get_cells(void* h, double* cells, double* cbuf) {
    Grid* g = (Grid*)h;
    for (size_t i = 0; i < g->map_buffer.size(); i++) {
        cells[i * 4] = g->map_buffer[i].z; cells[i * 4 + 1] = g->map_buffer[i].sigma;
        cells[i * 4 + 2] = g->map_buffer[i].zbx; cells[i * 4 + 3] = g->map_buffer[i].zby;
        if (cbuf) cbuf[i] = g->c_buffer[i];
    }
}
void orc_grid_index_to_pos(void* h, const int* id, double* pos) { ((Grid*)h)->indexToPos(id, pos); }
// pos: n x 3 (yaw already normalised by the caller if desired); values n x 7; grads n x 21
void orc_terrain_all_with_grad(void* h, const double* pos, int n, double* values, double* grads) {
    Grid* g = (Grid*)h;
    for (int i = 0; i < n; i++) {
        double gr[7][3];
        g->getAllWithGrad(pos + 3 * i, values + 7 * i, gr);
        std::memcpy(grads + 21 * i, gr, sizeof(gr));
    }
}
void orc_terrain_get(void* h, const double* pos, int n, double* rxs2 /* n x 4 */) {
    Grid* g = (Grid*)h;
    for (int i = 0; i < n; i++) {
        RXS2 v;
        g->getTerrain(pos + 3 * i, v);
        rxs2[i * 4] = v.z; rxs2[i * 4 + 1] = v.sigma; rxs2[i * 4 + 2] = v.zbx; rxs2[i * 4 + 3] = v.zby;
    }
}
void orc_terrain_variables(void* h, const double* pos, int n, double* values) {
    Grid* g = (Grid*)h;
    for (int i = 0; i < n; i++) g->getTerrainVariables(pos + 3 * i, values + 7 * i);
}

// ---------------- MINCO
// inPs D x (N-1) column-major, ts[N], head/tail D x 3 row-major -> c (6N) x D row-major
void orc_minco_generate(int N, int D, const double* inPs, const double* ts, const double* head, const double* tail, double* c_out,
                        double* jerk_cost) {
    MinJerk m;
    m.reset(N, D);
    m.generate(inPs, ts, head, tail);
    std::memcpy(c_out, m.c.data(), sizeof(double) * 6 * N * D);
    if (jerk_cost) *jerk_cost = m.getTrajJerkCost();
}
// given (q,T) and an arbitrary gdC, gdT: returns gdP (D x (N-1) col-major) and updated gdT
void orc_minco_grad_ct_to_qt(int N, int D, const double* inPs, const double* ts, const double* head, const double* tail,
                             const double* gdC, double* gdT_inout, double* gdP_out) {
    MinJerk m;
    m.reset(N, D);
    m.generate(inPs, ts, head, tail);
    Vec gC(gdC, gdC + 6 * N * D), gT(gdT_inout, gdT_inout + N), gP;
    m.calGradCTtoQT(gC, gT, gP);
    std::memcpy(gdT_inout, gT.data(), sizeof(double) * N);
    std::memcpy(gdP_out, gP.data(), sizeof(double) * D * (N - 1));
}
void orc_minco_jerk_grad(int N, int D, const double* inPs, const double* ts, const double* head, const double* tail, double* gdC, double* gdT) {
    MinJerk m;
    m.reset(N, D);
    m.generate(inPs, ts, head, tail);
    Vec gC, gT;
    m.calJerkGradCT(gC, gT);
    std::memcpy(gdC, gC.data(), sizeof(double) * 6 * N * D);
    std::memcpy(gdT, gT.data(), sizeof(double) * N);
}
// dense banded LU solve for tests: A given dense n x n row-major with bandwidth (p,q); b n x m -> x (and A^T x = b)
void orc_banded_solve(int n, int p, int q, const double* Adense, double* b, int m, int adjoint) {
    Banded B;
    B.create(n, p, q);
    for (int i = 0; i < n; i++)
        for (int j = std::max(0, i - p); j <= std::min(n - 1, i + q); j++) B(i, j) = Adense[(size_t)i * n + j];
    B.factorizeLU();
    if (adjoint) B.solveAdj(b, m); else B.solve(b, m);
}

// ---------------- L-BFGS (stand-alone check on the extended Rosenbrock function)
int orc_lbfgs_rosenbrock(int n, double* x, double* f_out, int mem_size, int past, double g_eps, double delta, int* iters, int* evals) {
    LbfgsParam lp;
    lp.mem_size = mem_size; lp.past = past; lp.g_epsilon = g_eps; lp.delta = delta; lp.min_step = 1e-32; lp.max_iterations = 10000;
    Vec xv(x, x + n);
    EvalFn fn = [n](const Vec& xx, Vec& g) {
        double fx = 0.0;
        for (int i = 0; i < n; i += 2) {
            double t1 = 1.0 - xx[i], t2 = 10.0 * (xx[i + 1] - xx[i] * xx[i]);
            g[i + 1] = 20.0 * t2;
            g[i] = -2.0 * (xx[i] * g[i + 1] + t1);
            fx += t1 * t1 + t2 * t2;
        }
        return fx;
    };
    LbfgsStats st;
    double f = 0;
    int r = lbfgs_optimize(xv, f, fn, nullptr, lp, &st);
    std::memcpy(x, xv.data(), sizeof(double) * n);
    *f_out = f; *iters = st.iters; *evals = st.evals;
    return r;
}

// L-BFGS on analytic test functions with the complete evaluation record: every point the algorithm evaluates (line-search trials
// included), in order, with its value -- trace[e] = {x (n), f}.  kind 0: extended Rosenbrock; kind 1: ill-conditioned quadratic + quartic
// coupling f = sum_i (1 + 3 i) (x_i - c_i)^2 + 0.05 sum_i (x_i x_{i+1})^2, c_i = sin(i + 1)
int orc_lbfgs_trace(int kind, int n, double* x, int mem_size, int past, double g_eps, double delta, int max_iter, double* trace, int cap, int* evals, int* iters, double* f_out) {
    LbfgsParam lp;
    lp.mem_size = mem_size; lp.past = past; lp.g_epsilon = g_eps; lp.delta = delta; lp.min_step = 1e-32; lp.max_iterations = max_iter;
    Vec xv(x, x + n);
    int ne = 0;
    EvalFn fn = [&](const Vec& xx, Vec& g) {
        double fx = 0.0;
        if (kind == 0) {
            for (int i = 0; i < n; i += 2) {
                double t1 = 1.0 - xx[i], t2 = 10.0 * (xx[i + 1] - xx[i] * xx[i]);
                g[i + 1] = 20.0 * t2;
                g[i] = -2.0 * (xx[i] * g[i + 1] + t1);
                fx += t1 * t1 + t2 * t2;
            }
        } else {
            for (int i = 0; i < n; i++) { const double w = 1.0 + 3.0 * i, d = xx[i] - std::sin(i + 1.0); fx += w * d * d; g[i] = 2.0 * w * d; }
            for (int i = 0; i + 1 < n; i++) { const double p = xx[i] * xx[i + 1]; fx += 0.05 * p * p; g[i] += 0.1 * p * xx[i + 1]; g[i + 1] += 0.1 * p * xx[i]; }
        }
        if (ne < cap) { std::memcpy(trace + (size_t)ne * (n + 1), xx.data(), sizeof(double) * n); trace[(size_t)ne * (n + 1) + n] = fx; }
        ne++;
        return fx;
    };
    LbfgsStats st;
    double f = 0;
    int r = lbfgs_optimize(xv, f, fn, nullptr, lp, &st);
    std::memcpy(x, xv.data(), sizeof(double) * n);
    *f_out = f; *iters = st.iters; *evals = ne;
    return r;
}

// ---------------- ALM optimiser
struct OrcAlm { AlmTrajOpt opt; Vec x_last; };
void* orc_alm_create(void* grid, const double* params21) {
    OrcAlm* o = new OrcAlm{AlmTrajOpt(paramsFrom(params21)), {}};
    o->opt.map = (Grid*)grid;
    return o;
}
void orc_alm_destroy(void* h) { delete (OrcAlm*)h; }
void orc_alm_set_rho(void* h, double rho) { ((OrcAlm*)h)->opt.rho = rho; }
double orc_alm_get_rho(void* h) { return ((OrcAlm*)h)->opt.rho; }
void orc_alm_set_flat_debug(void* h, int on) { ((OrcAlm*)h)->opt.flat_debug = on != 0; }

// set up a problem exactly as optimizeSE2Traj cpp:180-216 does, WITHOUT solving; writes x0 (n) and returns n
int orc_alm_setup(void* h, const double* init_xy, const double* end_xy, const double* inner_xy, int n_inner_xy,
                  const double* init_yaw, const double* end_yaw, const double* inner_yaw, int n_inner_yaw, double total_time, double* x0) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    a.piece_xy = n_inner_xy + 1; a.piece_yaw = n_inner_yaw + 1;
    a.minco.reset(a.piece_xy, a.piece_yaw);
    for (int k = 0; k < 6; k++) { a.init_xy[k] = init_xy[k]; a.end_xy[k] = end_xy[k]; }
    for (int k = 0; k < 3; k++) { a.init_yaw[k] = init_yaw[k]; a.end_yaw[k] = end_yaw[k]; }
    int n = 2 * (a.piece_xy - 1) + (a.piece_yaw - 1) + 1;
    a.equal_num = a.piece_xy * (a.p.int_K + 1);
    a.non_equal_num = a.piece_xy * (a.p.int_K + 1) * 6;
    a.hx.assign((size_t)a.equal_num, 0.0); a.lambda.assign((size_t)a.equal_num, 0.0);
    a.gx.assign((size_t)a.non_equal_num, 0.0); a.mu.assign((size_t)a.non_equal_num, 0.0);
    a.scale_fx = 1.0;
    a.scale_cx.assign((size_t)(a.equal_num + a.non_equal_num), 1.0);
    a.dim_T = 1;
    x0[0] = AlmTrajOpt::logC2(total_time);
    for (int i = 0; i < 2 * n_inner_xy; i++) x0[1 + i] = inner_xy[i];
    for (int i = 0; i < n_inner_yaw; i++) x0[1 + 2 * n_inner_xy + i] = inner_yaw[i];
    return n;
}
// overwrite dual / scaling state (any pointer may be null)
void orc_alm_set_state(void* h, const double* lambda, const double* mu, const double* scale_cx, const double* scale_fx) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    if (lambda) a.lambda.assign(lambda, lambda + a.lambda.size());
    if (mu) a.mu.assign(mu, mu + a.mu.size());
    if (scale_cx) a.scale_cx.assign(scale_cx, scale_cx + a.scale_cx.size());
    if (scale_fx) a.scale_fx = *scale_fx;
}
void orc_alm_get_state(void* h, double* lambda, double* mu, double* scale_cx, double* scale_fx, double* hx, double* gx) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    if (lambda) std::memcpy(lambda, a.lambda.data(), 8 * a.lambda.size());
    if (mu) std::memcpy(mu, a.mu.data(), 8 * a.mu.size());
    if (scale_cx) std::memcpy(scale_cx, a.scale_cx.data(), 8 * a.scale_cx.size());
    if (scale_fx) *scale_fx = a.scale_fx;
    if (hx) std::memcpy(hx, a.hx.data(), 8 * a.hx.size());
    if (gx) std::memcpy(gx, a.gx.data(), 8 * a.gx.size());
}
void orc_alm_init_scaling(void* h, const double* x0, int n) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    a.initScaling(Vec(x0, x0 + n));
}
// one objective evaluation (innerCallback): returns f; grad[n]; parts[3] = jerk term, constraint term, tau term
double orc_alm_eval(void* h, const double* x, int n, double* grad, double* parts) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    Vec xv(x, x + n), g(n, 0.0);
    double f = a.innerCallback(xv, g);
    std::memcpy(grad, g.data(), 8 * n);
    if (parts) { parts[0] = a.last_jerk_cost_term; parts[1] = a.last_constrain_cost; parts[2] = a.last_tau_cost; }
    return f;
}
// constraint part only, at the trajectory generated from x: cost, gdCxy (6Nxy x 2), gdTxy, gdCyaw, gdTyaw
double orc_alm_constrain(void* h, const double* x, int n, double* gdCxy, double* gdTxy, double* gdCyaw, double* gdTyaw) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    a.generateFromX(Vec(x, x + n));
    double cost;
    Vec a1, a2, a3, a4;
    a.calConstrainCostGrad(cost, a1, a2, a3, a4);
    std::memcpy(gdCxy, a1.data(), 8 * a1.size()); std::memcpy(gdTxy, a2.data(), 8 * a2.size());
    std::memcpy(gdCyaw, a3.data(), 8 * a3.size()); std::memcpy(gdTyaw, a4.data(), 8 * a4.size());
    return cost;
}
void orc_alm_get_coeffs(void* h, double* c_xy, double* c_yaw, double* T_xy, double* T_yaw, double* jerk_cost) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    if (c_xy) std::memcpy(c_xy, a.minco.pos.c.data(), 8 * a.minco.pos.c.size());
    if (c_yaw) std::memcpy(c_yaw, a.minco.yaw.c.data(), 8 * a.minco.yaw.c.size());
    if (T_xy) *T_xy = a.minco.pos.T1[0];
    if (T_yaw) *T_yaw = a.minco.yaw.T1[0];
    if (jerk_cost) *jerk_cost = a.minco.getTrajJerkCost();
}
// full solve == ALMTrajOpt::optimizeSE2Traj.  stats[6] = alm_iters, lbfgs_iters, evals, last_lbfgs_ret, inner_cost, wall_ms
int orc_alm_optimize(void* h, const double* init_xy, const double* end_xy, const double* inner_xy, int n_inner_xy,
                     const double* init_yaw, const double* end_yaw, const double* inner_yaw, int n_inner_yaw, double total_time,
                     double* x_final, double* stats) {
    OrcAlm* o = (OrcAlm*)h;
    auto t0 = std::chrono::steady_clock::now();
    Vec x;
    int ret = o->opt.optimizeSE2Traj(init_xy, end_xy, inner_xy, n_inner_xy, init_yaw, end_yaw, inner_yaw, n_inner_yaw, total_time, &x);
    auto t1 = std::chrono::steady_clock::now();
    if (x_final) std::memcpy(x_final, x.data(), 8 * x.size());
    if (stats) {
        stats[0] = o->opt.stats.alm_iters; stats[1] = o->opt.stats.lbfgs_iters; stats[2] = o->opt.stats.evals;
        stats[3] = o->opt.stats.last_lbfgs_ret; stats[4] = o->opt.stats.inner_cost;
        stats[5] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    }
    return ret;
}
// ---------------- teacher-forced test aids
// request an L-BFGS state capture at the top of iteration `k` of ALM pass `pass` (0-based) during the next orc_alm_optimize; record passes
void orc_alm_set_capture(void* h, int pass, int k, int record_passes) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    a.snap_pass = pass; a.snap_k = k; a.record_passes = record_passes != 0;
}
// iteration log of the last solve: rows of 6 ints {pass, k, ls, bound, end, updated}; returns the number of rows available
int orc_alm_get_iter_log(void* h, int* out, int cap_rows) {
    const AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    int pass = 0;
    const int n = (int)a.iter_log.size();
    for (int i = 0; i < n && i < cap_rows; i++) {
        while (pass + 1 < (int)a.pass_log_start.size() && a.pass_log_start[pass + 1] <= i) pass++;
        const LbfgsIterLog& r = a.iter_log[i];
        out[6 * i] = pass; out[6 * i + 1] = r.k; out[6 * i + 2] = r.ls; out[6 * i + 3] = r.bound; out[6 * i + 4] = r.end; out[6 * i + 5] = r.updated;
    }
    return n;
}
// L-BFGS state: vectors x, g, d [n], pf [past], lm_ys [m], lm_s / lm_y [m*n] (slot j at j*n), scal = {step, fx, k, end, bound}
static void stateOut(const LbfgsState& s, double* x, double* g, double* d, double* pf, double* lm_ys, double* lm_s, double* lm_y, double* scal) {
    const size_t n = s.x.size();
    if (x) std::memcpy(x, s.x.data(), 8 * n);
    if (g) std::memcpy(g, s.g.data(), 8 * n);
    if (d) std::memcpy(d, s.d.data(), 8 * n);
    if (pf) std::memcpy(pf, s.pf.data(), 8 * s.pf.size());
    if (lm_ys) std::memcpy(lm_ys, s.lm_ys.data(), 8 * s.lm_ys.size());
    if (lm_s) std::memcpy(lm_s, s.lm_s.data(), 8 * s.lm_s.size());
    if (lm_y) std::memcpy(lm_y, s.lm_y.data(), 8 * s.lm_y.size());
    if (scal) { scal[0] = s.step; scal[1] = s.fx; scal[2] = s.k; scal[3] = s.end; scal[4] = s.bound; }
}
// captured state of the last solve (returns 0 if the requested iteration was never reached) + the duals / rho in force during that pass
int orc_alm_get_capture(void* h, double* x, double* g, double* d, double* pf, double* lm_ys, double* lm_s, double* lm_y, double* scal,
                        double* lambda, double* mu, double* rho) {
    const AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    if (!a.snap_valid) return 0;
    stateOut(a.snap_state, x, g, d, pf, lm_ys, lm_s, lm_y, scal);
    if (a.record_passes && a.snap_pass < (int)a.passes.size()) {
        const AlmTrajOpt::PassRec& pr = a.passes[a.snap_pass];
        if (lambda) std::memcpy(lambda, pr.lambda_in.data(), 8 * pr.lambda_in.size());
        if (mu) std::memcpy(mu, pr.mu_in.data(), 8 * pr.mu_in.size());
        if (rho) *rho = pr.rho_in;
    }
    return 1;
}
// continue the iteration loop from the given state for at most `budget` iterations with the object's current duals / scales / rho;
// the state is updated in place; returns the L-BFGS code or 999 (LBFGS_RUNNING) when the budget ran out first
int orc_alm_lbfgs_resume(void* h, int n, double* x, double* g, double* d, double* pf, double* lm_ys, double* lm_s, double* lm_y, double* scal, int budget) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    const int m = a.p.mem_size, past = std::max(1, a.p.past);
    LbfgsState s;
    s.x.assign(x, x + n); s.g.assign(g, g + n); s.d.assign(d, d + n); s.pf.assign(pf, pf + past); s.lm_ys.assign(lm_ys, lm_ys + m);
    s.lm_s.assign(lm_s, lm_s + (size_t)m * n); s.lm_y.assign(lm_y, lm_y + (size_t)m * n);
    s.step = scal[0]; s.fx = scal[1]; s.k = (int)scal[2]; s.end = (int)scal[3]; s.bound = (int)scal[4];
    const int ret = a.lbfgsResume(s, budget);
    stateOut(s, x, g, d, pf, lm_ys, lm_s, lm_y, scal);
    return ret;
}
// number of recorded passes; pass record i: x_in/x_out [n], lambda_in/out [S], mu_in/out [6S], hx [S], gx [6S], scal = {rho_in, rho_out, cost, ret, k, converged}
int orc_alm_num_passes(void* h) { return (int)((OrcAlm*)h)->opt.passes.size(); }
int orc_alm_get_pass(void* h, int i, double* x_in, double* x_out, double* lambda_in, double* lambda_out, double* mu_in, double* mu_out, double* hx, double* gx, double* scal) {
    const AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    if (i < 0 || i >= (int)a.passes.size()) return -1;
    const AlmTrajOpt::PassRec& r = a.passes[i];
    auto cp = [](double* dst, const Vec& v) { if (dst) std::memcpy(dst, v.data(), 8 * v.size()); };
    cp(x_in, r.x_in); cp(x_out, r.x_out); cp(lambda_in, r.lambda_in); cp(lambda_out, r.lambda_out); cp(mu_in, r.mu_in); cp(mu_out, r.mu_out); cp(hx, r.hx); cp(gx, r.gx);
    if (scal) { scal[0] = r.rho_in; scal[1] = r.rho_out; scal[2] = r.cost; scal[3] = r.ret; scal[4] = r.k; scal[5] = r.converged; }
    return 0;
}
// updateDualVars + judgeConvergence on the object's current residuals (what the ALM loop does after an accepted L-BFGS code)
int orc_alm_finish_pass(void* h) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    a.updateDualVars();
    return a.judgeConvergence() ? 1 : 0;
}
// ONE ALM pass from x with the object's current duals / scales / rho: lbfgs_optimize, then (if the ALM accepts the code) updateDualVars and
// judgeConvergence.  out = {lbfgs code, k, accepted, converged, cost}
void orc_alm_pass(void* h, int n, double* x, double* out5) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    Vec xv(x, x + n);
    double cost = 0; int k = 0, acc = 0, conv = 0;
    const int ret = a.almPass(xv, cost, k, acc, conv);
    std::memcpy(x, xv.data(), 8 * n);
    out5[0] = ret; out5[1] = k; out5[2] = acc; out5[3] = conv; out5[4] = cost;
}

int orc_alm_get_trace(void* h, double* out, int cap) {
    const Vec& t = ((OrcAlm*)h)->opt.trace;
    int n = (int)std::min<size_t>(t.size(), (size_t)cap);
    for (int i = 0; i < n; i++) out[i] = t[i];
    return (int)t.size();
}
// install a trajectory (coefficients in the reference's internal layout, uniform piece durations) without generating it: lets the
// post-solve report be evaluated on coefficients that came from elsewhere (the device's stored trajectory)
void orc_alm_set_coeffs(void* h, const double* c_xy, const double* c_yaw, double T_xy, double T_yaw) {
    AlmTrajOpt& a = ((OrcAlm*)h)->opt;
    std::memcpy(a.minco.pos.c.data(), c_xy, 8 * a.minco.pos.c.size());
    std::memcpy(a.minco.yaw.c.data(), c_yaw, 8 * a.minco.yaw.c.size());
    std::fill(a.minco.pos.T1.begin(), a.minco.pos.T1.end(), T_xy);
    std::fill(a.minco.yaw.T1.begin(), a.minco.yaw.T1.end(), T_yaw);
}
void orc_alm_report(void* h, double* out7) { ((OrcAlm*)h)->opt.report(out7); }

// ---------------- map build
struct OrcMap { MapBuilder b; };
void* orc_mapbuilder_create(const float* xyz, long n, int apply_filters) {
    OrcMap* m = new OrcMap();
    Cloud c;
    c.x.resize(n); c.y.resize(n); c.z.resize(n);
    for (long i = 0; i < n; i++) { c.x[i] = xyz[3 * i]; c.y[i] = xyz[3 * i + 1]; c.z[i] = xyz[3 * i + 2]; }
    m->b.setCloud(c, apply_filters != 0);
    return m;
}
void* orc_mapbuilder_from_pcd(const char* path) {
    Cloud c;
    if (!readPCD(path, c)) return nullptr;
    OrcMap* m = new OrcMap();
    m->b.setCloud(c, true);
    return m;
}
void orc_mapbuilder_destroy(void* h) { delete (OrcMap*)h; }
long orc_mapbuilder_cloud_size(void* h) { return (long)((OrcMap*)h)->b.cloud.size(); }
void orc_mapbuilder_get_cloud(void* h, float* xyz) {
    const Cloud& c = ((OrcMap*)h)->b.cloud;
    for (size_t i = 0; i < c.size(); i++) { xyz[3 * i] = c.x[i]; xyz[3 * i + 1] = c.y[i]; xyz[3 * i + 2] = c.z[i]; }
}
// mp[11]: iter_num, size_x, size_y, ell_x, ell_y, ell_z, xy_res, yaw_res, min_cnormal, max_rho, gravity
static MapParams mapParamsFrom(const double* mp) {
    MapParams p;
    p.iter_num = (int)mp[0]; p.map_size_x = mp[1]; p.map_size_y = mp[2]; p.ellipsoid_x = mp[3]; p.ellipsoid_y = mp[4]; p.ellipsoid_z = mp[5];
    p.xy_resolution = mp[6]; p.yaw_resolution = mp[7]; p.min_cnormal = mp[8]; p.max_rho = mp[9]; p.gravity = mp[10];
    return p;
}
// runs constructMap on x-slab [x0,x1) of grid `g` (cells outside the slab untouched); then occupancy if do_occ
void orc_map_construct(void* builder, void* grid, const double* mp, int x0, int x1, int do_occ) {
    MapParams p = mapParamsFrom(mp);
    ((OrcMap*)builder)->b.construct(*(Grid*)grid, p, x0, x1);
    if (do_occ) computeOccupancy(*(Grid*)grid, p);
}
void orc_map_fit_cell(void* builder, void* grid, const double* mp, int x, int y, int yaw, double* cell4, double* c_out) {
    MapParams p = mapParamsFrom(mp);
    RXS2 cell; double cb = 1.0;
    ((OrcMap*)builder)->b.fitCell(*(Grid*)grid, p, x, y, yaw, cell, cb);
    cell4[0] = cell.z; cell4[1] = cell.sigma; cell4[2] = cell.zbx; cell4[3] = cell.zby; *c_out = cb;
}
void orc_grid_get_occ(void* h, char* occ, char* occ_r2) {
    Grid* g = (Grid*)h;
    if (occ) std::memcpy(occ, g->occ_buffer.data(), g->occ_buffer.size());
    if (occ_r2) std::memcpy(occ_r2, g->occ_r2_buffer.data(), g->occ_r2_buffer.size());
}
void orc_plane_filter(const double* pts, int n, double* cell4) {
    RXS2 r = planeFilter(std::vector<double>(pts, pts + 3 * n));
    cell4[0] = r.z; cell4[1] = r.sigma; cell4[2] = r.zbx; cell4[3] = r.zby;
}
int orc_map_write_csv(void* grid, const char* path) { return writeMapCSV(*(Grid*)grid, path) ? 0 : -1; }
int orc_map_read_csv(void* grid, const char* path) { return readMapCSV(*(Grid*)grid, path) ? 0 : -1; }

// plan_manager.cpp:62-132 for one path of M poses; inner_xy / inner_yaw sized by the caller (cap entries), counts returned in n2
void orc_resample(const double* path, int M, const double* mp5, double* init_xy, double* end_xy, double* init_yaw, double* end_yaw, double* inner_xy, double* inner_yaw,
                  int cap_xy, int cap_yaw, int* n2, double* total_time, double* unwrapped) {
    orc::ManagerParams mp;
    mp.piece_len = mp5[0]; mp.mean_vel = mp5[1]; mp.init_time_times = mp5[2]; mp.yaw_piece_times = mp5[3]; mp.init_sig_vel = mp5[4];
    // mp5[1] < 0 selects the test node's stage (alm_traj_opt.cpp:73-144) with max_vel = -mp5[1]
    orc::Resampled r = mp5[1] < 0.0 ? orc::resamplePathTest(std::vector<double>(path, path + 3 * (size_t)M), -mp5[1])
                                    : orc::resamplePath(std::vector<double>(path, path + 3 * (size_t)M), mp);
    for (int k = 0; k < 6; k++) { init_xy[k] = r.init_xy[k]; end_xy[k] = r.end_xy[k]; }
    for (int k = 0; k < 3; k++) { init_yaw[k] = r.init_yaw[k]; end_yaw[k] = r.end_yaw[k]; }
    n2[0] = (int)(r.inner_xy.size() / 2); n2[1] = (int)r.inner_yaw.size();
    for (int i = 0; i < std::min(n2[0], cap_xy) * 2; i++) inner_xy[i] = r.inner_xy[i];
    for (int i = 0; i < std::min(n2[1], cap_yaw); i++) inner_yaw[i] = r.inner_yaw[i];
    *total_time = r.total_time;
    if (unwrapped) for (int i = 0; i < M; i++) unwrapped[i] = r.yaw_unwrapped[i];
}

// occupancy of a grid whose cells were set directly (uneven_map.cpp:170-179); clears first (the reference fills fresh zero buffers)
void orc_grid_compute_occ(void* h, double min_cnormal, double max_rho) {
    Grid* g = (Grid*)h;
    std::fill(g->occ_buffer.begin(), g->occ_buffer.end(), 0);
    std::fill(g->occ_r2_buffer.begin(), g->occ_r2_buffer.end(), 0);
    MapParams p; p.min_cnormal = min_cnormal; p.max_rho = max_rho;
    computeOccupancy(*g, p);
}
void orc_grid_set_occ(void* h, const char* occ, const char* occ_r2) {
    Grid* g = (Grid*)h;
    if (occ) std::memcpy(g->occ_buffer.data(), occ, g->occ_buffer.size());
    if (occ_r2) std::memcpy(g->occ_r2_buffer.data(), occ_r2, g->occ_r2_buffer.size());
}

// UnevenMap::isOccupancy(pos) / isOccupancyXY(pos) (uneven_map.h:473-500): posToIndex, isInMap(idx) -> -1 outside, else the layer's value;
// getTerrainSig (:389-396).  Any output may be null.
void orc_grid_frontend_query(void* h, const double* pos, int n, double* sigma, int* occ, int* occ_xy) {
    Grid* g = (Grid*)h;
    for (int i = 0; i < n; i++) {
        const double* p = pos + 3 * i;
        int id[3] = {floorToInt((p[0] - g->map_origin[0]) * g->xy_resolution_inv), floorToInt((p[1] - g->map_origin[1]) * g->xy_resolution_inv),
                     floorToInt((p[2] - g->map_origin[2]) * g->yaw_resolution_inv)};
        const bool in = g->isInMapIdx(id);
        if (occ) occ[i] = in ? (int)g->occ_buffer[g->toAddress(id[0], id[1], id[2])] : -1;
        if (occ_xy) occ_xy[i] = in ? (int)g->occ_r2_buffer[(size_t)id[0] * g->voxel_num[1] + id[1]] : -1;
        if (sigma) { RXS2 v; g->getTerrain(p, v); sigma[i] = v.sigma; }
    }
}

// ---------------- front end: KinoAstar::plan (kino_astar.cpp:67-236) and the Dubins one-shot
// kp[13]: yaw_resolution, lambda_heu, weight_r2, weight_so2, weight_v_change, weight_delta_change, weight_sigma, time_interval,
//         collision_interval, oneshot_range, wheel_base, max_steer, max_vel   (rosparam order of kino_astar.cpp:7-19)
struct OrcKino { KinoAstar ka; };
void* orc_kino_create(void* grid, const double* kp) {
    KinoParams p;
    p.yaw_resolution = kp[0]; p.lambda_heu = kp[1]; p.weight_r2 = kp[2]; p.weight_so2 = kp[3]; p.weight_v_change = kp[4]; p.weight_delta_change = kp[5];
    p.weight_sigma = kp[6]; p.time_interval = kp[7]; p.collision_interval = kp[8]; p.oneshot_range = kp[9]; p.wheel_base = kp[10]; p.max_steer = kp[11]; p.max_vel = kp[12];
    OrcKino* k = new OrcKino();
    k->ka.init((Grid*)grid, p);
    return k;
}
void orc_kino_destroy(void* h) { delete (OrcKino*)h; }
// stats[4] = status, iter_num, use_node_num, n_shot; path: at most path_cap poses; expanded_index: at most exp_cap (ix, iy, iyaw) triples.
// returns the number of poses of front_end_path
int orc_kino_plan(void* h, const double* start3, const double* end3, int max_expand, double* path, int path_cap, int* stats, int* expanded_index, int exp_cap, int* n_expanded) {
    KinoResult r = ((OrcKino*)h)->ka.plan(start3, end3, max_expand);
    stats[0] = r.status; stats[1] = r.iter_num; stats[2] = r.use_node_num; stats[3] = r.n_shot;
    const int np = (int)r.path.size() / 3;
    if (path) for (int i = 0; i < std::min(np, path_cap) * 3; i++) path[i] = r.path[i];
    const int ne = (int)r.expanded_index.size() / 3;
    if (n_expanded) *n_expanded = ne;
    if (expanded_index) for (int i = 0; i < std::min(ne, exp_cap) * 3; i++) expanded_index[i] = r.expanded_index[i];
    return np;
}
// out[6] = path type (0 LSL 1 RSR 2 RSL 3 LSR 4 RLR 5 LRL), t, p, q (units of rho), distance, -
void orc_dubins(const double* from3, const double* to3, double rho, double* out) {
    const dubins::Path p = dubins::between(from3, to3, rho);
    out[0] = p.type; out[1] = p.len[0]; out[2] = p.len[1]; out[3] = p.len[2]; out[4] = dubins::distance(from3, to3, rho); out[5] = 0.0;
}
void orc_dubins_interpolate(const double* from3, const double* to3, double rho, const double* t, int n, double* out3) {
    for (int i = 0; i < n; i++) dubins::interpolate(from3, to3, t[i], rho, out3 + 3 * i);
}
void orc_kino_state_transit(void* h, const double* state0, const double* ctrl2, double T, double* state1) { ((OrcKino*)h)->ka.stateTransit(state0, state1, ctrl2, T); }
// The restated heap (KinoAstar::pushHeap / popHeap) against std::priority_queue itself on a random stream of pushes, pops and IN-PLACE
// changes of queued keys (what kino_astar.cpp:218-229 does), NaN keys included: returns the number of pops that differed.
// the dot product exactly as the L-BFGS restatement forms it: left to right in the default build, Eigen 3.3.7's redux order with -DORACLE_EIGEN_REDUX=1
double orc_dot(const double* a, const double* b, int n) { return vdotp(a, b, n); }
int orc_eigen_redux_enabled() { return ORACLE_EIGEN_REDUX; }
double orc_block_sum(const double* e, int rows, int dim) {      // e[r * dim + d]: the 6 x Dim / 3 x Dim block reduction of calGradCTtoQT in this build's order
#if ORACLE_EIGEN_REDUX
    return eigen_redux_block(rows, dim, [&](int r, int d) { return e[r * dim + d]; });
#else
    double s = 0.0;
    for (int d = 0; d < dim; d++) for (int r = 0; r < rows; r++) s += e[r * dim + d];
    return s;
#endif
}
int orc_heap_selfcheck(unsigned seed, int nops, int nan_every) {
    struct Cmp { const std::vector<KinoNode>* pool; bool operator()(int a, int b) const { return (*pool)[a].f_score > (*pool)[b].f_score; } };
    Grid g; g.init(1.0, 1.0, 0.05, 0.1);
    KinoAstar ka; ka.init(&g, KinoParams());
    ka.pool.assign((size_t)nops + 8, KinoNode());
    std::priority_queue<int, std::vector<int>, Cmp> pq(Cmp{&ka.pool});
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    int next = 0, bad = 0;
    std::vector<int> queued;
    for (int op = 0; op < nops; op++) {
        const double u = U(rng);
        if (u < 0.55 || ka.heap.empty()) {
            double f = std::floor(U(rng) * 64.0) / 8.0;                       // coarse keys: plenty of exact ties
            if (nan_every > 0 && next % nan_every == nan_every - 1) f = std::nan("");
            ka.pool[next].f_score = f;
            ka.pushHeap(next); pq.push(next); queued.push_back(next);
            next++;
        } else if (u < 0.8) {
            if (ka.heap[0] != pq.top()) bad++;
            const int t = pq.top();
            ka.popHeap(); pq.pop();
            for (size_t i = 0; i < queued.size(); i++) if (queued[i] == t) { queued[i] = queued.back(); queued.pop_back(); break; }
        } else if (!queued.empty()) {
            const int t = queued[(size_t)(U(rng) * queued.size()) % queued.size()];
            ka.pool[t].f_score -= std::floor(U(rng) * 16.0) / 8.0;          // lowered in place: no re-heapify on either side
        }
    }
    while (!ka.heap.empty()) { if (ka.heap[0] != pq.top()) bad++; ka.popHeap(); pq.pop(); }
    return bad;
}

}  // extern "C"
