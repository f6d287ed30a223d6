// ORACLE -- TEST INFRASTRUCTURE ONLY (see banded.hpp header).  PARITY UNPINNED.
//
// Restates back_end/include/utils/lbfgs.hpp:
//   parameters/defaults      :15-129
//   return codes             :135-184
//   line_search_lewisoverton :276-389  (incl. the non-upstream early accept :327-330)
//   lbfgs_optimize           :439-722
// Vectors are std::vector<double>; Eigen reductions are summed left to right (or, -DORACLE_EIGEN_REDUX=1, in Eigen 3.3.7's own order: eigen_redux.hpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

#include "eigen_redux.hpp"

namespace orc {

struct LbfgsParam {                 // lbfgs.hpp:15-129
    int mem_size = 8;
    double g_epsilon = 1.0e-5;
    int past = 3;
    double delta = 1.0e-6;
    int max_iterations = 0;
    int max_linesearch = 64;
    double min_step = 1.0e-20;
    double max_step = 1.0e+20;
    double f_dec_coeff = 1.0e-4;
    double s_curv_coeff = 0.9;
    double cautious_factor = 1.0e-6;
    double machine_prec = 1.0e-16;
};

enum {                              // lbfgs.hpp:135-184
    LBFGS_CONVERGENCE = 0,
    LBFGS_STOP,
    LBFGS_CANCELED,
    LBFGSERR_UNKNOWNERROR = -1024,
    LBFGSERR_INVALID_N,
    LBFGSERR_INVALID_MEMSIZE,
    LBFGSERR_INVALID_GEPSILON,
    LBFGSERR_INVALID_TESTPERIOD,
    LBFGSERR_INVALID_DELTA,
    LBFGSERR_INVALID_MINSTEP,
    LBFGSERR_INVALID_MAXSTEP,
    LBFGSERR_INVALID_FDECCOEFF,
    LBFGSERR_INVALID_SCURVCOEFF,
    LBFGSERR_INVALID_MACHINEPREC,
    LBFGSERR_INVALID_MAXLINESEARCH,
    LBFGSERR_INVALID_FUNCVAL,
    LBFGSERR_MINIMUMSTEP,
    LBFGSERR_MAXIMUMSTEP,
    LBFGSERR_MAXIMUMLINESEARCH,
    LBFGSERR_MAXIMUMITERATION,
    LBFGSERR_WIDTHTOOSMALL,
    LBFGSERR_INVALIDPARAMETERS,
    LBFGSERR_INCREASEGRADIENT,
};

using Vec = std::vector<double>;
using EvalFn = std::function<double(const Vec& x, Vec& g)>;
using ProgressFn = std::function<int(const Vec& x, const Vec& g, double fx, double step, int k, int ls)>;

// (default build: left to right; -DORACLE_EIGEN_REDUX=1: the association of Eigen 3.3.7's vectorised redux, eigen_redux.hpp)
inline double vdotp(const double* a, const double* b, int n) {
#if ORACLE_EIGEN_REDUX
    return eigen_redux_linear(n, [&](int i) { return a[i] * b[i]; });
#else
    double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s;
#endif
}
inline double vdot(const Vec& a, const Vec& b) { return vdotp(a.data(), b.data(), (int)a.size()); }
inline double vabsmax(const Vec& a) { double m = 0; for (double v : a) m = std::max(m, std::fabs(v)); return m; }

struct LbfgsStats { int iters = 0; int evals = 0; };

inline int line_search_lewisoverton(Vec& x, double& f, Vec& g, double& stp, const Vec& s, const Vec& xp, const Vec& gp,
                                    double stpmin, double stpmax, const EvalFn& eval, const LbfgsParam& param, LbfgsStats* st) {
    int count = 0;
    bool brackt = false, touched = false;
    double finit, dginit, dgtest, dstest;
    double mu = 0.0, nu = stpmax;
    if (!(stp > 0.0)) return LBFGSERR_INVALIDPARAMETERS;          // :291-294
    dginit = vdot(gp, s);                                         // :297
    if (0.0 < dginit) return LBFGSERR_INCREASEGRADIENT;           // :300-303
    finit = f;                                                    // :306-308
    dgtest = param.f_dec_coeff * dginit;
    dstest = param.s_curv_coeff * dginit;
    const size_t n = x.size();
    while (true) {
        for (size_t i = 0; i < n; i++) x[i] = xp[i] + stp * s[i];    // :312
        f = eval(x, g);                                           // :315-316
        ++count;
        if (st) st->evals++;
        if (std::isinf(f) || std::isnan(f)) return LBFGSERR_INVALID_FUNCVAL;      // :319-322
        if (param.past > 0 && std::fabs(finit - f) / (std::fabs(finit) + 1.0) < param.delta / param.past)   // :327-330
            return count;
        if (f > finit + stp * dgtest) {                           // :332-336
            nu = stp;
            brackt = true;
        } else {
            if (vdot(g, s) < dstest) mu = stp;                    // :340-343
            else return count;                                    // :346
        }
        if (param.max_linesearch <= count) return LBFGSERR_MAXIMUMLINESEARCH;     // :349-353
        if (brackt && (nu - mu) < param.machine_prec * nu) return LBFGSERR_WIDTHTOOSMALL;   // :354-358
        if (brackt) stp = 0.5 * (mu + nu);                        // :360-367
        else stp *= 2.0;
        if (stp < stpmin) return LBFGSERR_MINIMUMSTEP;            // :369-373
        if (stp > stpmax) {                                       // :374-387
            if (touched) return LBFGSERR_MAXIMUMSTEP;
            touched = true;
            stp = stpmax;
        }
    }
}

// Complete state of lbfgs_optimize at the top of its iteration loop (lbfgs.hpp:555), i.e. just before `xp = x; gp = g`.
// TEST AID (teacher-forced late-state tests): lets a test capture the state at a chosen iteration and lets both this
// restatement and the device continue from exactly that state for a bounded number of iterations.
struct LbfgsState {
    Vec x, g, d, pf, lm_ys;
    std::vector<double> lm_s, lm_y;     // column j at [j*n, (j+1)*n)
    double step = 0.0, fx = 0.0;
    int k = 0, end = 0, bound = 0;
};
struct LbfgsIterLog { int k, ls, bound, end, updated; };   // one record per completed iteration: ls = line-search return, updated = cautious test passed
constexpr int LBFGS_RUNNING = 999;      // lbfgs_loop stopped by its iteration budget (not a reference code)

// The iteration loop of lbfgs_optimize (lbfgs.hpp:555-715), continuing from `s`.  budget < 0: unlimited (the reference's behaviour);
// otherwise at most `budget` iterations are executed and LBFGS_RUNNING is returned with `s` at the next loop top.
// snap/snap_k: copy the state into *snap when the loop top is reached with k == snap_k.
inline int lbfgs_loop(LbfgsState& s, const EvalFn& eval, const ProgressFn& progress, const LbfgsParam& param, LbfgsStats* st,
                      int budget, LbfgsState* snap, int snap_k, std::vector<LbfgsIterLog>* log) {
    int ret, i, j, ls;
    double step_min, step_max, ys, yy, beta, rate, cau, gnorm_inf, xnorm_inf;
    Vec& x = s.x; Vec& g = s.g; Vec& d = s.d; Vec& pf = s.pf; Vec& lm_ys = s.lm_ys;
    std::vector<double>& lm_s = s.lm_s; std::vector<double>& lm_y = s.lm_y;
    double& step = s.step; double& fx = s.fx; int& k = s.k; int& end = s.end; int& bound = s.bound;
    const int n = (int)x.size();
    const int m = param.mem_size;
    Vec xp(n), gp(n), lm_alpha(m, 0.0);
    while (true) {
        if (snap && k == snap_k) *snap = s;
        if (budget == 0) return LBFGS_RUNNING;
        if (budget > 0) budget--;
        xp = x; gp = g;                                       // :557-558
        step_min = param.min_step;                            // :561-568 (no stepbound callback in the reference's call)
        step_max = param.max_step;
        ls = line_search_lewisoverton(x, fx, g, step, d, xp, gp, step_min, step_max, eval, param, st);   // :571
        if (ls < 0) {                                         // :573-580
            x = xp; g = gp;
            ret = ls;
            if (log) log->push_back({k, ls, bound, end, 0});
            break;
        }
        if (progress) {                                       // :583-590
            if (progress(x, g, fx, step, k, ls)) { ret = LBFGS_CANCELED; break; }
        }
        gnorm_inf = vabsmax(g);                               // :597-604
        xnorm_inf = vabsmax(x);
        if (gnorm_inf / std::max(1.0, xnorm_inf) < param.g_epsilon) { ret = LBFGS_CONVERGENCE; break; }
        if (0 < param.past) {                                 // :611-628
            if (param.past <= k) {
                rate = std::fabs(pf[k % param.past] - fx) / std::max(1.0, std::fabs(fx));
                if (rate < param.delta) { ret = LBFGS_STOP; break; }
            }
            pf[k % param.past] = fx;
        }
        if (param.max_iterations != 0 && param.max_iterations <= k) { ret = LBFGSERR_MAXIMUMITERATION; break; }   // :630-635
        ++k;                                                  // :638
        double* sc = &lm_s[(size_t)end * n];
        double* yc = &lm_y[(size_t)end * n];
        for (i = 0; i < n; i++) { sc[i] = x[i] - xp[i]; yc[i] = g[i] - gp[i]; }   // :645-646
        ys = vdotp(yc, sc, n);                                // :654-656
        yy = vdotp(yc, yc, n);
        lm_ys[end] = ys;
        for (i = 0; i < n; i++) d[i] = -g[i];                 // :659
        cau = vdotp(sc, sc, n) * std::sqrt(vdot(gp, gp)) * param.cautious_factor;   // :673
        if (log) log->push_back({k - 1, ls, bound, end, ys > cau ? 1 : 0});
        if (ys > cau) {                                       // :675-708
            ++bound;
            bound = m < bound ? m : bound;
            end = (end + 1) % m;
            j = end;
            for (i = 0; i < bound; ++i) {
                j = (j + m - 1) % m;
                lm_alpha[j] = vdotp(&lm_s[(size_t)j * n], d.data(), n) / lm_ys[j];
                const double a = -lm_alpha[j];
                const double* yj = &lm_y[(size_t)j * n];
                for (int t = 0; t < n; t++) d[t] += a * yj[t];
            }
            const double sc0 = ys / yy;
            for (int t = 0; t < n; t++) d[t] *= sc0;
            for (i = 0; i < bound; ++i) {
                beta = vdotp(&lm_y[(size_t)j * n], d.data(), n) / lm_ys[j];
                const double a = lm_alpha[j] - beta;
                const double* sj = &lm_s[(size_t)j * n];
                for (int t = 0; t < n; t++) d[t] += a * sj[t];
                j = (j + 1) % m;
            }
        }
        step = 1.0;                                           // :712
    }
    return ret;
}

inline int lbfgs_optimize(Vec& x, double& f, const EvalFn& eval, const ProgressFn& progress, const LbfgsParam& param, LbfgsStats* st,
                          LbfgsState* snap = nullptr, int snap_k = -1, std::vector<LbfgsIterLog>* log = nullptr) {
    int ret, i;
    double gnorm_inf, xnorm_inf;
    const int n = (int)x.size();
    const int m = param.mem_size;
    if (n <= 0) return LBFGSERR_INVALID_N;                         // :455-498
    if (m <= 0) return LBFGSERR_INVALID_MEMSIZE;
    if (param.g_epsilon < 0.0) return LBFGSERR_INVALID_GEPSILON;
    if (param.past < 0) return LBFGSERR_INVALID_TESTPERIOD;
    if (param.delta < 0.0) return LBFGSERR_INVALID_DELTA;
    if (param.min_step < 0.0) return LBFGSERR_INVALID_MINSTEP;
    if (param.max_step < param.min_step) return LBFGSERR_INVALID_MAXSTEP;
    if (!(param.f_dec_coeff > 0.0 && param.f_dec_coeff < 1.0)) return LBFGSERR_INVALID_FDECCOEFF;
    if (!(param.s_curv_coeff < 1.0 && param.s_curv_coeff > param.f_dec_coeff)) return LBFGSERR_INVALID_SCURVCOEFF;
    if (!(param.machine_prec > 0.0)) return LBFGSERR_INVALID_MACHINEPREC;
    if (param.max_linesearch <= 0) return LBFGSERR_INVALID_MAXLINESEARCH;

    LbfgsState s;                                                  // :501-511 (xp, gp, lm_alpha are scratch of the loop)
    s.x = x; s.g.assign(n, 0.0); s.d.assign(n, 0.0); s.pf.assign(std::max(1, param.past), 0.0);
    s.lm_ys.assign(m, 0.0);
    s.lm_s.assign((size_t)n * m, 0.0); s.lm_y.assign((size_t)n * m, 0.0);

    s.fx = eval(s.x, s.g);                                        // :521
    if (st) st->evals++;
    s.pf[0] = s.fx;                                               // :524
    for (i = 0; i < n; i++) s.d[i] = -s.g[i];                     // :530
    gnorm_inf = vabsmax(s.g);                                     // :535-536
    xnorm_inf = vabsmax(s.x);
    if (gnorm_inf / std::max(1.0, xnorm_inf) < param.g_epsilon) { // :538-542
        ret = LBFGS_CONVERGENCE;
    } else {
        s.step = 1.0 / std::sqrt(vdot(s.d, s.d));                 // :548
        s.k = 1; s.end = 0; s.bound = 0;
        ret = lbfgs_loop(s, eval, progress, param, st, -1, snap, snap_k, log);
    }
    x = s.x;
    f = s.fx;                                                     // :717
    if (st) st->iters = s.k;
    return ret;
}

}  // namespace orc
