// TEST SCAFFOLDING -- not part of the product, never linked into libunevenhip.so.
// Instantiates the single-source workgroup program (uneven_planner_amd/csrc/solver_program.hpp) with a sequential
// stand-in for the workgroup object so that the optimiser's state machine (evaluation, scaling, L-BFGS, ALM) can be
// checked against the oracle in the CPU-only test tier, before any GPU time is spent.  The reduction order mimics
// DevWG (per-lane strided partials -> 64-lane xor butterfly -> sequential over the 4 waves).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../uneven_planner_amd/csrc/minco_op_host.hpp"
#include "../../uneven_planner_amd/csrc/solver_program.hpp"

using namespace uph;

static int g_lanes = 256;
struct HostWG {
    static constexpr int NT = 256;      // partial-sum emulation width (upper bound of the lanes)
    static constexpr bool MFMA_SCATTER = false;     // the matrix-core xy scatter is a device path (DevWG::scatterXY17); the emulator runs the vector form
    int size() const { return g_lanes; }
    template <class F>
    void pfor(int n, F f) { for (int i = 0; i < n; i++) f(i); }
    void sync() {}
    long long clock() { return 0; }
    long long realtime() { return 0; }
    template <class F>
    void one(F f) { f(); }
    template <int M, class F>
    void sum(int n, double* out, F f) {
        static double part[NT][M];
        for (int t = 0; t < NT; t++) for (int m = 0; m < M; m++) part[t][m] = 0.0;
        const int L = g_lanes;
        for (int t = 0; t < L; t++) for (int i = t; i < n; i += L) f(i, part[t]);
        for (int m = 0; m < M; m++) {
            double total = 0.0;
            for (int w = 0; w < g_lanes / 64; w++) {
                double a[64], b[64];
                for (int l = 0; l < 64; l++) a[l] = part[w * 64 + l][m];
                for (int off = 32; off >= 1; off >>= 1) {
                    for (int l = 0; l < 64; l++) b[l] = a[l] + a[l ^ off];
                    std::memcpy(a, b, sizeof(a));
                }
                total = (w == 0) ? a[0] : total + a[0];
            }
            out[m] = total;
        }
    }
    // lbfgs.hpp:687-710, plain loops
    double bcast(double v) const { return v; }
    // the reduction split over several regions (DevWG::accBegin / accChunk / accEnd): per-lane partials kept across the chunks, one butterfly at the end
    double pacc_[NT][8];
    template <int M>
    void accBegin() { for (int t = 0; t < NT; t++) for (int m = 0; m < M; m++) pacc_[t][m] = 0.0; }
    template <int M, class F>
    void accChunk(int n, F f) { const int L = g_lanes; for (int t = 0; t < L; t++) for (int i = t; i < n; i += L) f(i, pacc_[t]); }
    template <int M>
    void accEnd(double* out) {
        for (int m = 0; m < M; m++) {
            double total = 0.0;
            for (int w = 0; w < g_lanes / 64; w++) {
                double a[64], b[64];
                for (int l = 0; l < 64; l++) a[l] = pacc_[w * 64 + l][m];
                for (int off = 32; off >= 1; off >>= 1) {
                    for (int l = 0; l < 64; l++) b[l] = a[l] + a[l ^ off];
                    std::memcpy(a, b, sizeof(a));
                }
                total = (w == 0) ? a[0] : total + a[0];
            }
            out[m] = total;
        }
    }
    template <int MS, int MM, class F>
    void sumMax(int n, double* outS, double* outM, F f) {
        static double part[NT][MS], pm[NT][MM];
        for (int t = 0; t < NT; t++) { for (int m = 0; m < MS; m++) part[t][m] = 0.0; for (int m = 0; m < MM; m++) pm[t][m] = 0.0; }
        const int L = g_lanes;
        for (int t = 0; t < L; t++) for (int i = t; i < n; i += L) f(i, part[t], pm[t]);
        for (int m = 0; m < MS; m++) {
            double total = 0.0;
            for (int w = 0; w < g_lanes / 64; w++) {
                double a[64], b[64];
                for (int l = 0; l < 64; l++) a[l] = part[w * 64 + l][m];
                for (int off = 32; off >= 1; off >>= 1) {
                    for (int l = 0; l < 64; l++) b[l] = a[l] + a[l ^ off];
                    std::memcpy(a, b, sizeof(a));
                }
                total = (w == 0) ? a[0] : total + a[0];
            }
            outS[m] = total;
        }
        for (int m = 0; m < MM; m++) { double v = 0.0; for (int t = 0; t < L; t++) v = pm[t][m] > v ? pm[t][m] : v; outM[m] = v; }
    }
    void twoLoop(double* d, const double* g, int n, const double* hist, double* dg_out, double* /*al_lds*/, int m, int end, int bound, double scale) {
        double lm_alpha[512];
        const int rowd = histRowDoubles(n), np = 64 * histNQ(n);
        int j = end;
        for (int i = 0; i < bound; ++i) {
            j = (j + m - 1) % m;
            const double* row = hist + (size_t)j * rowd;
            const double* sj = row + 2;
            const double* yj = row + 2 + np;
            double part[64] = {0};
            for (int t = 0; t < n; t++) part[t & 63] += sj[t] * d[t];
            double tot = 0.0;
            for (int l = 0; l < 64; l++) tot += part[l];
            const double al = tot / row[0];
            lm_alpha[j] = al;
            for (int t = 0; t < n; t++) d[t] += (-al) * yj[t];
        }
        for (int t = 0; t < n; t++) d[t] *= scale;
        for (int i = 0; i < bound; ++i) {
            const double* row = hist + (size_t)j * rowd;
            const double* sj = row + 2;
            const double* yj = row + 2 + np;
            double part[64] = {0};
            for (int t = 0; t < n; t++) part[t & 63] += yj[t] * d[t];
            double tot = 0.0;
            for (int l = 0; l < 64; l++) tot += part[l];
            const double beta = tot / row[0];
            const double a = lm_alpha[j] - beta;
            for (int t = 0; t < n; t++) d[t] += a * sj[t];
            j = (j + 1) % m;
        }
        double part[64] = {0};
        for (int t = 0; t < n; t++) part[t & 63] += g[t] * d[t];
        double tot = 0.0;
        for (int l = 0; l < 64; l++) tot += part[l];
        *dg_out = tot;
    }
    // MINCO knot sweeps (DevWG::thomas): the same per-knot arithmetic, plain serial loops
    template <bool ADJ>
    static void thomasChain(const double* tab, double* buf, int len, int ks, int cs) {
        double y0 = 0.0, y1 = 0.0;
        for (int j = 1; j <= len; j++) {
            double P[4], M1[4], Q[4], M2[4];
            thomasFactors<ADJ>(tab, j, P, M1, Q, M2);
            double* p = buf + knotOff(j - 1, ks);          // (padded per lane block of the device solve, uph_common.hpp)
            double P0, P1;
            mv2(P, p[0], p[cs], P0, P1);
            submv2(P0, P1, M1, y0, y1, y0, y1);
            p[0] = y0; p[cs] = y1;
        }
        double z0 = 0.0, z1 = 0.0;
        for (int j = len; j >= 1; j--) {
            double P[4], M1[4], Q[4], M2[4];
            thomasFactors<ADJ>(tab, j, P, M1, Q, M2);
            double* p = buf + knotOff(j - 1, ks);          // (padded per lane block of the device solve, uph_common.hpp)
            double q0, q1;
            mv2(Q, p[0], p[cs], q0, q1);
            submv2(q0, q1, M2, z0, z1, z0, z1);
            p[0] = z0; p[cs] = z1;
        }
    }
    void thomas(const double* tab, bool adj, double* bw, int lenW, double* bx, int lenX) {
        if (adj) { thomasChain<true>(tab, bw, lenW, 2, 1); thomasChain<true>(tab, bx, lenX, 4, 2); thomasChain<true>(tab, bx + 1, lenX, 4, 2); }
        else { thomasChain<false>(tab, bw, lenW, 2, 1); thomasChain<false>(tab, bx, lenX, 4, 2); thomasChain<false>(tab, bx + 1, lenX, 4, 2); }
    }
    template <class F>
    double maxv(int n, F f) {
        double m = 0.0;
        for (int i = 0; i < n; i++) { double v = f(i); if (v > m) m = v; }
        return m;
    }
};

static std::vector<double> g_trace;
// teacher-forced hooks (modes 4 / 5 of emu_run): pass cap, iteration budget, finish flag and the in/out state arrays
static int g_cap = 0, g_budget = 0, g_finish = 0;
static double *g_hg = nullptr, *g_hd = nullptr, *g_hpf = nullptr, *g_hs = nullptr, *g_hy = nullptr, *g_hys = nullptr, *g_hscal = nullptr;

struct Emu {
    GridDev grid;
    std::vector<double> cells;
    std::vector<float> cells32;
    OptParams P;
};

extern "C" {

void* emu_create(const double* mp11, const double* cells4, const double* op21) {
    Emu* e = new Emu();
    GridDev& g = e->grid;
    const double PI = 3.14159265358979323846;
    double size[3] = {mp11[1], mp11[2], 2.0 * PI + 5e-2};
    g.xy_res = mp11[6]; g.yaw_res = mp11[7]; g.xy_inv = 1.0 / g.xy_res; g.yaw_inv = 1.0 / g.yaw_res;
    for (int i = 0; i < 3; i++) { g.minb[i] = -size[i] / 2.0; g.maxb[i] = size[i] / 2.0; g.origin[i] = g.minb[i]; }
    finishGrid(g);
    g.nx = (int)std::ceil(size[0] / g.xy_res); g.ny = (int)std::ceil(size[1] / g.xy_res); g.nyaw = (int)std::ceil(size[2] / g.yaw_res);
    g.gravity = mp11[10];
    g.x_off = 0; g.nx_hold = g.nx;
    size_t nc = (size_t)g.nx * g.ny * g.nyaw;
    e->cells.assign(cells4, cells4 + 4 * nc);
    g.cells = e->cells.data(); g.cells32 = nullptr;
    OptParams& P = e->P;
    P.rho_T = op21[0]; P.rho_ter = op21[1]; P.max_vel = op21[2]; P.max_acc_lon = op21[3]; P.max_acc_lat = op21[4];
    P.max_kap = op21[5]; P.min_cxi = op21[6]; P.max_sig = op21[7]; P.use_scaling = op21[8] != 0.0;
    P.beta = op21[10]; P.gamma = op21[11]; P.epsilon_con = op21[12]; P.max_iter = op21[13];
    P.g_epsilon = op21[14]; P.min_step = op21[15]; P.inner_max_iter = (int)op21[16]; P.delta = op21[17];
    P.mem_size = (int)op21[18]; P.past = (int)op21[19]; P.int_K = (int)op21[20];
    P.max_linesearch = 64; P.max_step = 1e20; P.f_dec_coeff = 1e-4; P.s_curv_coeff = 0.9; P.cautious_factor = 1e-6; P.machine_prec = 1e-16;
    finishParams(P);
    return e;
}
void emu_destroy(void* h) { delete (Emu*)h; }
// fp32 cell storage (GridDev::cells32): the cells rounded to float, read through the same lookup code
void emu_store_f32(void* h) {
    Emu* e = (Emu*)h;
    e->cells32.resize(e->cells.size());
    for (size_t i = 0; i < e->cells.size(); i++) e->cells32[i] = (float)e->cells[i];
    e->grid.cells = nullptr; e->grid.cells32 = e->cells32.data();
}
void emu_set_lanes(int lanes) { g_lanes = lanes; }
// g, d [n]; pf [8]; lm_s / lm_y [mem][n]; lm_ys [mem]; scal [8]: in step, fx, k, end, bound -> out + code, accepted, converged
void emu_set_hook(int cap, int budget, int finish, double* g, double* d, double* pf8, double* lm_s, double* lm_y, double* lm_ys, double* scal8) {
    g_cap = cap; g_budget = budget; g_finish = finish; g_hg = g; g_hd = d; g_hpf = pf8; g_hs = lm_s; g_hy = lm_y; g_hys = lm_ys; g_hscal = scal8;
}

void emu_terrain(void* h, const double* pos, int n, double* values, double* grads) {
    Emu* e = (Emu*)h;
    for (int i = 0; i < n; i++) {
        double gr[7][3];
        const double yaw = pos[3 * i + 2];
        terrainAllWithGrad(e->grid, pos[3 * i], pos[3 * i + 1], yaw, std::cos(yaw), std::sin(yaw), values + 7 * i, gr);
        std::memcpy(grads + 21 * i, gr, sizeof(gr));
    }
}

// mode 0: eval at x (state as given); 1: initScaling at x; 2: full optimize from x; 3: optimize then report;
// 4: ALM passes from the given x / duals / scales / rho, at most g_cap passes (no reset, no initScaling); 5: resume the L-BFGS loop from the hook state
// state arrays in the reference's order: lambda[S], mu[6S] (sample-major), scale_cx[7S] (7 per sample); io = in/out
// scal[8]: in: rho, scale_fx; out: rho, scale_fx, f, jerk, Txy, Tyaw, (unused)   istat[6]: ret, alm_iters, lbfgs_iters, evals, last_ret, hist_reads
void emu_run(void* h, int mode, int n_inner_xy, int n_inner_yaw, const double* init_xy, const double* end_xy, const double* init_yaw,
             const double* end_yaw, double* x_io, double* g_out, double* lambda_io, double* mu_io, double* scale_cx_io, double* hx_out,
             double* gx_out, double* cxy_out, double* cyaw_out, double* scal, long long* istat, double* report7) {
    Emu* e = (Emu*)h;
    TrajDesc td;
    std::memset(&td, 0, sizeof(td));
    td.Nxy = n_inner_xy + 1; td.Nyaw = n_inner_yaw + 1;
    td.n = 2 * n_inner_xy + n_inner_yaw + 1; td.S = td.Nxy * (e->P.int_K + 1);
    td.op_xy = 0; td.op_yaw = 1;
    for (int k = 0; k < 6; k++) { td.init_xy[k] = init_xy[k]; td.end_xy[k] = end_xy[k]; }
    for (int k = 0; k < 3; k++) { td.init_yaw[k] = init_yaw[k]; td.end_yaw[k] = end_yaw[k]; }
    const int S = td.S, n = td.n;
    std::vector<double> Mt0, Mr0, Mt1, Mr1;
    buildMincoOp(td.Nxy, Mt0, Mr0);
    buildMincoOp(td.Nyaw, Mt1, Mr1);
    MincoOp ops[2] = {{td.Nxy, Mt0.data(), Mr0.data()}, {td.Nyaw, Mt1.data(), Mr1.data()}};
    TrajState st;
    std::memset(&st, 0, sizeof(st));
    st.rho = scal[0]; st.scale_fx = scal[1];
    std::vector<double> dual(7 * S), res(7 * S, 0.0), scl(7 * S), xg(x_io, x_io + n), gout(n, 0.0), cxy(12 * td.Nxy), cyaw(6 * td.Nyaw);
    std::vector<double> hist((size_t)e->P.mem_size * histRowDoubles(n), 0.0), rep(7, 0.0);
    g_trace.assign(20000, 0.0);
    for (int s = 0; s < S; s++) {
        dual[s] = lambda_io[s];
        for (int q = 0; q < 6; q++) dual[(q + 1) * S + s] = mu_io[6 * s + q];
        for (int q = 0; q < 7; q++) scl[q * S + s] = scale_cx_io[7 * s + q];
    }
    BatchDev bd;
    std::memset(&bd, 0, sizeof(bd));
    bd.B = 1; bd.desc = &td; bd.state = &st; bd.ops = ops; bd.x = xg.data(); std::vector<double> x0copy(xg); bd.x0 = x0copy.data(); bd.gout = gout.data(); bd.dual = dual.data(); bd.res = res.data();
    bd.scl = scl.data(); bd.cxy = cxy.data(); bd.cyaw = cyaw.data(); bd.hist = hist.data(); std::vector<double> thomasTab; buildThomasTable(thomasTab); bd.thomas = thomasTab.data(); bd.report = rep.data(); bd.trace = g_trace.data(); bd.trace_cap = (int)g_trace.size();
    std::vector<double> lds(Solver<HostWG>::ldsDoubles(td.Nxy, td.Nyaw, n, g_lanes, e->P.mem_size, e->P.int_K) + 64);
    HostWG wg;
    Solver<HostWG> sol(wg, e->grid, e->P, bd, 0, lds.data());
    std::vector<double> rsd(n, 0.0), rs(24, 0.0);
    bd.rs_d = rsd.data(); bd.rs = rs.data();
    if (mode == 0) sol.evalOnly(st, 1);
    else if (mode == 1) sol.scalingOnly(st);
    else if (mode == 4) sol.optimize(st, g_cap);
    else if (mode == 5) {
        const int mem = e->P.mem_size, rowd = histRowDoubles(n), np = 64 * histNQ(n);
        for (int j = 0; j < mem; j++) {
            double* row = hist.data() + (size_t)j * rowd;
            row[0] = g_hys[j]; row[1] = 1.0 / g_hys[j];
            std::memcpy(row + 2, g_hs + (size_t)j * n, 8 * n);
            std::memcpy(row + 2 + np, g_hy + (size_t)j * n, 8 * n);
        }
        std::memcpy(gout.data(), g_hg, 8 * n);
        std::memcpy(rsd.data(), g_hd, 8 * n);
        for (int q = 0; q < 5; q++) rs[q] = g_hscal[q];
        for (int q = 0; q < 8; q++) rs[8 + q] = g_hpf[q];
        sol.resumeHook(st, g_budget < 0 ? (1 << 28) : g_budget, g_finish);
        for (int j = 0; j < mem; j++) {
            const double* row = hist.data() + (size_t)j * rowd;
            g_hys[j] = row[0];
            std::memcpy(g_hs + (size_t)j * n, row + 2, 8 * n);
            std::memcpy(g_hy + (size_t)j * n, row + 2 + np, 8 * n);
        }
        std::memcpy(g_hg, gout.data(), 8 * n);
        std::memcpy(g_hd, rsd.data(), 8 * n);
        for (int q = 0; q < 8; q++) { g_hscal[q] = rs[q]; g_hpf[q] = rs[8 + q]; }
    }
    else { sol.prepare(st); sol.optimize(st); if (mode == 3) sol.report(st); }
    std::memcpy(x_io, xg.data(), 8 * n);
    std::memcpy(g_out, gout.data(), 8 * n);
    for (int s = 0; s < S; s++) {
        lambda_io[s] = dual[s]; hx_out[s] = res[s];
        for (int q = 0; q < 6; q++) { mu_io[6 * s + q] = dual[(q + 1) * S + s]; gx_out[6 * s + q] = res[(q + 1) * S + s]; }
        for (int q = 0; q < 7; q++) scale_cx_io[7 * s + q] = scl[q * S + s];
    }
    std::memcpy(cxy_out, cxy.data(), 8 * cxy.size());
    std::memcpy(cyaw_out, cyaw.data(), 8 * cyaw.size());
    scal[0] = st.rho; scal[1] = st.scale_fx; scal[2] = st.f; scal[3] = st.jerk_cost; scal[4] = st.T_xy; scal[5] = st.T_yaw;
    istat[0] = st.ret_code; istat[1] = st.alm_iters; istat[2] = st.lbfgs_iters; istat[3] = st.evals; istat[4] = st.last_lbfgs_ret; istat[5] = st.hist_reads;
    if (report7) std::memcpy(report7, rep.data(), 56);
    g_trace.resize(std::min<size_t>(g_trace.size(), (size_t)sol.trace_n));
}

int emu_get_trace(double* out, int cap) {
    int n = (int)std::min<size_t>(g_trace.size(), (size_t)cap);
    for (int i = 0; i < n; i++) out[i] = g_trace[i];
    int t = (int)g_trace.size();
    g_trace.clear();
    return t;
}

void emu_minco_op(int N, double* Mr) {
    std::vector<double> a, b;
    buildMincoOp(N, a, b);
    std::memcpy(Mr, b.data(), 8 * b.size());
}

}  // extern "C"
