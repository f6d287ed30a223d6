// The back-end optimiser as ONE workgroup program per trajectory: the whole PHR-ALM / L-BFGS / MINCO-SE(2) solve of
// ALMTrajOpt::optimizeSE2Traj runs inside a single persistent workgroup (unevenhip.hip launches one workgroup per
// trajectory of the batch), so divergent iteration counts cost nothing and no state leaves the CU between iterations.
//
// Reference functions realised here (paths under /root/reference/src/uneven_planner):
//   Solver::optimize     <- ALMTrajOpt::optimizeSE2Traj            back_end/src/alm_traj_opt.cpp:168-278
//   Solver::eval         <- innerCallback + calConstrainCostGrad   alm_traj_opt.cpp:280-347, 663-991
//   Solver::initScaling  <- ALMTrajOpt::initScaling                alm_traj_opt.cpp:349-661
//   Solver::lbfgs        <- lbfgs::lbfgs_optimize                  back_end/include/utils/lbfgs.hpp:439-722
//   Solver::lineSearch   <- line_search_lewisoverton               lbfgs.hpp:276-389
//   Solver::generate/adjoint <- MinJerkOpt::generate / calGradCTtoQT   back_end/include/utils/se2traj.hpp:595-680, 751-816
//   Solver::jerk*        <- getTrajJerkCost / calJerkGradCT        se2traj.hpp:697-747
//   Solver::report       <- getMaxVxAxAyCurAttSig + getNonHolError alm_traj_opt.h:170-229, se2traj.hpp:551-561
//
// MINCO as a knot system.  The reference gives every piece the same duration (calTfromTau, alm_traj_opt.h:257-261).
// In normalised time s = t/T (c~_k = c_k T^k) the banded system of se2traj.hpp:612-674 no longer depends on T:
// A(1) c~ = b~ with b~ = [P0, T V0, T^2 A0, ..q_i.., Pf, T Vf, T^2 Af].  A quintic piece is fixed by (p, v, a) at its two ends,
// so only the interior knots' z_j = (v_j, a_j) are solved for -- jerk and snap continuity at every interior knot, a block-
// tridiagonal system with constant 2x2 blocks whose block-LU factors do not depend on N and sit in LDS (minco_op_host.hpp).
// generate() = right-hand sides + one fused forward/backward sweep (wg.thomas) + the constant quintic Hermite map per piece;
// calGradCTtoQT() = the transposed Hermite map + the transposed sweeps.  The time gradient follows from c_k = c~_k T^-k:
//   sum_i dW/dT_i = sum_i dK/dT_i - sum_{i,k} (k c_ik / T) dK/dc_ik + <gamma, db~/dT>,  gamma = M^T (dK/dc . T^-k),
// M = (Hermite expansion) o (knot solve); identical to the reference's dK/dT_i - <B_i, lambda> in exact arithmetic.
// initScaling needs single rows of the inverse instead (one sparse adjoint per constraint): it gathers them from the dense knot
// operator W = the 2(N-1) x (N+5) restriction of A(1)^-1 (MincoOp, built once per N on the host).
//
// Wave-uniform values (reduction results, T, rho, scales, per-evaluation constants, line-search scalars) pass through
// wg.bcast(): on the device that is v_readfirstlane, which parks them in scalar registers -- see DevWG::uni in unevenhip.hip
// for why this matters (scratch reloads of "uniform" VGPRs drain the memory pipeline).
#pragma once
#include "terrain_dev.hpp"
#include "uph_common.hpp"

namespace uph {

// scatter batch widths (LDS reads in flight per lane): xy blocks hold K + 1 = 17 records, yaw candidates ~40
#ifndef UPH_GRID_FROM_MEM
#define UPH_GRID_FROM_MEM 1
#endif
// diagnostic builds only (tools/pmc_phases.sh): phases of an objective evaluation that run -- 1 generate + expand, 2 samples, 4 scatter, 8 adjoint.
// A masked build computes nonsense (stale coefficients / records / gradients); it exists so that hardware counters of the penalty kernel can be
// attributed to a phase as (all phases) - (all but one).  The shipped library has every phase: the conditions fold away.
#ifndef UPH_PHASE_MASK
#define UPH_PHASE_MASK 15
#endif
// columns of the knot operator fetched per batch in initScaling's row gathers (scalingGroup)
#ifndef UPH_SCALING_WB
#define UPH_SCALING_WB 8
#endif
#ifndef UPH_SC_XB
#define UPH_SC_XB 9
#endif
#ifndef UPH_SC_YB
#define UPH_SC_YB 14
#endif
// widening of a yaw piece's candidate-slot window beyond its exact-arithmetic bounds (slots before / after); the tag test decides membership
#ifndef UPH_SC_WLO
#define UPH_SC_WLO 2
#endif
#ifndef UPH_SC_WHI
#define UPH_SC_WHI 3
#endif

// SR = real type of the sample-phase arithmetic: double (the reference's, default) or f32r (fp32 sample mode, uph_common.hpp)
template <class WG, class SR = double>
struct Solver {
    WG& wg;
    const GridDev& grid;
    const OptParams& P;
    const BatchDev& bd;
    const TrajDesc& td;
    int Nxy, Nyaw, n, S, K, mem, CH, CHP, recd;
    int CHS;             // samples per chunk of the sample / scatter loop (<= CH, the record slots): see the constructor
    // workgroup-shared arrays (LDS)
    int* rtag;
    double *x, *xp, *g, *gp, *d, *cxy, *cyaw, *Gxy, *Gyaw, *gamxy, *gamyaw, *bt, *rec, *wtab, *ttab, *pf, *hd;
    // HBM
    double *dual, *res, *scl, *hist;
    int hrow, hnp;       // history row length and padded vector length in doubles (uph_common.hpp histRowDoubles)
    const double *Wr_xy, *Wr_yaw;                       // dense knot operators [row][col] (initScaling's row gathers)
    // uniform scalars (identical in every lane)
    double rho, scale_fx, Txy, Tyaw, last_jerk, last_gd;
    long long hist_reads;
    long long cyc[16];
    long long t_last_eval_end;
    long long* sub_t = nullptr;      // microbenchmark hook: sub-step ticks of generate() [0 rhs, 1 knot solve], expand() [2], adjoint() [4 transposed Hermite, 5 knot solve, 6 gamma]
    int evals, bidx, trace_n;
    float inv_k1;                           // 1 / (K + 1) for divSmall
    // per-evaluation wave-uniform constants of the sample loop, formed once and parked in scalar registers
    double ec_irho, ec_step, ec_invK, ec_iTyaw, ec_omega, ec_omega_h;

    // S is not part of the footprint: samples are processed in chunks of CH = workgroup size (records of one chunk only)
    static constexpr int REC_FIELDS = 12;   // per-sample record: grad_p, grad_v, grad_a (2 each) + its 6 yaw-block contributions (+ an int32 yaw-piece tag)
    static UPH_HD size_t ldsDoubles(int Nxy, int Nyaw, int n, int CH, int mem, int K) {
        size_t recd = (size_t)REC_FIELDS * (CH + 1) + (CH + 1) / 2;       // double fields (stride CH + 1: bank spread) + the int32 yaw-piece tags
        const size_t nvec = 2 * (Nxy + 5) + (Nyaw + 5);
        // the record buffer is idle outside the sample loop: knot states of generate(), knot gradients + direct parts of adjoint()
        const size_t knd = (size_t)knotBufDoubles(Nxy + 1, 4) + knotBufDoubles(Nyaw + 1, 2), adj = (size_t)knotBufDoubles(Nxy, 4) + knotBufDoubles(Nyaw, 2) + nvec;
        recd = recd < knd ? knd : recd;
        recd = recd < adj ? adj : recd;
        recd = recd < (size_t)mem ? (size_t)mem : recd;                  // ... and the two-loop's alphas
        const size_t td_ = (size_t)Nxy + K + 2;                          // sample-time tables; gamma lives in the same words after adjoint()
        return (size_t)5 * n + (nvec < td_ ? td_ : nvec) + 2 * (12 * Nxy + 6 * Nyaw) + recd + (size_t)6 * (K + 1) + THOMAS_DOUBLES + MAX_PAST + 8 + 18;
    }

    UPH_HD Solver(WG& w, const GridDev& gr, const OptParams& p, const BatchDev& b, int bi, double* lds)
        : wg(w), grid(gr), P(p), bd(b), td(b.desc[bi]) {
        bidx = bi;
        Nxy = td.Nxy; Nyaw = td.Nyaw; n = td.n; S = td.S; K = P.int_K; mem = P.mem_size;
        inv_k1 = 1.0f / (float)(K + 1);
        CH = wg.size(); CHP = CH + 1; recd = REC_FIELDS * CHP + (CH + 1) / 2;      // (ldsDoubles may have reserved more; only the size matters here)
        // Chunks of whole pieces where that costs no extra chunk.  A chunk of CH = 128 consecutive samples overlaps 8 or 9 pieces of K + 1 = 17 samples; with 9
        // the xy scatter needs a second 16-column MFMA tile for ONE piece -- a full 13-step pass on the wave that also sums the yaw blocks, while the other
        // wave waits.  floor(CH / (K + 1)) whole pieces per chunk (7 x 17 = 119 samples) never need it, and for most piece counts that is the same number
        // of chunks (e.g. 39 pieces: 6 either way; 22 pieces: 4 against 3 -- those keep the plain chunks, a chunk's sample pass costs the same however
        // full it is).  Which samples share a chunk decides the association of the per-chunk sums: results move at rounding level with this choice.
        CHS = CH;
#if !defined(UPH_ALIGNED_CHUNKS) || UPH_ALIGNED_CHUNKS
        {
            const int ppc = CH / (K + 1);
            if (ppc >= 1 && (Nxy + ppc - 1) / ppc == (S + CH - 1) / CH) CHS = ppc * (K + 1);
        }
#endif
        // field stride CH + 1 doubles: with stride CH (1 KB) the field rows of a piece start in the same LDS bank and the
        // scatter's per-(piece, field) lanes conflict
        double* q = lds;
        x = q; q += n; g = q; q += n; d = q; q += n;
        xp = q; q += n; gp = q; q += n;                      // iterate / gradient the line search starts from
        const int nvec = 2 * (Nxy + 5) + (Nyaw + 5);
        bt = q;                                              // sample-time tables (fillTimes), alive from expand() to adjoint()
        gamxy = q; gamyaw = q + 2 * (Nxy + 5);               // gamma (adjoint output) takes the same words afterwards
        q += nvec < Nxy + K + 2 ? Nxy + K + 2 : nvec;
        cxy = q; q += 12 * Nxy; cyaw = q; q += 6 * Nyaw;
        Gxy = q; q += 12 * Nxy; Gyaw = q; q += 6 * Nyaw;
        wtab = q; q += 6 * (K + 1);                          // powers s1^0 .. s1^5 of the K + 1 in-piece sample times: [j][k]
        rec = q;                                             // (wtab sits right before rec: scatterChunk's unmasked batch reads may run past either one's end by a few words)
        {
            size_t rd = (size_t)recd;
            const size_t knd = (size_t)knotBufDoubles(Nxy + 1, 4) + knotBufDoubles(Nyaw + 1, 2), adj = (size_t)knotBufDoubles(Nxy, 4) + knotBufDoubles(Nyaw, 2) + nvec;
            rd = rd < knd ? knd : rd; rd = rd < adj ? adj : rd; rd = rd < (size_t)mem ? (size_t)mem : rd;
            q += rd;
        }
        rtag = (int*)(rec + (size_t)REC_FIELDS * CHP);
        ttab = q; q += THOMAS_DOUBLES;                       // block-LU factors of the knot system
        pf = q; q += MAX_PAST + 8;
        hd = q; q += 18;                                     // head / tail states {P,V,A}: init_xy[6], end_xy[6], init_yaw[3], end_yaw[3]
        dual = bd.dual + 7 * td.off_s; res = bd.res + 7 * td.off_s; scl = bd.scl + 7 * td.off_s;
        hist = bd.hist + td.off_hist; hrow = histRowDoubles(n); hnp = 64 * histNQ(n);
        Wr_xy = bd.ops[td.op_xy].Wr; Wr_yaw = bd.ops[td.op_yaw].Wr;
        rho = 0; scale_fx = 1.0; Txy = Tyaw = 0; last_jerk = 0; hist_reads = 0; evals = 0; trace_n = 0;
        for (int q = 0; q < 16; q++) cyc[q] = 0;
        t_last_eval_end = 0;
        // read by every generate() / adjoint(): one copy per launch of the end states (descriptor) and the factor table (HBM) into LDS
        wg.pfor(18 + THOMAS_DOUBLES, [&](int t) {
            if (t < 18) hd[t] = t < 6 ? td.init_xy[t] : (t < 12 ? td.end_xy[t - 6] : (t < 15 ? td.init_yaw[t - 12] : td.end_yaw[t - 15]));
            else ttab[t - 18] = bd.thomas[t - 18];
        });
    }

    // optional diagnostic: cost after every accepted L-BFGS iteration (-1 marks the start of an ALM pass); off when bd.trace == nullptr
    UPH_HD void tracePush(double v) {
        if (bd.trace != nullptr && trace_n < bd.trace_cap) {
            const int at = trace_n;
            wg.one([&]() { bd.trace[(size_t)bidx * bd.trace_cap + at] = v; });
        }
        trace_n++;
    }

    // floor(a / b) for 0 <= a < 2^22, b > 0, given inv = 1.0f / b: float quotient, then one exact integer correction step each way
    static UPH_HD int divSmall(int a, int b, float inv) {
        int q = (int)((float)a * inv);
        const int r = a - q * b;
        if (r < 0) q--;
        else if (r >= b) q++;
        return q;
    }
    // column of beta = [P0, T V0, T^2 A0 | way-points | PN, T VN, T^2 AN] that holds the position of knot j
    static UPH_HD int knotCol(int j, int N) { return j == 0 ? 0 : (j == N ? N + 2 : j + 2); }
    // position of knot j (0..N) of dimension dd (0, 1: xy; 2: yaw): an end state or a way-point of x
    UPH_HD double knotPos(const double* xin, int j, int dd) const {
        if (dd < 2) return j == 0 ? hd[dd] : (j == Nxy ? hd[6 + dd] : xin[1 + 2 * (j - 1) + dd]);
        return j == 0 ? hd[12] : (j == Nyaw ? hd[15] : xin[1 + 2 * (Nxy - 1) + (j - 1)]);
    }

    // ------------------------------------------------------------------ small vector helpers
    UPH_HD double dot(const double* a, const double* b, int m) {
        double r[1];
        wg.template sum<1>(m, r, [&](int i, double* acc) { acc[0] += a[i] * b[i]; });
        return r[0];
    }
    UPH_HD double absmax(const double* a, int m) {
        return wg.maxv(m, [&](int i) { return fabs(a[i]); });
    }

    // ------------------------------------------------------------------ MINCO generate (se2traj.hpp:595-680), part 1: knot states
    // (v_j, a_j) of the interior knots from the block-tridiagonal jerk / snap continuity system (minco_op_host.hpp): right-hand
    // sides r_j = (20 (dl+ - dl-), -15 (dl+ + dl-)) in parallel -- the known end states z_0, z_N moved to the right (- A z_0 at
    // knot 1, - C z_N at knot N-1) --, then ONE fused forward / backward sweep with the 2x2 block factors (wg.thomas).  The knot
    // states [(N+1)][v,a][dim] sit in the record buffer, which is idle until the first sample chunk.
    // STEP: the line search's trial point x = xp + st d is formed here as well (every lane computes the positions it needs from xp and
    // d and stores its own knot), which spares the search a pass and a barrier per trial.
    template <bool STEP>
    UPH_HD void generate(double* xin, double st) {
        UPH_MARK("generate");
        const long long tsub_start = wg.clock();
        const double tau_ = STEP ? fma(st, d[0], xp[0]) : xin[0];
        const double Ttot = expC2(tau_);
        Txy = wg.bcast(Ttot / (double)Nxy);       // calTfromTau, alm_traj_opt.h:257-261 (wave-uniform: kept in scalar registers)
        Tyaw = wg.bcast(Ttot / (double)Nyaw);
        ec_iTyaw = wg.bcast(1.0 / Tyaw);
        const double Tx = Txy, Ty = Tyaw;
        double* zxy = rec;                                   // knot j at knotOffJ(j, ks): padded per lane block of the solve (uph_common.hpp)
        double* zyaw = rec + knotBufDoubles(Nxy + 1, 4);
        wg.pfor(2 * (Nxy + 1) + (Nyaw + 1), [&](int t) {
            const bool isxy = t < 2 * (Nxy + 1);
            const int j = isxy ? (t >> 1) : t - 2 * (Nxy + 1), dd = isxy ? (t & 1) : 2, N = isxy ? Nxy : Nyaw, os = isxy ? 2 : 1;
            const double T1 = isxy ? Tx : Ty;
            const double* h0 = isxy ? hd + (dd & 1) : hd + 12;       // {P, V, A} of the head at stride os; tail 6 (xy) / 3 (yaw) doubles further
            const double* h1 = isxy ? hd + 6 + (dd & 1) : hd + 15;
            double* z = isxy ? zxy + knotOffJ(j, 4) + (dd & 1) : zyaw + knotOffJ(j, 2);      // component stride os
            if (STEP && t == 0) xin[0] = tau_;
            if (j == 0) { z[0] = T1 * h0[os]; z[os] = T1 * T1 * h0[2 * os]; }
            else if (j == N) { z[0] = T1 * h1[os]; z[os] = T1 * T1 * h1[2 * os]; }
            else {
                double pm, p0, pp;
                if (STEP) {
                    const int ix = isxy ? 1 + 2 * (j - 1) + dd : 1 + 2 * (Nxy - 1) + (j - 1), ws = isxy ? 2 : 1;     // x index of knot j, stride between knots
                    p0 = fma(st, d[ix], xp[ix]);
                    pm = j == 1 ? h0[0] : fma(st, d[ix - ws], xp[ix - ws]);
                    pp = j == N - 1 ? h1[0] : fma(st, d[ix + ws], xp[ix + ws]);
                    xin[ix] = p0;
                } else { pm = knotPos(xin, j - 1, dd); p0 = knotPos(xin, j, dd); pp = knotPos(xin, j + 1, dd); }
                const double dp = pp - p0, dm = p0 - pm;
                double r0 = 20.0 * (dp - dm), r1 = -15.0 * (dp + dm);
                if (j == 1) { const double v0 = T1 * h0[os], a0 = T1 * T1 * h0[2 * os]; r0 += 8.0 * v0 + a0; r1 += 7.0 * v0 + a0; }
                if (j == N - 1) { const double vN = T1 * h1[os], aN = T1 * T1 * h1[2 * os]; r0 += -8.0 * vN + aN; r1 += 7.0 * vN - aN; }
                z[0] = r0; z[os] = r1;
            }
        });
        const long long tsub1 = wg.clock();
        wg.thomas(ttab, false, zyaw + knotOffJ(1, 2), Nyaw - 1, zxy + knotOffJ(1, 4), Nxy - 1);
        if (sub_t) { sub_t[0] += tsub1 - tsub_start; sub_t[1] += wg.clock() - tsub1; }
    }

    // ------------------------------------------------------------------ MINCO generate, part 2, fused with the jerk terms
    // One lane per (piece, dimension): quintic Hermite expansion from the piece's end states (normalised time), c_k = c~_k T^-k,
    // the piece's share of the jerk energy and of its direct T-derivative (se2traj.hpp:697-710, 736-745), and G <- jerk_w dJ/dc
    // (:722-734) -- the coefficients are still in registers.  Two more lanes fill the sample-time tables.
    // out[0] = energy_xy + energy_yaw, out[1] = sum_i gdT_xy(i), out[2] = sum_i gdT_yaw(i)   (unscaled)
    UPH_HD void expand(const double* xin, double jerk_w, double out[3]) {
        UPH_MARK("expand");
        const long long t0 = wg.clock();
        const double Tx = Txy, Ty = Tyaw, itx = wg.bcast(1.0 / Tx), ity = wg.bcast(1.0 / Ty);
        const double* zxy = rec;
        const double* zyaw = rec + knotBufDoubles(Nxy + 1, 4);
        const int np = 2 * Nxy + Nyaw;
        wg.template sum<3>(np + 2, out, [&](int t, double* acc) {
            if (t >= np) { fillTimes(t - np); return; }
            double p0, p1, v0, a0, v1, a1, it_, T1;
            double *oc, *og;
            int os;
            const bool isxy = t < 2 * Nxy;
            if (isxy) {
                const int i = t >> 1, dd = t & 1;
                p0 = knotPos(xin, i, dd); p1 = knotPos(xin, i + 1, dd);
                const double *k0 = zxy + knotOffJ(i, 4) + dd, *k1 = zxy + knotOffJ(i + 1, 4) + dd;
                v0 = k0[0]; a0 = k0[2]; v1 = k1[0]; a1 = k1[2];
                oc = cxy + 12 * i + dd; og = Gxy + 12 * i + dd; os = 2; it_ = itx; T1 = Tx;
            } else {
                const int m = t - 2 * Nxy;
                p0 = knotPos(xin, m, 2); p1 = knotPos(xin, m + 1, 2);
                const double *k0 = zyaw + knotOffJ(m, 2), *k1 = zyaw + knotOffJ(m + 1, 2);
                v0 = k0[0]; a0 = k0[1]; v1 = k1[0]; a1 = k1[1];
                oc = cyaw + 6 * m; og = Gyaw + 6 * m; os = 1; it_ = ity; T1 = Ty;
            }
            const double dl = p1 - p0;
            const double h3 = 10.0 * dl - 6.0 * v0 - 4.0 * v1 - 1.5 * a0 + 0.5 * a1;
            const double h4 = -15.0 * dl + 8.0 * v0 + 7.0 * v1 + 1.5 * a0 - a1;
            const double h5 = 6.0 * dl - 3.0 * v0 - 3.0 * v1 - 0.5 * a0 + 0.5 * a1;
            const double i2 = it_ * it_, i3 = i2 * it_, i4 = i3 * it_, i5 = i4 * it_;
            const double c3 = h3 * i3, c4 = h4 * i4, c5 = h5 * i5;
            oc[0] = p0; oc[os] = v0 * it_; oc[2 * os] = (0.5 * a0) * i2;
            oc[3 * os] = c3; oc[4 * os] = c4; oc[5 * os] = c5;
            const double T2 = T1 * T1, T3 = T2 * T1, T4 = T2 * T2, T5 = T4 * T1;
            og[0] = 0.0; og[os] = 0.0; og[2 * os] = 0.0;
            og[3 * os] = jerk_w * (72.0 * c3 * T1 + 144.0 * c4 * T2 + 240.0 * c5 * T3);
            og[4 * os] = jerk_w * (144.0 * c3 * T2 + 384.0 * c4 * T3 + 720.0 * c5 * T4);
            og[5 * os] = jerk_w * (240.0 * c3 * T3 + 720.0 * c4 * T4 + 1440.0 * c5 * T5);
            const double d33 = c3 * c3, d43 = c4 * c3, d44 = c4 * c4, d53 = c5 * c3, d54 = c5 * c4, d55 = c5 * c5;
            acc[0] += 36.0 * d33 * T1 + 144.0 * d43 * T2 + 192.0 * d44 * T3 + 240.0 * d53 * T3 + 720.0 * d54 * T4 + 720.0 * d55 * T5;
            const double gT = 36.0 * d33 + 288.0 * d43 * T1 + 576.0 * d44 * T2 + 720.0 * d53 * T2 + 2880.0 * d54 * T3 + 3600.0 * d55 * T4;
            acc[1] += isxy ? gT : 0.0;          // (selects, not `acc[isxy ? 1 : 2]`: a dynamically indexed accumulator array would live in scratch)
            acc[2] += isxy ? 0.0 : gT;
        });
        if (sub_t) sub_t[2] += wg.clock() - t0;
    }

    // ------------------------------------------------------------------ per-sample kinematics + terrain
    // the grid descriptor as this trajectory's lookups see it: the map's, or -- far from the map's origin -- its local frame's (TrajFrame)
    UPH_HD GridDev framedGrid() const {
        if (bd.grid_mem != nullptr && bd.grid_per_traj != 0) return bd.grid_mem[bidx];
        return grid;
    }

    template <class R>
    struct KinT {
        R b0[6], b1[6], b2[6], b3[6];
        R y0[6], y1[6], y2[6];
        R pos[2], vel[2], acc[2], jer[2];
        R yaw, dyaw, d2yaw, cyaw, syaw, v_norm, lon_acc, lat_acc, u, s1;
        R tv[7], tg[7][3];
        R vx, wz, ax, ay, curv_snorm, den, sq;
        R yawn, cw, sw;                 // wrapped yaw and its cos / sin (uneven_map.h:329-330)
        R zx, zy, gs[3], gzx[3], gzy[3]; // interpolated zb and the base gradients (penalty path)
        int yaw_idx;
    };
    typedef KinT<double> Kin;
    // penalty path: only (sigma, zb) and their gradients are gathered; the seven attitude terms follow from them here and their
    // gradients are never formed individually (sampleEval folds them into four scalar coefficients)
    template <class R>
    UPH_HD void terrainValuesOnly(KinT<R>& S_) const {
        R sg;
#if defined(__HIP_DEVICE_COMPILE__) && UPH_GRID_FROM_MEM
        // the grid descriptor through an opaque constant-address-space pointer: scalar loads issued here, nothing for the compiler to
        // keep live (and spill to VGPR lanes) across the solver's loops
        typedef const __attribute__((address_space(4))) unsigned long long* gw_t;
        // (one descriptor shared by the batch, or -- local frames, uph_common.hpp TrajFrame -- this trajectory's own: the same loads off another base)
        gw_t T = (gw_t)(const void*)(bd.grid_mem + (size_t)bidx * bd.grid_per_traj);
        asm volatile("" : "+s"(T));
        constexpr int NWORD = (int)(sizeof(GridDev) / 8);
        unsigned long long w[NWORD];
#pragma unroll
        for (int k = 0; k < NWORD; k++) w[k] = T[k];
        GridDev gl;
        __builtin_memcpy(&gl, w, sizeof(GridDev));
        terrainBase<R>(gl, S_.pos[0], S_.pos[1], S_.yawn, sg, S_.zx, S_.zy, S_.gs, S_.gzx, S_.gzy);
#else
        const GridDev gl = framedGrid();
        terrainBase<R>(gl, S_.pos[0], S_.pos[1], S_.yawn, sg, S_.zx, S_.zy, S_.gs, S_.gzx, S_.gzy);
#endif
        const R zx = S_.zx, zy = S_.zy;
        const R cc = sqrt(1.0 - zx * zx - zy * zy);                 // uneven_map.h:327-348
        const R inv_c = 1.0 / cc;
        const R t = S_.cw * zx + S_.sw * zy;
        const R s = -(-S_.sw * zx + S_.cw * zy);
        const R sq = sqrt(1.0 - t * t);
        const R r = 1.0 / sq;
        S_.sq = sq;
        S_.tv[0] = r; S_.tv[1] = -cc * t * r; S_.tv[2] = sq * inv_c; S_.tv[3] = s * r; S_.tv[4] = cc; S_.tv[5] = inv_c; S_.tv[6] = sg;
    }
    template <bool WITH_GRADS = true, class R = double>
    UPH_HD void kin(int i, int j, KinT<R>& S_) const {
        const R s1 = bt[Nxy + 1 + j];                              // :713-714,987: the s1 += step accumulation (Q2), tabulated by generate()
        S_.s1 = s1;
        const R s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;   // :734-741
        S_.b0[0] = 1.0; S_.b0[1] = s1; S_.b0[2] = s2; S_.b0[3] = s3; S_.b0[4] = s4; S_.b0[5] = s5;
        S_.b1[0] = 0.0; S_.b1[1] = 1.0; S_.b1[2] = 2.0 * s1; S_.b1[3] = 3.0 * s2; S_.b1[4] = 4.0 * s3; S_.b1[5] = 5.0 * s4;
        S_.b2[0] = 0.0; S_.b2[1] = 0.0; S_.b2[2] = 2.0; S_.b2[3] = 6.0 * s1; S_.b2[4] = 12.0 * s2; S_.b2[5] = 20.0 * s3;
        S_.b3[0] = 0.0; S_.b3[1] = 0.0; S_.b3[2] = 0.0; S_.b3[3] = 6.0; S_.b3[4] = 24.0 * s1; S_.b3[5] = 60.0 * s2;
        const double* c = cxy + 12 * i;
#pragma unroll
        for (int dd = 0; dd < 2; dd++) {                                // :742-745
            R a = R(0.0), b = R(0.0), cc = R(0.0), e = R(0.0);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const R cv = c[k * 2 + dd];
                a += cv * S_.b0[k]; b += cv * S_.b1[k]; cc += cv * S_.b2[k]; e += cv * S_.b3[k];
            }
            if (UPH_DIET && !WITH_GRADS) { pinv(a); pinv(b); pinv(cc); pinv(e); }      // formed HERE: the coefficients die before the gather
            S_.pos[dd] = a; S_.vel[dd] = b; S_.acc[dd] = cc; S_.jer[dd] = e;
        }
        const R now_time = s1 + bt[i];                             // :748-753
        int yi = toInt<R>(divR(now_time, R(Tyaw), R(ec_iTyaw)));
        if (yi >= Nyaw) yi = Nyaw - 1;
        if (yi < 0) yi = 0;                                             // (cannot happen for finite positive times; keeps indices in range)
        S_.yaw_idx = yi;
        const R u1 = now_time - yi * Tyaw;
        S_.u = u1;
        const R u2 = u1 * u1, u3 = u2 * u1, u4 = u2 * u2, u5 = u4 * u1;
        S_.y0[0] = 1.0; S_.y0[1] = u1; S_.y0[2] = u2; S_.y0[3] = u3; S_.y0[4] = u4; S_.y0[5] = u5;
        S_.y1[0] = 0.0; S_.y1[1] = 1.0; S_.y1[2] = 2.0 * u1; S_.y1[3] = 3.0 * u2; S_.y1[4] = 4.0 * u3; S_.y1[5] = 5.0 * u4;
        S_.y2[0] = 0.0; S_.y2[1] = 0.0; S_.y2[2] = 2.0; S_.y2[3] = 6.0 * u1; S_.y2[4] = 12.0 * u2; S_.y2[5] = 20.0 * u3;
        const double* cy = cyaw + 6 * yi;
        R yaw = R(0.0), dyaw = R(0.0), d2yaw = R(0.0);                            // :762-764
#pragma unroll
        for (int k = 0; k < 6; k++) { const R cv = cy[k]; yaw += cv * S_.y0[k]; dyaw += cv * S_.y1[k]; d2yaw += cv * S_.y2[k]; }
        if (UPH_DIET && !WITH_GRADS) { pinv(yaw); pinv(dyaw); pinv(d2yaw); }
        S_.yaw = yaw; S_.dyaw = dyaw; S_.d2yaw = d2yaw;
        const R yawn = normSO2(yaw);                               // :767-770
        sincosFast(yaw, S_.syaw, S_.cyaw);                              // one argument reduction for both
        // cos / sin of the WRAPPED yaw (uneven_map.h:329-330): yawn = yaw - 2 pi k differs from yaw by a rounding of ~1e-16 |yaw|,
        // so the same pair serves (the reference evaluates cos / sin a second time on the wrapped value)
        const R cw = S_.cyaw, sw = S_.syaw;
        S_.v_norm = sqrt(S_.vel[0] * S_.vel[0] + S_.vel[1] * S_.vel[1]);   // :771-775
        S_.lon_acc = S_.acc[0] * S_.cyaw + S_.acc[1] * S_.syaw;
        S_.lat_acc = S_.acc[0] * (-S_.syaw) + S_.acc[1] * S_.cyaw;
        S_.yawn = yawn; S_.cw = cw; S_.sw = sw;
        UPH_MARK("kin.terrain");
        if constexpr (WITH_GRADS) terrainAllWithGrad(framedGrid(), S_.pos[0], S_.pos[1], yawn, cw, sw, S_.tv, S_.tg);   // :778  (initScaling: R = double)
        else terrainValuesOnly<R>(S_);
        UPH_MARK("kin.after_terrain");
        S_.vx = S_.v_norm * S_.tv[0];                                   // :813-817
        S_.wz = dyaw * S_.tv[5];
        S_.ax = S_.lon_acc * S_.tv[0] + grid.gravity * S_.tv[1];
        S_.ay = S_.lat_acc * S_.tv[2] + grid.gravity * S_.tv[3];
        S_.den = 1.0 / (S_.vx * S_.vx + delta_sigl);
        S_.curv_snorm = divR(S_.wz * S_.wz, R(S_.vx * S_.vx + delta_sigl), S_.den);
    }

    template <class R> UPH_HD R augCost(R h, R lm) const { return h * (lm + 0.5 * rho * h); }   // alm_traj_opt.h:153-163
    template <class R> UPH_HD R augGrad(R h, R lm) const { return rho * h + lm; }

    // What the sample leaves for the per-piece reduction (alm_traj_opt.cpp:969-979):
    //   rec[0..5]  = grad_p, grad_v, grad_a (2 each): scatterChunk applies the basis weights beta0/1/2 of the sample's in-piece time
    //                (beta0_k = s1^k, beta1_k = k s1^(k-1), beta2_k = k (k-1) s1^(k-2)), which depend on j only: the powers are
    //                tabulated once per evaluation (wtab, written by the 17 samples of piece 0)
    //   rec[6+k]   = beta0_k(u) grad_yaw + beta1_k(u) grad_dyaw   -> gdCyaw block of its yaw piece (grad_d2yaw == 0, Q8)
    //   rtag       = that yaw piece (int32)
    template <class R>
    UPH_HD void putRec(int slot, int i, int j, const R gp_[2], const R gv_[2], const R ga_[2], R gyaw, R gdyaw, const KinT<R>& k) {
        rec[0 * CHP + slot] = gp_[0]; rec[1 * CHP + slot] = gp_[1];
        rec[2 * CHP + slot] = gv_[0]; rec[3 * CHP + slot] = gv_[1];
        rec[4 * CHP + slot] = ga_[0]; rec[5 * CHP + slot] = ga_[1];
        R u1 = k.u;
        if (UPH_DIET) pinv(u1);                          // (the powers are REBUILT here: not carried from kin() across the gather and the penalties)
        const R u2 = u1 * u1, u3 = u2 * u1, u4 = u2 * u2, u5 = u4 * u1;
        rec[6 * CHP + slot] = gyaw;
        rec[7 * CHP + slot] = (u1 * gyaw + gdyaw);
        rec[8 * CHP + slot] = (u2 * gyaw + 2.0 * u1 * gdyaw);
        rec[9 * CHP + slot] = (u3 * gyaw + 3.0 * u2 * gdyaw);
        rec[10 * CHP + slot] = (u4 * gyaw + 4.0 * u3 * gdyaw);
        rec[11 * CHP + slot] = (u5 * gyaw + 5.0 * u4 * gdyaw);
        rtag[slot] = k.yaw_idx;
        if (i == 0) {                                    // the powers behind beta0/1/2 of alm_traj_opt.cpp:738-740 at s1(j) (rebuilt from s1: not kept live across the sample)
            double* w = wtab + 6 * j;
            R s1 = k.s1;
            if (UPH_DIET) pinv(s1);
            const R s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
            w[0] = 1.0; w[1] = s1; w[2] = s2; w[3] = s3; w[4] = s4; w[5] = s5;
        }
    }

    // one constraint sample of calConstrainCostGrad (alm_traj_opt.cpp:716-988).  acc[0] += cost, acc[1] += gdTxy part, acc[2] += gdTyaw part.
    // The residuals hx / gx (alm_traj_opt.cpp:835, 846 ...) are consumed only by the dual update after an L-BFGS pass, so the
    // evaluations of the pass do not store them (7 stores per sample and their drain at the chunk barrier): RES = 1 is the
    // same code up to the residuals, run once over the last evaluated trajectory when the pass ends (refreshResiduals).
    // RES: 0 = cost and gradient, no residual stores (the solve's evaluations); 1 = residuals only; 2 = both -- calConstrainCostGrad as the
    // reference runs it, hx / gx written by every call (penaltyOnly, the A5-alone measurement entry).
    template <int RES, class R = double>
    UPH_HD void sampleEval(int s, int slot, double* acc) {
        constexpr bool RES_ONLY = RES == 1;
        UPH_MARK("sampleEval");
        const int i = divSmall(s, K + 1, inv_k1), j = s - i * (K + 1);
        // all 14 dual / scale operands are fetched up front: they are independent of the kinematics, their HBM/L2 latency overlaps
        // the polynomial evaluation and the terrain gather instead of serialising round trips
        R dl[7], sc7[7];
#if !defined(UPH_LATE_DUALS) || !UPH_LATE_DUALS
#pragma unroll
        for (int q = 0; q < 7; q++) { dl[q] = R(RES_ONLY ? 0.0 : dual[q * S + s]); sc7[q] = R(scl[q * S + s]); }
#endif
        KinT<R> k;
        kin<false, R>(i, j, k);
        UPH_MARK("sampleEval.penalties");
#if defined(UPH_LATE_DUALS) && UPH_LATE_DUALS
        {
            int s_ = s;
            asm volatile("" : "+v"(s_));           // (opaque: the loads cannot be hoisted above the gather)
#pragma unroll
            for (int q = 0; q < 7; q++) { dl[q] = R(RES_ONLY ? 0.0 : dual[q * S + s_]); sc7[q] = R(scl[q * S + s_]); }
        }
#endif
        const R icvx = k.tv[0], icvy = k.tv[2], cos_xi = k.tv[4], icxi = k.tv[5], sigma = k.tv[6];
        const R vx = k.vx, wz = k.wz, ax = k.ax, ay = k.ay, curv = k.curv_snorm;
        const R nh0 = k.syaw, nh1 = -k.cyaw;
        // residuals: non-holonomic :830-835, then g1..g6 :841-946 (Q6: without use_scaling only curvature and sigma take fixed scales)
#if defined(__HIP_DEVICE_COMPILE__) && UPH_GRID_FROM_MEM
        // the limits through an opaque constant-address-space pointer (scalar loads here instead of SGPRs held, and spilled, across the solve)
        typedef const __attribute__((address_space(4))) unsigned long long* pw_t;
        pw_t PT = (pw_t)(const void*)bd.params_mem;
        asm volatile("" : "+s"(PT));
        constexpr int PWORD = (int)(sizeof(OptParams) / 8);
        unsigned long long pw[PWORD];
#pragma unroll
        for (int q = 0; q < PWORD; q++) pw[q] = PT[q];
        OptParams Pm;
        __builtin_memcpy(&Pm, pw, sizeof(OptParams));
#else
        const OptParams& Pm = P;
#endif
        const R h = (k.vel[0] * nh0 + k.vel[1] * nh1) * sc7[0];
        const R g1 = (vx * vx - Pm.max_vel2) * sc7[1];
        const R g2 = (ax * ax - Pm.max_acc_lon2) * sc7[2];
        const R g3 = (ay * ay - Pm.max_acc_lat2) * sc7[3];
        const R sc4 = Pm.use_scaling ? sc7[4] : R(cur_scale);
        const R g4 = (curv - Pm.max_kap2) * sc4;
        const R g5 = (Pm.min_cxi - cos_xi) * sc7[5];
        const R sc6 = Pm.use_scaling ? sc7[6] : R(sig_scale);
        const R g6 = (sigma - Pm.max_sig) * sc6;
        if (RES != 0) {
            res[0 * S + s] = h; res[1 * S + s] = g1; res[2 * S + s] = g2; res[3 * S + s] = g3;
            res[4 * S + s] = g4; res[5 * S + s] = g5; res[6 * S + s] = g6;
            if (RES_ONLY) return;
        }
        const R alpha = ec_invK * j;                               // :718  (1.0 / K * j)
        const R gravity = grid.gravity;
        R grad_p[2] = {R(0.0), R(0.0)}, grad_v[2] = {R(0.0), R(0.0)}, grad_a[2] = {R(0.0), R(0.0)};
        R grad_yaw = R(0.0), grad_dyaw = R(0.0), grad_vx2 = R(0.0), grad_wz = R(0.0), grad_ax = R(0.0), grad_ay = R(0.0);
        R grad_se2[3] = {R(0.0), R(0.0), R(0.0)};
        // weights of the gradients of the seven terrain terms (invCosVphix, sinPhix, invCosVphiy, sinPhiy, cosXi, invCosXi, sigma):
        // grad_se2 = sum_q W[q] * grad(term_q); the sum is folded onto the three base gradients at the end
        R W[7] = {R(0.0), R(0.0), R(0.0), R(0.0), R(0.0), R(0.0), R(0.0)};
        R aug_grad, cost = R(0.0);
        const R irho = ec_irho;
        // user-defined cost: surface variation                          :819-827
        const R omega = R((j == 0 || j == K) ? ec_omega_h : ec_omega);     // (0.5 *) rho_ter * step * scale_fx
        const R user_cost = omega * sigma * sigma;
        cost += user_cost;
        W[6] += omega * sigma * 2.0;
        R tx = user_cost / K;                                      // Q3
        // non-holonomic                                                 :829-838
        {
            const R lm = dl[0], sc = sc7[0];
            cost += augCost<R>(h, lm);
            const R ng = augGrad<R>(h, lm) * sc;
            grad_v[0] += ng * nh0; grad_v[1] += ng * nh1;
            grad_yaw += ng * (k.vel[0] * k.cyaw + k.vel[1] * k.syaw);
        }
        // longitude velocity                                            :840-854
        {
            const R mu = dl[1], sc = sc7[1], gv = g1;
            if (rho * gv + mu > 0) { cost += augCost<R>(gv, mu); aug_grad = augGrad<R>(gv, mu) * sc; grad_vx2 += aug_grad; }
            else cost += divR(R(-0.5 * mu * mu), R(rho), irho);
        }
        // longitude acceleration                                        :856-870
        {
            const R mu = dl[2], sc = sc7[2], gv = g2;
            if (rho * gv + mu > 0) { cost += augCost<R>(gv, mu); aug_grad = augGrad<R>(gv, mu) * sc; grad_ax += aug_grad * 2.0 * ax; }
            else cost += divR(R(-0.5 * mu * mu), R(rho), irho);
        }
        // latitude acceleration                                         :872-886
        {
            const R mu = dl[3], sc = sc7[3], gv = g3;
            if (rho * gv + mu > 0) { cost += augCost<R>(gv, mu); aug_grad = augGrad<R>(gv, mu) * sc; grad_ay += aug_grad * 2.0 * ay; }
            else cost += divR(R(-0.5 * mu * mu), R(rho), irho);
        }
        // curvature                                                     :888-910  (Q6)
        {
            const R mu = dl[4], sc = sc4, gv = g4;
            if (rho * gv + mu > 0) {
                const R den = k.den;
                cost += augCost<R>(gv, mu);
                aug_grad = augGrad<R>(gv, mu) * sc;
                grad_wz += aug_grad * den * 2.0 * wz;
                grad_vx2 -= aug_grad * curv * den;
            } else cost += divR(R(-0.5 * mu * mu), R(rho), irho);
        }
        // attitude                                                      :912-925
        {
            const R mu = dl[5], sc = sc7[5], gv = g5;
            if (rho * gv + mu > 0) {
                cost += augCost<R>(gv, mu);
                const R ag = augGrad<R>(gv, mu);
                W[4] -= ag * sc;
            } else cost += divR(R(-0.5 * mu * mu), R(rho), irho);
        }
        // surface variation                                             :927-946  (Q6)
        {
            const R mu = dl[6], sc = sc6, gv = g6;
            if (rho * gv + mu > 0) {
                cost += augCost<R>(gv, mu);
                const R ag = augGrad<R>(gv, mu);
                W[6] += ag * sc;
            } else cost += divR(R(-0.5 * mu * mu), R(rho), irho);
        }
        // process with vx, wz, ax                                       :948-964
#pragma unroll
        for (int q = 0; q < 2; q++) grad_v[q] += grad_vx2 * icvx * icvx * 2.0 * k.vel[q];
        W[0] += grad_vx2 * k.v_norm * k.v_norm * 2.0 * icvx;
        grad_dyaw += grad_wz * icxi;
        W[5] += grad_wz * k.dyaw;
        grad_a[0] += grad_ax * icvx * k.cyaw; grad_a[1] += grad_ax * icvx * k.syaw;
        grad_yaw += grad_ax * icvx * k.lat_acc;
        W[1] += grad_ax * gravity; W[0] += grad_ax * k.lon_acc;
        grad_a[0] += grad_ay * icvy * (-k.syaw); grad_a[1] += grad_ay * icvy * k.cyaw;
        grad_yaw -= grad_ay * icvy * k.lon_acc;
        W[3] += grad_ay * gravity; W[2] += grad_ay * k.lat_acc;
        {
            // grad(term_q) in terms of dt = grad(t), ds = grad(s), gc = grad(c), gs = grad(sigma)  (uneven_map.h:338-355):
            //   g0 = t r^3 dt   g1 = -(t r gc + r^3 c dt)   g2 = -(t r dt / c + sq gc / c^2)   g3 = r ds + t r^3 s dt   g4 = gc   g5 = -gc / c^2   g6 = gs
            const R cc = k.tv[4], inv_c = k.tv[5], r = k.tv[0];
            const R t = k.cw * k.zx + k.sw * k.zy, s_ = -(-k.sw * k.zx + k.cw * k.zy);
            const R sq = k.sq, r3 = r * r * r;
            const R Cdt = W[0] * t * r3 - W[1] * r3 * cc - W[2] * inv_c * t * r + W[3] * t * r3 * s_;
            const R Cgc = -W[1] * t * r - W[2] * inv_c * inv_c * sq + W[4] - W[5] * inv_c * inv_c;
            const R Cds = W[3] * r;
            // dt = gzx cw + gzy sw (yaw: - s),  ds = gzx sw - gzy cw (yaw: + t),  gc = -(gzx zx + gzy zy) / c
            const R ax_ = Cdt * k.cw + Cds * k.sw - Cgc * k.zx * inv_c;      // coefficient of grad(zb.x)
            const R ay_ = Cdt * k.sw - Cds * k.cw - Cgc * k.zy * inv_c;      // coefficient of grad(zb.y)
#pragma unroll
            for (int q = 0; q < 3; q++) grad_se2[q] = ax_ * k.gzx[q] + ay_ * k.gzy[q] + W[6] * k.gs[q];
            grad_se2[2] += -Cdt * s_ + Cds * t;
        }
        grad_p[0] += grad_se2[0]; grad_p[1] += grad_se2[1];
        grad_yaw += grad_se2[2];
        // scatter terms (:966-985): the C-blocks are reduced per piece in scatter(); the T parts are summed here
        putRec<R>(slot, i, j, grad_p, grad_v, grad_a, grad_yaw, grad_dyaw, k);
        tx += ((grad_p[0] * k.vel[0] + grad_p[1] * k.vel[1]) + (grad_v[0] * k.acc[0] + grad_v[1] * k.acc[1]) +
               (grad_a[0] * k.jer[0] + grad_a[1] * k.jer[1])) * alpha;
        const R yawdot = (grad_yaw * k.dyaw + grad_dyaw * k.d2yaw);
        tx += yawdot * (alpha + i);
        acc[0] += cost;
        acc[1] += tx;
        acc[2] += -yawdot * k.yaw_idx;
    }

    // objective-only sample of initScaling (alm_traj_opt.cpp:507-519): rho_ter * int sigma^2, no scale_fx
    UPH_HD void sampleObjective(int s, int slot, double* acc) {
        const int i = divSmall(s, K + 1, inv_k1), j = s - i * (K + 1);
        Kin k;
        kin<false>(i, j, k);
        const double alpha = 1.0 / K * j;
        const double step = Txy / K;
        const double sigma = k.tv[6];
        const double omega = (j == 0 || j == K) ? 0.5 * P.rho_ter * step : P.rho_ter * step;
        const double user_cost = omega * sigma * sigma;
        double gse2[3];
#pragma unroll
        for (int q = 0; q < 3; q++) gse2[q] = omega * k.gs[q] * sigma * 2.0;
        const double zero2[2] = {0, 0};
        putRec(slot, i, j, gse2, zero2, zero2, gse2[2], 0.0, k);
        acc[0] += user_cost;
        acc[1] += user_cost / K + (gse2[0] * k.vel[0] + gse2[1] * k.vel[1]) * alpha + (gse2[2] * k.dyaw) * (alpha + i);
        acc[2] += -(gse2[2] * k.dyaw) * k.yaw_idx;
    }

    // ------------------------------------------------------------------ per-piece reduction of the sample records into dK/dc
    // G = jerk_w * dJ/dc (expand)  +  sum over samples of (beta0 (x) grad_p + beta1 (x) grad_v + beta2 (x) grad_a)   (:969-979).
    // Samples are produced in chunks of CH (= workgroup size) records; each chunk is folded into G right away.
    // sample-time tables of calConstrainCostGrad, built by the reference's own accumulations so that the roundings agree:
    //   bt[i] = base_time of piece i       (base += T1(i), alm_traj_opt.cpp:709,989)       i = 0..Nxy
    //   bt[Nxy+1+j] = in-piece time s1     (s1 += step, :713-714,987; Q2)                  j = 0..K
    // Alive from expand() until adjoint() writes gamma into the same words.
    UPH_HD void fillTimes(int u) {
        if (u == 0) {
            double base = 0.0;
            for (int i = 0; i <= Nxy; i++) { bt[i] = base; base += Txy; }
        } else {
            const double step = Txy / K;
            double s1 = 0.0;
            for (int j = 0; j <= K; j++) { bt[Nxy + 1 + j] = s1; s1 += step; }
        }
    }
    // fold the records of samples [s0, s0+cnt) into G (fixed summation order, no atomics).  The xy outputs and the yaw outputs go to
    // DIFFERENT waves so that neither wave runs both bodies: lanes [0, XYL) own (xy piece, dim, k-pair) -- three lanes per (piece,
    // dim), each summing its two k over the piece's <= K+1 samples in slot order -- and the lanes after them own (yaw piece, k-pair),
    // each summing its two record fields over the candidate slots whose tag names its piece.  XYL is a multiple of the wave size.
    // Every LDS operand is read at base + constant without clamping: reads next to a piece's slots land in neighbouring words of the
    // same allocation (wtab sits right before rec) and are discarded by the selects.
    UPH_HD void scatterChunk(int s0, int cnt) {
        UPH_MARK("scatterChunk");
        const int K1 = K + 1;
        const double xr = (double)Nxy / (double)Nyaw;       // xy pieces per yaw piece
        const int i0 = s0 / K1, i1 = (s0 + cnt - 1) / K1;
        const int nxyt = 6 * (i1 - i0 + 1);
        const int XYL = (nxyt + 63) & ~63;
        // yaw pieces that can receive samples of this chunk: from the first to the last sample's piece (monotone up to round-off) +-1
        int m0 = rtag[0] - 1, m1 = rtag[cnt - 1] + 1;
        if (m0 < 0) m0 = 0;
        if (m1 > Nyaw - 1) m1 = Nyaw - 1;
        // xy outputs: lane t owns (piece, dim, k-pair)
        auto xyTask = [&](int t) {
            const int pi = t / 6, r = t - 6 * pi, dd = r & 1, kp = r >> 1;      // r = 2 kp + dd
            const int i = i0 + pi;
            const int ja = i * K1 - s0;                  // slot of the piece's sample j = 0 (negative when the piece began in the previous chunk)
            const int jlo = ja < 0 ? -ja : 0, jhi = cnt - ja < K1 ? cnt - ja : K1;      // the piece's samples inside this chunk: j in [jlo, jhi)
            const double* r0 = rec + dd * CHP + ja;      // grad_p[dd] of sample j at r0[j]; grad_v[dd] two rows on, grad_a[dd] four
            // k = 2 kp and 2 kp + 1: beta0_k = s^k, beta1_k = k s^(k-1), beta2_k = k (k-1) s^(k-2) (alm_traj_opt.cpp:738-740) from the
            // power table; for kp = 0 the absent powers are read at index 0 and meet zero factors
            const int k0 = 2 * kp, k1 = k0 + 1;
            const int pa = k0 >= 2 ? k0 - 2 : 0, pb = k0 >= 1 ? k0 - 1 : 0;
            const double c1a = (double)k0, c2a = (double)(k0 * (k0 - 1)), c1b = (double)k1, c2b = (double)(k1 * k0);
            double a0 = 0.0, a1 = 0.0;
            const double* w = wtab;
#pragma unroll 1
            for (int jb = 0; jb < K1; jb += UPH_SC_XB) {
                double e0[UPH_SC_XB], e1[UPH_SC_XB], e2[UPH_SC_XB], pA[UPH_SC_XB], pB[UPH_SC_XB], pC[UPH_SC_XB], pD[UPH_SC_XB];
#pragma unroll
                for (int u = 0; u < UPH_SC_XB; u++) {
                    const int j = jb + u;
                    e0[u] = r0[j]; e1[u] = r0[2 * CHP + j]; e2[u] = r0[4 * CHP + j];
                    pA[u] = w[6 * j + pa]; pB[u] = w[6 * j + pb]; pC[u] = w[6 * j + k0]; pD[u] = w[6 * j + k1];
                }
#pragma unroll
                for (int u = 0; u < UPH_SC_XB; u++) {
                    const int j = jb + u;
                    const bool in = j >= jlo && j < jhi;
                    a0 += in ? (pC[u] * e0[u] + (c1a * pB[u]) * e1[u] + (c2a * pA[u]) * e2[u]) : 0.0;
                    a1 += in ? (pD[u] * e0[u] + (c1b * pC[u]) * e1[u] + (c2b * pB[u]) * e2[u]) : 0.0;
                }
            }
            Gxy[12 * i + 2 * k0 + dd] += a0;
            Gxy[12 * i + 2 * k1 + dd] += a1;
        };
        // yaw outputs: lane tt owns (yaw piece, k-pair)
        auto yawTask = [&](int tt) {
            const int mi = tt / 3, kp = tt - 3 * mi, m = m0 + mi;
            // candidate slots: sample (i, j) sits at time (i + j / K) Txy, yaw piece m covers [m, m + 1) Tyaw = [m, m + 1) (Nxy / Nyaw) Txy, i.e. in
            // exact arithmetic the slots from p_a K1 + ceil(f_a K) up to (not including) p_b K1 + ceil(f_b K), x_a = m xr = p_a + f_a.  The sample
            // times are accumulated sums (Q2), so a sample lying ON a yaw boundary -- every other boundary coincides with a piece boundary when
            // piece_yaw = 2 piece_xy, where the last sample of one piece and the first of the next share the time -- may fall to either side:
            // one slot each way, plus one slot of margin (bounds in double: their own rounding is far below a slot).  The tag test below
            // decides membership exactly; a window of real + 5 slots (8.5 + 5 for the usual two yaw pieces per position piece) is ONE batch.
            const double xa = (double)m * xr, xb = (double)(m + 1) * xr;
            const int pa = (int)xa, pb = (int)xb;
            int sa = pa * K1 + (int)((xa - (double)pa) * (double)K) - UPH_SC_WLO - s0;
            int sb = pb * K1 + (int)((xb - (double)pb) * (double)K) + UPH_SC_WHI - s0;
            if (m == Nyaw - 1) sb = cnt;                 // the last yaw piece also takes every clamped late sample (:751)
            if (sa < 0) sa = 0;
            if (sb > cnt) sb = cnt;
            if (sb < sa) sb = sa;
            const double* rv = rec + (6 + 2 * kp) * CHP + sa;
            const int* tg = rtag + sa;
            double a0 = 0.0, a1 = 0.0;
            for (int len = sb - sa; len > 0; len -= UPH_SC_YB, rv += UPH_SC_YB, tg += UPH_SC_YB) {
                int tg_[UPH_SC_YB];
                double v0[UPH_SC_YB], v1[UPH_SC_YB];
#pragma unroll
                for (int u = 0; u < UPH_SC_YB; u++) { tg_[u] = tg[u]; v0[u] = rv[u]; v1[u] = rv[CHP + u]; }
#pragma unroll
                for (int u = 0; u < UPH_SC_YB; u++) {
                    const bool in = (u < len) && (tg_[u] == m);
                    a0 += in ? v0[u] : 0.0;
                    a1 += in ? v1[u] : 0.0;
                }
            }
            Gyaw[6 * m + 2 * kp] += a0;
            Gyaw[6 * m + 2 * kp + 1] += a1;
        };
        if constexpr (WG::MFMA_SCATTER) {
            // The xy half is a dense contraction with a SHARED left operand: G(6 x 2P) += B(6 x 3 K1) R(3 K1 x 2P), B = the power table with the
            // factors k, k (k-1) of alm_traj_opt.cpp:738-740, R = the records of the chunk's P pieces (samples of a piece outside the chunk
            // masked to zero).  The device runs it on the matrix cores (DevWG::scatterXY17, v_mfma_f64_16x16x4_f64) for the reference's
            // int_K = 16 while the other wave(s) sum the yaw blocks; any other K takes the vector path below.
            if (K1 == 17) {
                UPH_MARK("scatter.mfma");
                wg.scatterXY17(rec, wtab, Gxy, i0, i1 - i0 + 1, s0, cnt);
                UPH_MARK("scatter.yaw");
                wg.pforRev(3 * (m1 - m0 + 1), yawTask);
                return;
            }
        }
        UPH_MARK("scatter.vector(cold for K=16)");
        wg.pfor(XYL + 3 * (m1 - m0 + 1), [&](int t) {
            if (t < XYL) { if (t < nxyt) xyTask(t); }
            else yawTask(t - XYL);
        });
    }

    // ------------------------------------------------------------------ adjoint: (dK/dc, direct dK/dT sums) -> gradient w.r.t. (q, T)
    // calGradCTtoQT (se2traj.hpp:751-816) through the knot system.  On return gamxy / gamyaw hold gamma = M^T (G T^-k) laid out
    // like beta; chain_xy / chain_yaw = sum_i dW/dT_i without the direct parts (header comment).
    // gout != nullptr: the way-point entries of the gradient are written straight from gamma (alm_traj_opt.cpp:336-337) and gd_out
    // receives sum_{t >= 1} gout[t] d[t] (the caller adds the tau entry): g . d for the line search without a pass of its own.
    UPH_HD void adjoint(double& chain_xy, double& chain_yaw, double* gout = nullptr, double* gd_out = nullptr) {
        UPH_MARK("adjoint");
        const double Tx = Txy, Ty = Tyaw, itx = wg.bcast(1.0 / Txy), ity = wg.bcast(1.0 / Tyaw);
        const long long ta0 = wg.clock();
        const int nbx = Nxy + 5, nby = Nyaw + 5;
        const int nvec = 2 * nbx + nby;
        double* gwxy = rec;                              // [Nxy-1][v,a][2] at knotOff(q, 4)   (records are consumed by scatterChunk before adjoint runs)
        double* gwyaw = gwxy + knotBufDoubles(Nxy, 4);   // [Nyaw-1][v,a] at knotOff(q, 2)
        double* gdir = gwyaw + knotBufDoubles(Nyaw, 2);  // [nvec] direct contributions, laid out like gamma
        // One lane per (knot, dimension): transposed Hermite expansion of G T^-k -- every knot collects from the piece it opens and
        // the piece it closes; interior (v, a) go to the transposed knot solve, everything that is itself an entry of beta (all
        // positions, the end knots' V and A) straight to its column -- plus the opened piece's share of -sum_k k c_k / T * G_k.
        double ch[2];
        wg.template sum<2>(2 * (Nxy + 1) + (Nyaw + 1), ch, [&](int t, double* acc) {
            const bool isxy = t < 2 * (Nxy + 1);
            const int j = isxy ? (t >> 1) : t - 2 * (Nxy + 1), dd = isxy ? (t & 1) : 0, N = isxy ? Nxy : Nyaw, os = isxy ? 2 : 1;
            const double* G = isxy ? Gxy + dd : Gyaw;
            const double* c = isxy ? cxy + dd : cyaw;
            const double it_ = isxy ? itx : ity;
            const double i2 = it_ * it_, i3 = i2 * it_, i4 = i3 * it_, i5 = i4 * it_;
            double dp = 0.0, dv = 0.0, da = 0.0;
            if (j < N) {
                const double* gl = G + (size_t)6 * j * os;
                const double* cl = c + (size_t)6 * j * os;
                const double r1 = gl[os], r2 = gl[2 * os], r3 = gl[3 * os], r4 = gl[4 * os], r5 = gl[5 * os];
                const double chain = -it_ * (cl[os] * r1 + 2.0 * (cl[2 * os] * r2) + 3.0 * (cl[3 * os] * r3) + 4.0 * (cl[4 * os] * r4) + 5.0 * (cl[5 * os] * r5));
                acc[0] += isxy ? chain : 0.0;
                acc[1] += isxy ? 0.0 : chain;
                const double g0 = gl[0], g1 = r1 * it_, g2 = r2 * i2, g3 = r3 * i3, g4 = r4 * i4, g5 = r5 * i5;
                dp += g0 - 10.0 * g3 + 15.0 * g4 - 6.0 * g5;
                dv += g1 - 6.0 * g3 + 8.0 * g4 - 3.0 * g5;
                da += 0.5 * g2 - 1.5 * g3 + 1.5 * g4 - 0.5 * g5;
            }
            if (j > 0) {
                const double* gr_ = G + (size_t)6 * (j - 1) * os;
                const double g3 = gr_[3 * os] * i3, g4 = gr_[4 * os] * i4, g5 = gr_[5 * os] * i5;
                dp += 10.0 * g3 - 15.0 * g4 + 6.0 * g5;
                dv += -4.0 * g3 + 7.0 * g4 - 3.0 * g5;
                da += 0.5 * g3 - g4 + 0.5 * g5;
            }
            double* gd = isxy ? gdir + dd : gdir + 2 * nbx;
            gd[knotCol(j, N) * os] = dp;
            if (j == 0) { gd[1 * os] = dv; gd[2 * os] = da; }
            else if (j == N) { gd[(N + 3) * os] = dv; gd[(N + 4) * os] = da; }
            else if (isxy) { double* kq = gwxy + knotOff(j - 1, 4) + dd; kq[0] = dv; kq[2] = da; }
            else { double* kq = gwyaw + knotOff(j - 1, 2); kq[0] = dv; kq[1] = da; }
        });
        const long long ta1 = wg.clock();
        // lambda = M^-T (knot gradients) by the transposed block sweeps, in place
        wg.thomas(ttab, true, gwyaw, Nyaw - 1, gwxy, Nxy - 1);
        const long long ta2 = wg.clock();
        // gamma = R^T lambda + direct parts: a way-point / end position p_k enters r_{k-1}, r_k, r_{k+1}; the end states enter r_1
        // (- A z_0) and r_{N-1} (- C z_N).  The four V / A columns also give <gamma, d b~/dT> (V scales with T, A with T^2).
        double hh[3];
        wg.template sum<3>(nvec, hh, [&](int t, double* acc) {
            const bool isxy = t < 2 * nbx;
            const int col = isxy ? (t >> 1) : t - 2 * nbx, dd = isxy ? (t & 1) : 0, N = isxy ? Nxy : Nyaw;
            const int ks = isxy ? 4 : 2, cs = isxy ? 2 : 1;
            const double* lam = isxy ? gwxy + dd : gwyaw;
            const double T1 = isxy ? Tx : Ty;
            const double* h0 = isxy ? hd + dd : hd + 12;
            const double* h1 = isxy ? hd + 6 + dd : hd + 15;
            auto L0 = [&](int j) { return (j >= 1 && j <= N - 1) ? lam[knotOff(j - 1, ks)] : 0.0; };
            auto L1 = [&](int j) { return (j >= 1 && j <= N - 1) ? lam[knotOff(j - 1, ks) + cs] : 0.0; };
            double a = gdir[t], hT = 0.0;
            if (col == 1) { a += 8.0 * L0(1) + 7.0 * L1(1); hT = a * h0[cs]; }                             // V0: (-A^T lambda_1)[0]
            else if (col == 2) { a += L0(1) + L1(1); hT = a * (2.0 * T1 * h0[2 * cs]); }                   // A0: (-A^T lambda_1)[1]
            else if (col == N + 3) { a += -8.0 * L0(N - 1) + 7.0 * L1(N - 1); hT = a * h1[cs]; }           // VN: (-C^T lambda_{N-1})[0]
            else if (col == N + 4) { a += L0(N - 1) - L1(N - 1); hT = a * (2.0 * T1 * h1[2 * cs]); }       // AN: (-C^T lambda_{N-1})[1]
            else {
                const int k = col == 0 ? 0 : (col == N + 2 ? N : col - 2);
                a += 20.0 * ((L0(k - 1) - L0(k)) - (L0(k) - L0(k + 1))) - 15.0 * (L1(k - 1) - L1(k + 1));
            }
            if (isxy) gamxy[t] = a; else gamyaw[t - 2 * nbx] = a;
            acc[0] += isxy ? hT : 0.0;
            acc[1] += isxy ? 0.0 : hT;
            if (gout != nullptr && col >= 3 && col <= N + 1) {
                const int ix = isxy ? 1 + 2 * (col - 3) + dd : 1 + 2 * (Nxy - 1) + (col - 3);
                gout[ix] = a;
                acc[2] += a * d[ix];
            }
        });
        if (gd_out) *gd_out = hh[2];
        if (sub_t) { sub_t[4] += ta1 - ta0; sub_t[5] += ta2 - ta1; sub_t[6] += wg.clock() - ta2; }
        chain_xy = ch[0] + hh[0];
        chain_yaw = ch[1] + hh[1];
    }

    UPH_HD void evalConsts() {
        ec_irho = wg.bcast(1.0 / rho);
        ec_step = wg.bcast(Txy / K);                                    // alm_traj_opt.cpp:713
        ec_invK = wg.bcast(1.0 / K);
        ec_omega = wg.bcast(P.rho_ter * ec_step * scale_fx);
        ec_omega_h = wg.bcast(0.5 * P.rho_ter * ec_step * scale_fx);
    }

    // hx / gx of the LAST evaluated trajectory (Q1: the coefficients in LDS are those of the last evaluation, also after a failed
    // line search restored x), for updateDualVars / judgeConvergence / the caller
    UPH_HD void refreshResiduals() {
        UPH_MARK("refreshResiduals");
        wg.pfor(2, [&](int u) { fillTimes(u); });          // adjoint() has overwritten the tables with gamma
        wg.pfor(S, [&](int s) { double dummy[3]; sampleEval<true>(s, 0, dummy); });
    }

    // ------------------------------------------------------------------ innerCallback (alm_traj_opt.cpp:280-347)
    // STEP: evaluates at x = xp + st d and leaves that point in xin (see generate).  last_gd = grad f . d afterwards.
    template <bool STEP>
    UPH_HD double eval(double* xin, double* gout, double st = 0.0) {
        evals++;
        long long t0 = wg.clock();
        if (t_last_eval_end) cyc[5] += t0 - t_last_eval_end;      // from the end of the previous evaluation (or of the two-loop) to here
        if (UPH_PHASE_MASK & 1) generate<STEP>(xin, st);
        // (values that stay alive across the evaluation's barriers are parked in scalar registers: a "uniform" double left in a
        // VGPR competes with the sample code for registers and ends up in scratch)
        const double tau = wg.bcast(xin[0]);
        const double jw = wg.bcast(P.use_scaling ? scale_trick_jerk * scale_fx : scale_fx);      // :308-310, 322-332
        double js[3] = {0.0, 0.0, 0.0};
        if (UPH_PHASE_MASK & 1) expand(xin, jw, js);
        long long t1 = wg.clock(); cyc[0] += t1 - t0;
        last_jerk = js[0];
        const double jerk_cost = wg.bcast(P.use_scaling ? js[0] * scale_fx * scale_trick_jerk : js[0] * scale_fx);
        evalConsts();
        // (cost, dT_xy, dT_yaw) of the samples: every lane keeps its partials across the chunks, ONE block reduction after the last chunk (wg.accEnd)
        double sm[3] = {0.0, 0.0, 0.0};
        wg.template accBegin<3>();
        for (int s0 = 0; s0 < S; s0 += CHS) {
            const int cnt = S - s0 < CHS ? S - s0 : CHS;
            t0 = wg.clock();
            if (UPH_PHASE_MASK & 2) wg.template accChunk<3>(cnt, [&](int t, double* acc) { sampleEval<false, SR>(s0 + t, t, acc); });
            t1 = wg.clock(); cyc[1] += t1 - t0;
            if (UPH_PHASE_MASK & 4) scatterChunk(s0, cnt);
            cyc[2] += wg.clock() - t1;
        }
        wg.template accEnd<3>(sm);
        t1 = wg.clock();
        double chx = 0.0, chy = 0.0, gdw = 0.0;
        if (UPH_PHASE_MASK & 8) adjoint(chx, chy, gout, &gdw);
        t0 = wg.clock(); cyc[3] += t0 - t1;
        const double gTx = js[1] * jw + sm[1] + chx;       // sum_i gdTxy(i) after calGradCTtoQT
        const double gTy = js[2] * jw + sm[2] + chy;
        const double grad_Tsum = P.rho_T * scale_fx + gTx / Nxy + gTy / Nyaw;             // :341-344
        const double g0 = wg.bcast(grad_Tsum * getTtoTauGrad(tau));
        wg.pfor(1, [&](int) { gout[0] = g0; });
        last_gd = wg.bcast(gdw + g0 * d[0]);
        const double tau_cost = P.rho_T * expC2(tau) * scale_fx;                          // :340
        t_last_eval_end = wg.clock();
        return wg.bcast(jerk_cost + sm[0] + tau_cost);                                    // :346
    }

    // ------------------------------------------------------------------ initScaling (alm_traj_opt.cpp:349-661)
    UPH_HD void initScaling(double* x0) {
        generate<false>(x0, 0.0);
        const double tau = x0[0];
        const double dTau = getTtoTauGrad(tau);
        // objective scale (:365-370, 507-519, 627-653): jerk + rho_ter*int sigma^2 + rho_T*T, no scale_trick_jerk
        double js[3];
        expand(x0, 1.0, js);
        double sm[3] = {0.0, 0.0, 0.0};
        for (int s0 = 0; s0 < S; s0 += CHS) {
            const int cnt = S - s0 < CHS ? S - s0 : CHS;
            double part[3];
            wg.template sum<3>(cnt, part, [&](int t, double* acc) { sampleObjective(s0 + t, t, acc); });
            sm[0] += part[0]; sm[1] += part[1]; sm[2] += part[2];
            scatterChunk(s0, cnt);
        }
        double chx, chy;
        adjoint(chx, chy);
        const double gTau_fx = (P.rho_T + (js[1] + sm[1] + chx) / Nxy + (js[2] + sm[2] + chy) / Nyaw) * dTau;
        const double mq = wg.maxv((Nxy - 1) * 2 + (Nyaw - 1), [&](int t) {
            return t < (Nxy - 1) * 2 ? fabs(gamxy[3 * 2 + t]) : fabs(gamyaw[3 + (t - (Nxy - 1) * 2)]);
        });
        scale_fx = wg.bcast(1.0 / dmax(1.0, dmax(mq, fabs(gTau_fx))));                    // :651-652
        // per-constraint scales (:521-620, 637-660): scale_cx(i) = 1 / max(1, |grad_x c_i|_inf)
        wg.pfor(2, [&](int u) { fillTimes(u); });          // adjoint() has overwritten the tables with gamma
        wg.pfor(S, [&](int s) { scalingSample(s, dTau); });
    }

    // one sample of the per-constraint scaling loop of initScaling.  The seven constraints of a sample read the SAME operator rows (those of the
    // sample's two knots): the rows are fetched once per batch of eight columns and applied to all seven weight sets (until round 5 every constraint
    // fetched them again: seven times the loads and their round trips); per constraint the arithmetic and its order are unchanged.
    UPH_HD void scalingSample(int s, double dTau) {
        const int i = divSmall(s, K + 1, inv_k1), j = s - i * (K + 1);
        Kin k;
        kin(i, j, k);
        // two groups of constraints (the rows are fetched twice per sample instead of seven times): all seven at once keep 150 doubles of per-constraint
        // state live next to the sample's kinematics and spill 340 registers even without a register cap (UPH_SCALING_GROUPS = 1: 8.8 ms per launch of
        // 16384 against 10.4 ms for the per-constraint form)
#if !defined(UPH_SCALING_GROUPS) || UPH_SCALING_GROUPS == 2
        scalingGroup<0, 4>(s, i, j, k, dTau);
        scalingGroup<4, 3>(s, i, j, k, dTau);
#elif UPH_SCALING_GROUPS == 3
        scalingGroup<0, 3>(s, i, j, k, dTau);
        scalingGroup<3, 2>(s, i, j, k, dTau);
        scalingGroup<5, 2>(s, i, j, k, dTau);
#else
        scalingGroup<0, 7>(s, i, j, k, dTau);
#endif
    }
    template <int Q0, int NQS>
    UPH_HD void scalingGroup(int s, int i, int j, const Kin& k, double dTau) {
        const int nbx = Nxy + 5, nby = Nyaw + 5;
        const double Tx = Txy, Ty = Tyaw, itx = 1.0 / Txy, ity = 1.0 / Tyaw;
        const double gravity = grid.gravity;
        const double alpha = 1.0 / K * j;
        const double icvx = k.tv[0], icvy = k.tv[2], icxi = k.tv[5];
        const int m = k.yaw_idx;
        // per constraint: the transposed Hermite expansion of its sparse dc/dC blocks to the piece's two knots (xy: dp / dv / da at the left and right
        // knot per dimension; yaw alike) and the scalar parts of its time gradient
        double dpL[NQS][2], dvL[NQS][2], daL[NQS][2], dpR[NQS][2], dvR[NQS][2], daR[NQS][2];
        double ypL[NQS], yvL[NQS], yaL[NQS], ypR[NQS], yvR[NQS], yaR[NQS];
        double sumx[NQS], sumy[NQS], mx[NQS];
#pragma unroll
        for (int qi = 0; qi < NQS; qi++) {
            const int q = Q0 + qi;                   // the constraint (alm_traj_opt.cpp:521-620)
            double gp_[2] = {0, 0}, gv_[2] = {0, 0}, ga_[2] = {0, 0}, gyaw = 0.0, gdyaw = 0.0, gse2[3];
            if (q == 0) {                    // non-holonomic :521-529
                gv_[0] = k.syaw; gv_[1] = -k.cyaw;
                gyaw = k.vel[0] * k.cyaw + k.vel[1] * k.syaw;
            } else if (q == 1) {             // longitude velocity :531-544
                for (int t = 0; t < 2; t++) gv_[t] = 1.0 * icvx * icvx * 2.0 * k.vel[t];
                for (int t = 0; t < 3; t++) gse2[t] = 1.0 * k.v_norm * k.v_norm * 2.0 * icvx * k.tg[0][t];
                gp_[0] = gse2[0]; gp_[1] = gse2[1]; gyaw = gse2[2];
            } else if (q == 2) {             // longitude acceleration :546-560
                const double gax = 2.0 * k.ax;
                ga_[0] = gax * icvx * k.cyaw; ga_[1] = gax * icvx * k.syaw;
                gyaw = gax * icvx * k.lat_acc;
                for (int t = 0; t < 3; t++) gse2[t] = gax * (gravity * k.tg[1][t] + k.tg[0][t] * k.lon_acc);
                gp_[0] = gse2[0]; gp_[1] = gse2[1]; gyaw += gse2[2];
            } else if (q == 3) {             // latitude acceleration :562-576
                const double gay = 2.0 * k.ay;
                ga_[0] = gay * icvy * (-k.syaw); ga_[1] = gay * icvy * k.cyaw;
                gyaw = -gay * icvy * k.lon_acc;
                for (int t = 0; t < 3; t++) gse2[t] = gay * (gravity * k.tg[3][t] + k.tg[2][t] * k.lat_acc);
                gp_[0] = gse2[0]; gp_[1] = gse2[1]; gyaw += gse2[2];
            } else if (q == 4) {             // curvature :578-598
                const double den = 1.0 / (k.vx * k.vx + delta_sigl);
                const double gwz = den * 2.0 * k.wz;
                const double gvx2 = -k.curv_snorm * den;
                gdyaw = gwz * icxi;
                for (int t = 0; t < 3; t++) gse2[t] = gwz * k.dyaw * k.tg[5][t];
                for (int t = 0; t < 2; t++) gv_[t] = gvx2 * icvx * icvx * 2.0 * k.vel[t];
                for (int t = 0; t < 3; t++) gse2[t] += gvx2 * k.v_norm * k.v_norm * 2.0 * icvx * k.tg[0][t];
                gp_[0] = gse2[0]; gp_[1] = gse2[1]; gyaw = gse2[2];
            } else if (q == 5) {             // attitude :600-609
                gp_[0] = -k.tg[4][0]; gp_[1] = -k.tg[4][1]; gyaw = -k.tg[4][2];
            } else {                         // surface variation :611-620
                gp_[0] = k.tg[6][0]; gp_[1] = k.tg[6][1]; gyaw = k.tg[6][2];
            }
            // sparse dc_i/dC blocks, already multiplied by T^-k, and the chain term -k c/T
            double gx_[6][2], gy_[6];
            double chain_x = 0.0, chain_y = 0.0, sx = 1.0, sy = 1.0;
            for (int kk = 0; kk < 6; kk++) {
                for (int t = 0; t < 2; t++) {
                    const double v = k.b0[kk] * gp_[t] + k.b1[kk] * gv_[t] + k.b2[kk] * ga_[t];
                    chain_x += -(double)kk * cxy[12 * i + kk * 2 + t] * itx * v;
                    gx_[kk][t] = v * sx;
                }
                const double vy = k.y0[kk] * gyaw + k.y1[kk] * gdyaw;
                chain_y += -(double)kk * cyaw[6 * m + kk] * ity * vy;
                gy_[kk] = vy * sy;
                sx *= itx; sy *= ity;
            }
            double tx = ((gp_[0] * k.vel[0] + gp_[1] * k.vel[1]) + (gv_[0] * k.acc[0] + gv_[1] * k.acc[1]) + (ga_[0] * k.jer[0] + ga_[1] * k.jer[1])) * alpha;
            const double yawdot = gyaw * k.dyaw + gdyaw * k.d2yaw;
            tx += yawdot * (alpha + i);
            const double ty = -yawdot * m;
            sumx[qi] = tx + chain_x;
            sumy[qi] = ty + chain_y;
            mx[qi] = 0.0;
            // M^T restricted to this sample's piece = (transposed Hermite expansion to the piece's two knots) followed by the
            // knot operator rows of those knots (interior ones) or the direct beta columns (end knots, all positions)
            for (int t = 0; t < 2; t++) {
                const double g0 = gx_[0][t], g1 = gx_[1][t], g2 = gx_[2][t], g3 = gx_[3][t], g4 = gx_[4][t], g5 = gx_[5][t];
                dpL[qi][t] = g0 - 10.0 * g3 + 15.0 * g4 - 6.0 * g5; dvL[qi][t] = g1 - 6.0 * g3 + 8.0 * g4 - 3.0 * g5; daL[qi][t] = 0.5 * g2 - 1.5 * g3 + 1.5 * g4 - 0.5 * g5;
                dpR[qi][t] = 10.0 * g3 - 15.0 * g4 + 6.0 * g5; dvR[qi][t] = -4.0 * g3 + 7.0 * g4 - 3.0 * g5; daR[qi][t] = 0.5 * g3 - g4 + 0.5 * g5;
            }
            {
                const double g0 = gy_[0], g1 = gy_[1], g2 = gy_[2], g3 = gy_[3], g4 = gy_[4], g5 = gy_[5];
                ypL[qi] = g0 - 10.0 * g3 + 15.0 * g4 - 6.0 * g5; yvL[qi] = g1 - 6.0 * g3 + 8.0 * g4 - 3.0 * g5; yaL[qi] = 0.5 * g2 - 1.5 * g3 + 1.5 * g4 - 0.5 * g5;
                ypR[qi] = 10.0 * g3 - 15.0 * g4 + 6.0 * g5; yvR[qi] = -4.0 * g3 + 7.0 * g4 - 3.0 * g5; yaR[qi] = 0.5 * g3 - g4 + 0.5 * g5;
            }
        }
        // ---- position blocks.  Only the way-point columns (max norm) and the four head/tail V, A columns (time gradient) are needed.  Operator
        // rows are read in unconditional batches of 8 columns (end knots carry zero weights instead of branches): a loop with one dependent load
        // per column pays one L2 round trip per column (the old form: 1700 round trips per sample).
        const bool inL = i >= 1, inR = i + 1 <= Nxy - 1;
        const auto WL = UPH_AS_GLOBAL(Wr_xy + (size_t)(inL ? 2 * (i - 1) : 0) * nbx);      // rows v_i, a_i
        const auto WR = UPH_AS_GLOBAL(Wr_xy + (size_t)(inR ? 2 * i : 0) * nbx);            // rows v_{i+1}, a_{i+1}
        const int pcL = knotCol(i, Nxy), pcR = knotCol(i + 1, Nxy);
        for (int c0 = 3; c0 < Nxy + 2; c0 += UPH_SCALING_WB) {
            double w[4][UPH_SCALING_WB];
#pragma unroll
            for (int u = 0; u < UPH_SCALING_WB; u++) {
                const int cc = c0 + u < Nxy + 2 ? c0 + u : Nxy + 1;
                w[0][u] = WL[cc]; w[1][u] = WL[nbx + cc]; w[2][u] = WR[cc]; w[3][u] = WR[nbx + cc];
            }
#pragma unroll
            for (int q = 0; q < NQS; q++) {
                // weights of rows (v_L, a_L, v_R, a_R); zero for an end knot
                const double c00 = inL ? dvL[q][0] : 0.0, c01 = inL ? dvL[q][1] : 0.0, c10 = inL ? daL[q][0] : 0.0, c11 = inL ? daL[q][1] : 0.0;
                const double c20 = inR ? dvR[q][0] : 0.0, c21 = inR ? dvR[q][1] : 0.0, c30 = inR ? daR[q][0] : 0.0, c31 = inR ? daR[q][1] : 0.0;
                double mq = mx[q];
#pragma unroll
                for (int u = 0; u < UPH_SCALING_WB; u++) {
                    const int col = c0 + u;
                    double a0 = (w[0][u] * c00 + w[1][u] * c10) + (w[2][u] * c20 + w[3][u] * c30);
                    double a1 = (w[0][u] * c01 + w[1][u] * c11) + (w[2][u] * c21 + w[3][u] * c31);
                    if (col == pcL) { a0 += dpL[q][0]; a1 += dpL[q][1]; }
                    if (col == pcR) { a0 += dpR[q][0]; a1 += dpR[q][1]; }
                    if (col < Nxy + 2) mq = dmax(mq, dmax(fabs(a0), fabs(a1)));
                }
                mx[q] = mq;
            }
        }
        double htx[NQS], hty[NQS];
        {
            const int sc[4] = {1, 2, Nxy + 3, Nxy + 4};
            double w[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) { w[0][u] = WL[sc[u]]; w[1][u] = WL[nbx + sc[u]]; w[2][u] = WR[sc[u]]; w[3][u] = WR[nbx + sc[u]]; }
#pragma unroll
            for (int q = 0; q < NQS; q++) {
                const double c00 = inL ? dvL[q][0] : 0.0, c01 = inL ? dvL[q][1] : 0.0, c10 = inL ? daL[q][0] : 0.0, c11 = inL ? daL[q][1] : 0.0;
                const double c20 = inR ? dvR[q][0] : 0.0, c21 = inR ? dvR[q][1] : 0.0, c30 = inR ? daR[q][0] : 0.0, c31 = inR ? daR[q][1] : 0.0;
                double a0[4], a1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    a0[u] = (w[0][u] * c00 + w[1][u] * c10) + (w[2][u] * c20 + w[3][u] * c30);
                    a1[u] = (w[0][u] * c01 + w[1][u] * c11) + (w[2][u] * c21 + w[3][u] * c31);
                }
                if (!inL) { a0[0] += dvL[q][0]; a1[0] += dvL[q][1]; a0[1] += daL[q][0]; a1[1] += daL[q][1]; }      // an end knot's V, A are beta columns themselves
                if (!inR) { a0[2] += dvR[q][0]; a1[2] += dvR[q][1]; a0[3] += daR[q][0]; a1[3] += daR[q][1]; }
                double headtail_x = 0.0;
                headtail_x += a0[0] * hd[2] + a1[0] * hd[3];
                headtail_x += 2.0 * Tx * (a0[1] * hd[4] + a1[1] * hd[5]);
                headtail_x += a0[2] * hd[8] + a1[2] * hd[9];
                headtail_x += 2.0 * Tx * (a0[3] * hd[10] + a1[3] * hd[11]);
                htx[q] = headtail_x;
            }
        }
        // ---- yaw blocks
        const bool yL = m >= 1, yR = m + 1 <= Nyaw - 1;
        const auto VL = UPH_AS_GLOBAL(Wr_yaw + (size_t)(yL ? 2 * (m - 1) : 0) * nby);
        const auto VR = UPH_AS_GLOBAL(Wr_yaw + (size_t)(yR ? 2 * m : 0) * nby);
        const int qL = knotCol(m, Nyaw), qR = knotCol(m + 1, Nyaw);
        for (int c0 = 3; c0 < Nyaw + 2; c0 += UPH_SCALING_WB) {
            double w[4][UPH_SCALING_WB];
#pragma unroll
            for (int u = 0; u < UPH_SCALING_WB; u++) {
                const int cc = c0 + u < Nyaw + 2 ? c0 + u : Nyaw + 1;
                w[0][u] = VL[cc]; w[1][u] = VL[nby + cc]; w[2][u] = VR[cc]; w[3][u] = VR[nby + cc];
            }
#pragma unroll
            for (int q = 0; q < NQS; q++) {
                const double y0_ = yL ? yvL[q] : 0.0, y1_ = yL ? yaL[q] : 0.0, y2_ = yR ? yvR[q] : 0.0, y3_ = yR ? yaR[q] : 0.0;
                double mq = mx[q];
#pragma unroll
                for (int u = 0; u < UPH_SCALING_WB; u++) {
                    const int col = c0 + u;
                    double a0 = (w[0][u] * y0_ + w[1][u] * y1_) + (w[2][u] * y2_ + w[3][u] * y3_);
                    if (col == qL) a0 += ypL[q];
                    if (col == qR) a0 += ypR[q];
                    if (col < Nyaw + 2) mq = dmax(mq, fabs(a0));
                }
                mx[q] = mq;
            }
        }
        {
            const int sc[4] = {1, 2, Nyaw + 3, Nyaw + 4};
            double w[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) { w[0][u] = VL[sc[u]]; w[1][u] = VL[nby + sc[u]]; w[2][u] = VR[sc[u]]; w[3][u] = VR[nby + sc[u]]; }
#pragma unroll
            for (int q = 0; q < NQS; q++) {
                const double y0_ = yL ? yvL[q] : 0.0, y1_ = yL ? yaL[q] : 0.0, y2_ = yR ? yvR[q] : 0.0, y3_ = yR ? yaR[q] : 0.0;
                double a0[4];
#pragma unroll
                for (int u = 0; u < 4; u++) a0[u] = (w[0][u] * y0_ + w[1][u] * y1_) + (w[2][u] * y2_ + w[3][u] * y3_);
                if (!yL) { a0[0] += yvL[q]; a0[1] += yaL[q]; }
                if (!yR) { a0[2] += yvR[q]; a0[3] += yaR[q]; }
                double headtail_y = 0.0;
                headtail_y += a0[0] * hd[13];
                headtail_y += 2.0 * Ty * a0[1] * hd[14];
                headtail_y += a0[2] * hd[16];
                headtail_y += 2.0 * Ty * a0[3] * hd[17];
                hty[q] = headtail_y;
            }
        }
#pragma unroll
        for (int q = 0; q < NQS; q++) {
            const double gTau = ((sumx[q] + htx[q]) / Nxy + (sumy[q] + hty[q]) / Nyaw) * dTau;                      // :642-644
            scl[(Q0 + q) * S + s] = 1.0 / dmax(1.0, dmax(mx[q], fabs(gTau)));                                                // :658-659
        }
    }

    // ------------------------------------------------------------------ line search (lbfgs.hpp:276-389)
    // dginit = gp . d is handed in: the caller gets it for free (from the two-loop's registers, or as -g.g when d = -g)
    UPH_HD int lineSearch(double& f, double& stp, double stpmin, double stpmax, double dginit) {
        int count = 0;
        bool brackt = false, touched = false;
        double mu = 0.0, nu = stpmax;
        if (!(stp > 0.0)) return LBFGSERR_INVALIDPARAMETERS;
        if (0.0 < dginit) return LBFGSERR_INCREASEGRADIENT;
        const double finit = wg.bcast(f);                               // wave-uniform scalars of the search live in scalar registers
        const double dgtest = wg.bcast(P.f_dec_coeff * dginit);
        const double dstest = wg.bcast(P.s_curv_coeff * dginit);
        while (true) {
            f = eval<true>(x, g, stp);                                  // x = xp + stp d, f(x), g(x), last_gd = g . d
            ++count;
            if (isinf(f) || isnan(f)) return LBFGSERR_INVALID_FUNCVAL;
            if (P.past > 0 && fabs(finit - f) / (fabs(finit) + 1.0) < P.delta / P.past) return count;   // :327-330
            if (f > finit + stp * dgtest) {
                nu = stp;
                brackt = true;
            } else {
                if (last_gd < dstest) mu = stp;
                else return count;
            }
            if (P.max_linesearch <= count) return LBFGSERR_MAXIMUMLINESEARCH;
            if (brackt && (nu - mu) < P.machine_prec * nu) return LBFGSERR_WIDTHTOOSMALL;
            if (brackt) stp = wg.bcast(0.5 * (mu + nu));
            else stp = wg.bcast(stp * 2.0);
            if (stp < stpmin) return LBFGSERR_MINIMUMSTEP;
            if (stp > stpmax) {
                if (touched) return LBFGSERR_MAXIMUMSTEP;
                touched = true;
                stp = stpmax;
            }
        }
    }

    // ------------------------------------------------------------------ L-BFGS (lbfgs.hpp:439-722); progress callback = earlyExit (alm_traj_opt.cpp:1016)
    // RESUME (test hook of the teacher-forced late-state tests): instead of starting at x, the loop is entered at its top with a given
    // state -- x = xp, g = gp, d, the history ring, pf (all placed by resumeHook) and the scalars of ResumeIO -- and left again after
    // at most `budget` iterations (return code LBFGS_RUNNING_HOOK) with the state of the next loop top written back.
    struct ResumeIO { double step, fx; int k, end, bound, budget; };
    static constexpr int LBFGS_RUNNING_HOOK = 999;
    template <bool RESUME>
    UPH_HD int lbfgs(double& f_out, int& k_out, ResumeIO* io = nullptr) {
        int ret = 0, k = 0, ls, end = 0, bound = 0, budget = 0;
        double step = 0.0, fx = 0.0, ys, yy, dginit = 0.0;
        const int m = mem;
        bool run = true;
        if constexpr (RESUME) {
            step = io->step; fx = io->fx; k = io->k; end = io->end; bound = io->bound; budget = io->budget;
            wg.pfor(n, [&](int i) { xp[i] = x[i]; gp[i] = g[i]; });
            dginit = dot(g, d, n);
        } else {
            fx = eval<false>(x, g);
            wg.pfor(n + 1, [&](int i) { if (i < n) { d[i] = -g[i]; xp[i] = x[i]; gp[i] = g[i]; } else pf[0] = fx; });
            const double gnorm0 = absmax(g, n), xnorm0 = absmax(x, n);
            if (gnorm0 / dmax(1.0, xnorm0) < P.g_epsilon) {
                ret = LBFGS_CONVERGENCE;
                run = false;
            } else {
                const double dd0 = dot(d, d, n);
                step = 1.0 / sqrt(dd0);
                dginit = -dd0;                                               // gp . d with d = -g
                k = 1; end = 0; bound = 0;
            }
        }
        double gnorm_inf, xnorm_inf;
        if (run) {
            while (true) {
                if constexpr (RESUME) {
                    if (budget == 0) { ret = LBFGS_RUNNING_HOOK; break; }
                    budget--;
                }
                // xp / gp hold the iterate the search starts from (initial copy above, afterwards the bookkeeping pass below)
                ls = lineSearch(fx, step, P.min_step, P.max_step, dginit);
                if (ls < 0) {
                    wg.pfor(n, [&](int i) { x[i] = xp[i]; g[i] = gp[i]; });
                    ret = ls;
                    break;
                }
                tracePush(fx);
                if (k > 1000) { ret = LBFGS_CANCELED; break; }               // earlyExit: return k > 1e3
                // one pass and ONE block reduction for everything lbfgs.hpp:560-640 needs from (x, g, xp, gp): the two infinity
                // norms of the convergence test, s = x - xp and y = g - gp (stored as history column `end`), y.s, y.y, s.s,
                // gp.gp, g.g, the steepest-descent direction, and the xp / gp update for the next search.  Writing column `end`
                // and xp / gp before the exit tests is harmless: every exit below ends this L-BFGS call.
                double* hr = hist + (size_t)end * hrow;
                double* sc = hr + 2;
                double* yc = hr + 2 + hnp;
                const double pf_old = (0 < P.past && P.past <= k) ? wg.bcast(pf[k % P.past]) : 0.0;      // read before the pass' barrier: lane 0 may overwrite it afterwards
                UPH_MARK("lbfgs.bookkeeping");
                double r5[5], mx2[2];
                wg.template sumMax<5, 2>(n, r5, mx2, [&](int i, double* acc, double* mx) {
                    const double xv = x[i], gv = g[i], xo = xp[i], go = gp[i];
                    const double sv = xv - xo, yv = gv - go;
                    mx[0] = dmax(mx[0], fabs(gv)); mx[1] = dmax(mx[1], fabs(xv));
                    sc[i] = sv; yc[i] = yv;
                    acc[0] += yv * sv; acc[1] += yv * yv; acc[2] += sv * sv; acc[3] += go * go; acc[4] += gv * gv;
                    xp[i] = xv; gp[i] = gv;
                    d[i] = -gv;
                });
                gnorm_inf = mx2[0];
                xnorm_inf = mx2[1];
                if (gnorm_inf / dmax(1.0, xnorm_inf) < P.g_epsilon) { ret = LBFGS_CONVERGENCE; break; }
                if (0 < P.past) {
                    if (P.past <= k) {
                        const double rate = fabs(pf_old - fx) / dmax(1.0, fabs(fx));
                        if (rate < P.delta) { ret = LBFGS_STOP; break; }
                    }
                }
                if (P.inner_max_iter != 0 && P.inner_max_iter <= k) { ret = LBFGSERR_MAXIMUMITERATION; break; }
                ys = r5[0]; yy = r5[1];
                const double cau = r5[2] * sqrt(r5[3]) * P.cautious_factor;
                {
                    const int kk = k, e0 = end;
                    const double fv = fx, ysv = ys;
                    wg.pfor(1, [&](int) {
                        if (0 < P.past) pf[kk % P.past] = fv;
                        double* h0 = hist + (size_t)e0 * hrow;
                        h0[0] = ysv; h0[1] = 1.0 / ysv;                      // (y.s, 1 / y.s) lead the pair's row: one 16-byte load per chain step
                    });
                }
                ++k;
                dginit = -r5[4];
                if (ys > cau) {
                    ++bound;
                    bound = m < bound ? m : bound;
                    end = (end + 1) % m;
                    // two-loop recursion (lbfgs.hpp:687-710): a serial chain of 2*bound dot/axpy steps over the history in HBM
                    const long long tq = wg.clock();
                    cyc[7] += tq - t_last_eval_end;                 // end of evaluation -> start of the two-loop
                    UPH_MARK("twoLoop");
                    wg.twoLoop(d, g, n, hist, pf + MAX_PAST, rec, m, end, bound, ys / yy);   // (the record buffer is idle here: it parks the alphas)
                    dginit = wg.bcast(pf[MAX_PAST]);                // g . d, left by the two-loop
                    t_last_eval_end = wg.clock();
                    cyc[4] += t_last_eval_end - tq;
                    hist_reads += (long long)4 * bound * n;
                }
                step = 1.0;
            }
        }
        if constexpr (RESUME) { io->step = step; io->fx = fx; io->k = k; io->end = end; io->bound = bound; }
        f_out = fx;
        k_out = k;
        return ret;
    }

    // ------------------------------------------------------------------ ALM helpers (alm_traj_opt.h:132-151)
    UPH_HD void updateDualVars() {
        UPH_MARK("updateDualVars");
        const double r = rho;
        wg.pfor(S, [&](int s) {
            dual[s] += r * res[s];
            for (int q = 1; q < 7; q++) dual[q * S + s] = dmax(dual[q * S + s] + r * res[q * S + s], 0.0);
        });
        rho = wg.bcast(dmin((1 + P.gamma) * rho, P.beta));
    }
    UPH_HD bool judgeConvergence() {
        const double r = rho;
        const double rh = wg.maxv(S, [&](int s) { return fabs(res[s]); });
        const double rg = wg.maxv(6 * S, [&](int t) { return fabs(dmax(res[S + t], -dual[S + t] / r)); });
        return dmax(rh, rg) < P.epsilon_con;
    }

    UPH_HD void storeTrajectory(TrajState& st) {
        double* oc = bd.cxy + td.off_cxy;
        double* oy = bd.cyaw + td.off_cyaw;
        wg.pfor(12 * Nxy + 6 * Nyaw, [&](int t) { if (t < 12 * Nxy) oc[t] = cxy[t]; else oy[t - 12 * Nxy] = cyaw[t - 12 * Nxy]; });
        wg.pfor(1, [&](int) {
            st.T_xy = Txy; st.T_yaw = Tyaw; st.jerk_cost = last_jerk; st.scale_fx = scale_fx; st.rho = rho;
            st.evals = evals; st.hist_reads = hist_reads;
#ifdef UPH_TL_PROF
            for (int q = 0; q < 6; q++) cyc[8 + q] = wg.tl[q];                  // two-loop profile of the device workgroup object
#endif
            for (int q = 0; q < 16; q++) st.cyc[q] = cyc[q];
        });
#ifdef UPH_BAR_PROF
        wg.sync();
        wg.dumpBar(&st.cyc[8]);
#endif
    }

    // ------------------------------------------------------------------ optimizeSE2Traj (alm_traj_opt.cpp:168-278)
    // first half of optimizeSE2Traj (alm_traj_opt.cpp:180-232): reset duals/scales, then initScaling.  Runs as its own kernel so
    // that the register-hungry scaling code does not set the register budget of the solve kernel.
    UPH_HD void prepare(TrajState& st) {
        const double* gx0 = bd.x0 + td.off_x;
        double* gx = bd.x + td.off_x;
        rho = wg.bcast(st.rho);                                           // Q7: rho persists; lambda, mu, scales reset
        scale_fx = 1.0;
        wg.pfor(S > n ? S : n, [&](int t) {
            if (t < n) { x[t] = gx0[t]; gx[t] = gx0[t]; }
            if (t < S) for (int q = 0; q < 7; q++) { dual[q * S + t] = 0.0; res[q * S + t] = 0.0; scl[q * S + t] = 1.0; }
        });
        const long long tstart = wg.clock();
        if (P.use_scaling) initScaling(x);                                // :231-232
        cyc[5] += wg.clock() - tstart;
        wg.pfor(1, [&](int) { st.scale_fx = scale_fx; st.cyc[5] = cyc[5]; });
    }

    // second half (alm_traj_opt.cpp:234-278): the ALM loop.  Expects prepare() to have run on this trajectory.
    // cap > 0 (test hook): stop after `cap` ALM passes with ret_code 3 unless the solve ended before
    UPH_HD void optimize(TrajState& st, int cap = 0) {
        double* gx0 = bd.x + td.off_x;
        rho = wg.bcast(st.rho);
        scale_fx = wg.bcast(st.scale_fx);
        wg.pfor(n, [&](int t) { x[t] = gx0[t]; });
        const long long tstart = wg.clock();
        cyc[14] = wg.realtime();                                         // residency timeline of the launch (tools/phase_breakdown.py): 100 MHz stamps
        int ret_code = 0, iter = 0, total_k = 0, last_ret = 0;
        double inner_cost = 0.0;
        while (true) {                                                    // :234-271
            int kk = 0;
            tracePush(-1.0);
            const int result = lbfgs<false>(inner_cost, kk);
            refreshResiduals();
            total_k += kk;
            last_ret = result;
            if (result == LBFGS_CONVERGENCE || result == LBFGS_CANCELED || result == LBFGS_STOP || result == LBFGSERR_MAXIMUMITERATION) {
            } else if (result == LBFGSERR_MAXIMUMLINESEARCH) {
            } else { ret_code = 1; break; }
            updateDualVars();                                             // :257
            if (judgeConvergence()) break;                                // :259
            if (++iter > P.max_iter) { ret_code = 2; break; }             // :265
            if (cap > 0 && iter >= cap) { ret_code = 3; break; }
        }
        wg.pfor(n, [&](int t) { gx0[t] = x[t]; bd.gout[td.off_x + t] = g[t]; });
        cyc[6] = wg.clock() - tstart;
        cyc[15] = wg.realtime();
        storeTrajectory(st);
        wg.pfor(1, [&](int) {
            st.ret_code = ret_code; st.alm_iters = iter; st.lbfgs_iters = total_k; st.last_lbfgs_ret = last_ret; st.f = inner_cost;
        });
    }

    // test hook: continue the L-BFGS iteration loop from a state placed in HBM by the host (uph_batch_set_lbfgs_state) for at most
    // `budget` iterations; finish != 0: when the loop ends by itself, react as the ALM loop would (accepted code -> updateDualVars +
    // judgeConvergence).  rs[24] per trajectory: 0 step, 1 fx, 2 k, 3 end, 4 bound | out: 5 code, 6 accepted, 7 converged | 8.. pf
    UPH_HD void resumeHook(TrajState& st, int budget, int finish) {
        double* gx = bd.x + td.off_x;
        double* gg = bd.gout + td.off_x;
        double* gd = bd.rs_d + td.off_x;
        double* rs = bd.rs + (size_t)bidx * 24;
        rho = wg.bcast(st.rho); scale_fx = wg.bcast(st.scale_fx);
        wg.pfor(n + MAX_PAST, [&](int t) { if (t < n) { x[t] = gx[t]; g[t] = gg[t]; d[t] = gd[t]; } else pf[t - n] = rs[8 + t - n]; });
        ResumeIO io;
        io.step = wg.bcast(rs[0]); io.fx = wg.bcast(rs[1]);
        io.k = (int)wg.bcast(rs[2]); io.end = (int)wg.bcast(rs[3]); io.bound = (int)wg.bcast(rs[4]); io.budget = budget;
        double f = 0.0;
        int kk = 0;
        const int ret = lbfgs<true>(f, kk, &io);
        int accepted = 0, conv = 0;
        if (evals > 0) refreshResiduals();
        if (finish && ret != LBFGS_RUNNING_HOOK) {
            accepted = (ret == LBFGS_CONVERGENCE || ret == LBFGS_CANCELED || ret == LBFGS_STOP || ret == LBFGSERR_MAXIMUMITERATION || ret == LBFGSERR_MAXIMUMLINESEARCH) ? 1 : 0;
            if (accepted) { updateDualVars(); conv = judgeConvergence() ? 1 : 0; }
        }
        wg.pfor(n + MAX_PAST, [&](int t) { if (t < n) { gx[t] = x[t]; gg[t] = g[t]; gd[t] = d[t]; } else rs[8 + t - n] = pf[t - n]; });
        wg.pfor(1, [&](int) {
            rs[0] = io.step; rs[1] = io.fx; rs[2] = io.k; rs[3] = io.end; rs[4] = io.bound; rs[5] = ret; rs[6] = accepted; rs[7] = conv;
            st.f = io.fx; st.last_lbfgs_ret = ret; st.lbfgs_iters = io.k;
        });
        storeTrajectory(st);
    }

    // diagnostic: shader-clock ticks per call of the phases of one evaluation (averaged over `reps`), written to st.cyc[0..7]
    UPH_HD void microbench(TrajState& st, int reps) {
        // phase-level: 0 generate (knot states)  1 expand (Hermite + jerk + G)  2 evalConsts  3 sampleEval chunk 0 (sum<3>)  4 scatterChunk(0)
        // 5 adjoint  6 L-BFGS bookkeeping  7 total; 8.. sub-steps (sub_t)
        const double* gx0 = bd.x + td.off_x;
        rho = wg.bcast(st.rho); scale_fx = wg.bcast(st.scale_fx);
        wg.pfor(n, [&](int t) { x[t] = gx0[t]; d[t] = 0.5; xp[t] = gx0[t]; gp[t] = 0.25; });
        long long a[9];
        long long sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        sub_t = sub;
        double acc = 0.0, js[3], part[3], c1, c2;
        const int cnt = S < CH ? S : CH;
        a[0] = wg.clock();
        for (int r = 0; r < reps; r++) generate<false>(x, 0.0);
        a[1] = wg.clock();
        for (int r = 0; r < reps; r++) { expand(x, 1.0, js); acc += js[0]; }
        a[2] = wg.clock();
        for (int r = 0; r < reps; r++) evalConsts();
        a[3] = wg.clock();
        for (int r = 0; r < reps; r++) { wg.template sum<3>(cnt, part, [&](int t, double* ac) { sampleEval<false>(t, t, ac); }); acc += part[0]; }
        a[4] = wg.clock();
        for (int r = 0; r < reps; r++) scatterChunk(0, cnt);
        a[5] = wg.clock();
        for (int r = 0; r < reps; r++) { adjoint(c1, c2); acc += c1 + c2; }
        a[6] = wg.clock();
        // the L-BFGS bookkeeping of one iteration (between the evaluation and the two-loop), with a moving ring position
        const int m = mem;
        for (int r = 0; r < reps; r++) {
            const int end = (r * 7) % m, bound = m;
            acc += absmax(g, n) + absmax(x, n);
            wg.sync();
            wg.pfor(1, [&](int) { pf[r % 3] = acc; });
            double* sc = hist + (size_t)end * hrow + 2;
            double* yc = sc + hnp;
            double r6[6] = {0, 0, 0, 0, 0, 0};
            wg.template sum<3>(n, r6, [&](int i, double* ac) {
                const double sv = x[i] - xp[i], yv = g[i] - gp[i];
                sc[i] = sv; yc[i] = yv;
                ac[0] += yv * sv; ac[1] += yv * yv; ac[2] += sv * sv;
                d[i] = -g[i];
            });
            acc += r6[0] + r6[3] + sqrt(dot(gp, gp, n));
            wg.sync();
            wg.pfor(1, [&](int) { hist[(size_t)end * hrow] = acc; });
            (void)bound;
        }
        a[7] = wg.clock();
        wg.pfor(1, [&](int) {
            for (int q = 0; q < 7; q++) st.cyc[q] = (a[q + 1] - a[q]) / reps;
            st.cyc[7] = a[7] - a[0];
            for (int q = 0; q < 8; q++) st.cyc[8 + q] = sub[q] / reps;
            st.f = acc;
        });
        sub_t = nullptr;
    }

    // test / bench hooks --------------------------------------------------------------------------------------------
    UPH_HD void evalOnly(TrajState& st, int repeat) {
        const double* gx0 = bd.x + td.off_x;
        rho = wg.bcast(st.rho); scale_fx = wg.bcast(st.scale_fx);
        wg.pfor(n, [&](int t) { x[t] = gx0[t]; });
        double f = 0.0;
        for (int r = 0; r < repeat; r++) f = eval<false>(x, g);
        refreshResiduals();
        wg.pfor(n, [&](int t) { bd.gout[td.off_x + t] = g[t]; });
        storeTrajectory(st);
        wg.pfor(1, [&](int) { st.f = f; });
    }
    // A5 ALONE -- calConstrainCostGrad (alm_traj_opt.cpp:663-991) and nothing else of innerCallback: the resident coefficients and piece durations of
    // the last evaluation (bd.cxy / bd.cyaw, TrajState::T_xy / T_yaw; duals, scales, rho, scale_fx resident as well) in -> cost, gdCxy, gdCyaw,
    // sum_i gdTxy(i), sum_i gdTyaw(i) out (uniform durations: only the sums enter the gradient, :341-344), and -- STORE_RES -- hx / gx written by
    // every call as the reference writes them (:835, 846 ...).  No MINCO generate / expand, no adjoint, no jerk terms: G starts from zero.
    // `repeat` calls per launch (SURVEY.md 8d's "penalty kernel" measured as north_star defines it: uph_penalty_batch).
    template <bool STORE_RES>
    UPH_HD void penaltyOnly(TrajState& st, int repeat) {
        rho = wg.bcast(st.rho); scale_fx = wg.bcast(st.scale_fx);
        Txy = wg.bcast(st.T_xy); Tyaw = wg.bcast(st.T_yaw);
        ec_iTyaw = wg.bcast(1.0 / Tyaw);
        const double* oc = bd.cxy + td.off_cxy;
        const double* oy = bd.cyaw + td.off_cyaw;
        const int nc = 12 * Nxy + 6 * Nyaw;
        wg.pfor(nc, [&](int t) { if (t < 12 * Nxy) cxy[t] = oc[t]; else cyaw[t - 12 * Nxy] = oy[t - 12 * Nxy]; });
        double sm[3] = {0.0, 0.0, 0.0};
        for (int r = 0; r < repeat; r++) {
            wg.pfor(nc + 2, [&](int t) { if (t < 12 * Nxy) Gxy[t] = 0.0; else if (t < nc) Gyaw[t - 12 * Nxy] = 0.0; else fillTimes(t - nc); });
            evalConsts();
            wg.template accBegin<3>();
            for (int s0 = 0; s0 < S; s0 += CHS) {
                const int cnt = S - s0 < CHS ? S - s0 : CHS;
                wg.template accChunk<3>(cnt, [&](int t, double* acc) { sampleEval<STORE_RES ? 2 : 0, SR>(s0 + t, t, acc); });
                scatterChunk(s0, cnt);
            }
            wg.template accEnd<3>(sm);
        }
        double* ogx = bd.pen_gxy + td.off_cxy;
        double* ogy = bd.pen_gyaw + td.off_cyaw;
        wg.pfor(nc + 1, [&](int t) {
            if (t < 12 * Nxy) ogx[t] = Gxy[t];
            else if (t < nc) ogy[t - 12 * Nxy] = Gyaw[t - 12 * Nxy];
            else { double* o = bd.pen_out + (size_t)bidx * 3; o[0] = sm[0]; o[1] = sm[1]; o[2] = sm[2]; }
        });
    }
    UPH_HD void scalingOnly(TrajState& st) {
        const double* gx0 = bd.x + td.off_x;
        rho = wg.bcast(st.rho); scale_fx = 1.0;
        wg.pfor(n, [&](int t) { x[t] = gx0[t]; });
        initScaling(x);
        storeTrajectory(st);
    }

    // ------------------------------------------------------------------ post-solve report (alm_traj_opt.h:170-229, se2traj.hpp:551-561)
    // evaluates the stored trajectory (coefficients of the last evaluation, Q1) every 0.01 s
    UPH_HD void report(const TrajState& st) {
        const double* oc = bd.cxy + td.off_cxy;
        const double* oy = bd.cyaw + td.off_cyaw;
        wg.pfor(12 * Nxy + 6 * Nyaw, [&](int t) { if (t < 12 * Nxy) cxy[t] = oc[t]; else cyaw[t - 12 * Nxy] = oy[t - 12 * Nxy]; });
        const double Tx = st.T_xy, Ty = st.T_yaw;
        double durx = 0.0, dury = 0.0;
        for (int i = 0; i < Nxy; i++) durx += Tx;
        for (int i = 0; i < Nyaw; i++) dury += Ty;
        const double total = dmin(durx, dury);
        // sample q of `for (t = 0; t < total; t += 0.01)` (alm_traj_opt.h:182); its time is the same running sum t += 0.01.
        // The count comes from one lane walking the loop; every lane then rebuilds the time of its own samples the same way.
        int cnt = 0;
        {
            double cnt_d[1];
            wg.template sum<1>(1, cnt_d, [&](int, double* acc) { double t = 0.0; int q = 0; for (; q < 200000 && t < total; q++) t += 0.01; acc[0] += (double)q; });
            cnt = (int)cnt_d[0];
        }
        const double gravity = grid.gravity;
        const GridDev fgrid = framedGrid();
        // sample q sits at the reference's running sum t += 0.01 (q additions): every lane walks its own samples in increasing order
        // and keeps adding where it stopped
        double tcur = 0.0;
        int qcur = 0;
        auto sampleAt = [&](int q, double out[7]) {
            if (q < qcur) { tcur = 0.0; qcur = 0; }
            for (; qcur < q; qcur++) tcur += 0.01;
            const double t = tcur;
            // locatePieceIdx (se2traj.hpp:343-361) with uniform durations
            double tl = t; int ix = 0;
            for (; ix < Nxy && tl > Tx; ix++) tl -= Tx;
            if (ix == Nxy) { ix--; tl += Tx; }
            double tw = t; int iw = 0;
            for (; iw < Nyaw && tw > Ty; iw++) tw -= Ty;
            if (iw == Nyaw) { iw--; tw += Ty; }
            double p[2], v[2], a[2];
            for (int dd = 0; dd < 2; dd++) {
                const double* c = cxy + 12 * ix + dd;
                double val = 0, tn = 1.0;
                for (int kk = 0; kk <= 5; kk++) { val += tn * c[kk * 2]; tn *= tl; }
                double dv = 0; tn = 1.0;
                for (int kk = 1; kk <= 5; kk++) { dv += kk * tn * c[kk * 2]; tn *= tl; }
                double da = 0; tn = 1.0;
                for (int kk = 2; kk <= 5; kk++) { da += (kk - 1) * kk * tn * c[kk * 2]; tn *= tl; }
                p[dd] = val; v[dd] = dv; a[dd] = da;
            }
            const double* c = cyaw + 6 * iw;
            double yaw = 0, tn = 1.0;
            for (int kk = 0; kk <= 5; kk++) { yaw += tn * c[kk]; tn *= tw; }
            double dyaw = 0; tn = 1.0;
            for (int kk = 1; kk <= 5; kk++) { dyaw += kk * tn * c[kk]; tn *= tw; }
            const double yawn = normSO2(yaw);
            double cy_, sy_;
            sincosFast(yaw, sy_, cy_);
            const double cw = cy_, sw = sy_;
            double tv[7];
            terrainVariables(fgrid, p[0], p[1], yawn, cw, sw, tv, nullptr);
            const double vnorm = sqrt(v[0] * v[0] + v[1] * v[1]);
            const double lon = a[0] * cy_ + a[1] * sy_;
            const double lat = -a[0] * sy_ + a[1] * cy_;
            const double vx = vnorm * tv[0];
            out[0] = vx;
            out[1] = lon * tv[0] + gravity * tv[1];
            out[2] = lat * tv[2] + gravity * tv[3];
            out[3] = (dyaw * tv[5]) / sqrt(vx * vx + delta_sigl);
            out[4] = -1.0 / tv[5];
            out[5] = tv[6];
            out[6] = fabs(v[0] * sy_ + v[1] * (-cy_));
        };
        // two passes: max of +v and of -v for vx, ax, ay, cur (signed value of largest magnitude, alm_traj_opt.h:201-216), att
        // (= -cos xi, started at -1, :177: shifted by one so that the floor is 0), sigma, and the non-holonomic error sum
        double o[7], e1[1], ma[7], mb[3], dum[1];
        wg.template sumMax<1, 7>(cnt, e1, ma, [&](int q, double* acc, double* mx) {
            double r[7];
            sampleAt(q, r);
            acc[0] += r[6];
            mx[0] = dmax(mx[0], r[0]); mx[1] = dmax(mx[1], -r[0]); mx[2] = dmax(mx[2], r[1]); mx[3] = dmax(mx[3], -r[1]);
            mx[4] = dmax(mx[4], r[2]); mx[5] = dmax(mx[5], -r[2]); mx[6] = dmax(mx[6], r[3]);
        });
        wg.template sumMax<1, 3>(cnt, dum, mb, [&](int q, double* acc, double* mx) {
            double r[7];
            sampleAt(q, r);
            acc[0] += 0.0;
            mx[0] = dmax(mx[0], -r[3]); mx[1] = dmax(mx[1], r[4] + 1.0); mx[2] = dmax(mx[2], r[5]);
        });
        o[0] = ma[0] >= ma[1] ? ma[0] : -ma[1];
        o[1] = ma[2] >= ma[3] ? ma[2] : -ma[3];
        o[2] = ma[4] >= ma[5] ? ma[4] : -ma[5];
        o[3] = ma[6] >= mb[0] ? ma[6] : -mb[0];
        o[4] = mb[1] - 1.0;
        o[5] = mb[2];
        o[6] = e1[0];
        wg.pfor(7, [&](int t) { bd.report[(size_t)bidx * 7 + t] = o[t]; });
    }
};

}  // namespace uph
