// ORACLE -- TEST INFRASTRUCTURE ONLY (see banded.hpp header).  PARITY UNPINNED.
//
// Restates back_end/include/utils/se2traj.hpp:564-870 (MinJerkOpt<Dim>, MINCO_SE2):
//   generate       :595-680   band assembly (rows: head PVA; per knot jerk/snap continuity, way-point,
//                             C0,C1,C2 continuity; tail PVA), no-pivot LU, solve
//   getTrajJerkCost:697-710
//   calJerkGradCT  :719-747
//   calGradCTtoQT  :751-816   adjoint solve; dW/dq_i = lambda(6i+5); dW/dT_i += <B, lambda>
// Layout: c is row-major (6N) x Dim, row 6i+k = coefficient of t^k of piece i (se2traj.hpp:570).
// Eigen reductions (.dot/.squaredNorm/.sum) have unspecified summation order (version-, SIMD- and
// alignment-dependent); plain left-to-right order is used here.
#pragma once
#include <cmath>
#include <vector>
#include "banded.hpp"
#include "eigen_redux.hpp"

namespace orc {

struct MinJerk {
    int N = 0, D = 1;
    std::vector<double> head, tail;   // D x 3 row-major: [d][0]=P, [d][1]=V, [d][2]=A
    Banded A;
    std::vector<double> c;            // (6N) x D
    std::vector<double> T1, T2, T3, T4, T5;

    void reset(int pieceNum, int dim) {                 // :581-592
        N = pieceNum; D = dim;
        A.create(6 * N, 6, 6);
        c.assign((size_t)6 * N * D, 0.0);
        T1.assign(N, 0); T2 = T1; T3 = T1; T4 = T1; T5 = T1;
    }
    double& C(int r, int d) { return c[(size_t)r * D + d]; }
    double C(int r, int d) const { return c[(size_t)r * D + d]; }

    // inPs: D x (N-1) column-major (Eigen MatrixXd), i.e. inPs[col*D + d]
    void generate(const double* inPs, const double* ts, const double* headState, const double* tailState) {
        head.assign(headState, headState + 3 * D);
        tail.assign(tailState, tailState + 3 * D);
        for (int i = 0; i < N; i++) {                   // :603-607
            T1[i] = ts[i];
            T2[i] = T1[i] * T1[i];
            T3[i] = T2[i] * T1[i];
            T4[i] = T2[i] * T2[i];
            T5[i] = T4[i] * T1[i];
        }
        A.reset();
        std::fill(c.begin(), c.end(), 0.0);
        A(0, 0) = 1.0; A(1, 1) = 1.0; A(2, 2) = 2.0;    // :612-617
        for (int d = 0; d < D; d++) { C(0, d) = head[d * 3 + 0]; C(1, d) = head[d * 3 + 1]; C(2, d) = head[d * 3 + 2]; }
        for (int i = 0; i < N - 1; i++) {               // :619-654
            A(6 * i + 3, 6 * i + 3) = 6.0;
            A(6 * i + 3, 6 * i + 4) = 24.0 * T1[i];
            A(6 * i + 3, 6 * i + 5) = 60.0 * T2[i];
            A(6 * i + 3, 6 * i + 9) = -6.0;
            A(6 * i + 4, 6 * i + 4) = 24.0;
            A(6 * i + 4, 6 * i + 5) = 120.0 * T1[i];
            A(6 * i + 4, 6 * i + 10) = -24.0;
            A(6 * i + 5, 6 * i) = 1.0;
            A(6 * i + 5, 6 * i + 1) = T1[i];
            A(6 * i + 5, 6 * i + 2) = T2[i];
            A(6 * i + 5, 6 * i + 3) = T3[i];
            A(6 * i + 5, 6 * i + 4) = T4[i];
            A(6 * i + 5, 6 * i + 5) = T5[i];
            A(6 * i + 6, 6 * i) = 1.0;
            A(6 * i + 6, 6 * i + 1) = T1[i];
            A(6 * i + 6, 6 * i + 2) = T2[i];
            A(6 * i + 6, 6 * i + 3) = T3[i];
            A(6 * i + 6, 6 * i + 4) = T4[i];
            A(6 * i + 6, 6 * i + 5) = T5[i];
            A(6 * i + 6, 6 * i + 6) = -1.0;
            A(6 * i + 7, 6 * i + 1) = 1.0;
            A(6 * i + 7, 6 * i + 2) = 2 * T1[i];
            A(6 * i + 7, 6 * i + 3) = 3 * T2[i];
            A(6 * i + 7, 6 * i + 4) = 4 * T3[i];
            A(6 * i + 7, 6 * i + 5) = 5 * T4[i];
            A(6 * i + 7, 6 * i + 7) = -1.0;
            A(6 * i + 8, 6 * i + 2) = 2.0;
            A(6 * i + 8, 6 * i + 3) = 6 * T1[i];
            A(6 * i + 8, 6 * i + 4) = 12 * T2[i];
            A(6 * i + 8, 6 * i + 5) = 20 * T3[i];
            A(6 * i + 8, 6 * i + 8) = -2.0;
            for (int d = 0; d < D; d++) C(6 * i + 5, d) = inPs[i * D + d];
        }
        A(6 * N - 3, 6 * N - 6) = 1.0;                  // :656-674
        A(6 * N - 3, 6 * N - 5) = T1[N - 1];
        A(6 * N - 3, 6 * N - 4) = T2[N - 1];
        A(6 * N - 3, 6 * N - 3) = T3[N - 1];
        A(6 * N - 3, 6 * N - 2) = T4[N - 1];
        A(6 * N - 3, 6 * N - 1) = T5[N - 1];
        A(6 * N - 2, 6 * N - 5) = 1.0;
        A(6 * N - 2, 6 * N - 4) = 2 * T1[N - 1];
        A(6 * N - 2, 6 * N - 3) = 3 * T2[N - 1];
        A(6 * N - 2, 6 * N - 2) = 4 * T3[N - 1];
        A(6 * N - 2, 6 * N - 1) = 5 * T4[N - 1];
        A(6 * N - 1, 6 * N - 4) = 2;
        A(6 * N - 1, 6 * N - 3) = 6 * T1[N - 1];
        A(6 * N - 1, 6 * N - 2) = 12 * T2[N - 1];
        A(6 * N - 1, 6 * N - 1) = 20 * T3[N - 1];
        for (int d = 0; d < D; d++) {
            C(6 * N - 3, d) = tail[d * 3 + 0];
            C(6 * N - 2, d) = tail[d * 3 + 1];
            C(6 * N - 1, d) = tail[d * 3 + 2];
        }
        A.factorizeLU();                                // :676-677
        A.solve(c.data(), D);
    }

    double rowdot(int r1, int r2) const { double s = 0; for (int d = 0; d < D; d++) s += C(r1, d) * C(r2, d); return s; }

    double getTrajJerkCost() const {                    // :697-710
        double energy = 0.0;
        for (int i = 0; i < N; i++) {
            energy += 36.0 * rowdot(6 * i + 3, 6 * i + 3) * T1[i] +
                      144.0 * rowdot(6 * i + 4, 6 * i + 3) * T2[i] +
                      192.0 * rowdot(6 * i + 4, 6 * i + 4) * T3[i] +
                      240.0 * rowdot(6 * i + 5, 6 * i + 3) * T3[i] +
                      720.0 * rowdot(6 * i + 5, 6 * i + 4) * T4[i] +
                      720.0 * rowdot(6 * i + 5, 6 * i + 5) * T5[i];
        }
        return energy;
    }

    void calJerkGradCT(std::vector<double>& gdC, std::vector<double>& gdT) const {   // :719-747
        gdC.assign((size_t)6 * N * D, 0.0);
        for (int i = 0; i < N; i++)
            for (int d = 0; d < D; d++) {
                gdC[(6 * i + 5) * D + d] = 240.0 * C(6 * i + 3, d) * T3[i] + 720.0 * C(6 * i + 4, d) * T4[i] + 1440.0 * C(6 * i + 5, d) * T5[i];
                gdC[(6 * i + 4) * D + d] = 144.0 * C(6 * i + 3, d) * T2[i] + 384.0 * C(6 * i + 4, d) * T3[i] + 720.0 * C(6 * i + 5, d) * T4[i];
                gdC[(6 * i + 3) * D + d] = 72.0 * C(6 * i + 3, d) * T1[i] + 144.0 * C(6 * i + 4, d) * T2[i] + 240.0 * C(6 * i + 5, d) * T3[i];
            }
        gdT.assign(N, 0.0);
        for (int i = 0; i < N; i++)
            gdT[i] = 36.0 * rowdot(6 * i + 3, 6 * i + 3) +
                     288.0 * rowdot(6 * i + 4, 6 * i + 3) * T1[i] +
                     576.0 * rowdot(6 * i + 4, 6 * i + 4) * T2[i] +
                     720.0 * rowdot(6 * i + 5, 6 * i + 3) * T2[i] +
                     2880.0 * rowdot(6 * i + 5, 6 * i + 4) * T3[i] +
                     3600.0 * rowdot(6 * i + 5, 6 * i + 5) * T4[i];
    }

    // gdC: (6N) x D row-major (in); gdT: N (in/out); gdP: D x (N-1) column-major (out)
    void calGradCTtoQT(const std::vector<double>& gdC, std::vector<double>& gdT, std::vector<double>& gdP) const {  // :751-816
        gdP.assign((size_t)D * (N - 1), 0.0);
        std::vector<double> adj = gdC;
        A.solveAdj(adj.data(), D);
        for (int i = 0; i < N - 1; i++)
            for (int d = 0; d < D; d++) gdP[i * D + d] = adj[(6 * i + 5) * D + d];
        std::vector<double> B1(6 * D);
        for (int i = 0; i < N - 1; i++) {
            for (int d = 0; d < D; d++) {
                // negative velocity
                B1[2 * D + d] = -(C(i * 6 + 1, d) + 2.0 * T1[i] * C(i * 6 + 2, d) + 3.0 * T2[i] * C(i * 6 + 3, d) +
                                  4.0 * T3[i] * C(i * 6 + 4, d) + 5.0 * T4[i] * C(i * 6 + 5, d));
                B1[3 * D + d] = B1[2 * D + d];
                // negative acceleration
                B1[4 * D + d] = -(2.0 * C(i * 6 + 2, d) + 6.0 * T1[i] * C(i * 6 + 3, d) + 12.0 * T2[i] * C(i * 6 + 4, d) +
                                  20.0 * T3[i] * C(i * 6 + 5, d));
                // negative jerk
                B1[5 * D + d] = -(6.0 * C(i * 6 + 3, d) + 24.0 * T1[i] * C(i * 6 + 4, d) + 60.0 * T2[i] * C(i * 6 + 5, d));
                // negative snap
                B1[0 * D + d] = -(24.0 * C(i * 6 + 4, d) + 120.0 * T1[i] * C(i * 6 + 5, d));
                // negative crackle
                B1[1 * D + d] = -120.0 * C(i * 6 + 5, d);
            }
            double s = 0.0;   // column-major traversal of the 6 x D product
#if ORACLE_EIGEN_REDUX
            s = eigen_redux_block(6, D, [&](int r, int d) { return B1[r * D + d] * adj[(6 * i + 3 + r) * D + d]; });
#else
            for (int d = 0; d < D; d++)
                for (int r = 0; r < 6; r++) s += B1[r * D + d] * adj[(6 * i + 3 + r) * D + d];
#endif
            gdT[i] += s;
        }
        double B2[3 * 4];
        for (int d = 0; d < D; d++) {
            B2[0 * D + d] = -(C(6 * N - 5, d) + 2.0 * T1[N - 1] * C(6 * N - 4, d) + 3.0 * T2[N - 1] * C(6 * N - 3, d) +
                              4.0 * T3[N - 1] * C(6 * N - 2, d) + 5.0 * T4[N - 1] * C(6 * N - 1, d));
            B2[1 * D + d] = -(2.0 * C(6 * N - 4, d) + 6.0 * T1[N - 1] * C(6 * N - 3, d) + 12.0 * T2[N - 1] * C(6 * N - 2, d) +
                              20.0 * T3[N - 1] * C(6 * N - 1, d));
            B2[2 * D + d] = -(6.0 * C(6 * N - 3, d) + 24.0 * T1[N - 1] * C(6 * N - 2, d) + 60.0 * T2[N - 1] * C(6 * N - 1, d));
        }
        double s = 0.0;
#if ORACLE_EIGEN_REDUX
        s = eigen_redux_block(3, D, [&](int r, int d) { return B2[r * D + d] * adj[(6 * N - 3 + r) * D + d]; });
#else
        for (int d = 0; d < D; d++)
            for (int r = 0; r < 3; r++) s += B2[r * D + d] * adj[(6 * N - 3 + r) * D + d];
#endif
        gdT[N - 1] += s;
    }
};

// se2traj.hpp:819-870
struct MincoSE2 {
    MinJerk pos, yaw;
    void reset(int piece_xy, int piece_yaw) { pos.reset(piece_xy, 2); yaw.reset(piece_yaw, 1); }
    double getTrajJerkCost() const { return pos.getTrajJerkCost() + yaw.getTrajJerkCost(); }
};

}  // namespace orc
