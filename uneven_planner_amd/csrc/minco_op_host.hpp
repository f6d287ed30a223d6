// Host-side construction of the MINCO knot operator for N uniform pieces in normalised time (see solver_program.hpp).
// A quintic piece is fixed by position / velocity / acceleration at its two ends (Hermite form), so all that has to be solved
// for is (v_j, a_j) at the N-1 interior knots: W = the rows (c1 of piece j, 2 c2 of piece j), j = 1..N-1, of the restricted
// inverse described below -- a 2(N-1) x (N+5) operator, three times smaller than the 6N x (N+5) coefficient operator.
// Rows/columns follow MinJerkOpt::generate (back_end/include/utils/se2traj.hpp:612-674) with every T_i = 1:
//   rows 0-2 head P,V,A; per knot i: 6i+3 jerk continuity, 6i+4 snap continuity, 6i+5 way-point, 6i+6..8 C0,C1,C2;
//   rows 6N-3..6N-1 tail P,V,A.
// The operator keeps the N+5 columns of A^-1 that can meet a non-zero right-hand side:
//   column order = [row 0, 1, 2 | rows 6i+5 (i = 0..N-2) | rows 6N-3, 6N-2, 6N-1].
// Solved once per N with partially pivoted banded elimination in long double, then rounded to fp64, so the operator
// itself carries no more than one rounding per entry.
#pragma once
#include <cmath>
#include <vector>

#include "uph_common.hpp"

namespace uph {

inline void buildMincoOp(int N, std::vector<double>& Wt /* [col][row] */, std::vector<double>& Wr /* [row][col] */) {
    const int n = 6 * N, nc = N + 5;
    typedef long double R;
    std::vector<R> A((size_t)n * n, 0.0L);
    auto a = [&](int i, int j) -> R& { return A[(size_t)i * n + j]; };
    a(0, 0) = 1; a(1, 1) = 1; a(2, 2) = 2;
    for (int i = 0; i < N - 1; i++) {
        a(6 * i + 3, 6 * i + 3) = 6; a(6 * i + 3, 6 * i + 4) = 24; a(6 * i + 3, 6 * i + 5) = 60; a(6 * i + 3, 6 * i + 9) = -6;
        a(6 * i + 4, 6 * i + 4) = 24; a(6 * i + 4, 6 * i + 5) = 120; a(6 * i + 4, 6 * i + 10) = -24;
        for (int k = 0; k < 6; k++) a(6 * i + 5, 6 * i + k) = 1;
        for (int k = 0; k < 6; k++) a(6 * i + 6, 6 * i + k) = 1;
        a(6 * i + 6, 6 * i + 6) = -1;
        for (int k = 1; k < 6; k++) a(6 * i + 7, 6 * i + k) = k;
        a(6 * i + 7, 6 * i + 7) = -1;
        a(6 * i + 8, 6 * i + 2) = 2; a(6 * i + 8, 6 * i + 3) = 6; a(6 * i + 8, 6 * i + 4) = 12; a(6 * i + 8, 6 * i + 5) = 20;
        a(6 * i + 8, 6 * i + 8) = -2;
    }
    for (int k = 0; k < 6; k++) a(6 * N - 3, 6 * N - 6 + k) = 1;
    for (int k = 1; k < 6; k++) a(6 * N - 2, 6 * N - 6 + k) = k;
    a(6 * N - 1, 6 * N - 4) = 2; a(6 * N - 1, 6 * N - 3) = 6; a(6 * N - 1, 6 * N - 2) = 12; a(6 * N - 1, 6 * N - 1) = 20;
    // right-hand sides: unit vectors of the live rows
    std::vector<int> live(nc);
    live[0] = 0; live[1] = 1; live[2] = 2;
    for (int i = 0; i < N - 1; i++) live[3 + i] = 6 * i + 5;
    live[N + 2] = 6 * N - 3; live[N + 3] = 6 * N - 2; live[N + 4] = 6 * N - 1;
    std::vector<R> X((size_t)n * nc, 0.0L);
    for (int c = 0; c < nc; c++) X[(size_t)live[c] * nc + c] = 1.0L;
    const int p = 6, q = 12;   // lower bandwidth 6; upper grows to 6+6 under row exchanges
    for (int k = 0; k < n; k++) {
        int piv = k;
        R best = fabsl(a(k, k));
        for (int i = k + 1; i <= k + p && i < n; i++)
            if (fabsl(a(i, k)) > best) { best = fabsl(a(i, k)); piv = i; }
        if (piv != k) {
            for (int j = k; j <= k + q && j < n; j++) std::swap(a(k, j), a(piv, j));
            for (int c = 0; c < nc; c++) std::swap(X[(size_t)k * nc + c], X[(size_t)piv * nc + c]);
        }
        const R d = a(k, k);
        for (int i = k + 1; i <= k + p && i < n; i++) {
            const R f = a(i, k) / d;
            if (f == 0.0L) continue;
            for (int j = k; j <= k + q && j < n; j++) a(i, j) -= f * a(k, j);
            for (int c = 0; c < nc; c++) X[(size_t)i * nc + c] -= f * X[(size_t)k * nc + c];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        for (int c = 0; c < nc; c++) {
            R s = X[(size_t)k * nc + c];
            for (int j = k + 1; j <= k + q && j < n; j++) s -= a(k, j) * X[(size_t)j * nc + c];
            X[(size_t)k * nc + c] = s / a(k, k);
        }
    }
    // knot rows: r = 2(j-1) -> v_j = c1 of piece j,  r = 2(j-1)+1 -> a_j = 2 c2 of piece j   (j = 1..N-1)
    // (a single piece has no interior knot: two all-zero rows are kept so that readers with unconditional row loads and zero weights
    // -- initScaling's per-sample gathers -- stay in bounds)
    const int nr = 2 * (N - 1);
    Wt.assign((size_t)(nr > 2 ? nr : 2) * nc, 0.0);
    Wr.assign((size_t)(nr > 2 ? nr : 2) * nc, 0.0);
    for (int j = 1; j < N; j++)
        for (int w = 0; w < 2; w++) {
            const int r = 2 * (j - 1) + w;
            for (int c = 0; c < nc; c++) {
                const R x = X[(size_t)(6 * j + 1 + w) * nc + c];
                const double v = (double)(w == 0 ? x : 2.0L * x);
                Wr[(size_t)r * nc + c] = v;
                Wt[(size_t)c * nr + r] = v;
            }
        }
}

}  // namespace uph

namespace uph {

// ------------------------------------------------------------------------------------------------------------------------
// Block-tridiagonal form of the same system (used by generate() / adjoint(); solver_program.hpp, uph_common.hpp thomasFactors).
// With z_j = (v_j, a_j) in normalised time, jerk and snap continuity at interior knot j read (rows divided by 3 and 24)
//     A z_{j-1} + B z_j + C z_{j+1} = r_j,      r_j = ( 20 (dl+ - dl-), -15 (dl+ + dl-) ),   dl+ = p_{j+1} - p_j,  dl- = p_j - p_{j-1},
//     A = [-8 -1; -7 -1],  B = [0 6; -16 0],  C = [8 -1; -7 1]
// (from the quintic Hermite form: end jerk 60 dl - 24 v0 - 36 v1 - 3 a0 + 9 a1, start jerk 60 dl - 36 v0 - 24 v1 - 9 a0 + 3 a1,
//  end snap 360 dl - 168 v0 - 192 v1 - 24 a0 + 36 a1, start snap -360 dl + 192 v0 + 168 v1 + 36 a0 - 24 a1).
// Block LU from the left: D_1 = B, L_j = A D_{j-1}^-1, D_j = B - L_j C.  The factors do not depend on N (the left end is always
// "z_0 known"), so ONE table serves every trajectory; ||L_j|| -> spectral radius 0.43, cond(D_j) ~ 3: no pivoting needed.
// Table entry j = knot index, 8 doubles: L_j (row-major 2x2; zero for j = 1) | D_j^-1.  The factors converge geometrically (rate
// 0.43^2 per knot) and are bit-constant in fp64 from j = 26 on, so the table stops there: knot j uses entry min(j, THOMAS_J).
// 27 x 8 doubles = 1.7 KB -- small enough to sit in LDS for the whole solve (uph::thomasFactors derives the products with C).
inline void buildThomasTable(std::vector<double>& tab) {
    typedef long double R;
    struct M2 { R a, b, c, d; };
    auto mul = [](const M2& x, const M2& y) { return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; };
    auto inv = [](const M2& x) { const R det = x.a * x.d - x.b * x.c; return M2{x.d / det, -x.b / det, -x.c / det, x.a / det}; };
    const M2 A{-8, -1, -7, -1}, B{0, 6, -16, 0}, C{8, -1, -7, 1}, Z{0, 0, 0, 0};
    tab.assign((size_t)(THOMAS_J + 1) * THOMAS_STRIDE, 0.0);
    M2 D = B, Di = inv(D), L = Z;
    for (int j = 1; j <= THOMAS_J; j++) {
        if (j > 1) {
            L = mul(A, Di);
            const M2 LC = mul(L, C);
            D = M2{B.a - LC.a, B.b - LC.b, B.c - LC.c, B.d - LC.d};
            Di = inv(D);
        }
        double* t = &tab[(size_t)j * THOMAS_STRIDE];
        t[0] = (double)L.a; t[1] = (double)L.b; t[2] = (double)L.c; t[3] = (double)L.d;
        t[4] = (double)Di.a; t[5] = (double)Di.b; t[6] = (double)Di.c; t[7] = (double)Di.d;
    }
}

}  // namespace uph
