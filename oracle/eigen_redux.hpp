// ORACLE -- TEST INFRASTRUCTURE ONLY.  The summation ORDER of Eigen's reductions, restated from the published source of Eigen 3.3.7 (the version
// ROS Noetic / Ubuntu 20.04 ships; the reference pins none: find_package(Eigen3 REQUIRED), back_end/CMakeLists.txt:17).  Eigen is absent from this
// image, so this is a restatement of its published algorithm, not a build of it ("parity unpinned" stands).
//
// The reference compiles with "-O3 -Wall -g" and no -march (back_end/CMakeLists.txt:7): on x86-64 that is SSE2, i.e. Eigen packets of TWO doubles and no
// fused multiply-add.  Reductions over dynamic vectors -- VectorXd::dot, squaredNorm, norm, sum (utils/lbfgs.hpp:297, 340, 548, 654-656, 673, 694-708;
// alm_traj_opt.cpp:342-343, 634-635, 642-643) -- run Eigen/src/Core/Redux.h, redux_impl<Func, Derived, LinearVectorizedTraversal, NoUnrolling>::run:
//     alignedStart = 0  (a.dot(b) reduces the CwiseBinaryOp a.conj() * b, squaredNorm a CwiseUnaryOp: no direct access, first_default_aligned() = 0)
//     two packet accumulators over strides of 2 * PacketSize = 4 coefficients, combined packet-wise, a possible fifth..  last whole packet added,
//     predux = lane 0 + lane 1, then the scalar tail in order.
// i.e. four interleaved partial sums  s0 = e0 + e4 + e8 .., s1 = e1 + e5 .., s2 = e2 + e6 .., s3 = e3 + e7 ..,  result ((s0 + s2 [+ last packet lane 0]) +
// (s1 + s3 [+ lane 1])) + tail -- against the plain left-to-right order of the default oracle build.  Every product e_i = a_i * b_i is rounded on its own.
// Fixed-size blocks -- the 6 x Dim / 3 x Dim products of MinJerkOpt::calGradCTtoQT (utils/se2traj.hpp:793, 814) -- are reduced by
//     Dim = 1 (yaw): LinearVectorizedTraversal + CompleteUnrolling, redux_vec_unroller: the packets split in halves recursively, p0 + (p1 + p2), predux;
//     Dim = 2 (xy) : the block of a dynamic matrix has no linear access -> SliceVectorizedTraversal: ONE packet accumulator over the columns' packets in
//                    order (column 0 rows 0-1, 2-3, 4-5, column 1 ..), predux (6 x 2); the 3 x 2 block is below the slice threshold (inner size >= 3 packets)
//                    and is summed coefficient by coefficient, column-major -- the default build's order.
// Built into the oracle with -DORACLE_EIGEN_REDUX=1 (tests/sensitivity.py solve_with_eigen_order_oracle): a second oracle whose only difference is this
// association, to show that the bucket tables of DESIGN.md section 6 do not depend on it.
#pragma once
#include <cstddef>

namespace orc {

#ifndef ORACLE_EIGEN_REDUX
#define ORACLE_EIGEN_REDUX 0
#endif

// sum_{i < size} coeff(i) in the order of redux_impl<.., LinearVectorizedTraversal, NoUnrolling> with PacketSize = 2, alignedStart = 0
template <class F>
inline double eigen_redux_linear(int size, F coeff) {
    const int ps = 2;
    const int alignedSize2 = (size / (2 * ps)) * (2 * ps), alignedSize = (size / ps) * ps;
    double res;
    if (alignedSize) {
        double p0[2] = {coeff(0), coeff(1)};
        if (alignedSize > ps) {
            double p1[2] = {coeff(2), coeff(3)};
            for (int i = 2 * ps; i < alignedSize2; i += 2 * ps) {
                p0[0] += coeff(i); p0[1] += coeff(i + 1);
                p1[0] += coeff(i + 2); p1[1] += coeff(i + 3);
            }
            p0[0] += p1[0]; p0[1] += p1[1];
            if (alignedSize > alignedSize2) { p0[0] += coeff(alignedSize2); p0[1] += coeff(alignedSize2 + 1); }
        }
        res = p0[0] + p0[1];
        for (int i = alignedSize; i < size; i++) res += coeff(i);
    } else {
        res = coeff(0);
        for (int i = 1; i < size; i++) res += coeff(i);
    }
    return res;
}

// rows x dim block products of calGradCTtoQT: e(r, d), r < rows (6 or 3), d < dim
template <class F>
inline double eigen_redux_block(int rows, int dim, F e) {
    if (dim == 1) {
        if (rows == 6) {                                   // three packets, complete unrolling: p0 + (p1 + p2)
            const double a0 = e(2, 0) + e(4, 0), a1 = e(3, 0) + e(5, 0);
            return (e(0, 0) + a0) + (e(1, 0) + a1);
        }
        // three coefficients: one packet (rows 0-1) then the scalar row 2  (size 3: alignedSize 2)
        return (e(0, 0) + e(1, 0)) + e(2, 0);
    }
    if (rows < 6) {                                        // MaySliceVectorize needs InnerMaxSize >= 3 * PacketSize: the 3 x 2 block takes the default traversal,
        double res = e(0, 0);                              // coefficient by coefficient in column-major order
        for (int r = 1; r < rows; r++) res += e(r, 0);
        for (int d = 1; d < dim; d++)
            for (int r = 0; r < rows; r++) res += e(r, d);
        return res;
    }
    // slice-vectorised: one packet accumulator over (column, row pairs), predux, then the rows that fill no packet, column by column
    const int packed = (rows / 2) * 2;
    double p[2] = {e(0, 0), e(1, 0)};
    for (int d = 0; d < dim; d++)
        for (int r = d == 0 ? 2 : 0; r < packed; r += 2) { p[0] += e(r, d); p[1] += e(r + 1, d); }
    double res = p[0] + p[1];
    for (int d = 0; d < dim; d++)
        for (int r = packed; r < rows; r++) res += e(r, d);
    return res;
}

}  // namespace orc
