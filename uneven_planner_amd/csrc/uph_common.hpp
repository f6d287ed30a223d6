// Shared POD types and helpers of the gfx950 back-end (device code + the host code that launches it).
// The algorithms in solver_program.hpp / terrain_dev.hpp are written once as "workgroup programs": scalar control
// flow that every lane executes identically, with data-parallel regions expressed through a workgroup object
// (pfor / sum / max).  On the GPU that object is DevWG (uph_kernels.hip: LDS + wave shuffles + s_barrier); the
// test-suite also instantiates the same source with a sequential HostWG (tests/emu/) to debug the state machine
// without a GPU.  HostWG is test scaffolding only -- libunevenhip.so contains no CPU path.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define UPH_HD __host__ __device__ __forceinline__
#define UPH_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define UPH_HD inline
#define UPH_NOINLINE __attribute__((noinline))
#endif

// pointers fetched from a struct in memory lose their address space; casting them back to the global space turns flat_load
// (which also ticks the LDS counter) into global_load.  Plain pointer on the host (emulator) build.
#if defined(__HIP_DEVICE_COMPILE__)
#define UPH_AS_GLOBAL(p) ((const __attribute__((address_space(1))) double*)(p))
#else
#define UPH_AS_GLOBAL(p) (p)
#endif

// diagnostic builds only (tools/one_kernel.sh ... -DUPH_ISA_MARKS=1): named comments in the compiler's assembly output, so that tools/isa_spills.py can say which
// phase of the workgroup program a scratch access / a stretch of instructions belongs to.  Empty in every shipped build.
#if defined(UPH_ISA_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define UPH_MARK(name) asm volatile("; UPHMARK " name)
#else
#define UPH_MARK(name) do { } while (0)
#endif

namespace uph {

constexpr int MAX_PIECE_XY = 128;
constexpr int MAX_PIECE_YAW = 256;
constexpr int MAX_MEM = 256;
constexpr int MAX_PAST = 8;

constexpr double delta_sigl = 0.01;        // back_end/include/back_end/alm_traj_opt.h:16-19
constexpr double cur_scale = 10.0;
constexpr double sig_scale = 1000.0;
constexpr double scale_trick_jerk = 1000.0;

// L-BFGS status codes, back_end/include/utils/lbfgs.hpp:135-184
enum {
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

// Terrain grid in HBM: SoA planes, reference cell order (x slowest, yaw fastest; uneven_map.h:427-435)
struct GridDev {
    int nx, ny, nyaw;             // dimensions of the WHOLE grid (index clamping, uneven_map.h:398-409)
    int x_off, nx_hold;           // rows held in memory: global x index of the first one and their number (x_off = 0, nx_hold = nx unless the map is a tile)
    int ix_off = 0, iy_off = 0;   // local frame of one trajectory (TrajFrame): the lookup's floor() index counts cells from the frame's corner, + these = the grid's index
    double xy_res, yaw_res, xy_inv, yaw_inv;
    double origin[3], minb[3], maxb[3];
    double lo[3], hi[3];          // minb + 1e-4, maxb - 1e-4 (isInMap), formed once on the host so that they stay scalar operands
    double half_xy, half_yaw;     // 0.5 * resolution
    double gravity;
    const double* cells;          // array of cells {z, sigma, zb.x, zb.y} (the build output, uneven_map.h:36-64 order): what every lookup gathers
    const float* cells32;         // non-null: the same cells stored as four floats (16 bytes); cells is null then (BASELINE.json configs[4])
};

// Optimiser parameters (alm_traj_opt.h:29-53) + L-BFGS defaults the reference does not override (lbfgs.hpp:76-128)
struct OptParams {
    double rho_T, rho_ter, max_vel, max_acc_lon, max_acc_lat, max_kap, min_cxi, max_sig;
    int use_scaling;
    double beta, gamma, epsilon_con, max_iter;
    double g_epsilon, min_step, delta;
    int inner_max_iter, mem_size, past, int_K;
    int max_linesearch;
    double max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec;
    double max_vel2, max_acc_lon2, max_acc_lat2, max_kap2;      // squares of the limits (host-computed: scalar operands on the device)
};

// derived fields, same expressions the device code used to evaluate per sample
inline void finishGrid(GridDev& g) {
    for (int i = 0; i < 3; i++) { g.lo[i] = g.minb[i] + 1e-4; g.hi[i] = g.maxb[i] - 1e-4; }
    g.half_xy = 0.5 * g.xy_res; g.half_yaw = 0.5 * g.yaw_res;
}
inline void finishParams(OptParams& P) {
    P.max_vel2 = P.max_vel * P.max_vel; P.max_acc_lon2 = P.max_acc_lon * P.max_acc_lon;
    P.max_acc_lat2 = P.max_acc_lat * P.max_acc_lat; P.max_kap2 = P.max_kap * P.max_kap;
}

// MINCO knot operator for N uniform pieces in normalised time (solver_program.hpp header, minco_op_host.hpp):
// the rows (v_j, a_j), j = 1..N-1, of A(T=1)^-1 restricted to the N+5 columns whose right-hand side can be non-zero
// (head P,V,A; way-points; tail P,V,A).
struct MincoOp {
    int N;
    const double* Wt;   // knot operator, [col][row]: 2(N-1) rows (v_j, a_j of the interior knots) x (N+5) columns of beta
    const double* Wr;   // the same, [row][col]
};

// Local frame of one trajectory on a grid whose coordinates are large (BASELINE.json configs[4]: +-500 m).  The reference's lookup forms
// (x - origin) and x - cell centre (uneven_map.h:275-283) in map coordinates; 400 m from the origin those differences carry 1e-13 of relative
// rounding where the 10 m maps of the reference carry 1e-15 -- the seed the optimiser then amplifies.  On such a grid every
// trajectory is solved in a frame translated by a WHOLE number of cells to a cell corner next to its own path (shift = origin + ioff * resolution:
// the translation of every way-point and end state is exact, the cells are the same cells): the lookups run with the frame's origin (0) and bounds
// and add ioff to the cell index -- through a grid descriptor of the trajectory's own (BatchDev::grid_mem holds one per trajectory then, patched on
// the host by applyFrame; the kernel reads "its" descriptor exactly as it reads the shared one).  Way-points and the t^0 coefficients return in map
// coordinates.  Grids within +-FRAME_EXTENT of the origin -- every map of the reference -- keep the map's own frame: one shared descriptor
// (BatchDev::grid_per_traj == 0), the arithmetic and the instructions of the plain lookup.
constexpr double FRAME_EXTENT = 32.0;
struct TrajFrame {
    double fo[2];               // frame origin that replaces GridDev::origin[0..1] in the lookup (0: the frame's corner IS a cell corner)
    double lo[2], hi[2];        // GridDev::lo / hi [0..1] in frame coordinates (isInMap)
    double shift[2];            // map coordinate of the frame's corner: frame = map - shift
    int ioff[2];                // cell index of the frame's corner
};
inline void applyFrame(GridDev& g, const TrajFrame& f) {
    g.origin[0] = f.fo[0]; g.origin[1] = f.fo[1];
    g.lo[0] = f.lo[0]; g.lo[1] = f.lo[1]; g.hi[0] = f.hi[0]; g.hi[1] = f.hi[1];
    g.ix_off = f.ioff[0]; g.iy_off = f.ioff[1];
}

// One trajectory of a batch: sizes and offsets into the packed batch arrays
struct TrajDesc {
    int Nxy, Nyaw, n, S;
    int op_xy, op_yaw;          // indices into the MincoOp table
    int64_t off_x;              // into x0 / x_out / g_out            [sum n]
    int64_t off_s;              // into per-sample SoA blocks: block b has 7*S entries at 7*off_s, plane k at 7*off_s + k*S
    int64_t off_cxy, off_cyaw;  // into c_xy [sum 12 Nxy], c_yaw [sum 6 Nyaw]
    int64_t off_hist;           // into hist [sum mem * histRowDoubles(n)]
    double init_xy[6], end_xy[6], init_yaw[3], end_yaw[3];   // xy: column-major 2x3 {P,V,A}
};

// Per-trajectory scalars in HBM
struct TrajState {
    double rho, scale_fx;
    double f, jerk_cost, T_xy, T_yaw;
    int ret_code, alm_iters, lbfgs_iters, evals, last_lbfgs_ret, pad;
    long long hist_reads;       // doubles read from the L-BFGS history (two-loop), for the roofline accounting
    long long cyc[16];          // shader-clock cycles per phase: 0 generate, 1 samples, 2 scatter, 3 adjoint, 4 two-loop, 5 scaling, 6 total; 8.. sub-steps (microbench)
};

struct BatchDev {
    int B;
    const TrajDesc* desc;
    TrajState* state;
    const MincoOp* ops;
    const double* x0;   // initial guess [tau | Pxy | Pyaw] of every trajectory (resident; never overwritten)   [sum n]
    double* x;          // working / final x                [sum n]
    double* gout;       // out: gradient of the last evaluation (eval mode)
    double* dual;       // [7*sumS]  plane 0 = lambda, 1..6 = mu_k
    double* res;        // [7*sumS]  plane 0 = hx, 1..6 = gx_k
    double* scl;        // [7*sumS]  scale_cx planes
    double* cxy;        // [sum 12 Nxy]
    double* cyaw;       // [sum 6 Nyaw]
    double* hist;       // L-BFGS history, per trajectory mem rows of histRowDoubles(n): [y.s, 1/(y.s) | s padded to 64 NQ | y padded]  (pads stay zero)
    double* rs_d;       // test hook (teacher-forced L-BFGS state): direction vectors [sum n]
    double* rs;         // ... and 24 scalars per trajectory (Solver::resumeHook)
    double* report;     // [B*7]
    double* pen_gxy;    // uph_penalty_batch (Solver::penaltyOnly): gdCxy [sum 12 Nxy], gdCyaw [sum 6 Nyaw], (cost, sum gdTxy, sum gdTyaw) [B*3]
    double* pen_gyaw;
    double* pen_out;
    double* trace;      // optional [B*trace_cap] diagnostic cost trace (nullptr = off)
    int trace_cap;
    const int* order;   // optional launch order: workgroup w solves trajectory order[w] (longest first)
    const OptParams* params_mem;   // the optimiser parameters in device memory, for the same reason as grid_mem
    const GridDev* grid_mem;   // the grid descriptor in device memory (same content as the kernel argument): the penalty kernel re-reads it with scalar loads per sample chunk instead of holding ~50 SGPRs across the whole solve
    int grid_per_traj;      // 0: grid_mem is ONE descriptor shared by the batch (the map's own frame); 1: one descriptor per trajectory, index = trajectory (local frames)
    const double* thomas;   // block-LU factors of the MINCO knot system, THOMAS_DOUBLES (minco_op_host.hpp); shared by every trajectory, copied to LDS per workgroup
};

// Register diet of the sample code (UPH_DIET, the three-waves-per-SIMD build): pinv(x) is an empty asm that "uses and redefines" x -- the value has to exist in a
// register AT THIS POINT of the program, so whatever it was computed from can die here instead of being carried to a later, cheaper-looking place
// (the scheduler otherwise sinks e.g. the acceleration / jerk sums to their single late use and keeps the piece's twelve coefficients alive across the whole
// terrain gather: tools/isa_pressure.py).  Also stops common-subexpression elimination across it: a value recomputed from a pinned input is really recomputed.
#if defined(__HIP_DEVICE_COMPILE__)
UPH_HD void pinv(double& v) { asm volatile("" : "+v"(v)); }
UPH_HD void pinv(float& v) { asm volatile("" : "+v"(v)); }
UPH_HD void pinv(int& v) { asm volatile("" : "+v"(v)); }
#else
UPH_HD void pinv(double&) {}
UPH_HD void pinv(float&) {}
UPH_HD void pinv(int&) {}
#endif
#ifndef UPH_DIET
#define UPH_DIET 0
#endif

UPH_HD double dmax(double a, double b) { return a > b ? a : b; }
UPH_HD double dmin(double a, double b) { return a < b ? a : b; }

// alm_traj_opt.h:232-253
UPH_HD double expC2(double tau) { return tau > 0.0 ? ((0.5 * tau + 1.0) * tau + 1.0) : 1.0 / ((0.5 * tau - 1.0) * tau + 1.0); }
// x / y from a stored r = RN(1 / y): q0 = x r, q = fma(fma(-q0, y, x), r, q0) -- the closing steps of the IEEE division sequence
// (Markstein); three FMAs instead of the full v_rcp / Newton / fixup expansion, for denominators that are reused or loop-invariant
UPH_HD double divR(double x, double y, double r) {
    const double q0 = x * r;
    return fma(fma(-q0, y, x), r, q0);
}
// 2x2 block helpers of the MINCO knot sweeps (M row-major).  Explicit fma so that the device and the CPU emulator round alike.
UPH_HD void mv2(const double* M, double v0, double v1, double& o0, double& o1) {
    o0 = fma(M[1], v1, M[0] * v0);
    o1 = fma(M[3], v1, M[2] * v0);
}
UPH_HD void submv2(double r0, double r1, const double* M, double p0, double p1, double& o0, double& o1) {   // r - M p
    o0 = fma(-M[1], p1, fma(-M[0], p0, r0));
    o1 = fma(-M[3], p1, fma(-M[2], p0, r1));
}
// L-BFGS history row of an n-variable trajectory: the pair's curvature y.s and its reciprocal, then s and y, each padded with
// zeros to NQ = ceil(n / 64) full 64-lane registers -- the two-loop streams a pair with unconditional 16-byte loads from ONE base
UPH_HD int histNQ(int n) { return (n + 63) >> 6; }
UPH_HD int histRowDoubles(int n) { return 2 + 128 * histNQ(n); }
// Block-LU factors of the MINCO knot system (minco_op_host.hpp): table entry j = {L_j, D_j^-1}, constant from j = THOMAS_J on.
constexpr int THOMAS_J = 26;
constexpr int THOMAS_STRIDE = 8;
constexpr int THOMAS_DOUBLES = (THOMAS_J + 1) * THOMAS_STRIDE;
// Knot buffers of the MINCO solve in LDS.  The blocked sweeps (DevWG::thomasWave) give every lane THOMAS_KPL consecutive knots; with the knots packed
// back to back a lane's block starts 4 KPL doubles (xy: 128 bytes, half the 256-byte LDS bank row) after its neighbour's, so the 32 lanes of a
// ds_read_b64 group hit TWO bank positions -- 16-way replays on every load and store of the solve (the largest single source of the penalty kernel's
// LDS bank conflicts, profiles/r05c_lds_knot_buffers.txt).  One pad double after every block makes the lane stride an odd number of doubles: knot q
// (0-based along its chain, ks doubles per knot) sits at knotOff(q, ks); the buffers of generate(), which also hold the two end knots, index by
// j = q + 1 through knotOffJ.  Producers and consumers (right-hand sides, Hermite expansion, adjoint) use the same two functions.
constexpr int THOMAS_KPL = 4;
UPH_HD int knotOff(int q, int ks) { return ks * q + (q >> 2); }
UPH_HD int knotOffJ(int j, int ks) { return ks * j + ((j + 3) >> 2); }        // knotOffJ(q + 1, ks) - knotOffJ(1, ks) == knotOff(q, ks); knotOffJ(0, ks) == 0
UPH_HD int knotBufDoubles(int nknots, int ks) { return ks * nknots + (nknots >> 2) + 2; }
// the four 2x2 matrices of knot j for the forward (M z = r) or adjoint (M^T lambda = g) sweeps, C = [8 -1; -7 1]
template <bool ADJ>
UPH_HD void thomasFactors(const double* tab, int j, double* P, double* M1, double* Q, double* M2) {
    const double* e = tab + (j < THOMAS_J ? j : THOMAS_J) * THOMAS_STRIDE;
    const double* en = tab + (j + 1 < THOMAS_J ? j + 1 : THOMAS_J) * THOMAS_STRIDE;
    const double D00 = e[4], D01 = e[5], D10 = e[6], D11 = e[7];
    if (!ADJ) {
        P[0] = 1.0; P[1] = 0.0; P[2] = 0.0; P[3] = 1.0;
        M1[0] = e[0]; M1[1] = e[1]; M1[2] = e[2]; M1[3] = e[3];                 // L_j (zero at j = 1)
        Q[0] = D00; Q[1] = D01; Q[2] = D10; Q[3] = D11;
        M2[0] = fma(8.0, D00, -7.0 * D01); M2[1] = D01 - D00; M2[2] = fma(8.0, D10, -7.0 * D11); M2[3] = D11 - D10;   // D^-1 C
    } else {
        P[0] = D00; P[1] = D10; P[2] = D01; P[3] = D11;                         // D^-T
        const double z = j == 1 ? 0.0 : 1.0;                                    // a chain's first knot has no predecessor
        M1[0] = z * fma(8.0, D00, -D10); M1[1] = z * fma(-7.0, D00, D10); M1[2] = z * fma(8.0, D01, -D11); M1[3] = z * fma(-7.0, D01, D11);   // (C D^-1)^T
        Q[0] = 1.0; Q[1] = 0.0; Q[2] = 0.0; Q[3] = 1.0;
        M2[0] = en[0]; M2[1] = en[2]; M2[2] = en[1]; M2[3] = en[3];             // L_{j+1}^T
    }
}
// sin and cos of one argument, < 0.8 ulp each (measured against long double on 1.4e7 arguments up to 1e9).  Why not the device
// library's sincos: its polynomial constants are loop-invariant 64-bit literals, the compiler hoists their materialisation out
// of the solver's loops into VGPRs and -- at the register cap of the solve kernel -- spills them to scratch, so that every
// sample paid nine serialised scratch reloads inside the Horner chain.  Here the coefficients come from a small constant-memory
// table through an opaque pointer: scalar loads issued at the call, nothing to hoist, no vector registers held.
// Reduction: q = rint(x 2/pi); r + rt = x - q pi/2 with pi/2 = P1 + P2 + P3 (every product exact inside its fma, the roundings
// of the two subtractions recovered into the tail rt); kernels: fdlibm's __kernel_sin / __kernel_cos with tail.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __constant__ const double UPH_SINCOS_TAB[16] = {
#else
static const double UPH_SINCOS_TAB[16] = {
#endif
    0x1.45f306dc9c883p-1,                                                    // 0  2/pi
    0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54, -0x1.f1976b7ed8fbcp-110,    // 1-3  pi/2 in three parts
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,     // 4-9  S1..S6
    2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,      // 10-15  C1..C6
    -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11};
UPH_HD void sincosFast(double x, double& sn, double& cs) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) double* ctab_t;
    ctab_t T = (ctab_t)UPH_SINCOS_TAB;
    asm volatile("" : "+s"(T));
#else
    const double* T = UPH_SINCOS_TAB;
#endif
    const double q = rint(x * T[0]);
    const double t = fma(-q, T[1], x);
    const double w = q * T[2];
    const double r = t - w;
    double rt = (t - r) - w;
    rt = fma(-q, T[2], w) + rt;
    rt = fma(-q, T[3], rt);
    const double z = r * r;
    const double v = z * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, T[9], T[8]), T[7]), T[6]), T[5]);
    const double sr = r - ((z * (0.5 * rt - v * ps) - rt) - v * T[4]);
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, T[15], T[14]), T[13]), T[12]), T[11]), T[10]);
    const double hz = 0.5 * z;
    const double w1 = 1.0 - hz;
    const double cr = w1 + (((1.0 - w1) - hz) + (z * pc - r * rt));
    const int n = (int)((long long)q & 3);       // |x| < 2^62 / (2/pi); beyond that the argument carries no angle information
    const double s0 = (n & 1) ? cr : sr, c0 = (n & 1) ? sr : cr;
    sn = (n & 2) ? -s0 : s0;
    cs = (n == 1 || n == 2) ? -c0 : c0;
}
// ---- fp32 sample arithmetic (BASELINE.json configs[4], "fp32"): a float that absorbs doubles.  The sample-phase code is written once,
// templated on its real type R; with R = f32r every mixed expression (literal, double member, LDS operand) rounds to float and the
// arithmetic issues as fp32 (twice the fp64 FMA rate on gfx950, single-instruction sqrt / rcp), with R = double nothing changes.
struct f32r {
    float v;
    UPH_HD f32r() : v(0.f) {}
    UPH_HD f32r(double d) : v((float)d) {}
    UPH_HD f32r(float f, int) : v(f) {}
    UPH_HD operator double() const { return (double)v; }
};
UPH_HD f32r mkf(float f) { return f32r(f, 0); }
UPH_HD void pinv(f32r& a) { pinv(a.v); }
#define UPH_F32_BINOP(op)                                                                   \
    UPH_HD f32r operator op(f32r a, f32r b) { return mkf(a.v op b.v); }                       \
    UPH_HD f32r operator op(f32r a, double b) { return mkf(a.v op (float)b); }                \
    UPH_HD f32r operator op(double a, f32r b) { return mkf((float)a op b.v); }                \
    UPH_HD f32r operator op(f32r a, int b) { return mkf(a.v op (float)b); }                   \
    UPH_HD f32r operator op(int a, f32r b) { return mkf((float)a op b.v); }
UPH_F32_BINOP(+) UPH_F32_BINOP(-) UPH_F32_BINOP(*) UPH_F32_BINOP(/)
#undef UPH_F32_BINOP
#define UPH_F32_CMP(op)                                                                     \
    UPH_HD bool operator op(f32r a, f32r b) { return a.v op b.v; }                            \
    UPH_HD bool operator op(f32r a, double b) { return a.v op (float)b; }                     \
    UPH_HD bool operator op(double a, f32r b) { return (float)a op b.v; }                     \
    UPH_HD bool operator op(f32r a, int b) { return a.v op (float)b; }
UPH_F32_CMP(<) UPH_F32_CMP(>) UPH_F32_CMP(<=) UPH_F32_CMP(>=)
#undef UPH_F32_CMP
UPH_HD f32r operator-(f32r a) { return mkf(-a.v); }
UPH_HD f32r& operator+=(f32r& a, f32r b) { a.v += b.v; return a; }
UPH_HD f32r& operator+=(f32r& a, double b) { a.v += (float)b; return a; }
UPH_HD f32r& operator-=(f32r& a, f32r b) { a.v -= b.v; return a; }
UPH_HD f32r& operator-=(f32r& a, double b) { a.v -= (float)b; return a; }
UPH_HD f32r& operator*=(f32r& a, f32r b) { a.v *= b.v; return a; }
UPH_HD double& operator+=(double& a, f32r b) { a += (double)b.v; return a; }      // accumulators stay double
// (overloads in this namespace hide the global math functions for unqualified calls: re-export the double versions next to them)
using ::sqrt; using ::fabs; using ::floor; using ::rint;
UPH_HD f32r sqrt(f32r a) { return mkf(sqrtf(a.v)); }
UPH_HD f32r fabs(f32r a) { return mkf(fabsf(a.v)); }
UPH_HD f32r floor(f32r a) { return mkf(floorf(a.v)); }
UPH_HD f32r rint(f32r a) { return mkf(rintf(a.v)); }
UPH_HD f32r divR(f32r x, f32r, f32r r) { return mkf(x.v * r.v); }               // stored reciprocal: one multiply is within an fp32 ulp or two
UPH_HD void sincosFast(f32r x, f32r& sn, f32r& cs) { float s_, c_; sincosf(x.v, &s_, &c_); sn = mkf(s_); cs = mkf(c_); }
UPH_HD f32r normSO2(f32r yaw) {
    const float PI = 3.14159265358979323846f;
    float y = yaw.v;
    for (int it = 0; it < 4096 && y < -PI; it++) y += 2 * PI;
    for (int it = 0; it < 4096 && y > PI; it++) y -= 2 * PI;
    return mkf(y);
}
template <class R> UPH_HD R toReal(float v);
template <> UPH_HD double toReal<double>(float v) { return (double)v; }
template <> UPH_HD f32r toReal<f32r>(float v) { return mkf(v); }
template <class R> UPH_HD int toInt(R v);
template <> UPH_HD int toInt<double>(double v) { return (int)v; }
template <> UPH_HD int toInt<f32r>(f32r v) { return (int)v.v; }

UPH_HD double logC2(double T) { return T > 1.0 ? (sqrt(2.0 * T - 1.0) - 1.0) : (1.0 - sqrt(2.0 / T - 1.0)); }
UPH_HD double getTtoTauGrad(double tau) {
    if (tau > 0) return tau + 1.0;
    double denSqrt = (0.5 * tau - 1.0) * tau + 1.0;
    return (1.0 - tau) / (denSqrt * denSqrt);
}

// UnevenMap::normSO2 (uneven_map.cpp:63-70); the loops are bounded so that a non-finite / absurd yaw cannot hang a wave
UPH_HD double normSO2(double yaw) {
    const double PI = 3.14159265358979323846;
    for (int it = 0; it < 4096 && yaw < -PI; it++) yaw += 2 * PI;
    for (int it = 0; it < 4096 && yaw > PI; it++) yaw -= 2 * PI;
    return yaw;
}

}  // namespace uph
