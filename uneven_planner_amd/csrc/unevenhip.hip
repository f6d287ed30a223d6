// libunevenhip.so -- optimiser half: gfx950 kernels + the C-ABI declared in include/uneven_hip.h.
// One persistent workgroup (64, 128 or 256 lanes) per trajectory runs the whole ALM / L-BFGS / MINCO solve (solver_program.hpp).
// The map half (plane-fit build) lives in map_build.hip.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/uneven_hip.h"
#include "minco_op_host.hpp"
#include "solver_program.hpp"
#include "uph_internal.hpp"

using namespace uph;

#ifndef UPH_THOMAS_KPL
#define UPH_THOMAS_KPL 4        // knots per lane of the knot solve: 64 lanes x 4 cover the UPH_MAX_PIECE_YAW - 1 = 255 knots of a yaw chain, 32 x 4 the 127 of an xy chain
#endif
#ifndef UPH_WPS128
#define UPH_WPS128 2
#endif
#ifndef UPH_TWOLOOP_PF
#define UPH_TWOLOOP_PF 5
#endif

// ------------------------------------------------------------------------------------------------ device workgroup object
// wave64 sum with DPP row rotations (no LDS traffic, no barrier): rotate-and-add inside each row of 16 lanes, then the four
// row totals are read from lanes 0/16/32/48 and added in a fixed order, so every lane gets the same bits.
template <int CTRL>
__device__ __forceinline__ double dppMov(double v) {
    // a row rotation writes every lane, so no "old" value has to be preserved: mov_dpp (undefined old) spares the two copies
    // per step that update_dpp(old = src) costs on the dependency chain of every reduction
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readLane(double v, int l) {      // l must be wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// wave-uniform values (block reduction results, ring positions, ...) are moved to SGPRs explicitly: the compiler cannot prove
// uniformity of anything that passed through LDS, and would otherwise keep loop bounds in VGPRs, branch through exec masks and
// -- worst -- park them in scratch, whose reload forces s_waitcnt vmcnt(0) and drains every prefetch in flight.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
typedef const __attribute__((address_space(1))) double* gcptr;       // read-only global pointer
__device__ __forceinline__ gcptr uniG(const double* p) {             // wave-uniform global pointer held in an SGPR pair
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (gcptr)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double writeLane(double v /*wave-uniform*/, int l /*wave-uniform*/, double old) {   // old with lane l replaced by v
    int hi = __double2hiint(old), lo = __double2loint(old);
    const int vh = __builtin_amdgcn_readfirstlane(__double2hiint(v)), vl = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int ls = __builtin_amdgcn_readfirstlane(l);
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(hi) : "s"(vh), "s"(ls) : "m0");
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(lo) : "s"(vl), "s"(ls) : "m0");
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double waveSum(double v) {
    v += dppMov<0x128>(v);   // row_ror:8
    v += dppMov<0x124>(v);   // row_ror:4
    v += dppMov<0x122>(v);   // row_ror:2
    v += dppMov<0x121>(v);   // row_ror:1
    return ((readLane(v, 0) + readLane(v, 16)) + readLane(v, 32)) + readLane(v, 48);
}

__device__ __forceinline__ double waveMax(double v) {
    double o;
    o = dppMov<0x128>(v); v = o > v ? o : v;
    o = dppMov<0x124>(v); v = o > v ? o : v;
    o = dppMov<0x122>(v); v = o > v ? o : v;
    o = dppMov<0x121>(v); v = o > v ? o : v;
    const double a = readLane(v, 0), b = readLane(v, 16), c = readLane(v, 32), d = readLane(v, 48);
    const double ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// NT lanes cooperate on one trajectory: NT = 64 (one wave, no cross-wave barrier; throughput mode, many trajectories per CU)
// or NT = 256 (four waves; lower latency for small batches).
template <int NT>
struct DevWG {
    static constexpr int NW = NT / 64;
    static constexpr int MAXM = 8;
    static constexpr int SCRATCH = 2 * NW * MAXM;   // doubles of LDS
    double* red;
    int tid, lane, wave, par;
#ifdef UPH_TL_PROF
    long long tl[6] = {0, 0, 0, 0, 0, 0};   // two-loop profile (tools/phase_breakdown.py): loop 1, loop 2, tail, steps, calls, first-row wait
#endif
#ifdef UPH_BAR_PROF
    // barrier profile (tools/phase_breakdown.py): cycles THIS wave spent from reaching a workgroup barrier (including the drain of its own
    // outstanding memory operations the compiler places before s_barrier) to leaving it, by the kind of region the barrier closes:
    // 0 parallel loops / reductions, 1 two-loop (the waves that do not run the chain), 2 knot solve, 3 scatter
    long long barw[4] = {0, 0, 0, 0};
    __device__ __forceinline__ void bar(int cls) {
        const long long t0 = __builtin_readcyclecounter();
        __syncthreads();
        barw[cls] += __builtin_readcyclecounter() - t0;
    }
    // wave 0 leaves its four sums at dst[0..3]; wave 1 leaves dst[4] = classes 0 + 3 (work it shares) and dst[5] = classes 1 + 2 (chains it only waits for)
    __device__ __forceinline__ void dumpBar(long long* dst) {
        if (lane == 0 && wave == 0) for (int q = 0; q < 4; q++) dst[q] = barw[q];
        if (lane == 0 && wave == 1) { dst[4] = barw[0] + barw[3]; dst[5] = barw[1] + barw[2]; }
    }
#else
    __device__ __forceinline__ void bar(int) { __syncthreads(); }
#endif
    __device__ DevWG(double* scratch) : red(scratch), tid(threadIdx.x), lane(threadIdx.x & 63), wave(uni((int)(threadIdx.x >> 6))), par(0) {}
    // The lane index, re-read through an opaque barrier at the start of every parallel region: everything derived from it (LDS
    // addresses, task decompositions) is invariant across the solver's outer loops, so the compiler would otherwise hoist those
    // computations out of the L-BFGS / ALM loops, keep them live through the register-starved sample code and spill them to
    // scratch -- a scratch reload (s_waitcnt vmcnt(0), memory latency) in place of a shift and an add.
    __device__ __forceinline__ int ftid() const { int v = tid; asm volatile("" : "+v"(v)); return v; }
    __device__ __forceinline__ int flane() const { int v = lane; asm volatile("" : "+v"(v)); return v; }

    template <class F>
    __device__ __forceinline__ void pfor(int n, F f) {
        for (int i = ftid(); i < n; i += NT) f(i);
        bar(0);
    }
    // pfor with the tasks dealt to the waves in REVERSE order (task 0 on the last wave): the companion of scatterXY17, whose tiles go to
    // the waves in forward order
    template <class F>
    __device__ __forceinline__ void pforRev(int n, F f) {
        for (int i = (NW - 1 - wave) * 64 + flane(); i < n; i += NT) f(i);
        bar(3);
    }
    // ---------------------------------------------------------------------------------------------------------------------------
    // xy half of Solver::scatterChunk on the matrix cores (alm_traj_opt.cpp:966-979 for all pieces of a sample chunk at once):
    //     G(6 x 2P) += B(6 x 51) R(51 x 2P),    v_mfma_f64_16x16x4_f64, 13 steps of four contraction indices per tile of 16 columns
    // contraction index q = 17 f + j (f = 0, 1, 2: grad_p, grad_v, grad_a; j = sample of the piece), column = 2 (piece of the chunk) + dim:
    //     B[k][q] = s_j^k, k s_j^(k-1), k (k-1) s_j^(k-2)  for f = 0, 1, 2   (power table wtab[j][0..5]; rows k >= 6 of the tile are zero)
    //     R[q][col] = record field 2 f + dim of the piece's sample j, zero when that sample is not in this chunk.
    // Operand layout of the instruction (tools/micro/mfma_f64_probe.hip): A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, one double each;
    // D[i][j] in lane 16 (i & 3) + j, register i >> 2.  Every out-of-range operand is SELECTED to zero (never multiplied by zero: the word read
    // instead belongs to a neighbouring array).  Tiles go to the waves round-robin; no barrier here (the caller's yaw pass ends with one).
    // Two accumulators alternate so that consecutive instructions do not wait for each other's result.
#ifndef UPH_MFMA_SCATTER
#define UPH_MFMA_SCATTER 1
#endif
    static constexpr bool MFMA_SCATTER = UPH_MFMA_SCATTER != 0;
    __device__ __forceinline__ void scatterXY17(const double* rec, const double* wtab, double* Gxy, int i0_, int P_, int s0_, int cnt_) {
        typedef double d4_t __attribute__((ext_vector_type(4)));
        constexpr int K1 = 17, NQ = 3 * K1, CHP = NT + 1;
        const int P = uni(P_), s0 = uni(s0_), cnt = uni(cnt_), i0 = uni(i0_);
        const int ntile = (2 * P + 15) >> 4;
        const int ln = flane();
        const int c = ln & 15, kk = (ln >> 4) & 3;
        const double c1 = (double)c, c2 = (double)(c * (c - 1));          // the row's factors k and k (k-1)
        for (int t = wave; t < ntile; t += NW) {
            const int col = 16 * t + c, pi = col >> 1, dd = col & 1;
            const int pbase = (i0 + pi) * K1 - s0;                       // slot of the piece's sample j = 0 (may lie before / beyond the chunk)
            const int jlo = pbase < 0 ? -pbase : 0;
            int jhi = cnt - pbase < K1 ? cnt - pbase : K1;
            if (pi >= P) jhi = 0;
            const double* rb = rec + dd * CHP + pbase;
            d4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int st = 0; st < (NQ + 3) / 4; st++) {
                const int q = 4 * st + kk;
                const int f = (q >= K1 ? 1 : 0) + (q >= 2 * K1 ? 1 : 0) + (q >= 3 * K1 ? 1 : 0);     // (f = 3: the pad q = 51, masked)
                const int j = q - K1 * f;
                const double wv = wtab[6 * j + c - f];
                const double rv = rb[2 * f * CHP + j];
                const double cf = f == 0 ? 1.0 : (f == 1 ? c1 : c2);
                const double a = (c < 6 && c >= f && q < NQ) ? cf * wv : 0.0;
                const double b = (j >= jlo && j < jhi && q < NQ) ? rv : 0.0;
                if (st & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            }
            if (pi < P) {
                double* gp = Gxy + 12 * (i0 + pi) + dd;
                gp[2 * kk] += acc0[0] + acc1[0];                         // row kk
                if (kk < 2) gp[2 * (4 + kk)] += acc0[1] + acc1[1];       // rows 4, 5
            }
        }
    }
    __device__ __forceinline__ void sync() { bar(0); }
    __device__ __forceinline__ int size() const { return NT; }
// In-kernel phase timers (uph_batch_cycles, uph_microbench_batch, tools/phase_breakdown.py): ~25 s_memtime stamps per objective evaluation, each followed
// by a wait for the scalar-memory return, and sixteen 64-bit accumulators held (and spilled) across the solve.  The shipped library compiles them out
// -- every stamp reads 0, the accumulation folds away: -1.7 % per solve launch, results bit for bit the same (profiles/r05b_timers_ab.txt); the
// diagnostic build is `tools/build_variants.sh cyc="-DUPH_CYC=1"`, selected with UNEVENHIP_LIB.
#ifndef UPH_CYC
#define UPH_CYC 0
#endif
#if UPH_CYC
    __device__ __forceinline__ long long clock() { return (long long)__builtin_readcyclecounter(); }
    __device__ __forceinline__ long long realtime() { return (long long)__builtin_amdgcn_s_memrealtime(); }      // constant 100 MHz
#else
    __device__ __forceinline__ long long clock() { return 0; }
    __device__ __forceinline__ long long realtime() { return 0; }
#endif
    template <class F>
    __device__ __forceinline__ void one(F f) { if (tid == 0) f(); }

    // deterministic block reduction: strided per-lane partials -> xor butterfly inside each wave64 -> the 4 wave totals
    // are added in wave order by every lane.  Two scratch slots alternate so one barrier per call suffices.
    template <int M, class F>
    __device__ __forceinline__ void sum(int n, double* out, F f) {
        double acc[M];
#pragma unroll
        for (int m = 0; m < M; m++) acc[m] = 0.0;
        for (int i = ftid(); i < n; i += NT) f(i, acc);
#pragma unroll
        for (int m = 0; m < M; m++) acc[m] = waveSum(acc[m]);          // DPP row rotations: ~3x cheaper than ds_bpermute shuffles
        double* r = red + par * (NW * MAXM);
        par ^= 1;
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; m++) r[wave * MAXM + m] = acc[m];
        }
        bar(0);
#pragma unroll
        for (int m = 0; m < M; m++) {
            double t = r[m];
#pragma unroll
            for (int w = 1; w < NW; w++) t += r[w * MAXM + m];
            out[m] = uni(t);
        }
    }
    // The same reduction split over several parallel regions: accBegin zeroes the lane's partials, every accChunk adds its region's terms to them (and ends
    // with the barrier the region needs anyway), accEnd is the wave sum / LDS exchange -- ONCE.  The sample loop of an objective evaluation runs 3 to 6
    // chunks; reducing (cost, dT_xy, dT_yaw) after each of them cost three wave sums and an LDS round trip per chunk that nothing was waiting for.
    double pacc[MAXM];
    template <int M>
    __device__ __forceinline__ void accBegin() {
#pragma unroll
        for (int m = 0; m < M; m++) pacc[m] = 0.0;
    }
    template <int M, class F>
    __device__ __forceinline__ void accChunk(int n, F f) {
        for (int i = ftid(); i < n; i += NT) f(i, pacc);
        bar(0);
    }
    template <int M>
    __device__ __forceinline__ void accEnd(double* out) {
        double acc[M];
#pragma unroll
        for (int m = 0; m < M; m++) acc[m] = waveSum(pacc[m]);
        double* r = red + par * (NW * MAXM);
        par ^= 1;
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; m++) r[wave * MAXM + m] = acc[m];
        }
        bar(0);
#pragma unroll
        for (int m = 0; m < M; m++) {
            double t = r[m];
#pragma unroll
            for (int w = 1; w < NW; w++) t += r[w * MAXM + m];
            out[m] = uni(t);
        }
    }
    // MS sums and MM maxima of non-negative values in one pass and one barrier (the L-BFGS bookkeeping pass)
    template <int MS, int MM, class F>
    __device__ __forceinline__ void sumMax(int n, double* outS, double* outM, F f) {
        static_assert(MS + MM <= MAXM, "reduction scratch too small");
        double acc[MS], mx[MM];
#pragma unroll
        for (int m = 0; m < MS; m++) acc[m] = 0.0;
#pragma unroll
        for (int m = 0; m < MM; m++) mx[m] = 0.0;
        for (int i = ftid(); i < n; i += NT) f(i, acc, mx);
#pragma unroll
        for (int m = 0; m < MS; m++) acc[m] = waveSum(acc[m]);
#pragma unroll
        for (int m = 0; m < MM; m++) mx[m] = waveMax(mx[m]);
        double* r = red + par * (NW * MAXM);
        par ^= 1;
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MS; m++) r[wave * MAXM + m] = acc[m];
#pragma unroll
            for (int m = 0; m < MM; m++) r[wave * MAXM + MS + m] = mx[m];
        }
        bar(0);
#pragma unroll
        for (int m = 0; m < MS; m++) {
            double t = r[m];
#pragma unroll
            for (int w = 1; w < NW; w++) t += r[w * MAXM + m];
            outS[m] = uni(t);
        }
#pragma unroll
        for (int m = 0; m < MM; m++) {
            double t = r[MS + m];
#pragma unroll
            for (int w = 1; w < NW; w++) t = r[w * MAXM + MS + m] > t ? r[w * MAXM + MS + m] : t;
            outM[m] = uni(t);
        }
    }
    __device__ __forceinline__ double bcast(double v) const { return uni(v); }   // a value every lane read from the same LDS word
    // L-BFGS two-loop recursion (lbfgs.hpp:687-710) by wave 0 alone: d lives in registers (n <= 512 -> NQ <= 8 per lane, NQ a
    // compile-time constant so that short problems carry no dead loads or FMAs; the paths of more than 19 m, NQ >= 5, run with a
    // prefetch ring of two rows instead of five so that ring + direction stay within 80 registers).  A history row holds one pair: (y.s, 1/(y.s)), then
    // s and y, both zero-padded to NQ full registers (uph_common.hpp histRowDoubles), so a chain step streams its pair with
    // unconditional 16-byte loads off ONE scalar base -- no exec masking for ragged rows, half the load instructions, one
    // address register.  Lane l holds elements (2l, 2l+1) of every 128-element group and, for odd NQ, element 64 (NQ-1) + l.
    // Rows are fetched PF chain steps ahead into a register ring, dot products are DPP wave sums, the alpha of chain step i is
    // parked in LDS (the idle record buffer) -- the 2*bound-step serial chain contains no barrier and no dependent memory access.
    // The quotient x / ys of every step is formed from the stored r = RN(1/ys) as q0 = x r, q = fma(fma(-q0, ys, x), r, q0): the
    // closing steps of the IEEE division sequence (Markstein), three dependent FMAs on the chain.
    static __device__ __forceinline__ double divByStored(double x, double ys, double r) {
        const double q0 = x * r;
        return fma(fma(-q0, ys, x), r, q0);
    }
    template <int NQ, int PF>
    __device__ __forceinline__ void twoLoopT(double* d, const double* g, double* dg_out, double* al_lds, int n_, const double* hist_, int m_, int end_, int bound_, double scale_) {
        typedef double dbl2_t __attribute__((ext_vector_type(2)));
        typedef const __attribute__((address_space(1))) char* gbytes;
        constexpr int NP2 = NQ / 2;                             // 16-byte register pairs per vector
        constexpr bool ODD = (NQ & 1) != 0;
        constexpr int NPAD = 64 * NQ, ROWB = (2 + 2 * NPAD) * 8;
        const gbytes hist = (gbytes)uniG(hist_);
        const int lane = flane();
        const int n = uni(n_), m = uni(m_), end = uni(end_), bound = uni(bound_);
        const double scale = uni(scale_);
        const unsigned l16 = 16u * (unsigned)lane, l8 = 8u * (unsigned)lane;
        // element index of register q of this lane
        auto eidx = [&](int q) { return q < 2 * NP2 ? 128 * (q >> 1) + 2 * lane + (q & 1) : 64 * (NQ - 1) + lane; };
        double dr[NQ], sr[PF][NQ], yr[PF][NQ], ysr[PF], rysr[PF];
#pragma unroll
        for (int q = 0; q < NQ; q++) { const int e = eidx(q); dr[q] = e < n ? d[e] : 0.0; }
        auto fetch = [&](int slot, int j) {
            const gbytes row = hist + (size_t)((unsigned)j * (unsigned)ROWB);
            const dbl2_t yr2 = *(const __attribute__((address_space(1))) dbl2_t*)row;       // (y.s, 1 / y.s)
            ysr[slot] = yr2.x; rysr[slot] = yr2.y;
#pragma unroll
            for (int p = 0; p < NP2; p++) {
                const dbl2_t s2 = *(const __attribute__((address_space(1))) dbl2_t*)(row + 16 + 1024 * p + l16);
                const dbl2_t y2 = *(const __attribute__((address_space(1))) dbl2_t*)(row + 16 + NPAD * 8 + 1024 * p + l16);
                sr[slot][2 * p] = s2.x; sr[slot][2 * p + 1] = s2.y; yr[slot][2 * p] = y2.x; yr[slot][2 * p + 1] = y2.y;
            }
            if (ODD) {
                sr[slot][NQ - 1] = *(const __attribute__((address_space(1))) double*)(row + 16 + 512 * (NQ - 1) + l8);
                yr[slot][NQ - 1] = *(const __attribute__((address_space(1))) double*)(row + 16 + NPAD * 8 + 512 * (NQ - 1) + l8);
            }
        };
        // explicit fma chains: the contraction the compiler would pick for a*b + c*d is not unique (fma(a,b,c*d) or fma(c,d,a*b)) and was
        // observed to change with the unroll depth, i.e. the results depended on PF at rounding level
        auto rowDot = [&](const double* a, const double* b_) {
            if (NQ == 1) return a[0] * b_[0];
            if (NQ == 2) return fma(a[1], b_[1], a[0] * b_[0]);
            if (NQ == 3) return fma(a[2], b_[2], fma(a[1], b_[1], a[0] * b_[0]));
            if (NQ == 4) return fma(a[1], b_[1], a[0] * b_[0]) + fma(a[3 % NQ], b_[3 % NQ], a[2 % NQ] * b_[2 % NQ]);
            // NQ = 5 .. 8: two fma chains over the even / odd registers (fixed order), then their sum
            double e = a[0] * b_[0], o = a[1 % NQ] * b_[1 % NQ];
#pragma unroll
            for (int q = 2; q < NQ; q += 2) e = fma(a[q], b_[q], e);
#pragma unroll
            for (int q = 3; q < NQ; q += 2) o = fma(a[q], b_[q], o);
            return e + o;
        };
        // Steady-state groups of PF steps carry no conditionals: every step re-fills its ring slot unconditionally (all m ring
        // rows exist, so running a few rows past `bound` is harmless), which lets the ring live in fixed registers with exact
        // s_waitcnt vmcnt(k) waits; the bound % PF left-over steps are peeled off behind uniform branches.
        auto step1 = [&](int u, int i) {
            const double al = divByStored(waveSum(rowDot(sr[u], dr)), ysr[u], rysr[u]);
            al_lds[i] = al;                                     // (uniform address and value: one fire-and-forget LDS write)
#pragma unroll
            for (int q = 0; q < NQ; q++) dr[q] = fma(-al, yr[u][q], dr[q]);
        };
        auto step2 = [&](int u, int i) {
            const double alpha = al_lds[bound - 1 - i];        // broadcast read, issued a whole reduction ahead of its use
            const double beta = divByStored(waveSum(rowDot(yr[u], dr)), ysr[u], rysr[u]);
            const double a = alpha - beta;
#pragma unroll
            for (int q = 0; q < NQ; q++) dr[q] = fma(a, sr[u][q], dr[q]);
        };
        // compiler fence that consumes the updated direction: the step's arithmetic cannot sink below it, the next loads cannot rise above it
        auto pin = [&]() {
#pragma unroll
            for (int q = 0; q < NQ; q++) asm volatile("" : "+v"(dr[q]) : : "memory");
        };
        // ---- first loop: newest -> oldest
#ifdef UPH_TL_PROF
        const long long tp0 = __builtin_readcyclecounter();
#endif
        int jf = end;                                            // row being fetched
#pragma unroll
        for (int u = 0; u < PF; u++) { jf = jf == 0 ? m - 1 : jf - 1; fetch(u, jf); }
#ifdef UPH_TL_PROF
        { double probe = sr[0][0]; asm volatile("s_waitcnt vmcnt(0)" : "+v"(probe) : : "memory"); tl[5] += __builtin_readcyclecounter() - tp0; }
#endif
        int i = 0;
        for (; i + PF <= bound; i += PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                step1(u, i + u);
                jf = jf == 0 ? m - 1 : jf - 1;
                pin();                                           // the slot's loads stay behind its last use: no register rotation
                fetch(u, jf);
            }
        }
        {
            const int rem = bound - i;
#pragma unroll
            for (int u = 0; u < PF - 1; u++) if (u < rem) step1(u, i + u);
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) dr[q] *= scale;
#ifdef UPH_TL_PROF
        pin();
        const long long tp1 = __builtin_readcyclecounter();
        tl[0] += tp1 - tp0; tl[3] += bound; tl[4] += 1;
#endif
        // ---- second loop: oldest -> newest, starting at the row the first loop ended on; step i2 pairs with first-loop step bound-1-i2
        jf = end - bound; jf = jf < 0 ? jf + m : jf;             // oldest stored pair
#pragma unroll
        for (int u = 0; u < PF; u++) { fetch(u, jf); jf = jf + 1 == m ? 0 : jf + 1; }
        for (i = 0; i + PF <= bound; i += PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                step2(u, i + u);
                pin();
                fetch(u, jf);
                jf = jf + 1 == m ? 0 : jf + 1;
            }
        }
        {
            const int rem = bound - i;
#pragma unroll
            for (int u = 0; u < PF - 1; u++) if (u < rem) step2(u, i + u);
        }
#ifdef UPH_TL_PROF
        pin();
        const long long tp2 = __builtin_readcyclecounter();
        tl[1] += tp2 - tp1;
#endif
        // g . d for the next line search, while d is still in registers
        double gd = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; q++) { const int e = eidx(q); if (e < n) { d[e] = dr[q]; gd = fma(g[e], dr[q], gd); } }
        gd = waveSum(gd);
        if (lane == 0) *dg_out = gd;
#ifdef UPH_TL_PROF
        tl[2] += __builtin_readcyclecounter() - tp2;
#endif
    }
    // ---------------------------------------------------------------------------------------------------------------------------
    // MINCO knot solve: block-tridiagonal sweeps with the precomputed 2x2 block-LU factors (minco_op_host.hpp; table in LDS).
    //     y_j = P_j r_j - M1_j y_{j-1}  (j ascending),      z_j = Q_j y_j - M2_j z_{j+1}  (j descending),   in place on LDS buffers.
    // The recurrence is serial in j, so it runs "systolically" across the lanes of one wave: lane l owns knots 2l+1 and 2l+2 with
    // their right-hand sides and factors in registers, every step hands the lane's last value to its neighbour with one
    // whole-wave DPP shift (no LDS, no memory access inside the loop) and all lanes recompute; after s steps the first s lanes hold
    // their final values, so ceil(len / 2) steps complete a sweep.  Knots beyond the chain carry all-zero factors (they produce
    // zeros, which is exactly what the last real knot must see).  The x and y chains share one wave (lanes 0-31 / 32-63): M1 of
    // every chain's first knot is zero, so lane 32 ignores what lane 31 hands over.  Wave 0 takes the yaw chain, wave 1 (if there
    // is one) the x / y chains concurrently.
    static __device__ __forceinline__ double shr1(double v) {   // lane l <- lane l-1, lane 0 <- 0
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ double shl1(double v) {   // lane l <- lane l+1, lane 63 <- 0
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    // Blocked form: lane l owns the KPL consecutive knots KPL l + 1 .. KPL l + KPL.  With p the value handed over by the neighbour, every
    // knot of the lane is affine in p, y_k = c_k + F_k p, with c_k = P_k r_k - M_k c_{k-1} and F_k = -M_k F_{k-1} (F_1 = -M_1) composed
    // locally -- all lanes in parallel, no neighbour involved.  The serial part then carries ONLY the lane's last knot: one DPP shift
    // and one 2x2 mat-vec per step, ceil(len / KPL) steps, instead of a shift and KPL dependent mat-vecs; the other knots follow from
    // the final hand-over in parallel.  Same recurrence, other association of the products (differences at the 1e-16 level).
    // LAYOUT: which chains a wave carries.  0 = the yaw chain on all 64 lanes (<= 255 knots); 1 = the x and y chains on lanes 0-31 / 32-63
    // (<= 127 knots each); 2 = all three on ONE wave -- yaw on lanes 0-31 (<= 127 knots), x on 32-47, y on 48-63 (<= 63 knots each) --, which
    // is every trajectory of up to 64 position / 128 yaw pieces: the other wave of the workgroup then skips the solve altogether instead of
    // executing the same ~450 instructions for a dozen active lanes (7 % of an evaluation's vector instructions).  Neighbouring chains do
    // not see each other: a chain's first knot has M1 = 0 (nothing is taken from the lane below), and a chain never fills its lane range
    // completely (len < lanes x KPL), so the slot after its last knot carries zero factors and nothing is taken from the lane above.
    template <int LAYOUT, bool ADJ, int KPL>
    __device__ __forceinline__ void thomasWave(const double* tab, double* bw, int lenW_, double* bx, int lenX_) {
        const int lane = flane();
        const bool isw = LAYOUT == 0 || (LAYOUT == 2 && lane < 32);                       // this lane works on the yaw chain
        const int hl = LAYOUT == 0 ? lane : (LAYOUT == 1 ? (lane & 31) : (lane < 32 ? lane : (lane & 15)));
        const int dd = LAYOUT == 1 ? (lane >> 5) : ((lane >> 4) & 1);                     // x or y (position lanes)
        const int len = LAYOUT == 0 ? uni(lenW_) : (LAYOUT == 1 ? uni(lenX_) : (isw ? uni(lenW_) : uni(lenX_)));
        const int j0 = KPL * hl + 1;                            // the lane's first knot, 1-based
        const int ks = isw ? 2 : 4, cs = isw ? 1 : 2;           // knot / component strides of the buffer
        static_assert(KPL == THOMAS_KPL, "the knot buffers are padded per block of THOMAS_KPL knots (uph_common.hpp knotOff)");
        double* pk = isw ? bw + knotOff(j0 - 1, 2) : bx + knotOff(j0 - 1, 4) + dd;      // the lane's block: KPL knots back to back (no pad inside a block)
        const int lmax = LAYOUT == 0 ? uni(lenW_) : (LAYOUT == 1 ? uni(lenX_) : (uni(lenW_) > uni(lenX_) ? uni(lenW_) : uni(lenX_)));
        double c[KPL][2], F[KPL][4], Qm[KPL][4], M2[KPL][4];
        // ---- forward composition (ascending knots)
#pragma unroll
        for (int k = 0; k < KPL; k++) {
            const int j = j0 + k;
            const bool v = j <= len;
            const double r0 = v ? pk[k * ks] : 0.0, r1 = v ? pk[k * ks + cs] : 0.0;
            double P[4], M1[4];
            thomasFactors<ADJ>(tab, v ? j : 1, P, M1, Qm[k], M2[k]);
#pragma unroll
            for (int q = 0; q < 4; q++) { if (!v) { P[q] = 0.0; M1[q] = 0.0; Qm[k][q] = 0.0; M2[k][q] = 0.0; } }
            double t0, t1;
            mv2(P, r0, r1, t0, t1);
            if (k == 0) {
                c[0][0] = t0; c[0][1] = t1;
#pragma unroll
                for (int q = 0; q < 4; q++) F[0][q] = -M1[q];
            } else {
                submv2(t0, t1, M1, c[k - 1][0], c[k - 1][1], c[k][0], c[k][1]);
                F[k][0] = -fma(M1[1], F[k - 1][2], M1[0] * F[k - 1][0]); F[k][1] = -fma(M1[1], F[k - 1][3], M1[0] * F[k - 1][1]);
                F[k][2] = -fma(M1[3], F[k - 1][2], M1[2] * F[k - 1][0]); F[k][3] = -fma(M1[3], F[k - 1][3], M1[2] * F[k - 1][1]);
            }
        }
        const int steps = (lmax + KPL - 1) / KPL;               // (uniform: the longest chain of the wave)
        double y0 = 0.0, y1 = 0.0;
        for (int s = 0; s < steps; s++) {
            const double p0 = shr1(y0), p1 = shr1(y1);
            y0 = fma(F[KPL - 1][1], p1, fma(F[KPL - 1][0], p0, c[KPL - 1][0]));
            y1 = fma(F[KPL - 1][3], p1, fma(F[KPL - 1][2], p0, c[KPL - 1][1]));
        }
        double yk[KPL][2];
        {
            const double p0 = shr1(y0), p1 = shr1(y1);
#pragma unroll
            for (int k = 0; k < KPL; k++) {
                yk[k][0] = fma(F[k][1], p1, fma(F[k][0], p0, c[k][0]));
                yk[k][1] = fma(F[k][3], p1, fma(F[k][2], p0, c[k][1]));
            }
        }
        // ---- backward composition (descending knots): z_k = q_k - M2_k z_{k+1}, the lane's FIRST knot is handed to the left
#pragma unroll
        for (int k = KPL - 1; k >= 0; k--) {
            double t0, t1;
            mv2(Qm[k], yk[k][0], yk[k][1], t0, t1);
            if (k == KPL - 1) {
                c[k][0] = t0; c[k][1] = t1;
#pragma unroll
                for (int q = 0; q < 4; q++) F[k][q] = -M2[k][q];
            } else {
                submv2(t0, t1, M2[k], c[k + 1][0], c[k + 1][1], c[k][0], c[k][1]);
                F[k][0] = -fma(M2[k][1], F[k + 1][2], M2[k][0] * F[k + 1][0]); F[k][1] = -fma(M2[k][1], F[k + 1][3], M2[k][0] * F[k + 1][1]);
                F[k][2] = -fma(M2[k][3], F[k + 1][2], M2[k][2] * F[k + 1][0]); F[k][3] = -fma(M2[k][3], F[k + 1][3], M2[k][2] * F[k + 1][1]);
            }
        }
        double z0 = 0.0, z1 = 0.0;
        for (int s = 0; s < steps; s++) {
            const double p0 = shl1(z0), p1 = shl1(z1);
            z0 = fma(F[0][1], p1, fma(F[0][0], p0, c[0][0]));
            z1 = fma(F[0][3], p1, fma(F[0][2], p0, c[0][1]));
        }
        {
            const double p0 = shl1(z0), p1 = shl1(z1);
#pragma unroll
            for (int k = 0; k < KPL; k++) {
                if (j0 + k <= len) {
                    pk[k * ks] = fma(F[k][1], p1, fma(F[k][0], p0, c[k][0]));
                    pk[k * ks + cs] = fma(F[k][3], p1, fma(F[k][2], p0, c[k][1]));
                }
            }
        }
    }
    template <bool ADJ>
    __device__ __forceinline__ void thomasT(const double* tab, double* bw, int lenW, double* bx, int lenX) {
        UPH_MARK("thomas");
        __builtin_amdgcn_s_setprio(3);              // dependency chains: let them win the issue arbitration
        const bool packed = uni(lenW) <= 32 * UPH_THOMAS_KPL - 1 && uni(lenX) <= 16 * UPH_THOMAS_KPL - 1;
        if (packed) {
            if (wave == 0) thomasWave<2, ADJ, UPH_THOMAS_KPL>(tab, bw, lenW, bx, lenX);
        } else if (NW >= 2) {
            if (wave == 0) thomasWave<0, ADJ, UPH_THOMAS_KPL>(tab, bw, lenW, bx, lenX);
            else if (wave == 1) thomasWave<1, ADJ, UPH_THOMAS_KPL>(tab, bw, lenW, bx, lenX);
        } else {
            thomasWave<0, ADJ, UPH_THOMAS_KPL>(tab, bw, lenW, bx, lenX);
            thomasWave<1, ADJ, UPH_THOMAS_KPL>(tab, bw, lenW, bx, lenX);
        }
        __builtin_amdgcn_s_setprio(0);
        bar(2);
    }
    __device__ __forceinline__ void thomas(const double* tab, bool adj, double* bw, int lenW, double* bx, int lenX) {
        if (adj) thomasT<true>(tab, bw, lenW, bx, lenX);
        else thomasT<false>(tab, bw, lenW, bx, lenX);
    }
    __device__ __forceinline__ void twoLoop(double* d, const double* g, int n, const double* __restrict__ hist, double* dg_out, double* al_lds, int m, int end, int bound, double scale) {
        if (wave == 0) {
            __builtin_amdgcn_s_setprio(3);          // a pure dependency chain: let it win the issue arbitration against throughput-bound waves
            constexpr int PF = UPH_TWOLOOP_PF;
            const int nq = uni((n + 63) >> 6);
            if (nq == 1) twoLoopT<1, PF>(d, g, dg_out, al_lds, n, hist, m, end, bound, scale);
            else if (nq == 2) twoLoopT<2, PF>(d, g, dg_out, al_lds, n, hist, m, end, bound, scale);
            else if (nq == 3) twoLoopT<3, PF>(d, g, dg_out, al_lds, n, hist, m, end, bound, scale);
            else if (nq == 4) twoLoopT<4, PF>(d, g, dg_out, al_lds, n, hist, m, end, bound, scale);
            else if (nq == 5) twoLoopT<5, 2>(d, g, dg_out, al_lds, n, hist, m, end, bound, scale);
            else if (nq == 6) twoLoopT<6, 2>(d, g, dg_out, al_lds, n, hist, m, end, bound, scale);
            else if (nq == 7) twoLoopT<7, 2>(d, g, dg_out, al_lds, n, hist, m, end, bound, scale);
            else twoLoopT<8, 2>(d, g, dg_out, al_lds, n, hist, m, end, bound, scale);
            __builtin_amdgcn_s_setprio(0);
        }
        bar(1);
    }
    template <class F>
    __device__ __forceinline__ double maxv(int n, F f) {
        double a = 0.0;
        for (int i = ftid(); i < n; i += NT) { const double v = f(i); a = v > a ? v : a; }
        a = waveMax(a);
        double* r = red + par * (NW * MAXM);
        par ^= 1;
        if (lane == 0) r[wave * MAXM] = a;
        bar(0);
        double t = r[0];
#pragma unroll
        for (int w = 1; w < NW; w++) t = r[w * MAXM] > t ? r[w * MAXM] : t;
        return uni(t);
    }
};

// MODE 0: one (or `repeat`) objective evaluation(s)   1: reset + initScaling   2: ALM / L-BFGS solve (repeat > 0: at most that many ALM passes)
// 3: post-solve report   4: initScaling only (test hook, keeps the resident duals)   5: phase microbenchmark
// 7: continue the L-BFGS loop from a host-given state (test hook; repeat = 2 * budget + finish_pass).  8: calConstrainCostGrad alone (uph_penalty_batch; repeat = 2 * calls + store_residuals).
// A compile-time MODE gives each phase its own register budget.
// F32S: the sample phase of every objective evaluation computes in fp32 (Solver<WG, f32r>; BASELINE.json configs[4] "fp32"), everything else
// -- MINCO, L-BFGS, ALM, the scatter and the accumulations -- stays fp64.  Own instantiations: the fp64 kernels are untouched by it.
template <int NT, int WPS, int MODE, bool F32S = false>
__global__ __launch_bounds__(NT, WPS) void uph_solver_kernel(GridDev grid, OptParams P, BatchDev bd, int repeat) {
    extern __shared__ double lds[];
    const int w = blockIdx.x;
    if (w >= bd.B) return;
    const int b = bd.order ? bd.order[w] : w;
    DevWG<NT> wg(lds);
    typedef typename std::conditional<F32S, f32r, double>::type SR;
    Solver<DevWG<NT>, SR> sol(wg, grid, P, bd, b, lds + DevWG<NT>::SCRATCH);
    TrajState& st = bd.state[b];
    if (MODE == 0) sol.evalOnly(st, repeat);
    else if (MODE == 1) sol.prepare(st);
    else if (MODE == 2) sol.optimize(st, repeat);
    else if (MODE == 7) sol.resumeHook(st, repeat >> 1, repeat & 1);
    else if (MODE == 8) { if (repeat & 1) sol.template penaltyOnly<true>(st, repeat >> 1); else sol.template penaltyOnly<false>(st, repeat >> 1); }
    else if (MODE == 3) sol.report(st);
    else if (MODE == 5) sol.microbench(st, repeat);
    else sol.scalingOnly(st);
}

// upload: scale_cx = 1 (alm_traj_opt.cpp:193-203) written on the device instead of shipping 7 x sum S ones over PCIe
__global__ void uph_fill_kernel(double* __restrict__ p, size_t n, double v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void uph_terrain_kernel(GridDev grid, const double* __restrict__ pos, int n, double* __restrict__ values, double* __restrict__ grads) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double yaw = pos[3 * i + 2];
    double v[7], g[7][3];
    terrainAllWithGrad(grid, pos[3 * i], pos[3 * i + 1], yaw, cos(yaw), sin(yaw), v, g);
    for (int k = 0; k < 7; k++) {
        values[7 * i + k] = v[k];
        for (int q = 0; q < 3; q++) grads[21 * i + 3 * k + q] = g[k][q];
    }
}

#ifdef UPH_ONE_KERNEL
// device-only build of ONE instantiation (tools/one_kernel.sh: registers, spills and ISA of a kernel in seconds instead of the whole library's minutes)
#ifndef UPH_OK_F32
#define UPH_OK_F32 false
#endif
template __global__ void uph_solver_kernel<UPH_OK_NT, UPH_OK_WPS, UPH_OK_MODE, UPH_OK_F32>(GridDev, OptParams, BatchDev, int);
#else
// ------------------------------------------------------------------------------------------------ host side
namespace uph {
thread_local std::string g_last_error;
void setError(const std::string& s) { g_last_error = s; }
}  // namespace uph

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { setError("hipMalloc failed"); return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() { return (T*)p; }
};

// grow-only pinned host staging (downloads run at the PCIe rate instead of the pageable-copy rate)
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { setError("hipHostMalloc failed"); return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() { return (T*)p; }
};

struct uph_ctx {
    uph_map* map = nullptr;
    int device = 0;                         // copied at creation: the context must never dereference the map during teardown
    OptParams P;
    double rho = 1.0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // MINCO operator cache
    std::map<int, int> op_index;            // N -> index
    std::vector<MincoOp> ops_host;          // device pointers inside
    std::vector<void*> op_allocs;
    DevBuf d_ops;
    bool ops_dirty = false;
    // batch
    int B = 0;
    std::vector<TrajDesc> desc;
    std::vector<int> order;
    int64_t sum_n = 0, sum_S = 0, sum_cxy = 0, sum_cyaw = 0, sum_hist = 0;
    size_t lds_bytes = 0;                   // dynamic LDS of the main launch (largest footprint among the trajectories below the residency limit)
    size_t lds_big = 0;                     // ... and of the oversize class, launched concurrently on stream2 (0 = no such class)
    int n_main = 0;                         // order[0, n_main) main class, order[n_main, B) oversize class
    std::vector<int> rejected;              // per problem: 0, or the status code that made it unsupported (solved as a placeholder, reported as UPH_RET_UNSUPPORTED)
    int n_rejected = 0;
    bool all_rejected = false;              // the last upload failed because EVERY problem was unsupported (not because of a misuse or a resource limit)
    std::vector<TrajFrame> frames;          // per-trajectory local frames of the uploaded batch (empty: the map's own frame, uph_common.hpp TrajFrame)
    std::vector<GridDev> grid_host_framed;  // ... and the per-trajectory grid descriptors made from them (source of the asynchronous copy)
    GridDev frames_grid;                    // the map's descriptor the frames were formed from (geometry check at launch)
    GridDev framed_from;                    // the map's descriptor the resident per-trajectory descriptors were made from
    bool framed_valid = false;              // d_gridmem holds the framed descriptors of the current batch
    std::vector<int> origin;                // batch loaded by uph_optimize_batch_multi: the caller's index of each problem of this context's share (empty: identity)
    bool sample_f32 = false;                // fp32 sample arithmetic (uph_ctx_set_sample_precision)
    int xcd_group = 0;                      // experiment knob (uph_ctx_set_xcd_locality): > 0 = permute the launch order inside groups of that many workgroups for per-XCD L2 locality
    hipStream_t stream2 = nullptr;
    hipEvent_t evp0 = nullptr, evp1 = nullptr;      // prepare launch of an asynchronous solve
    bool pending = false;                   // uph_batch_solve_async issued, uph_batch_wait not yet called
    hipEvent_t ev2 = nullptr;
    std::vector<size_t> fp_bytes;           // per-trajectory LDS footprint
    int lanes = 64;                         // lanes per trajectory of the current batch (64 or 256)
    int lanes_forced = 0;                   // 0 = choose from the batch size
    int wps = 1;                            // workgroups of 256 lanes per CU the kernel is compiled for (1 or 2)
    int wps_forced = 0;                     // experiment knob: register-capped (2) or uncapped (1) build regardless of batch size
    DevBuf d_thomas, d_rsd, d_rs, d_gridmem, d_parammem;
    GridDev grid_host;                      // source of the descriptor copy (outlives the asynchronous copy)
    DevBuf d_desc, d_state, d_x, d_x0, d_gout, d_dual, d_res, d_scl, d_cxy, d_cyaw, d_hist, d_report, d_order, d_trace;
    DevBuf d_pen_gxy, d_pen_gyaw, d_pen_out;      // uph_penalty_batch outputs (allocated at its first call)
    int trace_cap = 0;                      // requested for the next upload
    int trace_cap_up = 0;                   // what the uploaded batch's trace buffer was sized for
    std::vector<TrajState> state_host;
    HostBuf h_x, h_cxy, h_cyaw, h_dual, h_res, h_scl;      // download staging
    // stats of the last solve
    double last_ms = 0.0, last_prepare_ms = 0.0;
    int64_t last_evals = 0, last_sample_evals = 0, last_iters = 0, last_hist_bytes = 0;
};

#define HIPCHK(call)                                                                               \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            setError(std::string(#call) + ": " + hipGetErrorString(_e));                           \
            return UPH_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

static BatchDev makeBatchDev(uph_ctx* c) {
    BatchDev bd;
    std::memset(&bd, 0, sizeof(bd));
    bd.B = c->B;
    bd.desc = c->d_desc.as<TrajDesc>();
    bd.state = c->d_state.as<TrajState>();
    bd.ops = c->d_ops.as<MincoOp>();
    bd.x = c->d_x.as<double>(); bd.x0 = c->d_x0.as<double>(); bd.gout = c->d_gout.as<double>();
    bd.dual = c->d_dual.as<double>(); bd.res = c->d_res.as<double>(); bd.scl = c->d_scl.as<double>();
    bd.cxy = c->d_cxy.as<double>(); bd.cyaw = c->d_cyaw.as<double>();
    bd.hist = c->d_hist.as<double>();
    bd.report = c->d_report.as<double>();
    bd.pen_gxy = c->d_pen_gxy.as<double>(); bd.pen_gyaw = c->d_pen_gyaw.as<double>(); bd.pen_out = c->d_pen_out.as<double>();
    bd.trace = c->trace_cap_up > 0 ? c->d_trace.as<double>() : nullptr;
    bd.trace_cap = c->trace_cap_up;
    bd.order = c->d_order.as<int>();
    bd.thomas = c->d_thomas.as<double>();
    bd.grid_per_traj = c->frames.empty() ? 0 : 1;
    bd.grid_mem = c->d_gridmem.as<GridDev>();
    bd.params_mem = c->d_parammem.as<OptParams>();
    bd.rs_d = c->d_rsd.as<double>(); bd.rs = c->d_rs.as<double>();
    return bd;
}

static int ensureOp(uph_ctx* c, int N) {
    auto it = c->op_index.find(N);
    if (it != c->op_index.end()) return it->second;
    std::vector<double> Mt, Mr;
    buildMincoOp(N, Mt, Mr);
    double *dt = nullptr, *dr = nullptr;
    const size_t bytes = Mt.size() * sizeof(double);
    if (hipMalloc((void**)&dt, bytes) != hipSuccess || hipMalloc((void**)&dr, bytes) != hipSuccess) { setError("hipMalloc(MincoOp) failed"); return -1; }
    if (hipMemcpy(dt, Mt.data(), bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dr, Mr.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
        setError("hipMemcpy(MincoOp) failed");
        return -1;
    }
    c->op_allocs.push_back(dt); c->op_allocs.push_back(dr);
    MincoOp op; op.N = N; op.Wt = dt; op.Wr = dr;
    const int idx = (int)c->ops_host.size();
    c->ops_host.push_back(op);
    c->op_index[N] = idx;
    c->ops_dirty = true;
    return idx;
}

// async: enqueue only (events evb / eve bracket the launch on the context's stream); the caller synchronises and reads the time later
static int launchSolver(uph_ctx* c, int mode, int repeat, bool async = false, hipEvent_t evb = nullptr, hipEvent_t eve = nullptr) {
    if (!async && c->pending) { setError("an asynchronous solve is in flight on this context: call uph_batch_wait first"); return UPH_ERR_INVALID; }
    if (!evb) { evb = c->ev0; eve = c->ev1; }
    HIPCHK(hipSetDevice(uphMapDevice(c->map)));
    GridDev grid = uphMapGrid(c->map);
    c->grid_host = grid;
    if (c->frames.empty()) {
        c->framed_valid = false;
        if (c->d_gridmem.ensure(sizeof(GridDev))) return UPH_ERR_HIP;
        HIPCHK(hipMemcpyAsync(c->d_gridmem.p, &c->grid_host, sizeof(GridDev), hipMemcpyHostToDevice, c->stream));
    } else {
        // local frames: one descriptor per trajectory = the map's with the frame's origin, bounds and cell-index offset.  The frames were formed from the
        // grid's geometry at upload: a map that has since been re-created with another origin / resolution / size would meet stale shifts -> refused.
        // The descriptors (~250 B x B) are rebuilt and uploaded only when the map's descriptor changed since the last launch (a rebuilt map may have moved
        // its cell array), not twice per solve.
        const GridDev& fg = c->frames_grid;
        if (grid.nx != fg.nx || grid.ny != fg.ny || grid.nyaw != fg.nyaw || grid.xy_res != fg.xy_res || grid.origin[0] != fg.origin[0] || grid.origin[1] != fg.origin[1] ||
            grid.lo[0] != fg.lo[0] || grid.hi[0] != fg.hi[0] || grid.lo[1] != fg.lo[1] || grid.hi[1] != fg.hi[1]) {
            setError("the map's geometry changed since this batch was uploaded (its per-trajectory local frames were formed from the old one): upload the batch again");
            return UPH_ERR_INVALID;
        }
        if (!c->framed_valid || std::memcmp(&c->framed_from, &grid, sizeof(GridDev)) != 0) {
            if (c->d_gridmem.ensure(sizeof(GridDev) * c->frames.size())) return UPH_ERR_HIP;
            c->grid_host_framed.assign(c->frames.size(), grid);
            for (size_t b = 0; b < c->frames.size(); b++) applyFrame(c->grid_host_framed[b], c->frames[b]);
            HIPCHK(hipMemcpyAsync(c->d_gridmem.p, c->grid_host_framed.data(), sizeof(GridDev) * c->frames.size(), hipMemcpyHostToDevice, c->stream));
            c->framed_from = grid;
            c->framed_valid = true;
        }
    }
    if (c->d_parammem.ensure(sizeof(OptParams))) return UPH_ERR_HIP;
    HIPCHK(hipMemcpyAsync(c->d_parammem.p, &c->P, sizeof(OptParams), hipMemcpyHostToDevice, c->stream));
    BatchDev bd = makeBatchDev(c);
    // lanes/occupancy variants: <64,1> one wave per trajectory; <256,1> four waves, one workgroup per CU (no spills, lowest latency);
    // <256,2> four waves, registers capped at 256 so that two workgroups share a CU (best throughput for large batches)
#define UPH_LAUNCH(NTL, WPS, MODE)                                                                                                     \
    do {                                                                                                                             \
        const size_t ldsmax = 160 * 1024;   /* the opt-in ceiling, not the launch size: constant, so that contexts on other host threads never lower it under a launch */ \
        HIPCHK(hipFuncSetAttribute((const void*)uph_solver_kernel<NTL, WPS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax)); \
        HIPCHK(hipEventRecord(evb, c->stream));                                                                                   \
        BatchDev bm = bd;                                                                                                            \
        bm.B = c->n_main;                                                                                                            \
        if (c->n_main < c->B) {           /* oversize class: same kernel, own LDS size, concurrent on the high-priority stream2 and */ \
            BatchDev bb = bd;             /* submitted first: these are the longest solves of the batch (longest-first scheduling)  */ \
            bb.B = c->B - c->n_main;                                                                                                 \
            bb.order = bd.order + c->n_main;                                                                                         \
            HIPCHK(hipStreamWaitEvent(c->stream2, evb, 0));                                                                       \
            hipLaunchKernelGGL((uph_solver_kernel<NTL, WPS, MODE>), dim3(bb.B), dim3(NTL), c->lds_big, c->stream2, grid, c->P, bb, repeat); \
            HIPCHK(hipEventRecord(c->ev2, c->stream2));                                                                              \
        }                                                                                                                            \
        hipLaunchKernelGGL((uph_solver_kernel<NTL, WPS, MODE>), dim3(c->n_main), dim3(NTL), c->lds_bytes, c->stream, grid, c->P, bm, repeat); \
        if (c->n_main < c->B) HIPCHK(hipStreamWaitEvent(c->stream, c->ev2, 0));                                                      \
    } while (0)
#define UPH_LAUNCH_MODE(NTL, WPS)                                                                                                      \
    do {                                                                                                                             \
        if (mode == 0) UPH_LAUNCH(NTL, WPS, 0);                                                                                      \
        else if (mode == 1) UPH_LAUNCH(NTL, WPS, 1);                                                                                 \
        else if (mode == 2) UPH_LAUNCH(NTL, WPS, 2);                                                                                 \
        else if (mode == 3) UPH_LAUNCH(NTL, WPS, 3);                                                                                 \
        else if (mode == 5) UPH_LAUNCH(NTL, WPS, 5);                                                                                 \
        else UPH_LAUNCH(NTL, WPS, 4);                                                                                                \
    } while (0)
#define UPH_LAUNCH32(NTL, WPS, MODE)                                                                                                   \
    do {                                                                                                                             \
        const size_t ldsmax = 160 * 1024;   /* the opt-in ceiling, not the launch size: constant, so that contexts on other host threads never lower it under a launch */ \
        HIPCHK(hipFuncSetAttribute((const void*)uph_solver_kernel<NTL, WPS, MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax)); \
        HIPCHK(hipEventRecord(evb, c->stream));                                                                                   \
        BatchDev bm = bd;                                                                                                            \
        bm.B = c->n_main;                                                                                                            \
        if (c->n_main < c->B) {                                                                                                      \
            BatchDev bb = bd;                                                                                                        \
            bb.B = c->B - c->n_main;                                                                                                 \
            bb.order = bd.order + c->n_main;                                                                                         \
            HIPCHK(hipStreamWaitEvent(c->stream2, evb, 0));                                                                       \
            hipLaunchKernelGGL((uph_solver_kernel<NTL, WPS, MODE, true>), dim3(bb.B), dim3(NTL), c->lds_big, c->stream2, grid, c->P, bb, repeat); \
            HIPCHK(hipEventRecord(c->ev2, c->stream2));                                                                              \
        }                                                                                                                            \
        hipLaunchKernelGGL((uph_solver_kernel<NTL, WPS, MODE, true>), dim3(c->n_main), dim3(NTL), c->lds_bytes, c->stream, grid, c->P, bm, repeat); \
        if (c->n_main < c->B) HIPCHK(hipStreamWaitEvent(c->stream, c->ev2, 0));                                                      \
    } while (0)
    if (c->sample_f32 && (mode == 0 || mode == 2) && (c->lanes == 128 || c->lanes == 256 || c->lanes == 512)) {
        // fp32 sample arithmetic: evaluation and solve kernels of the three production lane counts (scaling / report / hooks stay fp64)
        if (c->lanes == 128) { if (mode == 0) UPH_LAUNCH32(128, 2, 0); else UPH_LAUNCH32(128, 2, 2); }
        else if (c->lanes == 512) { if (mode == 0) UPH_LAUNCH32(512, 1, 0); else UPH_LAUNCH32(512, 1, 2); }
        else if (mode == 0) UPH_LAUNCH32(256, 1, 0);
        else if (c->wps == 2 && c->lds_bytes <= 80 * 1024) UPH_LAUNCH32(256, 2, 2);
        else UPH_LAUNCH32(256, 1, 2);
    }
    else if (mode == 7) {
        if (c->lanes == 128) UPH_LAUNCH(128, 2, 7);
        else if (c->lanes == 256) UPH_LAUNCH(256, 1, 7);
        else { setError("the L-BFGS resume hook is built for 128 and 256 lanes"); return UPH_ERR_INVALID; }
    }
    else if (mode == 8) {
        if (c->lanes == 128) UPH_LAUNCH(128, 2, 8);
        else if (c->lanes == 256) UPH_LAUNCH(256, 1, 8);
        else if (c->lanes == 512) UPH_LAUNCH(512, 1, 8);
        else UPH_LAUNCH(64, 1, 8);
    }
    else if (c->lanes == 64 && c->wps_forced == 2) UPH_LAUNCH_MODE(64, 2);
    else if (c->lanes == 64) UPH_LAUNCH_MODE(64, 1);
    else if (c->lanes == 128 && mode == 1) UPH_LAUNCH(128, 1, 1);      // initScaling without the register cap: 410 VGPRs and no spills instead of 162 spilled at 256 (6.05 -> 5.39 ms at B = 8192)
    else if (c->lanes == 128) UPH_LAUNCH_MODE(128, UPH_WPS128);
    else if (c->lanes == 512) UPH_LAUNCH_MODE(512, 1);
    else if (c->wps == 2 && mode == 2 && c->lds_bytes <= 80 * 1024) UPH_LAUNCH(256, 2, 2);
    else if (c->wps == 2 && mode == 5 && c->lds_bytes <= 80 * 1024) UPH_LAUNCH(256, 2, 5);
    else UPH_LAUNCH_MODE(256, 1);
#undef UPH_LAUNCH_MODE
#undef UPH_LAUNCH32
#undef UPH_LAUNCH
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(eve, c->stream));
    if (async) return UPH_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, evb, eve));
    c->last_ms = ms;
    return UPH_OK;
}

// x of every trajectory, packed, in MAP coordinates -> the resident x (way-points translated into each trajectory's local frame, if any)
static int uploadPackedX(uph_ctx* c, const double* x_packed) {
    if (c->frames.empty()) { HIPCHK(hipMemcpy(c->d_x.p, x_packed, 8 * c->sum_n, hipMemcpyHostToDevice)); return UPH_OK; }
    std::vector<double> xl(x_packed, x_packed + c->sum_n);
    for (int b = 0; b < c->B; b++) {
        const TrajDesc& t = c->desc[b];
        for (int i = 0; i < 2 * (t.Nxy - 1); i++) xl[t.off_x + 1 + i] -= c->frames[b].shift[i & 1];
    }
    HIPCHK(hipMemcpy(c->d_x.p, xl.data(), 8 * c->sum_n, hipMemcpyHostToDevice));
    return UPH_OK;
}

extern "C" {

const char* uph_last_error(void) { return g_last_error.c_str(); }
const char* uph_version(void) { return "unevenhip 0.1 (gfx950)"; }
int uph_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int uph_ctx_create(uph_map* m, const uph_opt_params* p, uph_ctx** out) {
    if (!m || !p || !out) { setError("uph_ctx_create: null argument"); return UPH_ERR_INVALID; }
    if (p->mem_size < 1 || p->mem_size > UPH_MAX_MEM || p->past < 0 || p->past > UPH_MAX_PAST || p->int_K < 1 || p->int_K > 64) {
        setError("uph_ctx_create: mem_size/past/int_K outside compiled limits");
        return UPH_ERR_LIMIT;
    }
    HIPCHK(hipSetDevice(uphMapDevice(m)));
    uph_ctx* c = new uph_ctx();
    c->map = m;
    c->device = uphMapDevice(m);
    OptParams& P = c->P;
    P.rho_T = p->rho_T; P.rho_ter = p->rho_ter; P.max_vel = p->max_vel; P.max_acc_lon = p->max_acc_lon; P.max_acc_lat = p->max_acc_lat;
    P.max_kap = p->max_kap; P.min_cxi = p->min_cxi; P.max_sig = p->max_sig; P.use_scaling = p->use_scaling;
    P.beta = p->beta; P.gamma = p->gamma; P.epsilon_con = p->epsilon_con; P.max_iter = p->max_iter;
    P.g_epsilon = p->g_epsilon; P.min_step = p->min_step; P.delta = p->delta;
    P.inner_max_iter = (int)p->inner_max_iter; P.mem_size = p->mem_size; P.past = p->past; P.int_K = p->int_K;
    // lbfgs.hpp:76-128 defaults, not overridden at alm_traj_opt.cpp:219-225
    P.max_linesearch = 64; P.max_step = 1.0e20; P.f_dec_coeff = 1.0e-4; P.s_curv_coeff = 0.9; P.cautious_factor = 1.0e-6; P.machine_prec = 1.0e-16;
    finishParams(P);
    c->rho = p->rho;
    {
        std::vector<double> tab;
        buildThomasTable(tab);
        if (c->d_thomas.ensure(tab.size() * sizeof(double))) { delete c; return UPH_ERR_HIP; }
        if (hipMemcpy(c->d_thomas.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { setError("hipMemcpy(thomas table) failed"); c->d_thomas.release(); delete c; return UPH_ERR_HIP; }
    }
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        int prio_lo = 0, prio_hi = 0;
        hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);          // (numerically lowest = highest priority)
        HIPCHK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi));
    }
    HIPCHK(hipEventCreate(&c->ev2));
    HIPCHK(hipEventCreate(&c->evp0)); HIPCHK(hipEventCreate(&c->evp1));
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    *out = c;
    return UPH_OK;
}

void uph_ctx_destroy(uph_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);             // not via c->map: the map may already have been destroyed by the caller
    for (void* p : c->op_allocs) hipFree(p);
    DevBuf* bufs[] = {&c->d_ops, &c->d_desc, &c->d_state, &c->d_x, &c->d_gout, &c->d_dual, &c->d_res, &c->d_scl, &c->d_cxy, &c->d_cyaw,
                      &c->d_hist, &c->d_report, &c->d_order, &c->d_trace, &c->d_x0, &c->d_thomas, &c->d_rsd, &c->d_rs, &c->d_gridmem, &c->d_parammem,
                      &c->d_pen_gxy, &c->d_pen_gyaw, &c->d_pen_out};
    for (DevBuf* b : bufs) b->release();
    HostBuf* hbufs[] = {&c->h_x, &c->h_cxy, &c->h_cyaw, &c->h_dual, &c->h_res, &c->h_scl};
    for (HostBuf* b : hbufs) b->release();
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->ev2) hipEventDestroy(c->ev2);
    if (c->evp0) hipEventDestroy(c->evp0);
    if (c->evp1) hipEventDestroy(c->evp1);
    delete c;
}

// lanes cooperating on one trajectory: 64, 256, or 0 = pick from the batch size (takes effect at the next upload)
int uph_ctx_set_lanes(uph_ctx* c, int32_t lanes) {
    if (!c || (lanes != 0 && lanes != 64 && lanes != 128 && lanes != 256 && lanes != 512)) return UPH_ERR_INVALID;
    c->lanes_forced = lanes;
    return UPH_OK;
}
int uph_ctx_set_xcd_locality(uph_ctx* c, int32_t group) { if (!c || group < 0) return UPH_ERR_INVALID; c->xcd_group = group; return UPH_OK; }
int uph_ctx_set_wps(uph_ctx* c, int32_t wps) { if (!c || wps < 0 || wps > 2) return UPH_ERR_INVALID; c->wps_forced = wps; return UPH_OK; }
int uph_ctx_set_sample_precision(uph_ctx* c, int32_t bits) {
    if (!c || (bits != 32 && bits != 64)) { setError("uph_ctx_set_sample_precision: 32 or 64"); return UPH_ERR_INVALID; }
    c->sample_f32 = bits == 32;
    return UPH_OK;
}
int uph_ctx_set_rho(uph_ctx* c, double rho) { if (!c) return UPH_ERR_INVALID; c->rho = rho; return UPH_OK; }
int uph_ctx_get_rho(uph_ctx* c, double* rho) { if (!c || !rho) return UPH_ERR_INVALID; *rho = c->rho; return UPH_OK; }

// diagnostic: keep the first `cap` entries of every trajectory's cost trace (0 = off); read back with uph_ctx_get_trace
int uph_ctx_set_trace(uph_ctx* c, int32_t cap) { if (!c || cap < 0) return UPH_ERR_INVALID; c->trace_cap = cap; return UPH_OK; }
int uph_ctx_get_trace(uph_ctx* c, double* out /* B x cap */) {
    if (c && c->pending) { setError("an asynchronous solve is in flight on this context: call uph_batch_wait first"); return UPH_ERR_INVALID; }
    if (!c || !out || c->trace_cap_up <= 0 || c->B <= 0) return UPH_ERR_INVALID;
    HIPCHK(hipSetDevice(uphMapDevice(c->map)));
    HIPCHK(hipMemcpy(out, c->d_trace.p, sizeof(double) * (size_t)c->B * c->trace_cap_up, hipMemcpyDeviceToHost));
    return UPH_OK;
}

// a-priori cost of one solve (relative units): the launch order inside a batch and the split of a batch over several GPUs use it
static double predictedCost(const uph_problem& pr) {
    double turn = 0.0, kink = 0.0, prev = pr.init_yaw[0];
    for (int i = 0; i <= pr.n_inner_yaw; i++) {
        const double cur = i < pr.n_inner_yaw ? pr.inner_yaw[i] : pr.end_yaw[0];
        const double dy = std::fabs(cur - prev);
        turn += dy; kink = std::max(kink, dy); prev = cur;
    }
    const double n = 2.0 * pr.n_inner_xy + pr.n_inner_yaw + 1.0;
    return std::pow(n, 0.831) * std::exp(0.129 * turn) * std::pow(1.0 + kink, 0.408);
}

int uph_batch_upload(uph_ctx* c, int32_t B, const uph_problem* probs) {
    if (!c || B <= 0 || !probs) { setError("uph_batch_upload: bad arguments"); return UPH_ERR_INVALID; }
    if (c->pending) { setError("uph_batch_upload: an asynchronous solve is in flight (uph_batch_wait first)"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(uphMapDevice(c->map)));
    const int K1 = c->P.int_K + 1, mem = c->P.mem_size;
    c->B = 0;                       // the context holds no batch until this upload has succeeded as a whole
    c->origin.clear(); c->all_rejected = false;
    int first_rj = 0;
    c->desc.assign(B, TrajDesc());
    int64_t on = 0, os = 0, ocx = 0, ocy = 0, oh = 0;
    size_t lds_d = 0;
    // small batches: four waves per trajectory (latency); large batches: two waves per trajectory, up to four trajectories resident
    // per CU (throughput; crossover measured between 2048 and 2560 trajectories on 256 CUs: below it the batch time is the longest
    // trajectory's own latency, which four waves halve)
    c->lanes = c->lanes_forced ? c->lanes_forced : (B >= 2304 ? 128 : (B <= 256 ? 512 : 256));      // up to one trajectory per CU: eight waves each (shortest latency)
    c->wps = c->wps_forced ? c->wps_forced : ((B >= 512) ? 2 : 1);
    c->fp_bytes.assign(B, 0);
    // A problem outside the compiled limits (more pieces than UPH_MAX_PIECE_*; fewer yaw pieces than position pieces; on a tile map a
    // path outside the tile) does not fail its neighbours: a two-piece placeholder takes its slot and its result carries ret_code
    // UPH_RET_UNSUPPORTED.  Only a batch with no supported problem at all is an error.  (A goal closer than one piece length -- a single
    // quintic per block, no inner way-point -- IS solved, as the reference solves it.)
    static const double ph_inner_xy[2] = {0.3, 0.0}, ph_inner_yaw[1] = {0.0};
    uph_problem placeholder;
    std::memset(&placeholder, 0, sizeof(placeholder));
    placeholder.n_inner_xy = 1; placeholder.n_inner_yaw = 1; placeholder.inner_xy = ph_inner_xy; placeholder.inner_yaw = ph_inner_yaw;
    placeholder.init_xy[2] = 0.05; placeholder.end_xy[0] = 0.6; placeholder.end_xy[2] = 0.05; placeholder.total_time = 1.44;
    std::vector<const uph_problem*> pp(B);
    c->rejected.assign(B, 0); c->n_rejected = 0;
    const GridDev tg = uphMapGrid(c->map);
    const bool tiled = tg.nx_hold < tg.nx;
    const double tile_lo = tg.origin[0] + tg.x_off * tg.xy_res, tile_hi = tg.origin[0] + (tg.x_off + tg.nx_hold) * tg.xy_res;
    std::string why;
    for (int b = 0; b < B; b++) {
        const uph_problem& q = probs[b];
        int rj = 0;
        const char* msg = nullptr;
        // (a block without inner way-points is a single quintic piece -- a goal closer than one piece length; the reference solves it, and so does this)
        if (q.n_inner_xy < 0 || q.n_inner_yaw < 0 || (q.n_inner_xy > 0 && !q.inner_xy) || (q.n_inner_yaw > 0 && !q.inner_yaw)) { rj = UPH_ERR_INVALID; msg = "uph_batch_upload: negative way-point count or missing way-point array"; }
        else if (q.n_inner_xy + 1 > UPH_MAX_PIECE_XY || q.n_inner_yaw + 1 > UPH_MAX_PIECE_YAW) { rj = UPH_ERR_LIMIT; msg = "uph_batch_upload: piece count exceeds UPH_MAX_PIECE_*"; }
        else if (q.n_inner_yaw < q.n_inner_xy) { rj = UPH_ERR_INVALID; msg = "uph_batch_upload: piece_yaw < piece_xy (the reference indexes yaw_minco.T1 with the xy piece index, alm_traj_opt.cpp:749)"; }
        else if (tiled) {                // a tile map serves the problems routed to it: the initial path must lie well inside the held rows (the grid's own border is no tile border)
            double lo = std::min(q.init_xy[0], q.end_xy[0]), hi = std::max(q.init_xy[0], q.end_xy[0]);
            for (int i = 0; i < q.n_inner_xy; i++) { lo = std::min(lo, q.inner_xy[2 * i]); hi = std::max(hi, q.inner_xy[2 * i]); }
            if ((tg.x_off > 0 && lo < tile_lo + UPH_TILE_MARGIN) || (tg.x_off + tg.nx_hold < tg.nx && hi > tile_hi - UPH_TILE_MARGIN)) { rj = UPH_ERR_INVALID; msg = "uph_batch_upload: the path does not lie inside this map tile (route it to the tile's owner)"; }
        }
        c->rejected[b] = rj;
        pp[b] = rj ? &placeholder : &q;
        if (rj) { if (!c->n_rejected) { why = msg; first_rj = rj; } c->n_rejected++; }
    }
    if (c->n_rejected == B) { c->rejected.clear(); c->n_rejected = 0; c->all_rejected = true; setError(why); return first_rj; }
    // Local frames (uph_common.hpp TrajFrame): on a grid that reaches further than FRAME_EXTENT from its origin every trajectory is solved in
    // coordinates relative to the cell corner nearest the middle of its initial path's bounding box -- a whole number of cells away from the
    // grid's origin, so the translation of the inputs is exact and the lookups see the same cells.  The reference's maps (10 m x 10 m) stay
    // in the map's frame: frames empty, the plain lookup code path.
    c->frames.clear();
    c->framed_valid = false;
    c->frames_grid = tg;
    if (std::max(std::max(std::fabs(tg.minb[0]), std::fabs(tg.maxb[0])), std::max(std::fabs(tg.minb[1]), std::fabs(tg.maxb[1]))) > FRAME_EXTENT) {
        c->frames.resize(B);
        for (int b = 0; b < B; b++) {
            const uph_problem& q = *pp[b];
            double lo[2] = {std::min(q.init_xy[0], q.end_xy[0]), std::min(q.init_xy[1], q.end_xy[1])};
            double hi[2] = {std::max(q.init_xy[0], q.end_xy[0]), std::max(q.init_xy[1], q.end_xy[1])};
            for (int i = 0; i < q.n_inner_xy; i++)
                for (int d = 0; d < 2; d++) { lo[d] = std::min(lo[d], q.inner_xy[2 * i + d]); hi[d] = std::max(hi[d], q.inner_xy[2 * i + d]); }
            TrajFrame& f = c->frames[b];
            const int nn[2] = {tg.nx, tg.ny};
            for (int d = 0; d < 2; d++) {
                long long ci = std::llround((0.5 * (lo[d] + hi[d]) - tg.origin[d]) * tg.xy_inv);
                ci = ci < 0 ? 0 : (ci > nn[d] ? nn[d] : ci);                     // (a path outside the map: the frame stays at the map's edge)
                f.ioff[d] = (int)ci;
                f.shift[d] = tg.origin[d] + (double)ci * tg.xy_res;
                f.fo[d] = 0.0;
                f.lo[d] = tg.lo[d] - f.shift[d];
                f.hi[d] = tg.hi[d] - f.shift[d];
            }
        }
    }
    for (int b = 0; b < B; b++) {
        const uph_problem& pr = *pp[b];
        const int Nxy = pr.n_inner_xy + 1, Nyaw = pr.n_inner_yaw + 1;
        TrajDesc& t = c->desc[b];
        std::memset(&t, 0, sizeof(t));
        t.Nxy = Nxy; t.Nyaw = Nyaw; t.n = 2 * pr.n_inner_xy + pr.n_inner_yaw + 1; t.S = Nxy * K1;
        t.op_xy = ensureOp(c, Nxy);
        t.op_yaw = ensureOp(c, Nyaw);
        if (t.op_xy < 0 || t.op_yaw < 0) return UPH_ERR_HIP;
        t.off_x = on; t.off_s = os; t.off_cxy = ocx; t.off_cyaw = ocy; t.off_hist = oh;
        for (int k = 0; k < 6; k++) { t.init_xy[k] = pr.init_xy[k]; t.end_xy[k] = pr.end_xy[k]; }
        for (int k = 0; k < 3; k++) { t.init_yaw[k] = pr.init_yaw[k]; t.end_yaw[k] = pr.end_yaw[k]; }
        if (!c->frames.empty()) for (int d = 0; d < 2; d++) { t.init_xy[d] -= c->frames[b].shift[d]; t.end_xy[d] -= c->frames[b].shift[d]; }      // the positions P of {P, V, A}
        on += t.n; os += t.S; ocx += 12 * Nxy; ocy += 6 * Nyaw; oh += (int64_t)mem * histRowDoubles(t.n);
        c->fp_bytes[b] = (Solver<DevWG<64>>::ldsDoubles(Nxy, Nyaw, t.n, c->lanes, mem, c->P.int_K) + 2 * (c->lanes / 64) * DevWG<64>::MAXM) * sizeof(double);
        lds_d = std::max(lds_d, Solver<DevWG<64>>::ldsDoubles(Nxy, Nyaw, t.n, c->lanes, mem, c->P.int_K));
    }
    c->sum_n = on; c->sum_S = os; c->sum_cxy = ocx; c->sum_cyaw = ocy; c->sum_hist = oh;
    c->lds_bytes = (lds_d + 2 * (c->lanes / 64) * DevWG<64>::MAXM) * sizeof(double);     // program arrays + DevWG<lanes>::SCRATCH
    if (c->lds_bytes > 160 * 1024) { setError("uph_batch_upload: trajectory does not fit the 160 KiB LDS"); return UPH_ERR_LIMIT; }
    if (c->ops_dirty) {
        if (c->d_ops.ensure(sizeof(MincoOp) * c->ops_host.size())) return UPH_ERR_HIP;
        HIPCHK(hipMemcpy(c->d_ops.p, c->ops_host.data(), sizeof(MincoOp) * c->ops_host.size(), hipMemcpyHostToDevice));
        c->ops_dirty = false;
    }
    if (c->d_desc.ensure(sizeof(TrajDesc) * B) || c->d_state.ensure(sizeof(TrajState) * B) || c->d_x.ensure(8 * on) || c->d_x0.ensure(8 * on) || c->d_gout.ensure(8 * on) ||
        c->d_dual.ensure(8 * 7 * os) || c->d_res.ensure(8 * 7 * os) || c->d_scl.ensure(8 * 7 * os) || c->d_cxy.ensure(8 * ocx) || c->d_cyaw.ensure(8 * ocy) ||
        c->d_hist.ensure(8 * oh) || c->d_report.ensure(8 * 7 * B) || c->d_order.ensure(4 * B) ||
        c->d_trace.ensure(8 * (size_t)std::max(1, c->trace_cap) * B))
        return UPH_ERR_HIP;
    // x0 = [tau | Pxy | Pyaw]  (alm_traj_opt.cpp:206-216)
    std::vector<double> x0(on);
    for (int b = 0; b < B; b++) {
        const uph_problem& pr = *pp[b];
        double* x = x0.data() + c->desc[b].off_x;
        x[0] = logC2(pr.total_time);
        for (int i = 0; i < 2 * pr.n_inner_xy; i++) x[1 + i] = pr.inner_xy[i] - (c->frames.empty() ? 0.0 : c->frames[b].shift[i & 1]);
        for (int i = 0; i < pr.n_inner_yaw; i++) x[1 + 2 * pr.n_inner_xy + i] = pr.inner_yaw[i];
    }
    // Launch order: most expensive solves first (longest-processing-time list scheduling), so that the tail of a launch -- workgroups
    // finishing below full residency -- is made of short solves.  Predicted cost = n^0.83 exp(0.13 turn) (1 + kink)^0.41 with n = number of
    // variables, turn = total heading change of the initial path and kink = its largest heading change between two consecutive yaw
    // way-points (sharp corners of the initial path are what needs many ALM passes): a log-linear fit on 8192 hill-scene solves
    // (tools/collect_cost_features.py; R^2 0.83 on held-out problems, 0.77 without the kink term).  A simulated 1024-slot schedule of the
    // measured solve times at B = 8192 gives 164 ms for this order, 186 ms for the round-1 model n exp(0.23 turn), 196 ms for
    // n-descending and 163.4 ms for the unattainable perfect order; at B = 16384: 324.0 / 329.1 / 352.6 / 323.5 ms.  Placement never
    // affects results.  (Tried and dropped: cutting the sorted list into per-XCD chunks for L2 locality -- 8 % slower.)
    {
        std::vector<double> cost(B);
        for (int b = 0; b < B; b++) cost[b] = predictedCost(*pp[b]);
        c->order.resize(B);
        std::iota(c->order.begin(), c->order.end(), 0);
        std::stable_sort(c->order.begin(), c->order.end(), [&](int a, int b2) { return cost[a] > cost[b2]; });
        // Experiment: workgroup w is dispatched to XCD w % 8, each XCD has its own 4 MB L2, and the grid (82 MB) is shared.
        // Inside groups of xcd_group consecutive entries of the cost order (similar predicted cost: the LPT property survives), entries are dealt
        // to the residues w % 8 by the map octant (x half, y quarter) of their path's midpoint, so that the workgroups of one XCD read one region.
        if (c->xcd_group >= 16) {
            const GridDev gg = uphMapGrid(c->map);
            auto octant = [&](int b) {
                const uph_problem& q = *pp[b];
                const double mx = 0.5 * (q.init_xy[0] + q.end_xy[0]), my = 0.5 * (q.init_xy[1] + q.end_xy[1]);
                const int ix = mx < 0.5 * (gg.minb[0] + gg.maxb[0]) ? 0 : 1;
                int iy = (int)((my - gg.minb[1]) / (gg.maxb[1] - gg.minb[1]) * 4.0);
                iy = iy < 0 ? 0 : (iy > 3 ? 3 : iy);
                return ix * 4 + iy;
            };
            const int G = c->xcd_group & ~7;
            for (int g0 = 0; g0 + G <= B; g0 += G) {
                std::vector<int> bins[8], spill, placed((size_t)G, -1);
                for (int k = 0; k < G; k++) bins[octant(c->order[g0 + k])].push_back(c->order[g0 + k]);
                for (int r = 0; r < 8; r++) {                                      // residue r takes its octant's entries first (cost order kept inside a bin)
                    int at = r;
                    for (int b : bins[r]) { if (at < G) { placed[at] = b; at += 8; } else spill.push_back(b); }
                }
                size_t sp = 0;
                for (int k = 0; k < G; k++) if (placed[k] < 0) placed[k] = spill[sp++];
                for (int k = 0; k < G; k++) c->order[g0 + k] = placed[k];
            }
        }
        // Residency classes.  One launch has one LDS size, and the largest trajectory of a batch would set it for all: at 128 lanes
        // a single 41 KB trajectory among 8192 pushes everybody from four workgroups per CU to three (-16 %).  Trajectories above
        // the residency limit therefore form a second class that is launched concurrently on a second stream with its own size.
        const size_t limit = c->lanes == 128 ? 40960 : (c->lanes == 256 ? 81920 : (c->lanes == 512 ? 163840 : 32768));      // 4, 2, 1 and 5 workgroups per CU
        c->n_main = B; c->lds_big = 0;
        size_t mmain = 0, mbig = 0;
        int nbig = 0;
        for (int b = 0; b < B; b++) { if (c->fp_bytes[b] > limit) { nbig++; mbig = std::max(mbig, c->fp_bytes[b]); } else mmain = std::max(mmain, c->fp_bytes[b]); }
        if (nbig > 0 && nbig < B) {
            std::stable_partition(c->order.begin(), c->order.end(), [&](int a) { return c->fp_bytes[a] <= limit; });
            c->n_main = B - nbig; c->lds_bytes = mmain; c->lds_big = mbig;
        }
    }
    c->state_host.assign(B, TrajState());
    for (int b = 0; b < B; b++) { std::memset(&c->state_host[b], 0, sizeof(TrajState)); c->state_host[b].rho = c->rho; c->state_host[b].scale_fx = 1.0; }
    HIPCHK(hipMemcpy(c->d_desc.p, c->desc.data(), sizeof(TrajDesc) * B, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_x.p, x0.data(), 8 * on, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_x0.p, x0.data(), 8 * on, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_order.p, c->order.data(), 4 * B, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_state.p, c->state_host.data(), sizeof(TrajState) * B, hipMemcpyHostToDevice));
    // duals = 0, residuals = 0, scales = 1 (alm_traj_opt.cpp:193-203) so that the test hooks see a defined state
    // (all on the context's own stream, waited for with a STREAM synchronise: a device-wide one would block this host thread on every other
    // context's solve in flight on the device -- and uploading batch k+1 while batch k solves is what uph_batch_solve_async is for)
    HIPCHK(hipMemsetAsync(c->d_hist.p, 0, 8 * oh, c->stream));          // the pads of the history rows must be (and stay) zero
    HIPCHK(hipMemsetAsync(c->d_dual.p, 0, 8 * 7 * os, c->stream));
    HIPCHK(hipMemsetAsync(c->d_res.p, 0, 8 * 7 * os, c->stream));
    hipLaunchKernelGGL(uph_fill_kernel, dim3(1024), dim3(256), 0, c->stream, c->d_scl.as<double>(), (size_t)7 * os, 1.0);
    HIPCHK(hipGetLastError());
    if (c->trace_cap > 0) HIPCHK(hipMemsetAsync(c->d_trace.p, 0, 8 * (size_t)c->trace_cap * B, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->trace_cap_up = c->trace_cap;
    c->B = B;
    return UPH_OK;
}

static int refreshStates(uph_ctx* c) {
    HIPCHK(hipMemcpy(c->state_host.data(), c->d_state.p, sizeof(TrajState) * c->B, hipMemcpyDeviceToHost));
    return UPH_OK;
}

// enqueue one solve of the uploaded batch on the context's stream and return: reset + initScaling kernel, then the ALM kernel.  Two
// contexts driven this way overlap on the GPU -- the second batch's workgroups fill the CUs the first one's tail leaves idle.
int uph_batch_solve_async(uph_ctx* c) {
    if (!c || c->B <= 0) { setError("uph_batch_solve_async: no batch uploaded"); return UPH_ERR_INVALID; }
    if (c->pending) { setError("uph_batch_solve_async: the previous asynchronous solve has not been waited for"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(uphMapDevice(c->map)));
    // every problem starts from the context's rho (Q7)
    for (int b = 0; b < c->B; b++) { std::memset(&c->state_host[b], 0, sizeof(TrajState)); c->state_host[b].rho = c->rho; c->state_host[b].scale_fx = 1.0; }
    HIPCHK(hipMemcpyAsync(c->d_state.p, c->state_host.data(), sizeof(TrajState) * c->B, hipMemcpyHostToDevice, c->stream));
    int r = launchSolver(c, 1, 1, true, c->evp0, c->evp1);      // reset + initScaling (alm_traj_opt.cpp:193-203, 231-232)
    if (r != UPH_OK) { hipStreamSynchronize(c->stream); return r; }      // (whatever part of it was queued is drained: the context stays usable)
    r = launchSolver(c, 2, 0, true, c->ev0, c->ev1);            // ALM loop (alm_traj_opt.cpp:234-271)
    if (r != UPH_OK) { hipStreamSynchronize(c->stream); if (c->stream2) hipStreamSynchronize(c->stream2); return r; }
    c->pending = true;
    return UPH_OK;
}

int uph_batch_wait(uph_ctx* c) {
    if (!c || !c->pending) { setError("uph_batch_wait: no asynchronous solve in flight"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(uphMapDevice(c->map)));
    c->pending = false;
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->evp0, c->evp1));
    c->last_prepare_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    int r = refreshStates(c);
    if (r != UPH_OK) return r;
    c->last_evals = c->last_sample_evals = c->last_iters = c->last_hist_bytes = 0;
    for (int b = 0; b < c->B; b++) {
        if (c->rejected[b]) continue;
        const TrajState& s = c->state_host[b];
        c->last_evals += s.evals;
        c->last_sample_evals += (int64_t)s.evals * c->desc[b].S;
        c->last_iters += s.lbfgs_iters;
        c->last_hist_bytes += s.hist_reads * 8;
    }
    if (c->B == 1 && !c->rejected[0]) c->rho = c->state_host[0].rho;      // one optimizeSE2Traj call: rho persists into the next (Q7).  A batch has no "next": unchanged
    return UPH_OK;
}

int uph_batch_solve(uph_ctx* c) {
    const int r = uph_batch_solve_async(c);
    return r != UPH_OK ? r : uph_batch_wait(c);
}

int uph_batch_count(const uph_ctx* c) { return c ? c->B : UPH_ERR_INVALID; }
// for a batch loaded by uph_optimize_batch_multi: idx[k] = the caller's index of problem k of this context's share (identity otherwise)
int uph_batch_origin(const uph_ctx* c, int32_t* idx) {
    if (!c || !idx || c->B <= 0) { setError("uph_batch_origin: no batch uploaded"); return UPH_ERR_INVALID; }
    for (int b = 0; b < c->B; b++) idx[b] = c->origin.empty() ? b : c->origin[b];
    return UPH_OK;
}

int uph_batch_stats(uph_ctx* c, double* kernel_ms, int64_t* evals, int64_t* sample_evals, int64_t* lbfgs_iters, int64_t* hist_bytes) {
    if (!c) return UPH_ERR_INVALID;
    if (kernel_ms) *kernel_ms = c->last_ms;
    if (evals) *evals = c->last_evals;
    if (sample_evals) *sample_evals = c->last_sample_evals;
    if (lbfgs_iters) *lbfgs_iters = c->last_iters;
    if (hist_bytes) *hist_bytes = c->last_hist_bytes;
    return UPH_OK;
}

// kernel milliseconds of the reset+initScaling launch that precedes the solve kernel in uph_batch_solve
int uph_batch_prepare_ms(uph_ctx* c, double* ms) { if (!c || !ms) return UPH_ERR_INVALID; *ms = c->last_prepare_ms; return UPH_OK; }

// diagnostic: per-trajectory phase cycle counters of the last solve, out[B][8] (see TrajState::cyc)
int uph_batch_cycles(uph_ctx* c, long long* out) {
    if (!c || c->B <= 0 || !out) return UPH_ERR_INVALID;
    for (int b = 0; b < c->B; b++) for (int q = 0; q < 16; q++) out[b * 16 + q] = c->state_host[b].cyc[q];
    return UPH_OK;
}

int uph_batch_download(uph_ctx* c, uph_result* results) {
    if (!c || c->B <= 0 || !results) { setError("uph_batch_download: bad arguments"); return UPH_ERR_INVALID; }
    if (c->pending) { setError("uph_batch_download: an asynchronous solve is in flight (uph_batch_wait first)"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(uphMapDevice(c->map)));
    int r = refreshStates(c);
    if (r != UPH_OK) return r;
    const GridDev tg = uphMapGrid(c->map);
    const bool tiled = tg.nx_hold < tg.nx;
    const double tile_lo = tg.origin[0] + tg.x_off * tg.xy_res, tile_hi = tg.origin[0] + (tg.x_off + tg.nx_hold) * tg.xy_res;
    // only what the caller asked for crosses PCIe: the per-sample arrays (duals, residuals, scales) are 3 x 7 x sum S doubles -- 1 GB at
    // B = 16384 -- and a planner that pulls trajectories (alm_traj_opt.h:165-168) passes NULL for all of them
    bool want_x = tiled, want_cxy = false, want_cyaw = false, want_dual = false, want_res = false, want_scl = false;
    for (int b = 0; b < c->B; b++) {
        const uph_result& o = results[b];
        if (c->rejected[b]) continue;
        want_x |= o.x_final != nullptr; want_cxy |= o.c_xy != nullptr; want_cyaw |= o.c_yaw != nullptr;
        want_dual |= o.lambda != nullptr || o.mu != nullptr; want_res |= o.hx != nullptr || o.gx != nullptr; want_scl |= o.scale_cx != nullptr;
    }
    struct Pull { bool want; HostBuf* h; DevBuf* d; size_t bytes; };
    const Pull pulls[6] = {{want_x, &c->h_x, &c->d_x, (size_t)8 * c->sum_n}, {want_cxy, &c->h_cxy, &c->d_cxy, (size_t)8 * c->sum_cxy},
                           {want_cyaw, &c->h_cyaw, &c->d_cyaw, (size_t)8 * c->sum_cyaw}, {want_dual, &c->h_dual, &c->d_dual, (size_t)8 * 7 * c->sum_S},
                           {want_res, &c->h_res, &c->d_res, (size_t)8 * 7 * c->sum_S}, {want_scl, &c->h_scl, &c->d_scl, (size_t)8 * 7 * c->sum_S}};
    for (const Pull& q : pulls) {
        if (!q.want) continue;
        if (q.h->ensure(q.bytes)) return UPH_ERR_HIP;
        HIPCHK(hipMemcpyAsync(q.h->p, q.d->p, q.bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    const double *x = c->h_x.as<double>(), *cxy = c->h_cxy.as<double>(), *cyaw = c->h_cyaw.as<double>(), *dual = c->h_dual.as<double>(),
                 *res = c->h_res.as<double>(), *scl = c->h_scl.as<double>();
    for (int b = 0; b < c->B; b++) {
        const TrajDesc& t = c->desc[b];
        const TrajState& s = c->state_host[b];
        uph_result& o = results[b];
        if (c->rejected[b]) {             // the slot ran a placeholder: nothing of it is the caller's (arrays are left untouched)
            o.ret_code = UPH_RET_UNSUPPORTED; o.alm_iters = o.lbfgs_iters = o.evals = 0; o.last_lbfgs_ret = c->rejected[b];
            o.cost = o.jerk_cost = o.piece_T_xy = o.piece_T_yaw = 0.0; o.rho_final = c->rho; o.scale_fx = 1.0;
            continue;
        }
        o.ret_code = s.ret_code; o.alm_iters = s.alm_iters; o.lbfgs_iters = s.lbfgs_iters; o.evals = s.evals; o.last_lbfgs_ret = s.last_lbfgs_ret;
        o.cost = s.f; o.jerk_cost = s.jerk_cost; o.piece_T_xy = s.T_xy; o.piece_T_yaw = s.T_yaw; o.rho_final = s.rho; o.scale_fx = s.scale_fx;
        if (tiled) {                     // lookups outside the held rows were clamped to the tile: such a result is not the whole grid's
            const double* xf = x + t.off_x;
            const double sx = c->frames.empty() ? 0.0 : c->frames[b].shift[0];
            for (int i = 0; i < t.Nxy - 1; i++)
                if ((tg.x_off > 0 && xf[1 + 2 * i] + sx < tile_lo + 2.0 * tg.xy_res) || (tg.x_off + tg.nx_hold < tg.nx && xf[1 + 2 * i] + sx > tile_hi - 2.0 * tg.xy_res)) o.ret_code = UPH_RET_LEFT_TILE;
        }
        if (o.x_final) std::memcpy(o.x_final, x + t.off_x, 8 * t.n);
        if (o.c_xy) std::memcpy(o.c_xy, cxy + t.off_cxy, 8 * 12 * t.Nxy);
        if (!c->frames.empty()) {        // back into map coordinates: the way-points and every piece's constant coefficient (row 6 i of c_xy)
            const TrajFrame& f = c->frames[b];
            if (o.x_final) for (int i = 0; i < 2 * (t.Nxy - 1); i++) o.x_final[1 + i] += f.shift[i & 1];
            if (o.c_xy) for (int i = 0; i < t.Nxy; i++) for (int d = 0; d < 2; d++) o.c_xy[12 * i + d] += f.shift[d];
        }
        if (o.c_yaw) std::memcpy(o.c_yaw, cyaw + t.off_cyaw, 8 * 6 * t.Nyaw);
        if (!o.lambda && !o.mu && !o.hx && !o.gx && !o.scale_cx) continue;
        const int S = t.S;
        const double* dl = want_dual ? dual + 7 * t.off_s : nullptr;
        const double* rs = want_res ? res + 7 * t.off_s : nullptr;
        const double* sc = want_scl ? scl + 7 * t.off_s : nullptr;
        for (int i = 0; i < S; i++) {
            if (o.lambda) o.lambda[i] = dl[i];
            if (o.hx) o.hx[i] = rs[i];
            for (int q = 0; q < 6; q++) {
                if (o.mu) o.mu[6 * i + q] = dl[(q + 1) * S + i];
                if (o.gx) o.gx[6 * i + q] = rs[(q + 1) * S + i];
            }
            if (o.scale_cx) for (int q = 0; q < 7; q++) o.scale_cx[7 * i + q] = sc[q * S + i];
        }
    }
    return UPH_OK;
}

int uph_optimize_batch(uph_ctx* c, int32_t B, const uph_problem* probs, uph_result* results) {
    int r = uph_batch_upload(c, B, probs);
    if (r != UPH_OK) return r;
    r = uph_batch_solve(c);
    if (r != UPH_OK) return r;
    return uph_batch_download(c, results);
}

// The split of a batch over n contexts: problems in descending predicted cost, dealt round-robin (every device gets the same mix of long and
// short solves); inside a share the caller's order (the upload sorts by cost itself).  Pure host arithmetic: uph_multi_batch_plan exposes it
// so that the decisions of an 8-GPU run can be checked without 8 GPUs.
static std::vector<std::vector<int>> dealShares(int n_gpus, int B, const uph_problem* probs) {
    std::vector<int> idx(B);
    std::iota(idx.begin(), idx.end(), 0);
    std::vector<double> cost(B);
    for (int b = 0; b < B; b++) {
        const uph_problem& q = probs[b];
        const bool readable = q.n_inner_xy >= 0 && q.n_inner_yaw >= 0 && (q.n_inner_yaw == 0 || q.inner_yaw);
        cost[b] = readable ? predictedCost(q) : 0.0;      // (an invalid problem is rejected by its context's upload; it needs no balancing)
    }
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b2) { return cost[a] > cost[b2]; });
    std::vector<std::vector<int>> share(n_gpus);
    for (int k = 0; k < B; k++) share[k % n_gpus].push_back(idx[k]);
    for (auto& sh : share) std::sort(sh.begin(), sh.end());
    return share;
}

int uph_multi_batch_plan(int32_t n_gpus, int32_t B, const uph_problem* probs, int32_t* share_of, double* predicted_cost) {
    if (n_gpus < 1 || B <= 0 || !probs || !share_of) { setError("uph_multi_batch_plan: bad arguments"); return UPH_ERR_INVALID; }
    const std::vector<std::vector<int>> share = dealShares(n_gpus, B, probs);
    for (int g = 0; g < n_gpus; g++) for (int b : share[g]) share_of[b] = g;
    if (predicted_cost) for (int b = 0; b < B; b++) {
        const uph_problem& q = probs[b];
        predicted_cost[b] = (q.n_inner_xy >= 0 && q.n_inner_yaw >= 0 && (q.n_inner_yaw == 0 || q.inner_yaw)) ? predictedCost(q) : 0.0;
    }
    return UPH_OK;
}

// ---- one batch over several GPUs of this process (independent trajectories, replicated grid, no collective) --------
// Problems are dealt to the contexts in descending predicted cost, round-robin, so every device gets the same mix of long and short
// solves; one host thread per device runs upload -> solve -> download on its share (the calls block, the devices run concurrently).
int uph_optimize_batch_multi(uph_ctx* const* ctxs, int32_t n_gpus, int32_t B, const uph_problem* probs, uph_result* results) {
    if (!ctxs || n_gpus < 1 || B <= 0 || !probs || !results) { setError("uph_optimize_batch_multi: bad arguments"); return UPH_ERR_INVALID; }
    for (int g = 0; g < n_gpus; g++) {
        if (!ctxs[g]) { setError("uph_optimize_batch_multi: null context"); return UPH_ERR_INVALID; }
        for (int h = 0; h < g; h++) if (ctxs[h] == ctxs[g]) { setError("uph_optimize_batch_multi: the same context twice"); return UPH_ERR_INVALID; }
        if (ctxs[g]->pending) { setError("uph_optimize_batch_multi: context " + std::to_string(g) + " has an asynchronous solve in flight (uph_batch_wait first)"); return UPH_ERR_INVALID; }
    }
    if (n_gpus == 1) return uph_optimize_batch(ctxs[0], B, probs, results);
    int dev_on_entry = -1;
    (void)hipGetDevice(&dev_on_entry);                 // the worker threads set their own device; the caller's current device is left as it was
    struct DevRestore { int d; ~DevRestore() { if (d >= 0) (void)hipSetDevice(d); } } dev_restore{dev_on_entry};
    const std::vector<std::vector<int>> share = dealShares(n_gpus, B, probs);
    std::vector<int> rc(n_gpus, UPH_OK);
    std::vector<std::string> err(n_gpus);
    std::vector<std::vector<uph_problem>> pg(n_gpus);
    std::vector<std::vector<uph_result>> rg(n_gpus);
    std::vector<std::thread> th;
    for (int g = 0; g < n_gpus; g++) {
        if (share[g].empty()) { ctxs[g]->B = 0; ctxs[g]->origin.clear(); continue; }           // (fewer problems than contexts: this one holds no part of THIS batch, and says so)
        for (int b : share[g]) { pg[g].push_back(probs[b]); rg[g].push_back(results[b]); }      // shallow: the arrays stay the caller's
        auto work = [&, g]() {
            rc[g] = uph_optimize_batch(ctxs[g], (int32_t)pg[g].size(), pg[g].data(), rg[g].data());
            if (rc[g] != UPH_OK) err[g] = g_last_error;      // (thread-local in the worker)
        };
        try { th.emplace_back(work); }
        catch (...) { work(); }           // no thread to be had: this share runs here (nothing throws across the ABI)
    }
    for (auto& t : th) t.join();
    // A share whose problems are ALL unsupported (its upload said so: all_rejected) fails like a batch of its own would, and only that is
    // downgraded to per-problem UPH_RET_UNSUPPORTED; the other shares' results stand.  Any other failure of a share -- a misuse, a resource
    // limit, a HIP error -- is the call's error, reported after every successful share's results have been handed back.
    int solved_shares = 0, first_bad = -1, hard = -1;
    for (int g = 0; g < n_gpus; g++) {
        if (share[g].empty()) continue;
        if (rc[g] == UPH_OK) {
            solved_shares++;
            for (size_t k = 0; k < share[g].size(); k++) results[share[g][k]] = rg[g][k];
            ctxs[g]->origin = share[g];
        } else if (ctxs[g]->all_rejected) {
            if (first_bad < 0) first_bad = g;
            for (size_t k = 0; k < share[g].size(); k++) {
                uph_result& o = results[share[g][k]];
                o.ret_code = UPH_RET_UNSUPPORTED; o.alm_iters = o.lbfgs_iters = o.evals = 0; o.last_lbfgs_ret = rc[g];
                o.cost = o.jerk_cost = o.piece_T_xy = o.piece_T_yaw = 0.0; o.scale_fx = 1.0; o.rho_final = ctxs[g]->rho;
            }
        } else if (hard < 0) hard = g;
    }
    if (hard >= 0) { setError("uph_optimize_batch_multi: device share " + std::to_string(hard) + ": " + err[hard]); return rc[hard]; }
    if (!solved_shares) { setError("uph_optimize_batch_multi: " + err[first_bad]); return rc[first_bad]; }
    return UPH_OK;
}

int uph_batch_set_state(uph_ctx* c, const double* lambda, const double* mu, const double* scale_cx, const double* scale_fx, const double* rho) {
    if (c && c->pending) { setError("an asynchronous solve is in flight on this context: call uph_batch_wait first"); return UPH_ERR_INVALID; }
    if (c && c->n_rejected) { setError("packed-array hooks need a batch without unsupported problems (uph_result.ret_code == UPH_RET_UNSUPPORTED)"); return UPH_ERR_INVALID; }
    if (!c || c->B <= 0) { setError("uph_batch_set_state: no batch uploaded"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(uphMapDevice(c->map)));
    std::vector<double> dual(7 * c->sum_S), scl(7 * c->sum_S);
    HIPCHK(hipMemcpy(dual.data(), c->d_dual.p, 8 * 7 * c->sum_S, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(scl.data(), c->d_scl.p, 8 * 7 * c->sum_S, hipMemcpyDeviceToHost));
    int r = refreshStates(c);
    if (r != UPH_OK) return r;
    int64_t os = 0;
    for (int b = 0; b < c->B; b++) {
        const TrajDesc& t = c->desc[b];
        const int S = t.S;
        double* dl = dual.data() + 7 * t.off_s;
        double* sc = scl.data() + 7 * t.off_s;
        for (int i = 0; i < S; i++) {
            if (lambda) dl[i] = lambda[os + i];
            if (mu) for (int q = 0; q < 6; q++) dl[(q + 1) * S + i] = mu[6 * (os + i) + q];
            if (scale_cx) for (int q = 0; q < 7; q++) sc[q * S + i] = scale_cx[7 * (os + i) + q];
        }
        if (scale_fx) c->state_host[b].scale_fx = scale_fx[b];
        if (rho) c->state_host[b].rho = rho[b];
        os += S;
    }
    HIPCHK(hipMemcpy(c->d_dual.p, dual.data(), 8 * 7 * c->sum_S, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_scl.p, scl.data(), 8 * 7 * c->sum_S, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_state.p, c->state_host.data(), sizeof(TrajState) * c->B, hipMemcpyHostToDevice));
    return UPH_OK;
}

int uph_eval_batch(uph_ctx* c, const double* x_packed, double* f, double* grad_packed, int32_t repeat) {
    if (c && c->n_rejected) { setError("packed-array hooks need a batch without unsupported problems (uph_result.ret_code == UPH_RET_UNSUPPORTED)"); return UPH_ERR_INVALID; }
    if (!c || c->B <= 0 || repeat < 1) { setError("uph_eval_batch: bad arguments"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(uphMapDevice(c->map)));
    if (x_packed) { const int rx = uploadPackedX(c, x_packed); if (rx != UPH_OK) return rx; }
    int r = launchSolver(c, 0, repeat);
    if (r != UPH_OK) return r;
    r = refreshStates(c);
    if (r != UPH_OK) return r;
    if (f) for (int b = 0; b < c->B; b++) f[b] = c->state_host[b].f;
    if (grad_packed) HIPCHK(hipMemcpy(grad_packed, c->d_gout.p, 8 * c->sum_n, hipMemcpyDeviceToHost));
    c->last_evals = (int64_t)c->B * repeat;
    c->last_sample_evals = c->sum_S * repeat;
    c->last_iters = 0; c->last_hist_bytes = 0;
    return UPH_OK;
}

// A5 alone -- calConstrainCostGrad (alm_traj_opt.cpp:663-991): the coefficients, piece durations, duals, scales, rho and scale_fx RESIDENT on the device
// (i.e. those of the last uph_eval_batch / solve of this batch) in -> cost [B], gdCxy [sum 12 Nxy] (row 6 i + k, column dim: the reference's 6N x 2
// block row-major), gdCyaw [sum 6 Nyaw], gdT [B][2] = (sum_i gdTxy(i), sum_i gdTyaw(i)) out; store_residuals != 0: hx / gx are written by every call,
// as the reference's function writes them (read them with uph_batch_download).  `repeat` calls inside one launch (measurement); any output may be NULL.
int uph_penalty_batch(uph_ctx* c, int32_t repeat, int32_t store_residuals, double* cost, double* gdcxy_packed, double* gdcyaw_packed, double* gdT2) {
    if (c && c->n_rejected) { setError("packed-array hooks need a batch without unsupported problems (uph_result.ret_code == UPH_RET_UNSUPPORTED)"); return UPH_ERR_INVALID; }
    if (!c || c->B <= 0 || repeat < 1 || repeat > (1 << 20)) { setError("uph_penalty_batch: bad arguments"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(uphMapDevice(c->map)));
    if (c->d_pen_gxy.ensure(8 * c->sum_cxy) || c->d_pen_gyaw.ensure(8 * c->sum_cyaw) || c->d_pen_out.ensure(8 * 3 * (size_t)c->B)) return UPH_ERR_HIP;
    int r = launchSolver(c, 8, 2 * repeat + (store_residuals ? 1 : 0));
    if (r != UPH_OK) return r;
    if (cost || gdT2) {
        std::vector<double> o3((size_t)3 * c->B);
        HIPCHK(hipMemcpy(o3.data(), c->d_pen_out.p, 8 * o3.size(), hipMemcpyDeviceToHost));
        for (int b = 0; b < c->B; b++) {
            if (cost) cost[b] = o3[3 * (size_t)b];
            if (gdT2) { gdT2[2 * (size_t)b] = o3[3 * (size_t)b + 1]; gdT2[2 * (size_t)b + 1] = o3[3 * (size_t)b + 2]; }
        }
    }
    if (gdcxy_packed) HIPCHK(hipMemcpy(gdcxy_packed, c->d_pen_gxy.p, 8 * c->sum_cxy, hipMemcpyDeviceToHost));
    if (gdcyaw_packed) HIPCHK(hipMemcpy(gdcyaw_packed, c->d_pen_gyaw.p, 8 * c->sum_cyaw, hipMemcpyDeviceToHost));
    c->last_evals = (int64_t)c->B * repeat;
    c->last_sample_evals = c->sum_S * repeat;
    c->last_iters = 0; c->last_hist_bytes = 0;
    return UPH_OK;
}

// diagnostic: cost of the workgroup primitives (see Solver::microbench); read the result with uph_batch_cycles
int uph_microbench_batch(uph_ctx* c, int32_t reps) {
    if (!c || c->B <= 0 || reps < 1) return UPH_ERR_INVALID;
    int r = launchSolver(c, 5, reps);
    if (r != UPH_OK) return r;
    return refreshStates(c);
}

int uph_init_scaling_batch(uph_ctx* c) {
    if (!c || c->B <= 0) { setError("uph_init_scaling_batch: no batch uploaded"); return UPH_ERR_INVALID; }
    int r = launchSolver(c, 4, 1);
    if (r != UPH_OK) return r;
    return refreshStates(c);
}

int uph_report_batch(uph_ctx* c, double* out7) {
    if (!c || c->B <= 0 || !out7) { setError("uph_report_batch: bad arguments"); return UPH_ERR_INVALID; }
    int r = launchSolver(c, 3, 1);
    if (r != UPH_OK) return r;
    HIPCHK(hipMemcpy(out7, c->d_report.p, 8 * 7 * c->B, hipMemcpyDeviceToHost));
    return UPH_OK;
}

// ---- test hooks of the teacher-forced late-state tests -------------------------------------------------------------------------
static void collectSolveStats(uph_ctx* c) {
    c->last_evals = c->last_sample_evals = c->last_iters = c->last_hist_bytes = 0;
    for (int b = 0; b < c->B; b++) {
        if (c->rejected[b]) continue;
        const TrajState& s = c->state_host[b];
        c->last_evals += s.evals;
        c->last_sample_evals += (int64_t)s.evals * c->desc[b].S;
        c->last_iters += s.lbfgs_iters;
        c->last_hist_bytes += s.hist_reads * 8;
    }
}
int uph_batch_set_x(uph_ctx* c, const double* x_packed) {
    if (c && c->pending) { setError("an asynchronous solve is in flight on this context: call uph_batch_wait first"); return UPH_ERR_INVALID; }
    if (c && c->n_rejected) { setError("packed-array hooks need a batch without unsupported problems (uph_result.ret_code == UPH_RET_UNSUPPORTED)"); return UPH_ERR_INVALID; }
    if (!c || c->B <= 0 || !x_packed) { setError("uph_batch_set_x: bad arguments"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(c->device));
    return uploadPackedX(c, x_packed);
}
// the ALM loop of optimizeSE2Traj (alm_traj_opt.cpp:234-271) from the RESIDENT x, duals, scales and rho -- no reset, no initScaling --
// for at most max_passes passes (0 = until it ends); a solve stopped by the cap reports ret_code 3
int uph_batch_alm_passes(uph_ctx* c, int32_t max_passes) {
    if (!c || c->B <= 0 || max_passes < 0) { setError("uph_batch_alm_passes: bad arguments"); return UPH_ERR_INVALID; }
    int r = launchSolver(c, 2, max_passes);
    if (r != UPH_OK) return r;
    r = refreshStates(c);
    if (r != UPH_OK) return r;
    collectSolveStats(c);
    return UPH_OK;
}
// L-BFGS state at the top of the iteration loop (lbfgs.hpp:555), per trajectory: g, d packed like x; pf [B][UPH_MAX_PAST]; the history
// as the reference holds it -- lm_s / lm_y column j of trajectory b at hist[off_b + j*n] with off_b = mem * sum_{b' < b} n_b', lm_ys
// [B][mem] --; scal5 [B][5] = step, fx, k, end, bound.  x is the resident x (uph_batch_set_x).
int uph_batch_set_lbfgs_state(uph_ctx* c, const double* g, const double* d, const double* pf, const double* lm_s, const double* lm_y, const double* lm_ys, const double* scal5) {
    if (c && c->pending) { setError("an asynchronous solve is in flight on this context: call uph_batch_wait first"); return UPH_ERR_INVALID; }
    if (c && c->n_rejected) { setError("packed-array hooks need a batch without unsupported problems (uph_result.ret_code == UPH_RET_UNSUPPORTED)"); return UPH_ERR_INVALID; }
    if (!c || c->B <= 0 || !g || !d || !pf || !lm_s || !lm_y || !lm_ys || !scal5) { setError("uph_batch_set_lbfgs_state: bad arguments"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(c->device));
    const int mem = c->P.mem_size;
    if (c->d_rsd.ensure(8 * c->sum_n) || c->d_rs.ensure(8 * 24 * (size_t)c->B)) return UPH_ERR_HIP;
    std::vector<double> hist((size_t)c->sum_hist, 0.0), rs((size_t)24 * c->B, 0.0);
    int64_t off = 0;
    for (int b = 0; b < c->B; b++) {
        const TrajDesc& t = c->desc[b];
        const int rowd = histRowDoubles(t.n), np = 64 * histNQ(t.n);
        for (int j = 0; j < mem; j++) {
            double* row = hist.data() + t.off_hist + (size_t)j * rowd;
            const double ys = lm_ys[(size_t)b * mem + j];
            row[0] = ys; row[1] = 1.0 / ys;
            std::memcpy(row + 2, lm_s + off + (size_t)j * t.n, 8 * t.n);
            std::memcpy(row + 2 + np, lm_y + off + (size_t)j * t.n, 8 * t.n);
        }
        off += (int64_t)mem * t.n;
        for (int q = 0; q < 5; q++) rs[(size_t)24 * b + q] = scal5[5 * b + q];
        for (int q = 0; q < UPH_MAX_PAST; q++) rs[(size_t)24 * b + 8 + q] = pf[(size_t)b * UPH_MAX_PAST + q];
    }
    HIPCHK(hipMemcpy(c->d_hist.p, hist.data(), 8 * c->sum_hist, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_rs.p, rs.data(), 8 * rs.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_gout.p, g, 8 * c->sum_n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_rsd.p, d, 8 * c->sum_n, hipMemcpyHostToDevice));
    return UPH_OK;
}
// continue every trajectory's L-BFGS loop from the state set above for at most `budget` iterations (budget < 0: until it ends);
// finish_pass != 0: when the loop ends by itself, the ALM's reaction follows (accepted code -> updateDualVars + judgeConvergence)
int uph_batch_lbfgs_resume(uph_ctx* c, int32_t budget, int32_t finish_pass) {
    if (!c || c->B <= 0 || !c->d_rs.p) { setError("uph_batch_lbfgs_resume: no state set"); return UPH_ERR_INVALID; }
    const int bud = budget < 0 ? (1 << 28) : budget;
    int r = launchSolver(c, 7, 2 * bud + (finish_pass ? 1 : 0));
    if (r != UPH_OK) return r;
    r = refreshStates(c);
    if (r != UPH_OK) return r;
    collectSolveStats(c);
    return UPH_OK;
}
// the state after uph_batch_lbfgs_resume, same layout; scal8 [B][8] = step, fx, k, end, bound, L-BFGS code (999 = budget ran out),
// accepted, converged.  x / hx / gx / duals / rho through uph_batch_download.
int uph_batch_get_lbfgs_state(uph_ctx* c, double* g, double* d, double* pf, double* lm_s, double* lm_y, double* lm_ys, double* scal8) {
    if (c && c->pending) { setError("an asynchronous solve is in flight on this context: call uph_batch_wait first"); return UPH_ERR_INVALID; }
    if (!c || c->B <= 0 || !c->d_rs.p) { setError("uph_batch_get_lbfgs_state: no state set"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(c->device));
    const int mem = c->P.mem_size;
    std::vector<double> hist((size_t)c->sum_hist), rs((size_t)24 * c->B);
    HIPCHK(hipMemcpy(hist.data(), c->d_hist.p, 8 * c->sum_hist, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rs.data(), c->d_rs.p, 8 * rs.size(), hipMemcpyDeviceToHost));
    if (g) HIPCHK(hipMemcpy(g, c->d_gout.p, 8 * c->sum_n, hipMemcpyDeviceToHost));
    if (d) HIPCHK(hipMemcpy(d, c->d_rsd.p, 8 * c->sum_n, hipMemcpyDeviceToHost));
    int64_t off = 0;
    for (int b = 0; b < c->B; b++) {
        const TrajDesc& t = c->desc[b];
        const int rowd = histRowDoubles(t.n), np = 64 * histNQ(t.n);
        for (int j = 0; j < mem; j++) {
            const double* row = hist.data() + t.off_hist + (size_t)j * rowd;
            if (lm_ys) lm_ys[(size_t)b * mem + j] = row[0];
            if (lm_s) std::memcpy(lm_s + off + (size_t)j * t.n, row + 2, 8 * t.n);
            if (lm_y) std::memcpy(lm_y + off + (size_t)j * t.n, row + 2 + np, 8 * t.n);
        }
        off += (int64_t)mem * t.n;
        if (scal8) for (int q = 0; q < 8; q++) scal8[8 * b + q] = rs[(size_t)24 * b + q];
        if (pf) for (int q = 0; q < UPH_MAX_PAST; q++) pf[(size_t)b * UPH_MAX_PAST + q] = rs[(size_t)24 * b + 8 + q];
    }
    return UPH_OK;
}

int uph_terrain_query(uph_map* m, const double* pos, int32_t n, double* values7, double* grads21) {
    if (!m || !pos || n <= 0 || !values7 || !grads21) { setError("uph_terrain_query: bad arguments"); return UPH_ERR_INVALID; }
    HIPCHK(hipSetDevice(uphMapDevice(m)));
    UphPtr tp, tv, tg;
    tp.p = uphMapScratch(m, 0, 8 * 3 * (size_t)n); tv.p = uphMapScratch(m, 1, 8 * 7 * (size_t)n); tg.p = uphMapScratch(m, 2, 8 * 21 * (size_t)n);
    if (!tp.p || !tv.p || !tg.p) return UPH_ERR_HIP;
    HIPCHK(hipMemcpy(tp.p, pos, 8 * 3 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(uph_terrain_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, uphMapGrid(m), tp.as<double>(), n, tv.as<double>(), tg.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(values7, tv.p, 8 * 7 * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(grads21, tg.p, 8 * 21 * (size_t)n, hipMemcpyDeviceToHost));
    return UPH_OK;
}

}  // extern "C"
#endif      // UPH_ONE_KERNEL
