// fp64 matrix-core instructions of gfx950, as far as the xy scatter (solver_program.hpp scatterChunk) needs to know them:
//   1. operand layouts of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64, found by one-hot experiments (which A / B lanes reach which
//      (lane, register) of D) and printed as formulas that the host verifies over all 64 lanes;
//   2. issue cost: cycles per instruction in a dependent chain (one accumulator) and with two / four independent accumulators;
//   3. co-execution with fp64 vector arithmetic: a 512-lane workgroup (two waves per SIMD) whose waves 0-3 run MFMA chains while waves 4-7 run
//      fp64 FMA chains, against each half running alone -- does the matrix pipe take issue slots / DP units from the vector pipe?
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64_probe.hip -o build/micro/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void onehot16(double* out /* [2][64][64][4] */) {
    const int l = threadIdx.x;
    for (int which = 0; which < 2; which++)
        for (int src = 0; src < 64; src++) {
            const double a = which == 0 ? (l == src ? 1.0 : 0.0) : 1.0;
            const double b = which == 1 ? (l == src ? 1.0 : 0.0) : 1.0;
            d4 acc = {0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            for (int r = 0; r < 4; r++) out[((which * 64 + src) * 64 + l) * 4 + r] = acc[r];
        }
}
__global__ void onehot4(double* out /* [2][64][64] */) {
    const int l = threadIdx.x;
    for (int which = 0; which < 2; which++)
        for (int src = 0; src < 64; src++) {
            const double a = which == 0 ? (l == src ? 1.0 : 0.0) : 1.0;
            const double b = which == 1 ? (l == src ? 1.0 : 0.0) : 1.0;
            out[(which * 64 + src) * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        }
}

// mode 0: 16x16x4, NACC accumulators; mode 1: 4x4x4_4b, NACC accumulators; mode 2: fp64 FMA chains (8 accumulators)
template <int MODE, int NACC>
__device__ __forceinline__ double work(int iters, double seed) {
    const int l = threadIdx.x & 63;
    if (MODE == 0) {
        d4 acc[NACC];
        for (int q = 0; q < NACC; q++) acc[q] = d4{0, 0, 0, 0};
        const double a = seed + l, b = seed * 0.5 + l;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
        }
        double s = 0.0;
        for (int q = 0; q < NACC; q++) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
        return s;
    } else if (MODE == 1) {
        double acc[NACC];
        for (int q = 0; q < NACC; q++) acc[q] = 0.0;
        const double a = seed + l, b = seed * 0.5 + l;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
        }
        double s = 0.0;
        for (int q = 0; q < NACC; q++) s += acc[q];
        return s;
    } else {
        double a[8];
        for (int q = 0; q < 8; q++) a[q] = seed + q + l;
        const double x = 0.999, y = 0.5;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 8; q++) a[q] = fma(a[q], x, y);
        }
        double s = 0.0;
        for (int q = 0; q < 8; q++) s += a[q];
        return s;
    }
}
template <int MODE, int NACC>
__global__ void timeOne(long long* cyc, double* sink, int iters) {
    const long long t0 = __builtin_readcyclecounter();
    const double s = work<MODE, NACC>(iters, 1.0);
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = s;
}
// 512 lanes = 8 waves = two per SIMD.  sel bit 0: waves 0-3 run MFMA (MODE_M), bit 1: waves 4-7 run the FMA chains
template <int MODE_M, int NACC>
__global__ void coexec(long long* cyc, double* sink, int iters_m, int iters_v, int sel) {
    const int w = threadIdx.x >> 6;
    const long long t0 = __builtin_readcyclecounter();
    double s = 0.0;
    if (w < 4) { if (sel & 1) s = work<MODE_M, NACC>(iters_m, 1.0); }
    else { if (sel & 2) s = work<2, 1>(iters_v, 1.0); }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
    sink[threadIdx.x] = s;
}

int main() {
    double* d; hipMalloc((void**)&d, sizeof(double) * 2 * 64 * 64 * 4);
    std::vector<double> h(2 * 64 * 64 * 4);
    // ---- layout of 16x16x4: D(lane, reg) depends on which A lanes / B lanes
    hipLaunchKernelGGL(onehot16, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    {
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
            std::set<int> As, Bs;
            for (int s = 0; s < 64; s++) { if (h[((0 * 64 + s) * 64 + l) * 4 + r] != 0.0) As.insert(s); if (h[((1 * 64 + s) * 64 + l) * 4 + r] != 0.0) Bs.insert(s); }
            // hypothesis (cdna_hip_programming.md): A lane = k*16 + i, B lane = k*16 + j, D: col j = l & 15, row i = (l >> 4) + 4 r
            const int i = (l >> 4) + 4 * r, j = l & 15;
            std::set<int> Ae, Be;
            for (int k = 0; k < 4; k++) { Ae.insert(k * 16 + i); Be.insert(k * 16 + j); }
            if (As != Ae || Bs != Be) { if (bad < 4) { std::printf("16x16x4 mismatch at lane %d reg %d: A {", l, r); for (int s : As) std::printf("%d ", s); std::printf("} B {"); for (int s : Bs) std::printf("%d ", s); std::printf("}\n"); } bad++; }
        }
        std::printf("v_mfma_f64_16x16x4_f64: A[i][k] in lane 16k+i, B[k][j] in lane 16k+j, D[i][j] in lane 16(i&3)+j register i>>2 : %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    }
    // ---- layout of 4x4x4_4b
    hipLaunchKernelGGL(onehot4, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, sizeof(double) * 2 * 64 * 64, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
        std::printf("4x4x4_4b D lane %2d <- A lanes {", l);
        for (int s = 0; s < 64; s++) if (h[(0 * 64 + s) * 64 + l] != 0.0) std::printf("%d ", s);
        std::printf("} B lanes {");
        for (int s = 0; s < 64; s++) if (h[(1 * 64 + s) * 64 + l] != 0.0) std::printf("%d ", s);
        std::printf("}\n");
    }
    // ---- issue cost
    long long* dc; hipMalloc((void**)&dc, 64 * sizeof(long long));
    double* ds; hipMalloc((void**)&ds, 512 * sizeof(double));
    long long hc[8];
    const int iters = 4000;
#define TIME(MODE, NACC, name)                                                                                      \
    do {                                                                                                            \
        hipLaunchKernelGGL((timeOne<MODE, NACC>), dim3(1), dim3(64), 0, 0, dc, ds, 100);                            \
        hipLaunchKernelGGL((timeOne<MODE, NACC>), dim3(1), dim3(64), 0, 0, dc, ds, iters);                          \
        hipDeviceSynchronize();                                                                                     \
        hipMemcpy(hc, dc, sizeof(long long), hipMemcpyDeviceToHost);                                                \
        std::printf("%-34s %6.1f cycles per instruction (one wave alone)\n", name, (double)hc[0] / ((double)iters * (MODE == 2 ? 8 : NACC))); \
    } while (0)
    TIME(0, 1, "16x16x4   1 accumulator (chain)");
    TIME(0, 2, "16x16x4   2 accumulators");
    TIME(0, 4, "16x16x4   4 accumulators");
    TIME(1, 1, "4x4x4_4b  1 accumulator (chain)");
    TIME(1, 2, "4x4x4_4b  2 accumulators");
    TIME(1, 4, "4x4x4_4b  4 accumulators");
    TIME(2, 1, "v_fma_f64 8 chains");
    // ---- co-execution on shared SIMDs
    for (int mm = 0; mm < 2; mm++) {
        const int im = mm == 0 ? 2000 : 8000, iv = 4000;       // similar durations
        for (int sel = 1; sel <= 3; sel++) {
            if (mm == 0) { hipLaunchKernelGGL((coexec<0, 2>), dim3(1), dim3(512), 0, 0, dc, ds, 100, 100, sel); hipLaunchKernelGGL((coexec<0, 2>), dim3(1), dim3(512), 0, 0, dc, ds, im, iv, sel); }
            else { hipLaunchKernelGGL((coexec<1, 2>), dim3(1), dim3(512), 0, 0, dc, ds, 100, 100, sel); hipLaunchKernelGGL((coexec<1, 2>), dim3(1), dim3(512), 0, 0, dc, ds, im, iv, sel); }
            hipDeviceSynchronize();
            hipMemcpy(hc, dc, 8 * sizeof(long long), hipMemcpyDeviceToHost);
            std::printf("coexec %s  sel %d (1 = MFMA waves only, 2 = FMA waves only, 3 = both):  MFMA wave0 %.1f cyc/instr   FMA wave4 %.2f cyc/instr\n",
                        mm == 0 ? "16x16x4 " : "4x4x4_4b", sel, (double)hc[0] / (im * 2.0), (double)hc[4] / (iv * 8.0));
        }
    }
    return 0;
}
