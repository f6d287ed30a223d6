// Column-oriented triangular substitution carried by a scalar broadcast instead of a wave reduction -- the serial part of the compact
// (Gram-matrix) form of the L-BFGS direction.  One wave; the right-hand side lives one entry per lane (and register
// q for entries 64 q + lane); step i reads entry i with v_readlane, divides by the stored curvature (the three-FMA quotient of the
// product's two-loop), and subtracts alpha_i times row i of the product matrix, which is streamed from memory a few rows ahead.
// Prints cycles per step for bound = 48 / 128 / 256.   hipcc --offload-arch=gfx950 -O3 subst_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double readLaneD(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int QB, int PF, bool NOLOAD>
__global__ void k(const double* __restrict__ M, const double* __restrict__ hdr, const double* __restrict__ rhs, double* out, long long* cyc, int bound, int m, int reps) {
    const int lane = threadIdx.x;
    double b[QB], row[PF][QB], hy[PF], hr[PF];
    double sink = 0.0;
    int vz = 0;
    asm volatile("" : "+v"(vz));                 // an opaque per-lane zero: keeps the header loads on the vector path (vmcnt-ordered; scalar loads
                                                 // return out of order, so every use would wait for lgkmcnt(0) -- a full round trip per step)
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int q = 0; q < QB; q++) b[q] = rhs[64 * q + lane] + sink * 1e-300;
        auto fetch = [&](int u, int i) {
#pragma unroll
            for (int q = 0; q < QB; q++) row[u][q] = M[(size_t)i * m + 64 * q + lane];
            hy[u] = hdr[2 * i + vz]; hr[u] = hdr[2 * i + 1 + vz];
        };
#pragma unroll
        for (int u = 0; u < PF; u++) fetch(u, u);
        int i = 0;
        for (; i + PF <= bound; i += PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int ii = __builtin_amdgcn_readfirstlane(i + u);
                double bi;
                if (QB == 1) bi = readLaneD(b[0], ii & 63);
                else { const int qq = ii >> 6; bi = qq == 0 ? readLaneD(b[0], ii & 63) : (qq == 1 ? readLaneD(b[1 % QB], ii & 63) : (qq == 2 ? readLaneD(b[2 % QB], ii & 63) : readLaneD(b[3 % QB], ii & 63))); }
                const double q0 = bi * hr[u];
                const double al = fma(fma(-q0, hy[u], bi), hr[u], q0);
#pragma unroll
                for (int q = 0; q < QB; q++) b[q] = fma(-al, row[u][q], b[q]);
                sink += al * 1e-300;
#pragma unroll
                for (int q = 0; q < QB; q++) asm volatile("" : "+v"(b[q]) : : "memory");
                { int nx = i + u + PF; nx = nx >= m ? nx - m : nx; fetch(u, nx); }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = sink;
#pragma unroll
    for (int q = 0; q < QB; q++) s += b[q];
    out[lane] = s;
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    const int m = 256;
    std::vector<double> M((size_t)m * m), hdr(2 * m), rhs(m);
    srand(3);
    for (auto& x : M) x = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
    for (int i = 0; i < m; i++) { M[(size_t)i * m + i] = 1.0; hdr[2 * i] = 2.0 + i * 1e-3; hdr[2 * i + 1] = 1.0 / hdr[2 * i]; rhs[i] = 1.0 + 0.01 * i; }
    double *dM, *dh, *dr, *dout; long long* dc;
    (void)hipMalloc(&dM, 8 * M.size()); (void)hipMalloc(&dh, 8 * hdr.size()); (void)hipMalloc(&dr, 8 * rhs.size()); (void)hipMalloc(&dout, 512); (void)hipMalloc(&dc, 8);
    (void)hipMemcpy(dM, M.data(), 8 * M.size(), hipMemcpyHostToDevice); (void)hipMemcpy(dh, hdr.data(), 8 * hdr.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dr, rhs.data(), 8 * rhs.size(), hipMemcpyHostToDevice);
    const int reps = 200;
    auto run = [&](int qb, int bound) {
        if (qb == 1) k<1, 4, false><<<1, 64>>>(dM, dh, dr, dout, dc, bound, m, reps);
        else if (qb == 2) k<2, 4, false><<<1, 64>>>(dM, dh, dr, dout, dc, bound, m, reps);
        else if (qb == 4) k<4, 4, false><<<1, 64>>>(dM, dh, dr, dout, dc, bound, m, reps);
        else if (qb == 11) k<1, 4, true><<<1, 64>>>(dM, dh, dr, dout, dc, bound, m, reps);
        else k<1, 8, false><<<1, 64>>>(dM, dh, dr, dout, dc, bound, m, reps);
        long long c = 0;
        (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("QB %d bound %3d: %.1f cycles per substitution step\n", qb, bound, (double)c / reps / bound);
    };
    run(1, 48); run(1, 48); run(1, 64); run(2, 128); run(4, 256);
    printf("rows kept in registers (pure dependency chain):\n"); run(11, 48);
    printf("ring of eight rows:\n"); run(18, 48);
    return 0;
}
