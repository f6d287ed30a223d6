// Cycles per wave-wide fp64 sum for several instruction sequences, as ONE wave alone on a CU sees them in a dependent chain (the
// two-loop's situation).  hipcc --offload-arch=gfx950 -O3 reduce_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CTRL, bool BC = false>
__device__ __forceinline__ double dppMov(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, BC);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, BC);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readLane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double swz(double v, int pat) {   // ds_swizzle (LDS crossbar, no memory)
    return v;
}
template <int V> __device__ __forceinline__ double red(double v);
// 0: row_ror 8/4/2/1 + 8 readlanes + 3 adds (the product's waveSum)
template <> __device__ __forceinline__ double red<0>(double v) {
    v += dppMov<0x128>(v); v += dppMov<0x124>(v); v += dppMov<0x122>(v); v += dppMov<0x121>(v);
    return ((readLane(v, 0) + readLane(v, 16)) + readLane(v, 32)) + readLane(v, 48);
}
// 1: row_ror + row_bcast:15 + row_bcast:31 + readlane 63
template <> __device__ __forceinline__ double red<1>(double v) {
    v += dppMov<0x128>(v); v += dppMov<0x124>(v); v += dppMov<0x122>(v); v += dppMov<0x121>(v);
    v += dppMov<0x142, true>(v); v += dppMov<0x143, true>(v);
    return readLane(v, 63);
}
// 2: quad_perm xor1, quad_perm xor2, row_half_mirror, row_mirror + readlanes
template <> __device__ __forceinline__ double red<2>(double v) {
    v += dppMov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dppMov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dppMov<0x141>(v);   // row_half_mirror
    v += dppMov<0x140>(v);   // row_mirror
    return ((readLane(v, 0) + readLane(v, 16)) + readLane(v, 32)) + readLane(v, 48);
}
// 3: only the four row_ror stages (no cross-row part): cost of the DPP stages alone
template <> __device__ __forceinline__ double red<3>(double v) {
    v += dppMov<0x128>(v); v += dppMov<0x124>(v); v += dppMov<0x122>(v); v += dppMov<0x121>(v);
    return v;
}
// 4: only the readlane combine
template <> __device__ __forceinline__ double red<4>(double v) {
    return ((readLane(v, 0) + readLane(v, 16)) + readLane(v, 32)) + readLane(v, 48);
}
// 5: four dependent fp64 adds (latency calibration)
template <> __device__ __forceinline__ double red<5>(double v) {
    v += 1.0; v *= 1.0000001; v += 1.0; v *= 0.9999999;
    return v;
}
// 6: ds_bpermute butterfly (LDS crossbar) for all six stages
template <> __device__ __forceinline__ double red<6>(double v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int src = (lane ^ off) << 2;
        const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
        v += __hiloint2double(hi, lo);
    }
    return v;
}
// 7: permlane32_swap + permlane16_swap for the two upper stages, row_ror below
template <> __device__ __forceinline__ double red<7>(double v) {
    v += dppMov<0x128>(v); v += dppMov<0x124>(v); v += dppMov<0x122>(v); v += dppMov<0x121>(v);
    {
        int lo = __double2loint(v), hi = __double2hiint(v);
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        // after the swap element [0] holds, in the odd rows' positions, the even rows' values and vice versa
        const double o = __hiloint2double((int)(threadIdx.x & 16 ? b[0] : b[1]), (int)(threadIdx.x & 16 ? a[0] : a[1]));
        v += o;
    }
    {
        int lo = __double2loint(v), hi = __double2hiint(v);
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        const double o = __hiloint2double((int)(threadIdx.x & 32 ? b[0] : b[1]), (int)(threadIdx.x & 32 ? a[0] : a[1]));
        v += o;
    }
    return v;
}
// 8: TWO different sums in one pass: permlane32_swap hands the upper half its partner's copy of the second value (the lower half keeps
//    the first), four row_ror stages and one row_bcast:15 finish both; sum 1 ends in lane 31, sum 2 in lane 63
__device__ __forceinline__ void red2(double& u, double& v) {
    const bool up = (threadIdx.x & 32) != 0;
    const double keep = up ? v : u, send = up ? u : v;
    const unsigned lo = __double2loint(send), hi = __double2hiint(send);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    // a[0] = [lower half of send, lower half of send], a[1] = [upper, upper]: the partner's value is a[1] for the lower lanes, a[0] for the upper ones
    const double got = __hiloint2double((int)(up ? b[0] : b[1]), (int)(up ? a[0] : a[1]));
    double w = keep + got;
    w += dppMov<0x128>(w); w += dppMov<0x124>(w); w += dppMov<0x122>(w); w += dppMov<0x121>(w);
    w += dppMov<0x142, true>(w);
    u = readLane(w, 31); v = readLane(w, 63);
}
__global__ void k2(double* out, long long* cyc, int iters) {
    double u = 1.0 + threadIdx.x * 1e-3, v = 2.0 - threadIdx.x * 1e-3;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        red2(u, v);
        const double nu = u * 1e-2 + threadIdx.x * 1e-3, nv = v * 1e-2 - threadIdx.x * 1e-3;
        u = nu; v = nv;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = u + 3.0 * v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V>
__global__ void k(double* out, long long* cyc, int iters) {
    double v = 1.0 + threadIdx.x * 1e-3;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        v = red<V>(v) * 1e-2 + threadIdx.x * 1e-3;          // feed the result back: a dependent chain, like the two-loop's
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V> void run(const char* name, double* dout, long long* dcyc) {
    const int iters = 20000;
    k<V><<<1, 64>>>(dout, dcyc, iters);
    k<V><<<1, 64>>>(dout, dcyc, iters);
    long long c = 0; double o[64];
    (void)hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
    printf("%-70s %7.1f cycles per iteration   (v = %.6g)\n", name, (double)c / iters, o[0]);
}
int main() {
    double* dout; long long* dcyc;
    (void)hipMalloc(&dout, 512); (void)hipMalloc(&dcyc, 8);
    run<5>("4 dependent fp64 ops + the feedback fma", dout, dcyc);
    run<3>("4 row_ror stages only", dout, dcyc);
    run<4>("8 readlanes + 3 adds only", dout, dcyc);
    run<0>("row_ror x4 + 8 readlanes + 3 adds (product waveSum)", dout, dcyc);
    run<1>("row_ror x4 + row_bcast15 + row_bcast31 + readlane 63", dout, dcyc);
    run<2>("quad_perm x2 + row_half_mirror + row_mirror + 8 readlanes + 3 adds", dout, dcyc);
    run<6>("ds_bpermute butterfly x6", dout, dcyc);
    run<7>("row_ror x4 + permlane16_swap + permlane32_swap", dout, dcyc);
    {
        const int iters = 20000;
        k2<<<1, 64>>>(dout, dcyc, iters);
        k2<<<1, 64>>>(dout, dcyc, iters);
        long long c = 0; double o[64];
        (void)hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
        // reference for the first iteration is checked by value: u -> sum(1 + l/1000) = 66.016, v -> sum(2 - l/1000) = 125.984
        printf("%-70s %7.1f cycles per iteration   (out = %.6g)\n", "TWO sums: permlane32_swap split + row_ror x4 + row_bcast15", (double)c / iters, o[0]);
    }
    return 0;
}
