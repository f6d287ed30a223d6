// FETCH_SIZE / WRITE_SIZE calibration for the access patterns of the solve kernel (MI355X_MICROARCH.md, HBM section: "calibrate on a
// known byte count in your own access pattern").  Four kernels with exactly known byte counts, each run under rocprofv3 --pmc:
//   gather8    : every lane reads ONE 8-byte word at a random 8-byte-aligned address of a 4 GiB buffer (the terrain / dual loads)
//   rows16     : a wave reads 1 KiB rows (16 bytes per lane, contiguous) at random row addresses (the L-BFGS history rows)
//   stream16   : 16 bytes per lane, fully sequential (the guide's reference pattern: counted at 1/2)
//   stream8    : 8 bytes per lane, fully sequential
//   write8     : every lane writes one 8-byte word, sequential (residual / history stores)
// usage: fetch_calib <pattern> ; prints the bytes the kernel requested.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}
__global__ void gather8(const double* __restrict__ buf, size_t nwords, double* out, int per_lane) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a = 0.0;
    for (int i = 0; i < per_lane; i++) a += buf[mix(t * per_lane + i) % nwords];
    if (a == 12345.678) out[0] = a;
}
__global__ void rows16(const double2* __restrict__ buf, size_t nrows, double* out, int per_wave) {
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    double a = 0.0;
    for (int i = 0; i < per_wave; i++) { const double2 v = buf[(mix(w * per_wave + i) % nrows) * 64 + lane]; a += v.x + v.y; }
    if (a == 12345.678) out[0] = a;
}
__global__ void stream16(const double2* __restrict__ buf, size_t n, double* out) {
    double a = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = buf[i]; a += v.x + v.y; }
    if (a == 12345.678) out[0] = a;
}
__global__ void stream8(const double* __restrict__ buf, size_t n, double* out) {
    double a = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += buf[i];
    if (a == 12345.678) out[0] = a;
}
__global__ void write8(double* __restrict__ buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = (double)i;
}

int main(int argc, char** argv) {
    const char* pat = argc > 1 ? argv[1] : "gather8";
    const size_t bytes = 4ull << 30;                       // beyond L2 (32 MiB) and Infinity Cache (256 MiB)
    double* buf = nullptr;
    double* out = nullptr;
    CHK(hipMalloc((void**)&buf, bytes));
    CHK(hipMalloc((void**)&out, 8));
    CHK(hipMemset(buf, 0, bytes));
    CHK(hipDeviceSynchronize());
    unsigned long long req = 0;
    if (!std::strcmp(pat, "gather8")) { const int pl = 64; const size_t lanes = 4096ull * 256; hipLaunchKernelGGL(gather8, dim3(4096), dim3(256), 0, 0, buf, bytes / 8, out, pl); req = lanes * pl * 8; }
    else if (!std::strcmp(pat, "rows16")) { const int pw = 256; const size_t waves = 4096ull * 4; hipLaunchKernelGGL(rows16, dim3(4096), dim3(256), 0, 0, (const double2*)buf, bytes / 1024, out, pw); req = waves * pw * 1024; }
    else if (!std::strcmp(pat, "stream16")) { hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const double2*)buf, bytes / 16, out); req = bytes; }
    else if (!std::strcmp(pat, "stream8")) { hipLaunchKernelGGL(stream8, dim3(4096), dim3(256), 0, 0, buf, bytes / 8, out); req = bytes; }
    else if (!std::strcmp(pat, "write8")) { hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, buf, bytes / 8); req = bytes; }
    else { std::fprintf(stderr, "unknown pattern\n"); return 2; }
    CHK(hipGetLastError());
    CHK(hipDeviceSynchronize());
    std::printf("%s requested_bytes %llu\n", pat, req);
    return 0;
}
