// Checks the DPP wave reduction used by DevWG (row rotations + row_bcast:15 / row_bcast:31 combine) against a host sum, and that the raw
// buffer-load builtins address as expected (resource base + scalar offset + lane offset).  hipcc --offload-arch=gfx950 -O3 wave_reduce.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
template <int CTRL, bool BC>
__device__ __forceinline__ double dppMov(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, BC);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, BC);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readLane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ double waveSumNew(double v) {
    v += dppMov<0x128, false>(v); v += dppMov<0x124, false>(v); v += dppMov<0x122, false>(v); v += dppMov<0x121, false>(v);
    v += dppMov<0x142, true>(v);      // row_bcast:15, rows without a source read 0
    v += dppMov<0x143, true>(v);      // row_bcast:31
    return readLane(v, 63);
}
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* in, double* out, double* out2) {
    const int l = threadIdx.x;
    out[blockIdx.x * 64 + l] = waveSumNew(in[blockIdx.x * 64 + l]);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 64 * 8 * gridDim.x, 0x00020000);
    const unsigned soff = __builtin_amdgcn_readfirstlane(blockIdx.x * 512);
    u4 w = __builtin_amdgcn_raw_buffer_load_b128(r, (l >> 1) * 16, soff, 0);
    double2 d = *(double2*)&w;
    out2[blockIdx.x * 64 + l] = (l & 1) ? d.y : d.x;
}
int main() {
    const int G = 64;
    std::vector<double> h(64 * G), o(64 * G), o2(64 * G);
    srand(3);
    for (auto& x : h) x = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 40) - 20.0);
    double *di, *dout, *dout2;
    hipMalloc(&di, 8 * h.size()); hipMalloc(&dout, 8 * h.size()); hipMalloc(&dout2, 8 * h.size());
    hipMemcpy(di, h.data(), 8 * h.size(), hipMemcpyHostToDevice);
    k<<<G, 64>>>(di, dout, dout2);
    hipMemcpy(o.data(), dout, 8 * h.size(), hipMemcpyDeviceToHost);
    hipMemcpy(o2.data(), dout2, 8 * h.size(), hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0, exact = 0;
    for (int g = 0; g < G; g++) {
        // the device order: rotations inside each row of 16 (butterfly of rotations = fixed tree), then (r2 + r3) + (r0 + r1)
        double r[4];
        for (int q = 0; q < 4; q++) {
            double a[16], b[16];
            for (int t = 0; t < 16; t++) a[t] = h[g * 64 + q * 16 + t];
            for (int rot : {8, 4, 2, 1}) { for (int t = 0; t < 16; t++) b[t] = a[t] + a[(t + rot) % 16]; for (int t = 0; t < 16; t++) a[t] = b[t]; }
            r[q] = a[15];
        }
        const double want = (r[2] + r[3]) + (r[0] + r[1]);
        long double ref = 0; for (int t = 0; t < 64; t++) ref += h[g * 64 + t];
        for (int t = 0; t < 64; t++) {
            if (o[g * 64 + t] != o[g * 64]) bad++;
            if (o2[g * 64 + t] != h[g * 64 + t]) bad2++;
        }
        if (o[g * 64] == want) exact++;
        long double mag = 0; for (int t = 0; t < 64; t++) mag += fabsl(h[g * 64 + t]);
        if (fabsl(o[g * 64] - ref) > 1e-14L * mag) bad++;
    }
    printf("wave sums: %d groups, %d bit-equal to (r2+r3)+(r0+r1), %d wrong; buffer loads: %d wrong\n", G, exact, bad, bad2);
    return bad || bad2;
}
