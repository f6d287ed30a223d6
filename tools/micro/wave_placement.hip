// Where do the two waves of a 128-lane workgroup land?  Persistent-style launch of many workgroups with the solve kernel's footprint (40 KB of LDS: four
// workgroups per CU, 248-VGPR-like register pressure emulated by __launch_bounds__(128, 2)): every wave records HW_REG_HW_ID (SIMD, CU, SE) and the XCC id.
// Prints the histogram of (SIMD of wave 0, SIMD of wave 1) pairs and, per CU, how many wave-0s each SIMD got -- the wave that runs the serial chains of the
// solver (two-loop, knot solves) is wave 0 of its workgroup, so a placement that puts every wave 0 on SIMD 0 / 2 leaves SIMD 1 / 3 idle during those chains.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/wave_placement.hip -o build/micro/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(128, 2) void probe(unsigned* out, int spin) {
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that the whole grid's first wave of workgroups coexists
    double a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 1.0000001 + 1e-9;
    lds[threadIdx.x] = a;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (lds[0] > 1e300 ? 1 : 0);
    }
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 4096, spin = argc > 2 ? atoi(argv[2]) : 20000;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * nb * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(128), 40000, 0, d, spin);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d, sizeof(unsigned) * nb * 4, hipMemcpyDeviceToHost);
    std::map<std::pair<int, int>, int> pairs;
    std::map<long, std::vector<int>> percu;      // (xcc, se, sh, cu) -> wave-0 count per SIMD
    for (int b = 0; b < nb; b++) {
        const unsigned h0 = h[b * 4], h1 = h[b * 4 + 2], x0 = h[b * 4 + 1] & 0xf;
        const int s0 = (h0 >> 4) & 3, s1 = (h1 >> 4) & 3, cu = (h0 >> 8) & 15, sh = (h0 >> 12) & 1, se = (h0 >> 13) & 7;
        pairs[{s0, s1}]++;
        auto& v = percu[(((long)x0 * 8 + se) * 2 + sh) * 16 + cu];
        if (v.empty()) v.assign(4, 0);
        v[s0]++;
        if (b < 16) printf("block %3d: wave0 simd %d wave1 simd %d  cu %2d sh %d se %d xcc %u\n", b, s0, s1, cu, sh, se, x0);
    }
    for (auto& p : pairs) printf("(wave0 on SIMD %d, wave1 on SIMD %d): %d workgroups\n", p.first.first, p.first.second, p.second);
    long tot[4] = {0, 0, 0, 0};
    int ncu = 0, unbalanced = 0;
    for (auto& c : percu) {
        ncu++;
        for (int q = 0; q < 4; q++) tot[q] += c.second[q];
        const int lo = c.second[0] + c.second[2], hi = c.second[1] + c.second[3];
        if (abs(lo - hi) > (lo + hi) / 4) unbalanced++;
    }
    printf("%d distinct CUs seen; wave-0s per SIMD over the launch: %ld %ld %ld %ld; CUs whose even / odd SIMDs differ by more than 25 %%: %d\n", ncu, tot[0], tot[1], tot[2], tot[3], unbalanced);
    return 0;
}
