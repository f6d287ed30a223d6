// Issue rate of dependent-free FMA streams on gfx950: fp64, scalar fp32 and packed fp32 (two floats per lane), 8 independent accumulators
// per lane, enough waves to fill every SIMD.  Prints TFLOP/s per type.  (what fp32 arithmetic could buy.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <class T> __device__ T mk(float v);
template <> __device__ double mk<double>(float v) { return (double)v; }
template <> __device__ float mk<float>(float v) { return v; }
template <> __device__ f2 mk<f2>(float v) { return f2{v, v + 1.0f}; }
template <class T>
__global__ void k(T* out, int iters, float seed) {
    T a[8];
    const T x = mk<T>(seed), y = mk<T>(0.5f);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = mk<T>((float)(threadIdx.x + i));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = a[i] * x + y;
    }
    T s = a[0];
#pragma unroll
    for (int i = 1; i < 8; i++) s = s + a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class T>
double run(const char* name, int flops_per_elem) {
    const int blocks = 256 * 8, threads = 256, iters = 20000;
    T* d; hipMalloc((void**)&d, sizeof(T) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, 100, 0.999f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * threads * iters * 8 * flops_per_elem / (ms * 1e-3) / 1e12;
    std::printf("%-12s %8.2f ms  %7.1f TFLOP/s\n", name, ms, tf);
    hipFree(d);
    return tf;
}
int main() {
    run<double>("fp64 fma", 2);
    run<float>("fp32 fma", 2);
    run<f2>("fp32 packed", 4);
    return 0;
}
