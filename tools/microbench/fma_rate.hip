// Issue rate of v_fma_f64 / v_fma_f32 / v_pk_fma_f32 on gfx950: 16 independent chains per lane, 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/fma_rate.hip -o /tmp/fma_rate && /tmp/fma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <class T> __global__ __launch_bounds__(256) void k(T *out, T a, T b, int iters)
{
    T v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = a * (T)(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = v[i] * a + b;
    }
    T s = v[0];
#pragma unroll
    for (int i = 1; i < 16; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class T> static double run(const char *name, double flop_per_op)
{
    T *d;
    const int blocks = 256 * 4, iters = 4096;
    hipMalloc(&d, sizeof(T) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    T a, b;
    if constexpr (sizeof(T) == 8 && !__is_floating_point(T)) { a = T{1.0000001f, 0.9999999f}; b = T{1e-7f, -1e-7f}; }
    else { a = (T)1.0000001; b = (T)1e-7; }
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d, a, b, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 10; w++) hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d, a, b, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = 10.0 * blocks * 256 * (double)iters * 16;
    printf("%-14s %8.2f T instr-lanes/s  %8.2f TFLOP/s\n", name, ops / (ms * 1e-3) / 1e12, ops * flop_per_op / (ms * 1e-3) / 1e12);
    hipFree(d);
    return 0;
}
int main()
{
    run<double>("v_fma_f64", 2);
    run<float>("v_fma_f32", 2);
    run<f32x2>("v_pk_fma_f32", 4);
    return 0;
}
