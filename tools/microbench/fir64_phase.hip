// What does one sample of the IIR's matrix FIR cost?  12 v_fma_f64 with a scalar (SMEM-loaded) tap each + 2 int16 -> f64
// conversions, variants: taps as kernel-constant SGPRs, taps from a table by scalar loads, taps in VGPRs, no conversions.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/fir64_phase.hip -o tools/microbench/fir64_phase
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(4))) double cdouble_t;
#define NS 4096   // samples per lane
template <int VAR>
__global__ __launch_bounds__(64) void k(const double *__restrict__ G, const uint32_t *__restrict__ x, double *out, unsigned long long *cyc)
{
    const cdouble_t *g = (const cdouble_t *)G;
    double v[12];
    for (int i = 0; i < 12; i++) v[i] = 0.0;
    double tv[6];
    for (int r = 0; r < 6; r++) tv[r] = G[r] * (threadIdx.x + 1);
    uint32_t w = x[threadIdx.x];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
    for (int k = 0; k < NS; k++) {
        double xi, xq;
        if (VAR == 3) { xi = __builtin_bit_cast(double, (unsigned long long)w | 0x3ff0000000000000ull); xq = xi; }   // no conversion
        else { xi = (double)(int16_t)(w & 0xFFFF); xq = (double)(int16_t)(w >> 16); }
        const cdouble_t *gk = g + (k & 63) * 8;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const double gr = VAR == 2 ? tv[r] : gk[r];
            v[r] = __builtin_fma(gr, xi, v[r]);
            v[6 + r] = __builtin_fma(gr, xq, v[6 + r]);
        }
        w = w * 1664525u + 1013904223u;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 12; i++) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int VAR> static void run(const char *name, int blocks)
{
    double *G, *out; uint32_t *x; unsigned long long *cyc;
    hipMalloc(&G, 64 * 8 * 8); hipMalloc(&out, 8 * 64 * blocks); hipMalloc(&x, 4 * 64); hipMalloc(&cyc, 8 * blocks);
    hipMemset(G, 0, 64 * 8 * 8); hipMemset(x, 1, 4 * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(64), 0, 0, G, x, out, cyc);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(64), 0, 0, G, x, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    unsigned long long h[8]; hipMemcpy(h, cyc, 8 * (blocks < 8 ? blocks : 8), hipMemcpyDeviceToHost);
    printf("%-32s blocks %5d: %6.1f cycles/sample in wave 0; kernel %7.1f us -> %5.1f ns per sample per wave, %6.2f T DFMA lanes/s\n", name, blocks,
           (double)h[0] / NS, ms * 1e3, ms * 1e6 / NS, 12.0 * 64 * NS * blocks / (ms * 1e-3) / 1e12);
    hipFree(G); hipFree(out); hipFree(x); hipFree(cyc);
}
int main()
{
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<1>("taps by scalar loads + 2 cvt", blocks);
        run<2>("taps in VGPRs + 2 cvt", blocks);
        run<3>("taps by scalar loads, no cvt", blocks);
    }
    return 0;
}
