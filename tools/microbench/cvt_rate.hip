// Issue cost (cycles per wave instruction on one SIMD) of the conversions the IIR kernels need, measured with
// s_memtime around an unrolled block of independent instructions, four waves per SIMD (one 1024-thread workgroup on one CU).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/cvt_rate.hip -o /tmp/cvt_rate && /tmp/cvt_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
#define BODY(NAME, ASM, OUTC, INC, TOUT, TIN)                                                         \
__global__ void NAME(unsigned long long *cyc, TOUT *sink, TIN seed)                                   \
{                                                                                                     \
    TIN a[8]; TOUT r[8];                                                                              \
    for (int i = 0; i < 8; i++) a[i] = seed + (TIN)(threadIdx.x + i);                                 \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
    for (int it = 0; it < 256; it++) {                                                                \
        _Pragma("unroll") for (int k = 0; k < REP / 8; k++)                                           \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : OUTC(r[i]) : INC(a[i])); \
    }                                                                                                 \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                  \
    TOUT s = r[0]; for (int i = 1; i < 8; i++) s += r[i];                                             \
    sink[blockIdx.x * 1024 + threadIdx.x] = s;                                                          \
}
BODY(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1", "=v", "v", double, int)
BODY(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %1", "=v", "v", float, double)
BODY(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %1", "=v", "v", int, float)
BODY(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1", "=v", "v", int, double)
BODY(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %1", "=v", "v", float, int)
BODY(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1", "=v", "v", double, float)
BODY(k_add_f64, "v_add_f64 %0, %1, %1", "=v", "v", double, double)
BODY(k_mul_f64, "v_mul_f64 %0, %1, %1", "=v", "v", double, double)
BODY(k_fma_f64, "v_fma_f64 %0, %1, %1, %1", "=v", "v", double, double)
BODY(k_fma_f32, "v_fma_f32 %0, %1, %1, %1", "=v", "v", float, float)
BODY(k_bfe, "v_bfe_i32 %0, %1, 0, 16", "=v", "v", int, int)
BODY(k_mov64, "v_mov_b64 %0, %1", "=v", "v", double, double)
template <class F, class TO, class TI> static void run(const char *n, F f, TO *, TI seed)
{
    unsigned long long *c; TO *s;
    hipMalloc(&c, 8 * 1024); hipMalloc(&s, sizeof(TO) * 1024 * 1024);
    hipLaunchKernelGGL(f, dim3(1), dim3(1024), 0, 0, c, s, seed);
    hipLaunchKernelGGL(f, dim3(1), dim3(1024), 0, 0, c, s, seed);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz on gfx9?  report ticks per instruction and let fma_f32 calibrate (4 cycles)
    printf("%-16s %8.3f ticks/instr with 4 waves per SIMD -> %.2f cycles of SIMD issue each\n", n, (double)h / (256.0 * REP), (double)h / (256.0 * REP) / 4);
    hipFree(c); hipFree(s);
}
int main()
{
    run("v_fma_f32", k_fma_f32, (float *)0, 1.0f);
    run("v_fma_f64", k_fma_f64, (double *)0, 1.0);
    run("v_add_f64", k_add_f64, (double *)0, 1.0);
    run("v_mul_f64", k_mul_f64, (double *)0, 1.0);
    run("v_cvt_f64_i32", k_cvt_f64_i32, (double *)0, 1);
    run("v_cvt_f64_f32", k_cvt_f64_f32, (double *)0, 1.0f);
    run("v_cvt_f32_f64", k_cvt_f32_f64, (float *)0, 1.0);
    run("v_cvt_i32_f64", k_cvt_i32_f64, (int *)0, 1.0);
    run("v_cvt_i32_f32", k_cvt_i32_f32, (int *)0, 1.0f);
    run("v_cvt_f32_i32", k_cvt_f32_i32, (float *)0, 1);
    run("v_bfe_i32", k_bfe, (int *)0, 1);
    run("v_mov_b64", k_mov64, (double *)0, 1.0);
    return 0;
}
