// How a persistent workgroup with the fused RX kernel's memory pattern overlaps its traffic with its arithmetic:
// per tile and lane, 5 prefetched 16-byte loads (the next tile's words), a block of dependent packed FMAs standing for
// the FIR (SPIN of them per lane), then 12 stores of 16 bytes in 1 KiB-contiguous wave runs -- config 2's 1:3 shape.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/pipe_shape.hip -o tools/microbench/pipe_shape
// Variants: where the wait for the prefetched words sits (before / after the arithmetic), workgroup barriers or
// none, how many workgroups per CU.  Prints ms per 2^28-sample pass next to the arithmetic-only and traffic-only times.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // bit 0: loads on, bit 1: stores on, bit 2: barriers, bit 3: wait for the loads BEFORE the arithmetic
__global__ __launch_bounds__(256, 4) void pipe_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, long n_tiles, int spin, float a, float b)
{
    extern __shared__ unsigned char lds[];
    const int t = threadIdx.x;
    u32x4 r[5];
    long tile = blockIdx.x;
    if (MODE & 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = __builtin_nontemporal_load(in + tile * 1024 + k * 256 + t);
        r[4] = r[0];
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) r[k] = u32x4{(uint32_t)t, 1u, 2u, 3u};
    }
    while (tile < n_tiles) {
        u32x4 cur[5];
#pragma unroll
        for (int k = 0; k < 5; k++) { cur[k] = r[k]; asm volatile("" : "+v"(cur[k])); }
        if (MODE & 4) { ((u32x4 *)lds)[t] = cur[0]; __syncthreads(); cur[0] = ((u32x4 *)lds)[(t + 1) & 255]; }
        const long next = tile + gridDim.x;
        if ((MODE & 1) && next < n_tiles) {
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = __builtin_nontemporal_load(in + next * 1024 + k * 256 + t);
            r[4] = r[0];
        }
        if (MODE & 8) {
#pragma unroll
            for (int k = 0; k < 5; k++) asm volatile("" : "+v"(r[k]));
        }
        // the "FIR": 16 independent accumulators, `spin` rounds of 16 packed FMAs
        f32x2 acc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { acc[i].x = __builtin_bit_cast(float, cur[i % 5][i % 4] & 0x3fffffffu); acc[i].y = acc[i].x; }
        for (int s = 0; s < spin; s++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = acc[i] * a + b;
        }
        if (!(MODE & 8)) {
#pragma unroll
            for (int k = 0; k < 5; k++) asm volatile("" : "+v"(r[k]));
        }
        if (MODE & 4) __syncthreads();
        if (MODE & 2) {
#pragma unroll
            for (int k = 0; k < 12; k++) {
                u32x4 o = {__builtin_bit_cast(uint32_t, acc[k].x), __builtin_bit_cast(uint32_t, acc[k].y), __builtin_bit_cast(uint32_t, acc[(k + 4) & 15].x), (uint32_t)k};
                __builtin_nontemporal_store(o, out + tile * 3072 + k * 256 + t);
            }
        } else if (acc[0].x == 12345.f) out[t] = cur[1];
        if (MODE & 4) __syncthreads();
        tile = next;
    }
}

template <int MODE>
static float run(const u32x4 *in, u32x4 *out, long n_tiles, int grid, int spin)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 30; w++) hipLaunchKernelGGL((pipe_kernel<MODE>), dim3(grid), dim3(256), 4096, 0, in, out, n_tiles, spin, 1.0000001f, 1e-7f);
    hipEventRecord(e0);
    const int reps = 50;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((pipe_kernel<MODE>), dim3(grid), dim3(256), 4096, 0, in, out, n_tiles, spin, 1.0000001f, 1e-7f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const long n = 1L << 28;
    const long n_tiles = n / 4096;                         // a tile: 4096 samples = 16 KiB in, 48 KiB out
    u32x4 *in, *out;
    hipMalloc(&in, n * 4 + 65536); hipMalloc(&out, n * 12 + 65536);
    hipMemset(in, 1, n * 4); hipMemset(out, 2, n * 12);
    printf("spin = packed FMAs per lane and tile / 16 (the fused kernel: ~1000 packed FMAs per lane and tile -> spin 64)\n");
    for (int grid : {1024, 2048}) {
        for (int spin : {0, 32, 64, 96}) {
            const float t_ar = run<0>(in, out, n_tiles, grid, spin);
            const float t_mem = run<3>(in, out, n_tiles, grid, 0);
            const float t_after = run<3>(in, out, n_tiles, grid, spin);
            const float t_before = run<11>(in, out, n_tiles, grid, spin);
            const float t_bar = run<7>(in, out, n_tiles, grid, spin);
            const float t_ld = run<1>(in, out, n_tiles, grid, spin);
            const float t_st = run<2>(in, out, n_tiles, grid, spin);
            printf("grid %5d spin %3d: arithmetic only %.3f | traffic only %.3f | both, wait after the arithmetic %.3f | wait before %.3f | + barriers %.3f | loads only + arith %.3f | stores only + arith %.3f ms\n",
                   grid, spin, t_ar, t_mem, t_after, t_before, t_bar, t_ld, t_st);
        }
    }
    return 0;
}
