// What the memory system gives a kernel with config 2's traffic shape (read 4 B, write 12 B per sample) when the kernel
// does nothing else: every wave store instruction writes 1 KiB contiguous (64 lanes x 16 B), loads are 16 B per lane.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/write_shape.hip -o tools/microbench/write_shape && tools/microbench/write_shape
// Variants: pure write / pure read / copy (1:1) / 1:3; plain vs non-temporal stores; many short workgroups vs
// persistent ones; 2^28 samples (1 GiB in, 3 GiB out: outside the 256 MB Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// each workgroup iteration: `rd` 16-byte loads and `wr` 16-byte stores per lane, 256 lanes; chunk c of the grid-stride loop
template <int RD, int WR, bool NT>
__global__ __launch_bounds__(256) void shape_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, long n_chunks)
{
    const int t = threadIdx.x;
    for (long c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        u32x4 v[RD > 0 ? RD : 1];
        u32x4 acc = {(uint32_t)c, 1u, 2u, 3u};
#pragma unroll
        for (int k = 0; k < RD; k++) {
            v[k] = __builtin_nontemporal_load(in + (c * RD + k) * 256 + t);
            acc ^= v[k];
        }
#pragma unroll
        for (int k = 0; k < WR; k++) {
            u32x4 o = acc + (uint32_t)k;
            if (NT) __builtin_nontemporal_store(o, out + (c * WR + k) * 256 + t);
            else out[(c * WR + k) * 256 + t] = o;
        }
        if (WR == 0 && acc.x == 0x12345678u) out[t] = acc;     // keep the loads alive
    }
}

template <int RD, int WR, bool NT>
static void run(const char *name, const u32x4 *in, u32x4 *out, long n_chunks, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL((shape_kernel<RD, WR, NT>), dim3(grid), dim3(256), 0, 0, in, out, n_chunks);
    hipEventRecord(e0);
    const int reps = 30;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((shape_kernel<RD, WR, NT>), dim3(grid), dim3(256), 0, 0, in, out, n_chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double rb = (double)n_chunks * RD * 4096, wb = (double)n_chunks * WR * 4096;
    printf("%-44s grid %6d  %7.3f ms  read %6.2f  write %6.2f  total %6.2f TB/s\n", name, grid, ms, rb / ms / 1e9, wb / ms / 1e9, (rb + wb) / ms / 1e9);
}

int main()
{
    const long n = 1L << 28;                              // samples
    u32x4 *in, *out;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 12);
    hipMemset(in, 1, n * 4); hipMemset(out, 2, n * 12);
    const long c13 = n * 4 / 4096;                       // chunks when a chunk reads 4 KiB (1 load per lane)
    for (int grid : {1024, 2048, 4096, 65536}) {
        run<1, 3, true>("1:3  (config 2's shape), NT stores", in, out, c13, grid);
        run<1, 3, false>("1:3  (config 2's shape), plain stores", in, out, c13, grid);
    }
    // the same, every wave's 1 KiB store instruction straddling 128-byte lines (base + 32 bytes): 7 whole lines + 2 partial ones
    run<1, 3, true>("1:3, NT stores, output base + 32 B", in, out + 2, c13 - 1, 1024);
    run<1, 3, false>("1:3, plain stores, output base + 32 B", in, out + 2, c13 - 1, 1024);
    run<1, 3, true>("1:3, NT stores, output base + 32 B", in, out + 2, c13 - 1, 4096);
    run<1, 3, true>("1:3, NT, input base + 96 B, output + 32 B", in + 6, out + 2, c13 - 1, 1024);
    run<1, 3, true>("1:3, NT, input base + 96 B", in + 6, out, c13 - 1, 1024);
    run<0, 3, true>("pure write 3 GiB, NT", in, out, c13, 4096);
    run<0, 3, false>("pure write 3 GiB, plain", in, out, c13, 4096);
    run<0, 3, false>("pure write 3 GiB, plain", in, out, c13, 65536);
    run<1, 0, false>("pure read 1 GiB", in, out, c13, 4096);
    run<1, 1, true>("copy 1 GiB -> 1 GiB, NT", in, out, c13, 4096);
    run<1, 1, false>("copy 1 GiB -> 1 GiB, plain", in, out, c13, 4096);
    run<3, 1, true>("3:1 (read 3 GiB, write 1 GiB), NT", out, in, c13, 4096);
    hipFree(in); hipFree(out);
    return 0;
}
