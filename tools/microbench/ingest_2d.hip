// Many small reads across PCIe while a kernel stores across it the other way: how should N members' 512 KiB batches, which lie in
// separate pinned buffers, reach the device?  (cl_group_readStream's timeline: profiles/r04/group_call_timeline_c2.txt)
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/ingest_2d.hip -o tools/microbench/ingest_2d && tools/microbench/ingest_2d
//   A  32 x hipMemcpyAsync(512 KiB), two streams taking turns                 (what the group does)
//   B  8 x hipMemcpy2DAsync(4 rows x 512 KiB, source pitch 4 MiB)             (the batches at one stride in ONE pinned slab)
//   C  1 x hipMemcpy2DAsync(32 rows)                                          (the whole call at once)
//   D  8 x hipMemcpyAsync(2 MiB)                                              (contiguous sub-batches: the ceiling of the idea)
// each alone, and with a kernel storing 48 MiB into mapped pinned memory at the same time (the sub-batches' launches).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
static double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
__global__ __launch_bounds__(256) void expand3(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n16)
{
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n16; g += (size_t)gridDim.x * 256) {
        const u32x4 w = in[g];
        out[3 * g] = w; out[3 * g + 1] = w + 1u; out[3 * g + 2] = w + 2u;
    }
}
int main()
{
    const size_t MB = 1 << 20, row = 512 << 10, pitch = 4 * MB, n = 32;
    uint8_t *slab, *h_out, *d_in, *d_src, *m_out;
    hipHostMalloc((void **)&slab, n * pitch, hipHostMallocMapped); hipHostMalloc((void **)&h_out, 48 * MB, hipHostMallocMapped);
    hipMalloc((void **)&d_in, n * row + 256); hipMalloc((void **)&d_src, 16 * MB);
    hipHostGetDevicePointer((void **)&m_out, h_out, 0);
    memset(slab, 3, n * pitch);
    hipStream_t s[2], sk;
    hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking); hipStreamCreateWithFlags(&sk, hipStreamNonBlocking);
    auto A = [&] { for (size_t i = 0; i < n; i++) hipMemcpyAsync(d_in + i * row, slab + i * pitch, row, hipMemcpyHostToDevice, s[i & 1]); };
    auto B = [&] { for (size_t b = 0; b < n / 4; b++) hipMemcpy2DAsync(d_in + 4 * b * row, row, slab + 4 * b * pitch, pitch, row, 4, hipMemcpyHostToDevice, s[b & 1]); };
    auto C = [&] { hipMemcpy2DAsync(d_in, row, slab, pitch, row, n, hipMemcpyHostToDevice, s[0]); };
    auto D = [&] { for (size_t b = 0; b < n / 4; b++) hipMemcpyAsync(d_in + 4 * b * row, slab + b * pitch, 4 * row > pitch ? pitch : 4 * row, hipMemcpyHostToDevice, s[b & 1]); };
    auto K = [&] { hipLaunchKernelGGL(expand3, dim3(1024), dim3(256), 0, sk, (const u32x4 *)d_src, (u32x4 *)m_out, 16 * MB / 16); };
    const int reps = 20;
    auto timeit = [&](auto fn) { fn(); hipDeviceSynchronize(); const double t0 = now_s(); for (int r = 0; r < reps; r++) { fn(); hipDeviceSynchronize(); } return (now_s() - t0) / reps * 1e3; };
    printf("{\"kernel_stores_48MiB_alone_ms\": %.3f", timeit(K));
    const char *names[4] = {"A_32_copies_512K", "B_8_copies_2D_4rows", "C_1_copy_2D_32rows", "D_8_copies_2MiB_contiguous"};
    for (int v = 0; v < 4; v++) {
        auto in = [&] { if (v == 0) A(); else if (v == 1) B(); else if (v == 2) C(); else D(); };
        const double alone = timeit(in);
        const double both = timeit([&] { K(); in(); });
        printf(", \"%s\": {\"alone_ms\": %.3f, \"with_kernel_ms\": %.3f}", names[v], alone, both);
    }
    printf("}\n");
    return 0;
}
