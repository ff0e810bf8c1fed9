// What can cross PCIe on this box, and how fast can the host move what arrived into pageable memory?  The ceilings a stream group at the
// Soapy boundary (cl_group_readStream) is priced against: N native batches in (4 B per sample), N outputs back (12 B per sample for
// FIR64 + 3/2), the last hop a memcpy from the pinned mirror into the client's pageable buffers.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/pcie_duplex.hip -o tools/microbench/pcie_duplex -lpthread && tools/microbench/pcie_duplex
//   1  copy engine, pinned memory: H2D alone, D2H alone, both at once (two streams), 16 / 48 MiB and 4 / 12 MiB messages
//   2  a kernel storing into MAPPED pinned memory (no copy engine) while the copy engine brings the next input: the other way out
//   3  memcpy pinned -> pageable with 1 .. 16 threads (1.5 MiB pieces, the destination touched before), plain and non-temporal
// One JSON object on stdout.
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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

struct CopyJob { uint8_t *dst; const uint8_t *src; size_t bytes; int nt; };
static void copy_nt(uint8_t *dst, const uint8_t *src, size_t n)
{
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m256i a = _mm256_load_si256((const __m256i *)(src + i)), b = _mm256_load_si256((const __m256i *)(src + i + 32));
        _mm256_stream_si256((__m256i *)(dst + i), a); _mm256_stream_si256((__m256i *)(dst + i + 32), b);
    }
    _mm_sfence();
    if (i < n) memcpy(dst + i, src + i, n - i);
}
static void *copy_thread(void *p)
{
    CopyJob *j = (CopyJob *)p;
    const size_t piece = 3 << 19;                                       // 1.5 MiB: one stream's output
    for (size_t o = 0; o < j->bytes; o += piece) {
        const size_t n = j->bytes - o < piece ? j->bytes - o : piece;
        if (j->nt) copy_nt(j->dst + o, j->src + o, n); else memcpy(j->dst + o, j->src + o, n);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    // pcie_duplex <in_MiB> <out_MiB>: only the copy-engine ceilings of that shape (tools/bench_group.py prices its rows with it)
    const bool shape_only = argc == 3;
    const size_t MB = 1 << 20, in_b = (shape_only ? (size_t)atoi(argv[1]) : 16) * MB, out_b = (shape_only ? (size_t)atoi(argv[2]) : 48) * MB;
    uint8_t *h_in, *h_out, *d_in, *d_out;
    if (hipHostMalloc((void **)&h_in, in_b, hipHostMallocMapped) != hipSuccess || hipHostMalloc((void **)&h_out, out_b, hipHostMallocMapped) != hipSuccess ||
        hipMalloc((void **)&d_in, in_b) != hipSuccess || hipMalloc((void **)&d_out, out_b) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    memset(h_in, 3, in_b); memset(h_out, 0, out_b);
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    const int reps = 20;
    auto timeit = [&](auto fn) { fn(); hipDeviceSynchronize(); const double t0 = now_s(); for (int r = 0; r < reps; r++) fn(); hipDeviceSynchronize(); return (now_s() - t0) / reps; };
    printf("{");
    for (int small = 0; small < (shape_only ? 1 : 2); small++) {
        const size_t ib = small ? in_b / 4 : in_b, ob = small ? out_b / 4 : out_b;
        const double t_h2d = timeit([&] { hipMemcpyAsync(d_in, h_in, ib, hipMemcpyHostToDevice, s0); });
        const double t_d2h = timeit([&] { hipMemcpyAsync(h_out, d_out, ob, hipMemcpyDeviceToHost, s1); });
        const double t_both = timeit([&] { hipMemcpyAsync(d_in, h_in, ib, hipMemcpyHostToDevice, s0); hipMemcpyAsync(h_out, d_out, ob, hipMemcpyDeviceToHost, s1); });
        printf("\"copy_engine_%zu_in_%zu_out_MiB\": {\"h2d_GBs\": %.1f, \"d2h_GBs\": %.1f, \"duplex_ms\": %.3f, \"duplex_total_GBs\": %.1f, \"duplex_out_GBs\": %.1f}, ",
               ib / MB, ob / MB, ib / t_h2d / 1e9, ob / t_d2h / 1e9, t_both * 1e3, (ib + ob) / t_both / 1e9, ob / t_both / 1e9);
        if (shape_only) { printf("\"shape\": {\"in_MiB\": %zu, \"out_MiB\": %zu, \"h2d_GBs\": %.2f, \"d2h_GBs\": %.2f, \"duplex_ms\": %.4f}}\n", ib / MB, ob / MB, ib / t_h2d / 1e9, ob / t_d2h / 1e9, t_both * 1e3); return 0; }
    }
    {   // the kernel's own stores across PCIe (mapped pinned target), the copy engine bringing input at the same time
        void *m_out = nullptr;
        hipHostGetDevicePointer(&m_out, h_out, 0);
        const double t_k = timeit([&] { hipLaunchKernelGGL(expand3, dim3(1024), dim3(256), 0, s1, (const u32x4 *)d_in, (u32x4 *)m_out, in_b / 16); });
        const double t_kc = timeit([&] { hipMemcpyAsync(d_in, h_in, in_b, hipMemcpyHostToDevice, s0);
                                          hipLaunchKernelGGL(expand3, dim3(1024), dim3(256), 0, s1, (const u32x4 *)d_in, (u32x4 *)m_out, in_b / 16); });
        printf("\"kernel_stores_mapped_48_MiB\": {\"alone_GBs\": %.1f, \"with_h2d_16_MiB_ms\": %.3f, \"with_h2d_out_GBs\": %.1f}, ", out_b / t_k / 1e9, t_kc * 1e3, out_b / t_kc / 1e9);
    }
    {   // the last hop: pinned mirror -> pageable client buffers
        uint8_t *page = (uint8_t *)aligned_alloc(4096, out_b);
        memset(page, 1, out_b);
        printf("\"memcpy_pinned_to_pageable_48_MiB_GBs\": {");
        const int counts[] = {1, 2, 4, 8, 12, 16};
        for (int nt = 0; nt < 2; nt++)
            for (int ci = 0; ci < 6; ci++) {
                const int T = counts[ci];
                double best = 1e9;
                for (int r = 0; r < 6; r++) {
                    hipMemcpyAsync(h_out, d_out, out_b, hipMemcpyDeviceToHost, s1);      // the lines have just been written by the device
                    hipStreamSynchronize(s1);
                    pthread_t th[16]; CopyJob jobs[16];
                    const size_t per = (out_b / T + 4095) & ~(size_t)4095;
                    const double t0 = now_s();
                    for (int t = 0; t < T; t++) {
                        const size_t o = (size_t)t * per, n = o >= out_b ? 0 : (out_b - o < per ? out_b - o : per);
                        jobs[t] = CopyJob{page + o, h_out + o, n, nt};
                        pthread_create(&th[t], NULL, copy_thread, &jobs[t]);
                    }
                    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
                    const double dt = now_s() - t0;
                    if (dt < best) best = dt;
                }
                printf("\"%s_%d_threads\": %.1f%s", nt ? "nontemporal" : "memcpy", T, out_b / best / 1e9, (nt == 1 && ci == 5) ? "" : ", ");
            }
        printf("}");
        free(page);
    }
    printf("}\n");
    return 0;
}
