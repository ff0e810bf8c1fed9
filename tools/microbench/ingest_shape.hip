// How should one native batch (512 KiB of SMI words in pinned host memory) become samples in the client's pageable buffer?
// Host wall time per call (launches + the one synchronisation + the final memcpy), output 4 / 8 / 12 bytes per sample.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/ingest_shape.hip -o tools/microbench/ingest_shape && tools/microbench/ingest_shape
//   A  H2D copy -> kernel (device -> device) -> D2H copy into the pinned mirror -> sync -> memcpy to pageable
//   B  H2D copy -> kernel storing into the MAPPED pinned mirror -> sync -> memcpy
//   C  kernel LOADING the mapped pinned staging bytes itself and storing into the mapped mirror -> sync -> memcpy (no copy engine)
//   D  as C, the client's buffer registered (hipHostRegister) and written directly: no memcpy
//   E  as A with the D2H copy aimed at the pageable buffer (what the runtime does with pageable memory)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

template <int OUT16>   // 16-byte stores per 16-byte load
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, u32x4 *__restrict__ keep, int n16)
{
    for (int g = blockIdx.x * 256 + threadIdx.x; g < n16; g += gridDim.x * 256) {
        u32x4 w = in[g];
        if (keep) keep[g] = w;
#pragma unroll
        for (int k2 = 0; k2 < OUT16; k2++) out[g * OUT16 + k2] = w + (uint32_t)k2;
    }
}

template <int OUT16>
static void run(int grid)
{
    const size_t nin = 512 << 10, nout = nin * OUT16;
    void *h_in, *h_mirror, *d_in, *d_out, *d_keep;
    hipHostMalloc(&h_in, nin, hipHostMallocMapped); hipHostMalloc(&h_mirror, nout, hipHostMallocMapped);
    hipMalloc(&d_in, nin); hipMalloc(&d_out, nout); hipMalloc(&d_keep, nin);
    void *page = aligned_alloc(4096, nout); memset(page, 1, nout); memset(h_in, 3, nin);
    void *m_in, *m_mirror, *m_page = nullptr;
    hipHostGetDevicePointer(&m_in, h_in, 0); hipHostGetDevicePointer(&m_mirror, h_mirror, 0);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int n16 = nin / 16, reps = 200;
    double t[5] = {0, 0, 0, 0, 0};
    for (int v = 0; v < 5; v++) {
        if (v == 3) { if (hipHostRegister(page, nout, hipHostRegisterMapped) != hipSuccess || hipHostGetDevicePointer(&m_page, page, 0) != hipSuccess) { t[3] = -1; continue; } }
        for (int r = -20; r < reps; r++) {
            ((volatile char *)h_in)[(r & 1023) * 64] = (char)r;     // the feeder touched the bytes
            const double t0 = now_us();
            switch (v) {
            case 0: hipMemcpyAsync(d_in, h_in, nin, hipMemcpyHostToDevice, s);
                    hipLaunchKernelGGL(k<OUT16>, dim3(grid), dim3(256), 0, s, (const u32x4 *)d_in, (u32x4 *)d_out, (u32x4 *)nullptr, n16);
                    hipMemcpyAsync(h_mirror, d_out, nout, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); memcpy(page, h_mirror, nout); break;
            case 1: hipMemcpyAsync(d_in, h_in, nin, hipMemcpyHostToDevice, s);
                    hipLaunchKernelGGL(k<OUT16>, dim3(grid), dim3(256), 0, s, (const u32x4 *)d_in, (u32x4 *)m_mirror, (u32x4 *)nullptr, n16);
                    hipStreamSynchronize(s); memcpy(page, h_mirror, nout); break;
            case 2: hipLaunchKernelGGL(k<OUT16>, dim3(grid), dim3(256), 0, s, (const u32x4 *)m_in, (u32x4 *)m_mirror, (u32x4 *)d_keep, n16);
                    hipStreamSynchronize(s); memcpy(page, h_mirror, nout); break;
            case 3: hipLaunchKernelGGL(k<OUT16>, dim3(grid), dim3(256), 0, s, (const u32x4 *)m_in, (u32x4 *)m_page, (u32x4 *)d_keep, n16);
                    hipStreamSynchronize(s); break;
            case 4: hipMemcpyAsync(d_in, h_in, nin, hipMemcpyHostToDevice, s);
                    hipLaunchKernelGGL(k<OUT16>, dim3(grid), dim3(256), 0, s, (const u32x4 *)d_in, (u32x4 *)d_out, (u32x4 *)nullptr, n16);
                    hipMemcpyAsync(page, d_out, nout, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); break;
            }
            if (r >= 0) t[v] += now_us() - t0;
        }
        if (v == 3) hipHostUnregister(page);
    }
    double tm = 0;
    for (int r = 0; r < reps; r++) { const double t0 = now_us(); memcpy(page, h_mirror, nout); tm += now_us() - t0; }
    printf("out %2d B/sample grid %4d: A copies %6.1f | B mapped stores %6.1f | C mapped loads+stores %6.1f | D registered client buffer %6.1f | E D2H to pageable %6.1f | memcpy alone %5.1f us\n",
           4 * OUT16, grid, t[0] / reps, t[1] / reps, t[2] / reps, t[3] < 0 ? -1.0 : t[3] / reps, t[4] / reps, tm / reps);
    hipStreamDestroy(s); hipHostFree(h_in); hipHostFree(h_mirror); hipFree(d_in); hipFree(d_out); hipFree(d_keep); free(page);
}

int main()
{
    for (int grid : {32, 128, 512}) { run<1>(grid); run<2>(grid); run<3>(grid); }
    return 0;
}
