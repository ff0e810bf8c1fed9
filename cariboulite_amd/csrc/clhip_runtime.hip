// clhip_runtime.hip -- runtime plumbing of the C-ABI shim: device selection,
// memory, streams, events, error string.  No kernels here.
#include <stdarg.h>
#include <string.h>

#include "clhip_common.h"

static thread_local char g_err[512] = "";

extern "C" void clhip_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *clhip_last_error(void) { return g_err; }

extern "C" int clhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int clhip_set_device(int device)
{
    CLHIP_CHECK(hipSetDevice(device));
    return 0;
}

extern "C" const char *clhip_arch_name(void)
{
    static thread_local char name[256];
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return "";
    strncpy(name, prop.gcnArchName, sizeof name - 1);
    name[sizeof name - 1] = 0;
    return name;
}

extern "C" void *clhip_malloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        clhip_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

extern "C" void clhip_free(void *p) { if (p) (void)hipFree(p); }

extern "C" void *clhip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        clhip_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

extern "C" void clhip_host_free(void *p) { if (p) (void)hipHostFree(p); }

// the address kernels use for pinned host memory of clhip_host_alloc (NULL when the device cannot reach it)
extern "C" void *clhip_host_device_ptr(void *h)
{
    void *d = nullptr;
    if (!h || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) return nullptr;
    return d;
}

// Pin and map memory the CALLER owns (a client's sample buffer) so that kernels can store into it: returns the address kernels
// use, NULL when the range cannot be registered (already registered in part, not mapped, limits).  Until clhip_host_unregister
// the range has to stay mapped in the process.
extern "C" void *clhip_host_register(void *h, size_t bytes)
{
    void *d = nullptr;
    if (!h || !bytes) return nullptr;
    if (hipHostRegister(h, bytes, hipHostRegisterMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostUnregister(h); return nullptr; }
    return d;
}

extern "C" void clhip_host_unregister(void *h) { if (h && hipHostUnregister(h) != hipSuccess) (void)hipGetLastError(); }

// Copies between device memory and host memory the CALLER owns.  The HIP runtime copies pageable host memory of 1 MiB and
// more (GPU_PINNED_MIN_XFER_SIZE) by pinning the caller's pages in place -- the copy engine then reads or writes the
// process's heap through a user-pointer mapping, and the runtime keeps such pinnings cached by address.  In a long-lived
// process whose heap is freed and reused that path has ended test sessions with "Memory access fault by GPU ... on address
// <a host heap address>" and the HSA runtime's abort() (robustness record: five located cases, all
// inside or right behind a 1.5 MiB pageable copy; none in 18 full runs once that path was closed).  A sample path that runs
// for hours cannot afford that, so host memory that is not page-locked (hipHostMalloc / hipHostRegister: asked of the
// runtime, ~1 us) is copied in pieces below the threshold: each piece goes through the runtime's own pinned staging buffers
// and the device never touches the caller's pages.  Order on the stream is kept; pinned memory is copied in one piece.
#define CLHIP_PAGEABLE_PIECE ((size_t)512 << 10)
static bool clhip_host_is_pinned(const void *h)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, h) != hipSuccess) { (void)hipGetLastError(); return false; }   // unknown to the runtime: plain pageable memory
    return a.type == hipMemoryTypeHost;
}
static bool clhip_copy_in_pieces(const void *h, size_t n)
{
    static const bool off = getenv("CLHIP_PAGEABLE_WHOLE") && atoi(getenv("CLHIP_PAGEABLE_WHOLE"));   // A/B: what round 2 did
    return !off && n > CLHIP_PAGEABLE_PIECE && !clhip_host_is_pinned(h);
}

extern "C" int clhip_memcpy_h2d(void *d, const void *h, size_t n, void *s)
{
    if (clhip_copy_in_pieces(h, n)) {
        for (size_t o = 0; o < n; o += CLHIP_PAGEABLE_PIECE) {
            const size_t m = n - o < CLHIP_PAGEABLE_PIECE ? n - o : CLHIP_PAGEABLE_PIECE;
            CLHIP_CHECK(hipMemcpyAsync((char *)d + o, (const char *)h + o, m, hipMemcpyHostToDevice, (hipStream_t)s));
        }
        return 0;
    }
    CLHIP_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s));
    return 0;
}
extern "C" int clhip_memcpy_d2h(void *h, const void *d, size_t n, void *s)
{
    if (clhip_copy_in_pieces(h, n)) {
        for (size_t o = 0; o < n; o += CLHIP_PAGEABLE_PIECE) {
            const size_t m = n - o < CLHIP_PAGEABLE_PIECE ? n - o : CLHIP_PAGEABLE_PIECE;
            CLHIP_CHECK(hipMemcpyAsync((char *)h + o, (const char *)d + o, m, hipMemcpyDeviceToHost, (hipStream_t)s));
        }
        return 0;
    }
    CLHIP_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s));
    return 0;
}
extern "C" int clhip_memcpy_d2d(void *dd, const void *ds, size_t n, void *s)
{
    CLHIP_CHECK(hipMemcpyAsync(dd, ds, n, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return 0;
}
extern "C" int clhip_memset(void *d, int v, size_t n, void *s)
{
    CLHIP_CHECK(hipMemsetAsync(d, v, n, (hipStream_t)s));
    return 0;
}

extern "C" void *clhip_stream_create(void)
{
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        clhip_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    return (void *)s;
}
extern "C" void clhip_stream_destroy(void *s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
extern "C" int clhip_stream_sync(void *s)
{
    CLHIP_CHECK(hipStreamSynchronize((hipStream_t)s));
    return 0;
}

extern "C" void *clhip_event_create(void)
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void *)e;
}
extern "C" void clhip_event_destroy(void *e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
extern "C" int clhip_event_record(void *e, void *s)
{
    CLHIP_CHECK(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
    return 0;
}
extern "C" int clhip_stream_wait_event(void *s, void *e)
{
    CLHIP_CHECK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0));
    return 0;
}
extern "C" float clhip_event_elapsed_ms(void *a, void *b)
{
    float ms = -1.0f;
    if (hipEventSynchronize((hipEvent_t)b) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.0f;
    return ms;
}
