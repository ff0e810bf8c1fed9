// clhip_runtime.hip -- runtime plumbing of the C-ABI shim: device selection,
// memory, streams, events, error string.  No kernels here.
#include <stdarg.h>
#include <string.h>
#include <unistd.h>

#include <atomic>

#include "clhip_common.h"

// ---- what this library has asked the runtime to do with HOST memory it does not own, most recent 256 operations ----
// The robustness record of round 3 ends on a GPU page fault at a host heap address that could not be
// set against the library's own registrations and copies, because nobody had written those down.  Every registration,
// release and copy of caller-owned host memory is noted here (a store and a relaxed counter: nothing a call could feel);
// clhip_debug_ops() reads the ring, clhip_debug_ops_dump(fd) writes it with write(2) only -- safe to call from the SIGABRT
// handler the HSA runtime's fault handler ends in (tests/cpp/abrt_trace.c does), so that a fault address is decidable against
// these ranges the next time one is seen.
struct clhip_op_slot { std::atomic<uint64_t> seq; uint32_t op; uintptr_t base; size_t len; };
static clhip_op_slot g_ops[256];
static std::atomic<uint64_t> g_op_seq{0};
static std::atomic<uint64_t> g_copy_ctr[4];      // [0] pageable copies made in pieces, [1] page-locked copies made whole,
                                                 // [2] largest pageable range handed to ONE hipMemcpyAsync, [3] pageable bytes copied
static void clhip_note_op(uint32_t op, const void *base, size_t len)
{
    const uint64_t n = g_op_seq.fetch_add(1, std::memory_order_relaxed) + 1;
    clhip_op_slot &sl = g_ops[n & 255];
    sl.seq.store(0, std::memory_order_relaxed);                  // being written
    sl.op = op; sl.base = (uintptr_t)base; sl.len = len;
    sl.seq.store(n, std::memory_order_release);
}
extern "C" size_t clhip_debug_ops(clhip_op_record *out, size_t max)
{
    const uint64_t last = g_op_seq.load(std::memory_order_acquire);
    const uint64_t first = last > 256 ? last - 255 : 1;
    size_t k = 0;
    for (uint64_t n = first; n <= last && k < max; n++) {
        const clhip_op_slot &sl = g_ops[n & 255];
        if (sl.seq.load(std::memory_order_acquire) != n) continue;    // overwritten meanwhile, or half written
        out[k].seq = n; out[k].op = sl.op; out[k].base = (uint64_t)sl.base; out[k].len = (uint64_t)sl.len;
        k++;
    }
    return k;
}
static size_t hex_u64(char *dst, uint64_t v)
{
    char tmp[16];
    int n = 0;
    do { tmp[n++] = "0123456789abcdef"[v & 15]; v >>= 4; } while (v);
    for (int i = 0; i < n; i++) dst[i] = tmp[n - 1 - i];
    return (size_t)n;
}
extern "C" void clhip_debug_ops_dump(int fd)
{
    static const char *const names[] = {"?", "register", "register-failed", "unregister", "h2d-pageable-in-pieces", "d2h-pageable-in-pieces",
                                        "h2d-page-locked", "d2h-page-locked", "h2d-pageable-small", "d2h-pageable-small"};
    const char head[] = "[clhip] host-memory operations of libcariboulite_hip.so, oldest first (seq op base len, hex):\n";
    if (write(fd, head, sizeof head - 1) < 0) return;
    const uint64_t last = g_op_seq.load(std::memory_order_acquire);
    for (uint64_t n = last > 256 ? last - 255 : 1; n <= last; n++) {
        const clhip_op_slot &sl = g_ops[n & 255];
        if (sl.seq.load(std::memory_order_acquire) != n) continue;
        char line[160];
        size_t k = 0;
        line[k++] = ' '; line[k++] = ' ';
        k += hex_u64(line + k, n); line[k++] = ' ';
        const char *nm = names[sl.op < 10 ? sl.op : 0];
        for (; *nm; nm++) line[k++] = *nm;
        line[k++] = ' '; line[k++] = '0'; line[k++] = 'x'; k += hex_u64(line + k, (uint64_t)sl.base);
        line[k++] = ' '; line[k++] = '+'; line[k++] = '0'; line[k++] = 'x'; k += hex_u64(line + k, (uint64_t)sl.len);
        line[k++] = '\n';
        if (write(fd, line, k) < 0) return;
    }
}
extern "C" void clhip_debug_copy_counters(uint64_t out[4])
{
    for (int i = 0; i < 4; i++) out[i] = g_copy_ctr[i].load(std::memory_order_relaxed);
}

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

// diagnostic: the runtime's sticky "last error" of the calling thread (consumed by the call; 0 = none).  A HIP call that fails and is
// handled must not leave it behind -- the next kernel launch's check would report it as its own.
extern "C" int clhip_debug_sticky_error(void) { return (int)hipGetLastError(); }

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
    hipError_t e = hipHostRegister(h, bytes, hipHostRegisterMapped);
    if (e != hipSuccess) { clhip_set_error("hipHostRegister(%p, %zu): %s", h, bytes, hipGetErrorString(e)); (void)hipGetLastError(); clhip_note_op(CLHIP_OP_REGISTER_FAILED, h, bytes); return nullptr; }
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) {
        (void)hipGetLastError(); (void)hipHostUnregister(h);
        clhip_set_error("hipHostGetDevicePointer(%p): the device cannot reach registered host memory", h);
        clhip_note_op(CLHIP_OP_REGISTER_FAILED, h, bytes);
        return nullptr;
    }
    clhip_note_op(CLHIP_OP_REGISTER, h, bytes);
    return d;
}

extern "C" void clhip_host_unregister(void *h)
{
    if (!h) return;
    clhip_note_op(CLHIP_OP_UNREGISTER, h, 0);
    if (hipHostUnregister(h) != hipSuccess) (void)hipGetLastError();
}

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
static bool clhip_host_byte_is_pinned(const void *h)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, h) != hipSuccess) { (void)hipGetLastError(); return false; }   // unknown to the runtime: plain pageable memory
    return a.type == hipMemoryTypeHost;
}
// The WHOLE range, not its first byte: registrations are whole pages, so a neighbouring registration (somebody else's buffer that ends
// in the page this one starts in) makes the head of a pageable buffer look page-locked -- a copy of the whole range then fails in the
// runtime ("invalid argument"; found by the group's random walk: the lone devices' buffers lie next to the registered ones).
static bool clhip_host_is_pinned(const void *h, size_t n)
{
    return clhip_host_byte_is_pinned(h) && clhip_host_byte_is_pinned((const uint8_t *)h + (n ? n - 1 : 0)) &&
           clhip_host_byte_is_pinned((const uint8_t *)h + n / 2);
}
// what kind of copy this is, noted in the operation ring and the counters (clhip_debug_copy_counters: a test asserts that no
// pageable range above one piece ever reaches a single hipMemcpyAsync)
static bool clhip_copy_in_pieces(const void *h, size_t n, bool h2d)
{
    if (n <= CLHIP_PAGEABLE_PIECE) {
        // (small copies are staged by the runtime whatever the memory is; asking what it is would cost more than the copy's set-up)
        return false;
    }
    const bool pinned = clhip_host_is_pinned(h, n);
    clhip_note_op(pinned ? (h2d ? CLHIP_OP_H2D_LOCKED : CLHIP_OP_D2H_LOCKED) : (h2d ? CLHIP_OP_H2D_PIECES : CLHIP_OP_D2H_PIECES), h, n);
    if (pinned) { g_copy_ctr[1].fetch_add(1, std::memory_order_relaxed); return false; }
    g_copy_ctr[0].fetch_add(1, std::memory_order_relaxed);
    g_copy_ctr[3].fetch_add(n, std::memory_order_relaxed);
    return true;
}
static void clhip_note_pageable_piece(size_t m)
{
    uint64_t cur = g_copy_ctr[2].load(std::memory_order_relaxed);
    while (m > cur && !g_copy_ctr[2].compare_exchange_weak(cur, m, std::memory_order_relaxed)) {}
}

// A host range that the runtime refuses as one copy ("invalid argument"): it STRADDLES the edge of somebody's registration --
// registrations are whole pages, so a registered buffer that ends inside a page pins the head of whatever the heap placed behind it,
// and a copy that starts in that page and runs on into pageable memory is neither one thing nor the other to the runtime (found by
// the stream group's random walk: the lone devices' client buffers lay right behind the group's registered ones in the heap).  Such
// a range is walked page by page and copied in runs of one kind: page-locked runs whole, pageable runs in pieces.  Rare, hence on the
// error path only: the normal copy costs nothing for it.
static int clhip_copy_straddling(void *dev, void *host, size_t n, bool h2d, hipStream_t s)
{
    (void)hipGetLastError();
    const uintptr_t page = 4096;
    size_t o = 0;
    while (o < n) {
        const bool pinned = clhip_host_byte_is_pinned((char *)host + o);
        size_t e = (size_t)((((uintptr_t)host + o) | (page - 1)) + 1 - (uintptr_t)host);      // the end of this byte's page
        while (e < n && clhip_host_byte_is_pinned((char *)host + e) == pinned && (pinned || e - o < CLHIP_PAGEABLE_PIECE)) e += page;
        if (e > n) e = n;
        if (!pinned) clhip_note_pageable_piece(e - o);
        CLHIP_CHECK(h2d ? hipMemcpyAsync((char *)dev + o, (char *)host + o, e - o, hipMemcpyHostToDevice, s)
                        : hipMemcpyAsync((char *)host + o, (char *)dev + o, e - o, hipMemcpyDeviceToHost, s));
        o = e;
    }
    return 0;
}
#define CLHIP_COPY_OR_STRADDLE(call, dev, host, n, h2d, s)                                             \
    do {                                                                                               \
        const hipError_t first_ = (call);                                                              \
        if (first_ == hipErrorInvalidValue) return clhip_copy_straddling((void *)(dev), (void *)(host), (n), (h2d), (hipStream_t)(s)); \
        CLHIP_CHECK(first_);                                                                           \
    } while (0)

extern "C" int clhip_memcpy_h2d(void *d, const void *h, size_t n, void *s)
{
    if (clhip_copy_in_pieces(h, n, true)) {
        for (size_t o = 0; o < n; o += CLHIP_PAGEABLE_PIECE) {
            const size_t m = n - o < CLHIP_PAGEABLE_PIECE ? n - o : CLHIP_PAGEABLE_PIECE;
            clhip_note_pageable_piece(m);
            CLHIP_COPY_OR_STRADDLE(hipMemcpyAsync((char *)d + o, (const char *)h + o, m, hipMemcpyHostToDevice, (hipStream_t)s), (char *)d + o, (const char *)h + o, n - o, true, s);
        }
        return 0;
    }
    CLHIP_COPY_OR_STRADDLE(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s), d, h, n, true, s);
    return 0;
}
extern "C" int clhip_memcpy_d2h(void *h, const void *d, size_t n, void *s)
{
    if (clhip_copy_in_pieces(h, n, false)) {
        for (size_t o = 0; o < n; o += CLHIP_PAGEABLE_PIECE) {
            const size_t m = n - o < CLHIP_PAGEABLE_PIECE ? n - o : CLHIP_PAGEABLE_PIECE;
            clhip_note_pageable_piece(m);
            CLHIP_COPY_OR_STRADDLE(hipMemcpyAsync((char *)h + o, (const char *)d + o, m, hipMemcpyDeviceToHost, (hipStream_t)s), (const char *)d + o, (char *)h + o, n - o, false, s);
        }
        return 0;
    }
    CLHIP_COPY_OR_STRADDLE(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s), d, h, n, false, s);
    return 0;
}
// `height` rows of `width` bytes from PAGE-LOCKED host memory (rows h_pitch apart) to device rows d_pitch apart, one runtime call: the
// way N equally long batches that lie at one stride in one pinned slab reach the device -- one copy-engine operation instead of N
// (tools/microbench/ingest_2d.hip: 32 x 512 KiB in 0.30 ms instead of 0.84, and hidden behind a kernel that stores across the link the
// other way where the N small copies are not).  Refuses pageable memory (the rule of clhip_memcpy_h2d: the runtime never pins the
// caller's pages in place).
extern "C" int clhip_memcpy2d_h2d(void *d, size_t d_pitch, const void *h, size_t h_pitch, size_t width, size_t height, void *s)
{
    if (!width || !height) return 0;
    if (!clhip_host_is_pinned(h, h_pitch * (height - 1) + width)) { clhip_set_error("clhip_memcpy2d_h2d: the source is not page-locked memory"); return -1; }
    clhip_note_op(CLHIP_OP_H2D_LOCKED, h, h_pitch * (height - 1) + width);
    g_copy_ctr[1].fetch_add(1, std::memory_order_relaxed);
    CLHIP_CHECK(hipMemcpy2DAsync(d, d_pitch, h, h_pitch, width, height, hipMemcpyHostToDevice, (hipStream_t)s));
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
extern "C" int clhip_event_sync(void *e)
{
    CLHIP_CHECK(hipEventSynchronize((hipEvent_t)e));
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
