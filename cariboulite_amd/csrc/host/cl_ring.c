/* cl_ring.c -- the sample ring of the ASYNC stream mode.
 *
 * The reference's compiled-out USE_ASYNC path (soapy_api/CaribouliteStream.cpp:16-49,70-75) parks every native
 * read in a host-memory circular_buffer<T> (datatypes/circular_buffer.h:16-164).  On the MI355X the reader
 * thread's samples are ALREADY on the GPU (the unpack kernel wrote them), and the consumer's next stages (IIR,
 * conversions, FIR / resampler) run there too, so a host ring would carry every sample over PCIe twice more.
 * This ring therefore splits the two concerns:
 *
 *   storage      one power-of-two array of elements in DEVICE memory (or host memory for the CPU-only uses);
 *   bookkeeping  two monotonically increasing element counters on the host (written / released), the
 *                overwrite-oldest rule and the whole-request blocking rule, under one mutex + condvar.
 *
 * The ring never touches sample data on the device path: cl_ring_put_begin / cl_ring_get_begin hand out a SPAN --
 * at most two linear pieces of the storage -- and the caller moves the data with whatever engine owns it
 * (hipMemcpyAsync device-to-device on its HIP stream, a kernel, memcpy) before calling the matching _end.  The
 * mutex is NOT held in between (the data movement includes a stream synchronisation: tens of microseconds during which
 * the other side would stand still): an open put owns the elements behind `written`, which no get can see before
 * _end publishes them; an open get owns the oldest elements, and the one thing that could touch those -- a put that
 * has to displace the oldest elements of a full ring -- waits for the get to end.  One producer, one consumer.
 * cl_ring_put / cl_ring_get are the one-call forms with host data on either kind
 * of storage; their observable behaviour (accepted / returned counts, order, fill level) replays the op
 * sequences recorded from the reference's template (tests/golden/ring_cases.npz).
 */
#include <errno.h>
#include <pthread.h>
#include <time.h>

#include "cl_internal.h"

struct cl_ring {
    uint8_t *store;              /* cap * elem bytes, device or host */
    int on_device, device;
    size_t elem, cap;            /* cap is a power of two */
    uint64_t written, released;  /* written - released = elements held, never more than cap */
    int put_open, get_open;      /* a span is handed out and its _end has not been called yet */
    uint64_t released_before_put, released_after_drop;   /* around the displacement of the open put (cl_ring_put_cancel) */
    int drop_oldest;             /* a put that does not fit discards the oldest elements instead of being cut */
    int whole_requests;          /* a get waits for its full length and yields nothing otherwise */
    void *xfer;                  /* HIP stream of the host-data convenience calls on device storage */
    pthread_mutex_t mu;
    pthread_cond_t grown;        /* a put was published (whole-request gets wait for it) */
    pthread_cond_t get_done;     /* an open get ended (a displacing put waits for it) */
};

static cl_ring *ring_new(size_t size_elems, size_t elem_bytes, int override_write, int block_read, int on_device, int device)
{
    if (!elem_bytes) return NULL;
    cl_ring *r = (cl_ring *)calloc(1, sizeof *r);
    if (!r) return NULL;
    r->cap = 1;
    while (r->cap < size_elems) r->cap <<= 1;          /* circular_buffer.h:21-25: next power of two */
    r->elem = elem_bytes; r->on_device = on_device; r->device = device;
    r->drop_oldest = override_write != 0; r->whole_requests = block_read != 0;
    if (on_device) {
        if (clhip_set_device(device) == 0) {
            r->store = (uint8_t *)clhip_malloc(r->cap * elem_bytes);
            r->xfer = clhip_stream_create();
        }
        if (!r->store || !r->xfer) { clhip_free(r->store); clhip_stream_destroy(r->xfer); free(r); return NULL; }
    } else {
        r->store = (uint8_t *)malloc(r->cap * elem_bytes);
        if (!r->store) { free(r); return NULL; }
    }
    pthread_mutex_init(&r->mu, NULL);
    pthread_cond_init(&r->grown, NULL);
    pthread_cond_init(&r->get_done, NULL);
    return r;
}

cl_ring *cl_ring_create(size_t size_elems, size_t elem_bytes, int override_write, int block_read)
{
    return ring_new(size_elems, elem_bytes, override_write, block_read, 0, 0);
}

cl_ring *cl_ring_create_device(int device, size_t size_elems, size_t elem_bytes, int override_write, int block_read)
{
    return ring_new(size_elems, elem_bytes, override_write, block_read, 1, device);
}

void cl_ring_destroy(cl_ring *r)
{
    if (!r) return;
    if (r->on_device) {
        clhip_set_device(r->device);
        clhip_stream_sync(r->xfer);
        clhip_stream_destroy(r->xfer);
        clhip_free(r->store);
    } else
        free(r->store);
    pthread_cond_destroy(&r->grown);
    pthread_cond_destroy(&r->get_done);
    pthread_mutex_destroy(&r->mu);
    free(r);
}

void  *cl_ring_storage(const cl_ring *r) { return r->store; }
int    cl_ring_on_device(const cl_ring *r) { return r->on_device; }
size_t cl_ring_capacity(const cl_ring *r) { return r->cap; }

/* `count` elements starting at absolute element index `at` as one or two linear pieces of the storage */
static void ring_span(const cl_ring *r, uint64_t at, size_t count, cl_ring_span *sp)
{
    const size_t first = (size_t)(at & (r->cap - 1));
    const size_t run = r->cap - first;                 /* elements up to the end of the array */
    sp->pos[0] = first; sp->len[0] = count < run ? count : run;
    sp->pos[1] = 0;     sp->len[1] = count - sp->len[0];
}

/* Reserve room for `length` elements.  When they do not fit: a drop_oldest ring releases just enough of its
 * oldest elements (circular_buffer.h:41-45) -- after an open get, which owns exactly those, has ended -- any other
 * ring accepts only what fits (:47).  Returns the number accepted and their span; nobody sees them before
 * cl_ring_put_end. */
size_t cl_ring_put_begin(cl_ring *r, size_t length, cl_ring_span *sp)
{
    pthread_mutex_lock(&r->mu);
    size_t held = (size_t)(r->written - r->released);
    r->released_before_put = r->released;
    if (r->drop_oldest && length > r->cap - held) {
        while (r->get_open) pthread_cond_wait(&r->get_done, &r->mu);
        held = (size_t)(r->written - r->released);
        r->released_before_put = r->released;
        if (length > r->cap - held) {
            size_t drop = length - (r->cap - held);
            if (drop > held) drop = held;              /* a request beyond the capacity keeps its first cap elements */
            r->released += drop;
            held -= drop;
        }
    }
    r->released_after_drop = r->released;
    const size_t take = length < r->cap - held ? length : r->cap - held;
    ring_span(r, r->written, take, sp);
    r->put_open = 1;
    pthread_mutex_unlock(&r->mu);
    return take;
}

void cl_ring_put_end(cl_ring *r, size_t accepted)
{
    pthread_mutex_lock(&r->mu);
    r->written += accepted;
    r->put_open = 0;
    if (r->whole_requests) pthread_cond_signal(&r->grown);
    pthread_mutex_unlock(&r->mu);
}

/* Give up an open put whose span has NOT been written to: nothing is published, and the elements a full ring displaced
 * for it are held again -- unless a get has claimed elements in the meantime (then they stay displaced). */
void cl_ring_put_cancel(cl_ring *r)
{
    pthread_mutex_lock(&r->mu);
    if (!r->get_open && r->released == r->released_after_drop) r->released = r->released_before_put;
    r->put_open = 0;
    pthread_mutex_unlock(&r->mu);
}

/* Give up an open put whose span may have been written to (a copy was queued before the caller learnt that it has to give
 * up): nothing is published; what a full ring displaced for it is gone, its slots may hold the abandoned data. */
void cl_ring_put_abandon(cl_ring *r)
{
    pthread_mutex_lock(&r->mu);
    r->put_open = 0;
    pthread_mutex_unlock(&r->mu);
}

/* Claim up to `length` of the oldest elements.  whole_requests: wait up to timeout_us until all `length` are
 * held; if they never are, nothing is claimed and what is held stays queued (circular_buffer.h:68-82).  Returns
 * the number claimed and their span (0 = nothing claimed); the elements stay the caller's until cl_ring_get_end. */
size_t cl_ring_get_begin(cl_ring *r, size_t length, int timeout_us, cl_ring_span *sp)
{
    pthread_mutex_lock(&r->mu);
    if (r->whole_requests) {
        struct timespec until;
        clock_gettime(CLOCK_REALTIME, &until);
        until.tv_sec += timeout_us / 1000000;
        until.tv_nsec += (long)(timeout_us % 1000000) * 1000L;
        if (until.tv_nsec >= 1000000000L) { until.tv_sec++; until.tv_nsec -= 1000000000L; }
        int expired = 0;
        while (r->written - r->released < length && !expired)
            expired = pthread_cond_timedwait(&r->grown, &r->mu, &until) == ETIMEDOUT;
        if (r->written - r->released < length) { pthread_mutex_unlock(&r->mu); return 0; }
    }
    const size_t held = (size_t)(r->written - r->released);
    const size_t take = length < held ? length : held;
    if (!take) { pthread_mutex_unlock(&r->mu); return 0; }
    ring_span(r, r->released, take, sp);
    r->get_open = 1;
    pthread_mutex_unlock(&r->mu);
    return take;
}

void cl_ring_get_end(cl_ring *r, size_t claimed)
{
    pthread_mutex_lock(&r->mu);
    r->released += claimed;
    r->get_open = 0;
    pthread_cond_broadcast(&r->get_done);
    pthread_mutex_unlock(&r->mu);
}

/* host data <-> one span of the storage */
static int ring_move(cl_ring *r, const cl_ring_span *sp, uint8_t *host, int to_ring)
{
    size_t done = 0;
    for (int k = 0; k < 2; k++) {
        const size_t nb = sp->len[k] * r->elem;
        if (!nb) continue;
        uint8_t *slot = r->store + sp->pos[k] * r->elem;
        if (!r->on_device) { if (to_ring) memcpy(slot, host + done, nb); else memcpy(host + done, slot, nb); }
        else if (to_ring ? clhip_memcpy_h2d(slot, host + done, nb, r->xfer) : clhip_memcpy_d2h(host + done, slot, nb, r->xfer)) return -1;
        done += nb;
    }
    return r->on_device ? clhip_stream_sync(r->xfer) : 0;
}

size_t cl_ring_put(cl_ring *r, const void *data, size_t length)
{
    cl_ring_span sp;
    const size_t n = cl_ring_put_begin(r, length, &sp);
    const int bad = n && ring_move(r, &sp, (uint8_t *)(uintptr_t)data, 1);
    cl_ring_put_end(r, bad ? 0 : n);
    return bad ? 0 : n;
}

size_t cl_ring_get(cl_ring *r, void *data, size_t length, int timeout_us)
{
    cl_ring_span sp;
    const size_t n = cl_ring_get_begin(r, length, timeout_us, &sp);
    if (!n) return 0;
    const int bad = data && ring_move(r, &sp, (uint8_t *)data, 0);     /* data == NULL: discard (circular_buffer.h:87) */
    cl_ring_get_end(r, n);
    return bad ? 0 : n;
}

void cl_ring_reset(cl_ring *r)
{
    pthread_mutex_lock(&r->mu);
    r->released = r->written;
    pthread_mutex_unlock(&r->mu);
}

size_t cl_ring_size(cl_ring *r)
{
    pthread_mutex_lock(&r->mu);
    const size_t held = (size_t)(r->written - r->released);
    pthread_mutex_unlock(&r->mu);
    return held;
}
