/* cl_ring.c -- the reference's circular_buffer<T> (datatypes/circular_buffer.h:16-164) in C:
 * power-of-two capacity, overwrite-oldest put, blocking get with a timeout that returns 0 unless
 * the whole request is present.  Used by the ASYNC stream mode (the compiled-out USE_ASYNC reader
 * thread of soapy_api/CaribouliteStream.cpp:16-49,70-75) -- SURVEY.md section 8(f) rank 2. */
#include <errno.h>
#include <pthread.h>
#include <time.h>

#include "cl_internal.h"

struct cl_ring {
    uint8_t *buf;
    size_t elem, max_size, head, tail;
    int override_write, block_read;
    pthread_mutex_t mu;
    pthread_cond_t cv;
};

cl_ring *cl_ring_create(size_t size_elems, size_t elem_bytes, int override_write, int block_read)
{
    if (!elem_bytes) return NULL;
    cl_ring *r = (cl_ring *)calloc(1, sizeof *r);
    if (!r) return NULL;
    size_t cap = 1;                                   /* :21-25 next power of two */
    while (cap < size_elems) cap <<= 1;
    r->buf = (uint8_t *)malloc(cap * elem_bytes);
    if (!r->buf) { free(r); return NULL; }
    r->elem = elem_bytes; r->max_size = cap; r->override_write = override_write; r->block_read = block_read;
    pthread_mutex_init(&r->mu, NULL);
    pthread_cond_init(&r->cv, NULL);
    return r;
}

void cl_ring_destroy(cl_ring *r)
{
    if (!r) return;
    pthread_mutex_destroy(&r->mu);
    pthread_cond_destroy(&r->cv);
    free(r->buf);
    free(r);
}

/* :37-62 */
size_t cl_ring_put(cl_ring *r, const void *data, size_t length)
{
    pthread_mutex_lock(&r->mu);
    const size_t sz = r->head - r->tail;
    if ((r->max_size - sz) < length && r->override_write) r->tail += length - (r->max_size - sz);
    size_t len = length < r->max_size - r->head + r->tail ? length : r->max_size - r->head + r->tail;
    const size_t hi = r->head & (r->max_size - 1);
    const size_t l = len < r->max_size - hi ? len : r->max_size - hi;
    memcpy(r->buf + hi * r->elem, data, l * r->elem);
    memcpy(r->buf, (const uint8_t *)data + l * r->elem, (len - l) * r->elem);
    r->head += len;
    if (r->block_read) pthread_cond_signal(&r->cv);
    pthread_mutex_unlock(&r->mu);
    return len;
}

/* :64-93 */
size_t cl_ring_get(cl_ring *r, void *data, size_t length, int timeout_us)
{
    pthread_mutex_lock(&r->mu);
    if (r->block_read) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        ts.tv_sec += timeout_us / 1000000;
        ts.tv_nsec += (long)(timeout_us % 1000000) * 1000L;
        if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
        while (r->head - r->tail < length)
            if (pthread_cond_timedwait(&r->cv, &r->mu, &ts) == ETIMEDOUT) break;
        if (r->head - r->tail < length) { pthread_mutex_unlock(&r->mu); return 0; }
    }
    const size_t sz = r->head - r->tail;
    const size_t len = length < sz ? length : sz;
    const size_t ti = r->tail & (r->max_size - 1);
    const size_t l = len < r->max_size - ti ? len : r->max_size - ti;
    if (data) {
        memcpy(data, r->buf + ti * r->elem, l * r->elem);
        memcpy((uint8_t *)data + l * r->elem, r->buf, (len - l) * r->elem);
    }
    r->tail += len;
    pthread_mutex_unlock(&r->mu);
    return len;
}

void cl_ring_reset(cl_ring *r) { pthread_mutex_lock(&r->mu); r->head = r->tail = 0; pthread_mutex_unlock(&r->mu); }
size_t cl_ring_size(cl_ring *r) { pthread_mutex_lock(&r->mu); size_t s = r->head - r->tail; pthread_mutex_unlock(&r->mu); return s; }
size_t cl_ring_capacity(const cl_ring *r) { return r->max_size; }
