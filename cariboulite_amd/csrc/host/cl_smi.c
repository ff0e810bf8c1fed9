/* cl_smi.c -- the SMI user-driver seam (caribou_smi/caribou_smi.h:85-106) on
 * the GPU: same arguments, chunking, return codes and "untouched slot"
 * behaviour as caribou_smi_read / caribou_smi_write, with the /dev/smi fd
 * replaced by an injected byte FIFO and the per-chunk analysis
 * (caribou_smi.c:235-393) / packing (:684-717) done by HIP kernels. */
#include <sys/time.h>
#include <time.h>

#include "cl_internal.h"

void cl_seterr(char *dst, size_t n, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, n, fmt, ap);
    va_end(ap);
}

/* ------------------------------------------------------------------ FIFO */
void cl_fifo_free(cl_fifo *f)
{
    if (f->external) f->data = NULL;
    if (f->pinned) clhip_host_free(f->data); else free(f->data);
    free(f->front);
    memset(f, 0, sizeof *f);
}

/* make room for n more bytes behind the pending ones; the bytes from `keep` on survive a move */
uint8_t *cl_fifo_reserve(cl_fifo *f, size_t n)
{
    if (f->len == 0 && f->keep == f->head) f->keep = f->head = 0;          /* empty: start over at the front */
    if (f->head + f->len + n > f->cap) {
        const size_t live = f->head - f->keep + f->len;                      /* staged-unconfirmed + pending */
        for (int k = 0; k < CL_FIFO_DMA_STREAMS; k++)                                          /* copies may still be reading this memory in place */
            if (f->dma_stream[k]) clhip_stream_sync(f->dma_stream[k]);
        if (live + n <= f->cap && f->keep) {                                /* compact */
            memmove(f->data, f->data + f->keep, live);
        } else {                                                            /* grow */
            size_t cap = f->cap ? f->cap : (size_t)1 << 20;
            while (cap < live + n) cap *= 2;
            uint8_t *p = f->pinned ? (uint8_t *)clhip_host_alloc(cap) : (uint8_t *)malloc(cap);
            if (!p) return NULL;
            if (live) memcpy(p, f->data + f->keep, live);
            if (f->external) f->external = 0;                                /* (outgrew the slice it was lent: a buffer of its own from here on) */
            else if (f->pinned) clhip_host_free(f->data); else free(f->data);
            f->data = p; f->cap = cap;
        }
        f->head -= f->keep; f->keep = 0;
    }
    return f->data + f->head + f->len;
}

void cl_fifo_commit(cl_fifo *f, size_t n) { f->len += n; }

/* Under the owner's lock, with no reservation open: live bytes (staged-unconfirmed + pending) move to the front of `slice`. */
int cl_fifo_adopt(cl_fifo *f, uint8_t *slice, size_t cap)
{
    const size_t live = f->head - f->keep + f->len;
    if (live > cap || f->reserved) return -1;               /* (a producer is writing into the buffer through a pointer it was handed) */
    for (int k = 0; k < CL_FIFO_DMA_STREAMS; k++)
        if (f->dma_stream[k]) clhip_stream_sync(f->dma_stream[k]);
    if (live) memcpy(slice, f->data + f->keep, live);
    if (!f->external) { if (f->pinned) clhip_host_free(f->data); else free(f->data); }
    f->data = slice; f->cap = cap; f->external = 1;
    f->head -= f->keep; f->keep = 0;
    return 0;
}

int cl_fifo_leave(cl_fifo *f)
{
    if (!f->external) return 0;
    if (f->reserved) return -1;
    const size_t live = f->head - f->keep + f->len;
    size_t cap = (size_t)1 << 20;
    while (cap < live) cap *= 2;
    for (int k = 0; k < CL_FIFO_DMA_STREAMS; k++)
        if (f->dma_stream[k]) clhip_stream_sync(f->dma_stream[k]);
    uint8_t *p = f->pinned ? (uint8_t *)clhip_host_alloc(cap) : (uint8_t *)malloc(cap);
    if (!p) return -1;
    if (live) memcpy(p, f->data + f->keep, live);
    f->data = p; f->cap = cap; f->external = 0;
    f->head -= f->keep; f->keep = 0;
    return 0;
}

int cl_fifo_push(cl_fifo *f, const uint8_t *src, size_t n)
{
    uint8_t *dst = cl_fifo_reserve(f, n);
    if (!dst) return -1;
    memcpy(dst, src, n);
    f->len += n;
    return 0;
}

/* the address kernels use for a byte of a pinned FIFO's buffer (NULL: not pinned, or the device cannot reach it) */
void *cl_fifo_device_ptr(const cl_fifo *f, const uint8_t *at)
{
    if (!f->pinned || !f->data) return NULL;
    uint8_t *d = (uint8_t *)clhip_host_device_ptr(f->data);
    return d ? d + (at - f->data) : NULL;
}

/* largest write call (bytes in) whose kernel reads the pinned samples and stores into the pinned room itself, across PCIe (round 3, three
 * alternations on one box: CS16 55-58 -> 40-43 us per MTU call, CF32 79-82 -> 69-70, FM + 2/3 81-88 -> 70-71); above it: copy engine both ways */
size_t cl_write_mapped_max(void) { return (size_t)2 << 20; }

size_t cl_fifo_pop(cl_fifo *f, uint8_t *dst, size_t n)
{
    size_t done = 0;
    const size_t fl = cl_fifo_front_len(f);
    if (fl) {                                           /* what a reader gave back is older than anything in `data` */
        done = n < fl ? n : fl;
        if (dst) memcpy(dst, f->front + f->front_head, done);
        f->front_head += done;
        if (dst) dst += done;
        n -= done;
    }
    if (n > f->len) n = f->len;
    if (dst && n) memcpy(dst, f->data + f->head, n);
    if (f->keep == f->head) f->keep += n;               /* nothing staged in front: consumed for good */
    f->head += n; f->len -= n;
    return done + n;
}

size_t cl_fifo_stage(cl_fifo *f, size_t n, uint8_t **where)
{
    if (cl_fifo_front_len(f)) { *where = NULL; return 0; }   /* older bytes wait in the front stash: callers take the copying route */
    if (n > f->len) n = f->len;
    *where = f->data + f->head;
    f->head += n; f->len -= n;
    return n;
}

void cl_fifo_confirm(cl_fifo *f, size_t n) { f->keep += n; if (f->keep > f->head) f->keep = f->head; }

void cl_fifo_unstage(cl_fifo *f, size_t n)
{
    if (n > f->head - f->keep) n = f->head - f->keep;
    f->head -= n; f->len += n;
}

/* put bytes back at the FRONT of the pending ones (a batched call read them with a copy and did not consume them).
 * `data` is never moved or reallocated here: a producer may be writing into a reservation it holds (read(fd) straight
 * into the pinned buffer, outside the lock), and in-place copies may be reading it.  Bytes that do not fit in front of
 * `head` go to the front stash. */
int cl_fifo_unpop(cl_fifo *f, const uint8_t *src, size_t n)
{
    if (n == 0) return 0;
    if (!cl_fifo_front_len(f) && f->keep == f->head && f->head >= n) {   /* room in front of `head`, nothing staged there, no older bytes */
        f->head -= n; f->keep = f->head; f->len += n;
        memcpy(f->data + f->head, src, n);
        return 0;
    }
    if (f->keep != f->head) return -1;                  /* staged bytes are older than what is given back: callers confirm or unstage first */
    if (f->front_head < n) {                            /* grow the stash at its front */
        const size_t live = cl_fifo_front_len(f);
        size_t cap = f->front_cap ? f->front_cap : (size_t)1 << 16;
        while (cap < 2 * (live + n)) cap *= 2;
        uint8_t *p = (uint8_t *)malloc(cap);
        if (!p) return -1;
        if (live) memcpy(p + cap - live, f->front + f->front_head, live);
        free(f->front);
        f->front = p; f->front_cap = cap; f->front_head = cap - live;
    }
    f->front_head -= n;
    memcpy(f->front + f->front_head, src, n);
    return 0;
}

/* ------------------------------------------------------------ allocation */
int cl_ensure(void **p, size_t *cap, size_t need, size_t elem, int pinned)
{
    if (need <= *cap) return 0;
    size_t ncap = *cap ? *cap : 1024;
    while (ncap < need) ncap *= 2;
    void *np = pinned == 1 ? clhip_host_alloc(ncap * elem) : pinned == 2 ? malloc(ncap * elem) : clhip_malloc(ncap * elem);
    if (!np) return -1;
    if (*p) { if (pinned == 1) clhip_host_free(*p); else if (pinned == 2) free(*p); else clhip_free(*p); }
    *p = np; *cap = ncap;
    return 0;
}

/* The seam's persistent int16 buffer (it stands where the Stream's interm_native_buffer stands: an overlay of every call so far --
 * a re-synchronised read() leaves slots untouched, caribou_smi.c:382-389): starts as zeros, and growing keeps what it holds. */
int cl_smi_ensure_iq(cl_smi *dev, size_t samples)
{
    if (samples <= dev->iq_cap) return 0;
    size_t ncap = dev->iq_cap ? dev->iq_cap : CL_NATIVE_MTU_SAMPLES + 8;
    while (ncap < samples) ncap *= 2;
    int16_t *np = (int16_t *)clhip_malloc(ncap * 4);
    if (!np) return -1;
    int bad = clhip_memset(np, 0, ncap * 4, dev->stream);
    if (!bad && dev->d_iq) bad = clhip_memcpy_d2d(np, dev->d_iq, dev->iq_cap * 4, dev->stream);
    if (clhip_stream_sync(dev->stream) || bad) { clhip_free(np); return -1; }
    clhip_free(dev->d_iq);
    dev->d_iq = np; dev->iq_cap = ncap;
    return 0;
}

/* ----------------------------------------------------------- init / close */
cl_smi *cl_smi_init(int device)
{
    if (clhip_device_count() <= 0 || clhip_set_device(device) != 0) return NULL;   /* no GPU: fail loudly */
    cl_smi *dev = (cl_smi *)calloc(1, sizeof *dev);
    if (!dev) return NULL;
    dev->device = device;
    dev->native_batch_len = CL_NATIVE_BATCH_LEN;       /* caribou_smi.c:78 */
    dev->sample_rate = CL_SAMPLE_RATE;
    dev->tx_mode = CL_TX_DOCUMENTED;
    dev->stream = clhip_stream_create();
    if (!dev->stream) { free(dev); return NULL; }
    pthread_mutex_init(&dev->fifo_mu, NULL);
    pthread_cond_init(&dev->fifo_fed, NULL);
    dev->rx.pinned = 1;                                /* the feeder writes where the DMA engine reads */
    dev->tx.pinned = 1;                                /* ... and the TX side drains what the DMA engine wrote */
    return dev;
}

int cl_smi_close(cl_smi *dev)
{
    if (!dev) return -1;
    clhip_set_device(dev->device);
    clhip_stream_sync(dev->stream);
    cl_smi_readahead_cancel(dev);
    if (dev->cstream) { clhip_stream_sync(dev->cstream); clhip_stream_destroy(dev->cstream); }
    for (int k = 0; k < CL_RA_SLOTS; k++) { clhip_event_destroy(dev->ev_copied[k]); clhip_free(dev->d_slot[k]); }
    clhip_free(dev->d_bytes); clhip_free(dev->d_iq); clhip_free(dev->d_meta); clhip_free(dev->d_offs);
    clhip_host_free(dev->h_stage); clhip_host_free(dev->h_txin); clhip_host_free(dev->h_offs); clhip_free(dev->d_dbg); clhip_host_free(dev->h_dbg);
    free(dev->chunks);
    cl_fifo_free(&dev->rx); cl_fifo_free(&dev->tx);
    clhip_stream_destroy(dev->stream);
    free(dev);
    return 0;
}

/* Zero-copy feed: a pointer into the pinned RX FIFO with room for n bytes -- read(fd, p, n) straight into it -- and the
 * commit of what actually arrived.  One producer at a time; the pointer is valid until the commit. */
uint8_t *cl_smi_feed_reserve(cl_smi *dev, size_t n)
{
    clhip_set_device(dev->device);
    pthread_mutex_lock(&dev->fifo_mu);
    uint8_t *p = cl_fifo_reserve(&dev->rx, n);
    dev->rx.reserved = p != NULL;
    pthread_mutex_unlock(&dev->fifo_mu);
    return p;
}

int cl_smi_feed_commit(cl_smi *dev, size_t n)
{
    pthread_mutex_lock(&dev->fifo_mu);
    cl_fifo_commit(&dev->rx, n);
    dev->rx.reserved = 0;
    pthread_cond_broadcast(&dev->fifo_fed);
    pthread_mutex_unlock(&dev->fifo_mu);
    return 0;
}

int cl_smi_feed_bytes(cl_smi *dev, const uint8_t *b, size_t n)
{
    clhip_set_device(dev->device);
    pthread_mutex_lock(&dev->fifo_mu);
    int rc = cl_fifo_push(&dev->rx, b, n);
    pthread_cond_broadcast(&dev->fifo_fed);
    pthread_mutex_unlock(&dev->fifo_mu);
    return rc;
}

/* caribou_smi_timeout_read's poll(POLLIN, timeout) (caribou_smi.c:466-492) on the injected stream */
int cl_smi_wait_bytes(cl_smi *dev, long timeout_us)
{
    struct timespec until;
    clock_gettime(CLOCK_REALTIME, &until);
    until.tv_sec += timeout_us / 1000000;
    until.tv_nsec += (timeout_us % 1000000) * 1000L;
    if (until.tv_nsec >= 1000000000L) { until.tv_sec++; until.tv_nsec -= 1000000000L; }
    pthread_mutex_lock(&dev->fifo_mu);
    int expired = 0;
    while (cl_fifo_pending(&dev->rx) == 0 && !__atomic_load_n(&dev->ahead_bytes, __ATOMIC_RELAXED) && !expired)
        expired = pthread_cond_timedwait(&dev->fifo_fed, &dev->fifo_mu, &until) != 0;
    const int ready = cl_fifo_pending(&dev->rx) != 0 || __atomic_load_n(&dev->ahead_bytes, __ATOMIC_RELAXED);
    pthread_mutex_unlock(&dev->fifo_mu);
    return ready;
}

/* Any thread may ask (a feeder pacing itself, a monitor): the FIFO's counters are read under its lock, what the consumer has staged
 * ahead of its client -- still pending as far as anybody outside can tell -- from a word the consumer publishes. */
size_t cl_smi_pending_bytes(const cl_smi *dev)
{
    cl_smi *d = (cl_smi *)dev;
    pthread_mutex_lock(&d->fifo_mu);
    const size_t n = cl_fifo_pending(&d->rx);
    pthread_mutex_unlock(&d->fifo_mu);
    return n + __atomic_load_n(&d->ahead_bytes, __ATOMIC_RELAXED);
}

/* the consumer publishes what it holds staged ahead (its own read-ahead + a stream group's) after every change */
void cl_smi_ahead_note(cl_smi *dev)
{
    __atomic_store_n(&dev->ahead_bytes, (dev->ahead.valid ? dev->ahead.len : 0) + dev->foreign_ahead, __ATOMIC_RELAXED);
}
void   cl_smi_set_max_read(cl_smi *dev, size_t m) { dev->max_read = m; }
/* The TX FIFO has one producer (the write calls: reserve, fill by DMA or kernel, commit) and one consumer (the drain calls),
 * which may be two threads: its bookkeeping moves under fifo_mu; the bytes of an open reservation lie behind everything a pop
 * can touch, and only the producer's reserve ever moves the buffer. */
uint8_t *cl_smi_tx_reserve_raw(cl_smi *dev, size_t n)
{
    pthread_mutex_lock(&dev->fifo_mu);
    uint8_t *room = cl_fifo_reserve(&dev->tx, n);
    pthread_mutex_unlock(&dev->fifo_mu);
    return room;
}
/* (a stream group may have words of this seam in flight -- launched over, not committed: it lands them before anybody else looks at
 * or adds to the TX FIFO) */
static void tx_settle(cl_smi *dev)
{
    void (*fn)(void *, int) = dev->tx_settle;
    if (fn) fn(dev->tx_settle_ctx, dev->tx_settle_member);
}
uint8_t *cl_smi_tx_reserve(cl_smi *dev, size_t n)
{
    tx_settle(dev);
    pthread_mutex_lock(&dev->fifo_mu);
    uint8_t *room = cl_fifo_reserve(&dev->tx, n);
    pthread_mutex_unlock(&dev->fifo_mu);
    return room;
}
void cl_smi_tx_commit(cl_smi *dev, size_t n)
{
    pthread_mutex_lock(&dev->fifo_mu);
    cl_fifo_commit(&dev->tx, n);
    pthread_mutex_unlock(&dev->fifo_mu);
}
static size_t tx_pop(cl_smi *dev, uint8_t *b, size_t max)
{
    tx_settle(dev);
    pthread_mutex_lock(&dev->fifo_mu);
    const size_t n = cl_fifo_pop(&dev->tx, b, max);
    pthread_mutex_unlock(&dev->fifo_mu);
    return n;
}
size_t cl_smi_drain_bytes(cl_smi *dev, uint8_t *b, size_t max) { return tx_pop(dev, b, max); }
void   cl_smi_set_tx_mode(cl_smi *dev, int mode) { dev->tx_mode = mode; }
size_t cl_smi_get_native_batch_samples(cl_smi *dev) { return dev->native_batch_len / CL_BYTES_PER_SAMPLE; }
void   cl_smi_set_debug_mode(cl_smi *dev, int mode) { dev->debug_mode = mode; }       /* caribou_smi.c:612-615 */
const cl_smi_debug_data *cl_smi_get_debug_data(const cl_smi *dev) { return &dev->debug_data; }
void   cl_smi_set_debug_clock(cl_smi *dev, cl_smi_clock_fn now, void *user) { if (dev) { dev->debug_clock = now; dev->debug_clock_user = user; } }

/* smi_calculate_performance (smi_utils.c:233-244), called once per analysed chunk (caribou_smi.c:210): the reference forms
 * the elapsed time in SECONDS (it calls it elapsed_us), so bytes * 8 / elapsed / 1e6 is Mbit/s; 0.98 : 0.02 blend */
static void smi_debug_bitrate(cl_smi *dev, size_t bytes)
{
    cl_smi_debug_data *d = &dev->debug_data;
    long sec, usec;
    if (dev->debug_clock) dev->debug_clock(dev->debug_clock_user, &sec, &usec);
    else { struct timeval t; gettimeofday(&t, NULL); sec = (long)t.tv_sec; usec = (long)t.tv_usec; }
    const double elapsed_us = (sec - d->last_time_sec) + ((double)(usec - d->last_time_usec)) / 1000000.0;
    const double speed_mbps = (double)(bytes * 8) / elapsed_us / 1e6;
    d->last_time_sec = sec; d->last_time_usec = usec;
    d->bitrate = d->bitrate * 0.98 + speed_mbps * 0.02;
}
void cl_smi_get_stats(const cl_smi *dev, cl_smi_stats *out)
{
    if (!out) return;
    memset(out, 0, sizeof *out);
    if (!dev) return;
    out->samples_read = dev->stat_samples; out->resyncs = dev->stat_resyncs; out->sync_losses = dev->stat_sync_failures;
    out->timeouts = dev->stat_timeouts; out->io_errors = dev->stat_io_errors; out->samples_written = dev->stat_written;
}
/* every read call's return passes through here on its way out */
static int smi_count(cl_smi *dev, int ret)
{
    if (ret == 0) dev->stat_timeouts++;
    else if (ret == CL_SMI_ERR_IO) dev->stat_io_errors++;
    return ret;
}

/* debug modes: one chunk is read and analysed, then the call returns -2 (caribou_smi.c:650-675) */
static int cl_smi_read_debug(cl_smi *dev, size_t length_samples)
{
    clhip_set_device(dev->device);
    cl_smi_readahead_cancel(dev);
    size_t left = length_samples * CL_BYTES_PER_SAMPLE;
    if (!left) return 0;
    size_t cur = left > dev->native_batch_len ? dev->native_batch_len : left;
    if (dev->max_read && cur > dev->max_read) cur = dev->max_read;
    if (cl_ensure((void **)&dev->h_stage, &dev->h_stage_cap, cur + 256, 1, 1) ||
        cl_ensure((void **)&dev->d_bytes, &dev->bytes_cap, cur + 256, 1, 0))
        return CL_SMI_ERR_IO;
    if (!dev->d_dbg) { dev->d_dbg = (int32_t *)clhip_malloc(16); dev->h_dbg = (int32_t *)clhip_host_alloc(16); }
    if (!dev->d_dbg || !dev->h_dbg) return CL_SMI_ERR_IO;
    pthread_mutex_lock(&dev->fifo_mu);
    const size_t ret = cl_fifo_pop(&dev->rx, dev->h_stage, cur);
    pthread_mutex_unlock(&dev->fifo_mu);
    if (ret == 0) return 0;                                   /* "Reading timed-out" */
    if (clhip_memcpy_h2d(dev->d_bytes, dev->h_stage, ret, dev->stream) ||
        clhip_smi_debug_analyze(dev->debug_mode, dev->d_bytes, ret, dev->debug_data.last_correct_byte, dev->d_dbg, dev->stream) ||
        clhip_memcpy_d2h(dev->h_dbg, dev->d_dbg, 16, dev->stream) || clhip_stream_sync(dev->stream))
        return CL_SMI_ERR_IO;
    const int offs = dev->h_dbg[0];
    if (offs < 0) { dev->stat_sync_failures++; return CL_SMI_ERR_SYNC; }            /* :665-668 */
    const size_t shortening = offs > 0 ? (size_t)(offs / 4 + 1) : 0;
    const size_t alen = ret - 4 * shortening;
    cl_smi_debug_data *d = &dev->debug_data;                  /* caribou_smi.c:188-214 */
    d->cur_err_cnt = (uint32_t)dev->h_dbg[1];
    d->error_accum_counter += d->cur_err_cnt;
    d->last_correct_byte = (uint8_t)dev->h_dbg[3];
    smi_debug_bitrate(dev, alen);                             /* :210, before the error rate like the reference */
    d->error_rate = d->error_rate * 0.9 + (double)d->cur_err_cnt / (double)alen * 0.1;
    if (d->error_rate < 1e-8) d->error_rate = 0.0;
    return CL_SMI_ERR_DEBUGMODE;
}

/* ------------------------------------------------- file / wire replay front-end */
/* a /dev/smi-shaped byte source.  Bytes are taken from any fd (regular
 * file with a recorded capture, pipe, socket) with the reference's own read pattern -- read() of at most
 * one native batch at a time (caribou_smi.c:466-492), short reads and lengths that are not a multiple of
 * 4 included -- and queued exactly as the kernel kfifo would hand them on.  Returns bytes queued
 * (0 at EOF / EAGAIN), -1 on a read error. */
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

long cl_smi_feed_fd(cl_smi *dev, int fd, size_t max_bytes)
{
    if (!dev || fd < 0) return -1;
    size_t total = 0;
    while (total < max_bytes) {
        const size_t want = max_bytes - total < dev->native_batch_len ? max_bytes - total : dev->native_batch_len;
        uint8_t *slot = cl_smi_feed_reserve(dev, want);        /* read() lands in pinned memory the DMA engine reads from */
        if (!slot) return -1;
        ssize_t r = read(fd, slot, want);
        if (r <= 0) {
            const int e = errno;
            cl_smi_feed_commit(dev, 0);                        /* (the reservation ends whatever read() said) */
            if (r == 0) break;
            if (e == EINTR) continue;
            if (e == EAGAIN || e == EWOULDBLOCK) break;
            return -1;
        }
        cl_smi_feed_commit(dev, (size_t)r);
        total += (size_t)r;
    }
    return (long)total;
}

long cl_smi_feed_file(cl_smi *dev, const char *path, size_t offset_bytes, size_t max_bytes)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    if (offset_bytes && lseek(fd, (off_t)offset_bytes, SEEK_SET) < 0) { close(fd); return -1; }
    long n = cl_smi_feed_fd(dev, fd, max_bytes);
    close(fd);
    return n;
}

/* the TX mirror: hand the packed bytes to an fd with write() of at most one native batch
 * (caribou_smi.c:444-463,738-759) */
long cl_smi_drain_to_fd(cl_smi *dev, int fd, size_t max_bytes)
{
    if (!dev || fd < 0) return -1;
    uint8_t *tmp = (uint8_t *)malloc(dev->native_batch_len);
    if (!tmp) return -1;
    size_t total = 0;
    while (total < max_bytes) {
        size_t want = max_bytes - total < dev->native_batch_len ? max_bytes - total : dev->native_batch_len;
        size_t got = tx_pop(dev, tmp, want);
        if (!got) break;
        size_t off = 0;
        while (off < got) {
            ssize_t w = write(fd, tmp + off, got - off);
            if (w < 0) { if (errno == EINTR) continue; free(tmp); return -1; }
            off += (size_t)w;
        }
        total += got;
    }
    free(tmp);
    return (long)total;
}

/* --------------------------------------------------------------- RX path */
int cl_smi_read_device(cl_smi *dev, int channel, size_t length_samples, int want_meta, int *all_aligned)
{
    if (cl_smi_ensure_iq(dev, length_samples + 8) ||
        (want_meta && cl_ensure((void **)&dev->d_meta, &dev->meta_cap, length_samples + 8, 1, 0)))
        return CL_SMI_ERR_IO;
    return cl_smi_read_device_to(dev, channel, length_samples, dev->d_iq, want_meta ? dev->d_meta : NULL, all_aligned);
}

/* The reference's chunk loop (caribou_smi.c:643-679; read() = FIFO pop) up to the point where bytes are analysed:
 * every read() of the call is popped into the pinned staging buffer and the lot goes to the device in one copy.
 * Fills dev->chunks; *contiguous = the staged bytes are the call's word sequence back to back (whole samples, no
 * staging gaps).  Returns read_so_far (0: nothing pending) or CL_SMI_ERR_IO. */
static long smi_stage_call(cl_smi *dev, size_t length_samples, int *contiguous)
{
    size_t left = length_samples * CL_BYTES_PER_SAMPLE, read_so_far = 0, stage_off = 0;
    dev->n_chunks = 0;
    *contiguous = 1;
    /* worst-case staging: every chunk rounded up to 256 B */
    const size_t max_chunks = left / (dev->max_read && dev->max_read < dev->native_batch_len ? dev->max_read : dev->native_batch_len) + 2;
    if (cl_ensure((void **)&dev->chunks, &dev->chunks_cap, max_chunks, sizeof(cl_chunk), 2)) return CL_SMI_ERR_IO;
    const size_t need_bytes = left + 256 * max_chunks + 256;
    if (cl_ensure((void **)&dev->h_stage, &dev->h_stage_cap, need_bytes, 1, 1) ||
        cl_ensure((void **)&dev->d_bytes, &dev->bytes_cap, need_bytes, 1, 0) ||
        cl_ensure((void **)&dev->d_offs, &dev->offs_cap, max_chunks, 4, 0) ||
        cl_ensure((void **)&dev->h_offs, &dev->h_offs_cap, max_chunks, 4, 1))
        return CL_SMI_ERR_IO;
    if (left <= dev->native_batch_len && !(dev->max_read && dev->max_read < left)) {
        /* the call is ONE read(): its bytes go to the device straight from the pinned FIFO (no staging copy).  They are
         * consumed whatever the analysis says (a failed chunk is consumed too, :665-668) -- but only once the copy has
         * RUN: until the caller has synchronised dev->stream they stay staged (smi_inplace_done), so that a feeder
         * cannot be handed their memory (an empty FIFO restarts at the front of its buffer).  The copy is enqueued
         * under the FIFO lock: a feeder that must move the buffer waits for this stream first. */
        uint8_t *src = NULL;
        int bad = 0;
        pthread_mutex_lock(&dev->fifo_mu);
        dev->rx.dma_stream[1] = dev->stream;
        size_t got = cl_fifo_stage(&dev->rx, left, &src);
        if (got & 3) { cl_fifo_unstage(&dev->rx, got); got = 0; bad = 2; }          /* ragged: the copying loop below */
        else if (got) {
            bad = clhip_memcpy_h2d(dev->d_bytes, src, got, dev->stream);
            if (bad) cl_fifo_unstage(&dev->rx, got); else dev->inplace_len = got;
        }
        else if (cl_fifo_front_len(&dev->rx)) bad = 2;                               /* given-back bytes first: the copying loop below */
        pthread_mutex_unlock(&dev->fifo_mu);
        if (bad == 1) return CL_SMI_ERR_IO;
        if (bad == 0) {
            if (got) {
                cl_chunk *c = &dev->chunks[dev->n_chunks++];
                c->stage_off = 0; c->len = got; c->slot0 = 0; c->offs = 0;
            }
            return (long)(got / CL_BYTES_PER_SAMPLE);
        }
    }
    while (left) {
        size_t want = left > dev->native_batch_len ? dev->native_batch_len : left;
        if (dev->max_read && want > dev->max_read) want = dev->max_read;
        pthread_mutex_lock(&dev->fifo_mu);
        size_t ret = cl_fifo_pop(&dev->rx, dev->h_stage + stage_off, want);
        pthread_mutex_unlock(&dev->fifo_mu);
        if (ret == 0) break;                                   /* :657-661 "Reading timed-out" */
        cl_chunk *c = &dev->chunks[dev->n_chunks++];
        c->stage_off = stage_off; c->len = ret; c->slot0 = read_so_far; c->offs = 0;
        if ((ret & 3) || stage_off != 4 * read_so_far || (left > ret && ret != dev->native_batch_len)) *contiguous = 0;
        stage_off += (ret + 255) & ~(size_t)255;
        read_so_far += ret / CL_BYTES_PER_SAMPLE;              /* :677 */
        left -= ret;                                           /* :678 */
    }
    if (dev->n_chunks && clhip_memcpy_h2d(dev->d_bytes, dev->h_stage, stage_off, dev->stream)) return CL_SMI_ERR_IO;
    return (long)read_so_far;
}

/* dev->stream has been synchronised (or is, here): the bytes a one-read() call staged in place are consumed for good */
static void smi_inplace_done(cl_smi *dev, int synced)
{
    if (!dev->inplace_len) return;
    if (!synced) clhip_stream_sync(dev->stream);
    pthread_mutex_lock(&dev->fifo_mu);
    cl_fifo_confirm(&dev->rx, dev->inplace_len);
    pthread_mutex_unlock(&dev->fifo_mu);
    dev->inplace_len = 0;
}

/* The sync-search results of a staged call are on the host (dev->h_offs): statistics, and the reference's exit
 * at the first chunk without sync (:665-668 -> -3) -- what was read after that chunk goes back to the FIFO. */
static int smi_call_verdict(cl_smi *dev, int *all_aligned)
{
    for (size_t i = 0; i < dev->n_chunks; i++) {
        dev->chunks[i].offs = dev->h_offs[i];
        /* "aligned" = the staged bytes ARE the sample sequence: in sync, whole samples, no staging gaps */
        if (dev->h_offs[i] != 0 || (dev->chunks[i].len & 3) || dev->chunks[i].stage_off != 4 * dev->chunks[i].slot0) { if (all_aligned) *all_aligned = 0; }
        if (dev->h_offs[i] > 0) dev->stat_resyncs++;
        if (dev->h_offs[i] < 0) {
            dev->stat_sync_failures++;
            pthread_mutex_lock(&dev->fifo_mu);
            for (size_t k = dev->n_chunks; k-- > i + 1;)
                if (cl_fifo_unpop(&dev->rx, dev->h_stage + dev->chunks[k].stage_off, dev->chunks[k].len)) { pthread_mutex_unlock(&dev->fifo_mu); return CL_SMI_ERR_IO; }
            pthread_mutex_unlock(&dev->fifo_mu);
            dev->n_chunks = i;                                  /* chunks before the failure were delivered */
            return CL_SMI_ERR_SYNC;
        }
    }
    return 0;
}

int cl_smi_read_device_to(cl_smi *dev, int channel, size_t length_samples, int16_t *d_iq, uint8_t *d_meta, int *all_aligned)
{
    clhip_set_device(dev->device);
    cl_smi_readahead_cancel(dev);
    if (d_iq == dev->d_iq && cl_smi_restore_prev_words(dev, channel)) return CL_SMI_ERR_IO;   /* the seam's int16 buffer is brought up to date first */
    if (all_aligned) *all_aligned = 1;
    int contiguous;
    const long read_so_far = smi_stage_call(dev, length_samples, &contiguous);
    if (read_so_far < 0) return (int)read_so_far;
    if (dev->n_chunks == 0) return 0;

    /* runs of full native chunks go out as ONE batched launch each */
    size_t i = 0;
    while (i < dev->n_chunks) {
        size_t j = i + 1;
        if (dev->chunks[i].len == dev->native_batch_len)
            while (j < dev->n_chunks && dev->chunks[j].len == dev->native_batch_len) j++;
        const cl_chunk *c = &dev->chunks[i];
        const size_t stride = dev->native_batch_len, total = (j - i - 1) * stride + dev->chunks[j - 1].len;
        if (clhip_smi_find_offsets(dev->d_bytes + c->stage_off, total, stride, stride, (int)(j - i), dev->d_offs + i, dev->stream) ||
            clhip_smi_unpack(channel, dev->d_bytes + c->stage_off, total, stride, stride, (int)(j - i), dev->d_offs + i,
                             CL_FORMAT_CS16, d_iq + 2 * c->slot0, d_meta ? d_meta + c->slot0 : NULL, dev->stream)) {
            smi_inplace_done(dev, 0);
            return CL_SMI_ERR_IO;
        }
        i = j;
    }
    const int bad_sync = clhip_memcpy_d2h(dev->h_offs, dev->d_offs, dev->n_chunks * 4, dev->stream) || clhip_stream_sync(dev->stream);
    smi_inplace_done(dev, !bad_sync);
    if (bad_sync) return CL_SMI_ERR_IO;
    const int v = smi_call_verdict(dev, all_aligned);
    if (v) return v;
    dev->stat_samples += (uint64_t)read_so_far;
    return (int)read_so_far;
}

/* caribou_smi_find_buffer_offset (caribou_smi.c:235-292) returns 0 exactly when the chunk is at most 16 bytes long or its
 * words at byte offsets 0, 4, 8, 12 all carry the sync pattern (0 is the smallest candidate offset).  The staged bytes
 * are in pinned host memory, so the host knows BEFORE the device has looked whether a chunk will come back in sync --
 * and may then let results flow straight into the client's buffer, every slot of which such a chunk writes. */
int cl_smi_head_in_sync(const uint8_t *chunk, size_t len)
{
    if (len <= 16) return 1;
    for (int k = 0; k < 4; k++) {
        uint32_t w;
        memcpy(&w, chunk + 4 * k, 4);
        if ((w & 0xC001C000u) != 0x80004000u) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------- read-ahead reader */
/* Take the next read() IN PLACE from the pinned FIFO and start its host-to-device copy on the copy stream (enqueued
 * under the FIFO lock: a feeder that has to move the buffer waits for that stream first).  *in_sync = the host's
 * verdict on the chunk head (cl_smi_head_in_sync), taken while the bytes cannot move. */
static size_t ra_stage(cl_smi *dev, int slot, size_t want, int *in_sync)
{
    uint8_t *src = NULL;
    pthread_mutex_lock(&dev->fifo_mu);
    dev->rx.dma_stream[0] = dev->cstream;
    const size_t got = cl_fifo_stage(&dev->rx, want, &src);
    int bad = 0;
    if (got) {
        *in_sync = !(got & 3) && cl_smi_head_in_sync(src, got);
        bad = clhip_memcpy_h2d(dev->d_slot[slot], src, got, dev->cstream) || clhip_event_record(dev->ev_copied[slot], dev->cstream);
        if (bad) cl_fifo_unstage(&dev->rx, got);
    }
    pthread_mutex_unlock(&dev->fifo_mu);
    return bad ? 0 : got;
}

void cl_smi_foreign_cancel(cl_smi *dev)
{
    if (!dev->foreign_ahead) return;
    /* (the group's copy of these bytes may still be reading them: once they are pending again somebody else consumes them and a feeder
     * may write over them -- the stale copy is waited for first; giving up a read-ahead is rare) */
    if (dev->rx.dma_stream[2]) clhip_stream_sync(dev->rx.dma_stream[2]);
    pthread_mutex_lock(&dev->fifo_mu);
    cl_fifo_unstage(&dev->rx, dev->foreign_ahead);             /* the newest staged bytes: still in place, pending again */
    pthread_mutex_unlock(&dev->fifo_mu);
    dev->foreign_ahead = 0;
    dev->foreign_epoch++;
    cl_smi_ahead_note(dev);
}

void cl_smi_readahead_cancel(cl_smi *dev)
{
    cl_smi_foreign_cancel(dev);
    if (!dev->ahead.valid) return;
    if (dev->cstream) clhip_stream_sync(dev->cstream);          /* (the copy that is reading them, as above) */
    pthread_mutex_lock(&dev->fifo_mu);
    cl_fifo_unstage(&dev->rx, dev->ahead.len);                 /* still in place: pending again */
    pthread_mutex_unlock(&dev->fifo_mu);
    dev->ahead.valid = 0;
    cl_smi_ahead_note(dev);
}

/* caribou_smi_read's chunk loop (caribou_smi.c:643-679) for a reader thread: chunk k is analysed on the seam's
 * stream while the bytes of read() k+1 -- also the first read() of the NEXT call -- are already popped into the
 * other pinned slot and on their way to the device on a second HIP stream.  Same chunks, slots and return codes
 * as cl_smi_read_device; a read() staged ahead that the loop turns out not to want goes back to the FIFO.
 *
 * Two halves, so that the caller can queue its own work on the seam's stream (the ring's device-to-device copy,
 * a conversion) behind the analysis and pay for ONE synchronisation per call:
 *   cl_smi_ra_launch  runs the loop; the analysis of the call's LAST read() is launched but not waited for.
 *                     Returns the samples the call yields if that last chunk is in sync (0: nothing pending).
 *   cl_smi_ra_finish  synchronises, applies the last chunk's verdict, returns what caribou_smi_read returns. */
static int ra_chunk_verdict(cl_smi *dev)
{
    const int32_t offs = dev->h_offs[0];
    dev->chunks[dev->n_chunks].offs = offs;
    pthread_mutex_lock(&dev->fifo_mu);                         /* this read() is consumed for good, whatever it held */
    cl_fifo_confirm(&dev->rx, dev->chunks[dev->n_chunks].len);
    pthread_mutex_unlock(&dev->fifo_mu);
    if (offs >= 0) dev->n_chunks++;                            /* chunks before a failure were delivered */
    if (offs > 0) dev->stat_resyncs++;
    if (offs < 0) {                                            /* :665-668 -> -3; nothing after this read() is consumed */
        dev->stat_sync_failures++;
        cl_smi_readahead_cancel(dev);
        return CL_SMI_ERR_SYNC;
    }
    return 0;
}

/* The seam's persistent int16 buffer (dev->d_iq: it stands where the Stream's interm_native_buffer stands, so the slots a
 * re-synchronised chunk leaves untouched keep what the call before left there, caribou_smi.c:382-389) is not written by a
 * call whose raw words went straight into the caller's own kernel (cl_smi_ra_launch with want_words: the IIR's input
 * conversion, the fused pipe).  Such a call leaves its words where they are -- the read-ahead rotates THREE device slots, so
 * the slot of the previous call survives the next one's staging -- and the first call that needs the int16 samples (any call
 * that is not one in-sync read()) unpacks them first.  Costs nothing until then. */
int cl_smi_restore_prev_words(cl_smi *dev, int channel)
{
    if (!dev->prev_words) return 0;
    const uint8_t *w = dev->prev_words;
    const size_t n = dev->prev_words_len;
    const int cs16 = dev->prev_is_cs16;
    dev->prev_words = NULL; dev->prev_is_cs16 = 0;
    if (cl_smi_ensure_iq(dev, n / 4 + 8)) return -1;
    /* (synchronised: the copy stream may stage into that slot as soon as this call moves on) */
    if (cs16) return clhip_memcpy_d2d(dev->d_iq, w, n, dev->stream) || clhip_stream_sync(dev->stream) ? -1 : 0;
    return clhip_smi_unpack_aligned(channel, w, n, CL_FORMAT_CS16, dev->d_iq, NULL, dev->stream) || clhip_stream_sync(dev->stream) ? -1 : 0;
}

/* `w` (len bytes of raw words, in sync, on the device) stand in for the persistent buffer from now on.  The buffer is an OVERLAY of
 * every call so far -- a call writes the slots of its own length, the slots behind them keep what longer calls left there -- so words
 * that are shorter than the ones they replace may only do so once those have been unpacked. */
int cl_smi_set_prev_words(cl_smi *dev, int channel, const uint8_t *w, size_t len)
{
    if (dev->prev_words && dev->prev_words_len > len && cl_smi_restore_prev_words(dev, channel)) return -1;
    dev->prev_words = w; dev->prev_words_len = len; dev->prev_is_cs16 = 0;
    return 0;
}

/* The same with int16 SAMPLES (n of them, device-readable): what a call that ran the low-pass delivered.  The reference filters in
 * place in the buffer it read into (CaribouliteStream.cpp:282-301: interm_native_buffer2, or the client's own buffer for CS16), so
 * the slots a later re-synchronised read() leaves untouched hold FILTERED samples there -- and go through the filter again. */
int cl_smi_set_prev_cs16(cl_smi *dev, int channel, const int16_t *p, size_t n_samples)
{
    if (dev->prev_words && dev->prev_words_len > 4 * n_samples && cl_smi_restore_prev_words(dev, channel)) return -1;
    dev->prev_words = (const uint8_t *)p; dev->prev_words_len = 4 * n_samples; dev->prev_is_cs16 = 1;
    return 0;
}

/* A runtime error inside the loop, behind a staged read(): everything queued is waited for, the read() in hand (`got` bytes, the
 * oldest staged ones) counts as consumed -- like a read() whose analysis failed, caribou_smi.c:665-668 -- and what was staged ahead
 * of it is pending again: the FIFO is never left with bytes that are neither consumed nor pending. */
static long ra_fail(cl_smi *dev, size_t got)
{
    clhip_stream_sync(dev->stream);
    if (dev->cstream) clhip_stream_sync(dev->cstream);
    cl_smi_readahead_cancel(dev);
    pthread_mutex_lock(&dev->fifo_mu);
    cl_fifo_confirm(&dev->rx, got);
    pthread_mutex_unlock(&dev->fifo_mu);
    dev->ra_pending = 0;
    return smi_count(dev, CL_SMI_ERR_IO);
}

long cl_smi_ra_launch(cl_smi *dev, int channel, size_t length_samples, int16_t *d_iq)
{
    clhip_set_device(dev->device);
    const size_t nb = dev->native_batch_len;
    cl_smi_foreign_cancel(dev);                                /* (a group's read-ahead: this reader comes first) */
    const int want_words = dev->want_words;
    dev->want_words = 0;
    dev->ra_pending = 0;
    dev->fast_used = 0;
    if (!dev->cstream) {
        dev->cstream = clhip_stream_create();
        int bad = !dev->cstream;
        for (int k = 0; k < CL_RA_SLOTS; k++) {
            dev->ev_copied[k] = clhip_event_create();
            dev->d_slot[k] = (uint8_t *)clhip_malloc(nb + 256);
            bad |= !dev->ev_copied[k] || !dev->d_slot[k];
        }
        dev->slot_cap = nb + 256;
        if (bad) return CL_SMI_ERR_IO;
    }
    const int own = !d_iq;                                     /* results in the seam's persistent int16 buffer */
    if (own) {
        if (cl_smi_ensure_iq(dev, length_samples + 8)) return CL_SMI_ERR_IO;
        d_iq = dev->d_iq;
    }
    if (cl_ensure((void **)&dev->d_offs, &dev->offs_cap, 4, 4, 0) || cl_ensure((void **)&dev->h_offs, &dev->h_offs_cap, 4, 4, 1))
        return CL_SMI_ERR_IO;
    if (cl_fifo_front_len(&dev->rx)) {
        /* bytes a batched call gave back wait in the front stash (after a "-3", a ragged call): this call takes the
         * copying chunk loop -- same chunks, slots and return codes -- and the in-place reader resumes behind it */
        return smi_count(dev, cl_smi_read_device_to(dev, channel, length_samples, d_iq, NULL, NULL));
    }
    const size_t cap_read = dev->max_read && dev->max_read < nb ? dev->max_read : nb;
    size_t left = length_samples * CL_BYTES_PER_SAMPLE, read_so_far = 0;
    if (cl_ensure((void **)&dev->chunks, &dev->chunks_cap, left / cap_read + 2, sizeof(cl_chunk), 2)) return CL_SMI_ERR_IO;
    dev->n_chunks = 0;
    while (left) {
        const size_t want = left < cap_read ? left : cap_read;
        size_t got; int slot, head_ok = 0;
        if (dev->ahead.valid && dev->ahead.len < want) {
            /* staged ahead when the FIFO held less than this read() asks for: if more has arrived since, the read() of the call
             * (caribou_smi.c:655: as many bytes as there are, up to len) would return more than what was staged -- and where a
             * read() ends decides where the next one looks for the sync pattern.  Staged again, now. */
            pthread_mutex_lock(&dev->fifo_mu);
            const int more = cl_fifo_pending(&dev->rx) != 0;
            pthread_mutex_unlock(&dev->fifo_mu);
            if (more) {
                if (dev->cstream) clhip_stream_sync(dev->cstream);
                pthread_mutex_lock(&dev->fifo_mu);
                cl_fifo_unstage(&dev->rx, dev->ahead.len);
                pthread_mutex_unlock(&dev->fifo_mu);
                dev->ahead.valid = 0;
                cl_smi_ahead_note(dev);
            }
        }
        if (dev->ahead.valid) {
            slot = dev->ahead.slot; got = dev->ahead.len; head_ok = dev->ahead.head_ok; dev->ahead.valid = 0;
            cl_smi_ahead_note(dev);
            if (got > want) {                                  /* staged for a longer read than this one: the tail is pending again */
                pthread_mutex_lock(&dev->fifo_mu);
                cl_fifo_unstage(&dev->rx, got - want);         /* (the newest staged bytes: nothing was staged behind them) */
                pthread_mutex_unlock(&dev->fifo_mu);
                got = want;
                head_ok = head_ok && want > 16;                /* a chunk of <= 16 bytes is in sync by definition, but keep it simple */
            }
        } else {
            slot = dev->next_slot;
            got = ra_stage(dev, slot, want, &head_ok);
            if (!got) break;                                   /* :657-661 "Reading timed-out" */
        }
        const int slot_next = (slot + 1) % CL_RA_SLOTS;
        dev->next_slot = slot_next;
        /* the read() after this one: the rest of this call, or the head of the next call */
        const size_t rest = left - got, want_next = rest ? (rest < cap_read ? rest : cap_read) : cap_read;
        int ahead_ok = 0;
        const size_t a = ra_stage(dev, slot_next, want_next, &ahead_ok);
        if (a) { dev->ahead.valid = 1; dev->ahead.slot = slot_next; dev->ahead.len = a; dev->ahead.head_ok = ahead_ok; cl_smi_ahead_note(dev); }
        if (want_words && own && read_so_far == 0 && got == left && head_ok && !(got & 15)) {
            /* The call is this one read(), and the host has seen the sync pattern on its first four words: offset 0
             * (caribou_smi.c:235-292) without asking the device, every slot written.  No launch here at all: the caller's
             * own first kernel reads the raw words (dev->fast_words, ready on dev->stream) -- the unpack in the client's
             * format, the IIR's input conversion, the fused pipe -- and cl_smi_ra_finish is the call's one synchronisation. */
            if (clhip_stream_wait_event(dev->stream, dev->ev_copied[slot])) return ra_fail(dev, got);
            dev->fast_words = dev->d_slot[slot];
            if (cl_smi_set_prev_words(dev, channel, dev->d_slot[slot], got)) return ra_fail(dev, got);   /* (a caller that writes dev->d_iq itself clears this) */
            dev->h_offs[0] = 0;
            dev->fast_used = 1;
            cl_chunk *c = &dev->chunks[dev->n_chunks];
            c->stage_off = 0; c->len = got; c->slot0 = 0; c->offs = 0;
            dev->ra_pending = 1; dev->ra_samples = got / CL_BYTES_PER_SAMPLE;
            return (long)dev->ra_samples;
        }
        if (own && read_so_far == 0 && cl_smi_restore_prev_words(dev, channel)) return ra_fail(dev, got);
        if (clhip_stream_wait_event(dev->stream, dev->ev_copied[slot]) ||
            clhip_smi_find_offsets(dev->d_slot[slot], got, nb, nb, 1, dev->d_offs, dev->stream) ||
            clhip_smi_unpack(channel, dev->d_slot[slot], got, nb, nb, 1, dev->d_offs, CL_FORMAT_CS16, d_iq + 2 * read_so_far, NULL, dev->stream) ||
            clhip_memcpy_d2h(dev->h_offs, dev->d_offs, 4, dev->stream))
            return ra_fail(dev, got);
        cl_chunk *c = &dev->chunks[dev->n_chunks];             /* published (n_chunks++) once its verdict is in */
        c->stage_off = 0; c->len = got; c->slot0 = read_so_far; c->offs = 0;
        read_so_far += got / CL_BYTES_PER_SAMPLE;              /* :677 */
        left -= got;                                           /* :678 */
        if (left == 0 || !a) {                                 /* the call's last read(): a further one would time out */
            dev->ra_pending = 1; dev->ra_samples = read_so_far;
            return (long)read_so_far;
        }
        if (clhip_stream_sync(dev->stream)) return ra_fail(dev, got);
        const int v = ra_chunk_verdict(dev);
        if (v) return v;
    }
    dev->stat_samples += read_so_far;
    if (!read_so_far) dev->stat_timeouts++;
    return (long)read_so_far;
}

int cl_smi_ra_finish(cl_smi *dev)
{
    if (!dev->ra_pending) return 0;
    dev->ra_pending = 0;
    if (clhip_stream_sync(dev->stream)) return (int)ra_fail(dev, dev->chunks[dev->n_chunks].len);
    const int v = ra_chunk_verdict(dev);
    if (v) return v;
    dev->stat_samples += dev->ra_samples;
    return (int)dev->ra_samples;
}

int cl_smi_read_device_ra(cl_smi *dev, int channel, size_t length_samples, int16_t *d_iq)
{
    const long exp = cl_smi_ra_launch(dev, channel, length_samples, d_iq);
    if (exp < 0 || !dev->ra_pending) return (int)exp;          /* (both halves count their own exits) */
    return cl_smi_ra_finish(dev);
}

/* Copy exactly the slots the reference writes (caribou_smi.c:344-389): n unpacked
 * samples per chunk, plus one extrapolated I/Q sample (no meta) after a re-sync. */
int cl_smi_copy_out(cl_smi *dev, cl_sample_complex_int16 *buffer, cl_sample_meta *metadata, int upto_chunk)
{
    const size_t nch = upto_chunk < 0 ? dev->n_chunks : (size_t)upto_chunk;
    if (nch == 0) return 0;
    int simple = 1;
    for (size_t i = 0; i < nch; i++)
        if (dev->chunks[i].offs != 0 || (dev->chunks[i].len & 3)) simple = 0;
    if (simple) {                       /* aligned stream: every slot is written, one copy */
        const cl_chunk *l = &dev->chunks[nch - 1];
        const size_t n = l->slot0 + l->len / 4;
        if (buffer && clhip_memcpy_d2h(buffer, dev->d_iq, n * 4, dev->stream)) return -1;
        if (metadata && clhip_memcpy_d2h(metadata, dev->d_meta, n, dev->stream)) return -1;
        return clhip_stream_sync(dev->stream);
    }
    for (size_t i = 0; i < nch; i++) {
        const cl_chunk *c = &dev->chunks[i];
        const size_t shortening = c->offs > 0 ? (size_t)(c->offs / 4 + 1) : 0;
        const size_t n = (c->len - 4 * shortening) / 4;
        const size_t n_iq = n + (shortening > 0 && n >= 2 ? 1 : 0);
        if (buffer && n_iq && clhip_memcpy_d2h(buffer + c->slot0, dev->d_iq + 2 * c->slot0, n_iq * 4, dev->stream)) return -1;
        if (metadata && n && clhip_memcpy_d2h(metadata + c->slot0, dev->d_meta + c->slot0, n, dev->stream)) return -1;
    }
    return clhip_stream_sync(dev->stream);
}

/* caribou_smi_read  caribou_smi.c:632-682 */
int cl_smi_read(cl_smi *dev, int channel, cl_sample_complex_int16 *buffer, cl_sample_meta *metadata, size_t length_samples)
{
    if (!dev) return CL_SMI_ERR_IO;
    if (dev->debug_mode != CL_SMI_DEBUG_NONE) return cl_smi_read_debug(dev, length_samples);
    int ret = cl_smi_read_device(dev, channel, length_samples, metadata != NULL, NULL);
    if (ret == CL_SMI_ERR_SYNC) { cl_smi_copy_out(dev, buffer, metadata, -1); return ret; }
    if (ret <= 0) return smi_count(dev, ret);
    if (cl_smi_copy_out(dev, buffer, metadata, -1)) return smi_count(dev, CL_SMI_ERR_IO);
    return ret;
}

/* caribou_smi_read with the caller's buffers in DEVICE memory (length_samples + 1 slots each; metadata may be
 * NULL): same chunk loop, slots and return codes; the results are complete when the call returns */
int cl_smi_read_to_device(cl_smi *dev, int channel, int16_t *d_iq, uint8_t *d_meta, size_t length_samples)
{
    if (!dev || !d_iq) return CL_SMI_ERR_IO;
    if (dev->debug_mode != CL_SMI_DEBUG_NONE) return cl_smi_read_debug(dev, length_samples);
    return smi_count(dev, cl_smi_read_device_to(dev, channel, length_samples, d_iq, d_meta, NULL));
}

/* caribou_smi_flush_fifo caribou_smi.c:772-783: drop what the driver FIFO holds (and what was staged ahead of it) */
int cl_smi_flush_fifo(cl_smi *dev)
{
    if (!dev) return -1;
    cl_smi_readahead_cancel(dev);
    pthread_mutex_lock(&dev->fifo_mu);
    cl_fifo_pop(&dev->rx, NULL, cl_fifo_pending(&dev->rx));
    pthread_mutex_unlock(&dev->fifo_mu);
    return 0;
}

void *cl_smi_stream(cl_smi *dev) { return dev ? dev->stream : NULL; }
int   cl_smi_device(const cl_smi *dev) { return dev ? dev->device : -1; }

/* --------------------------------------------------------------- TX path */
static int smi_write_core(cl_smi *dev, const cl_sample_complex_int16 *h_buffer, const int16_t *d_src, size_t length_samples)
{
    clhip_set_device(dev->device);
    size_t left = length_samples * CL_BYTES_PER_SAMPLE, written_so_far = 0;
    if (length_samples == 0) return 0;
    if ((h_buffer && cl_smi_ensure_iq(dev, length_samples + 8)) ||
        cl_ensure((void **)&dev->d_bytes, &dev->bytes_cap, left + 256, 1, 0))
        return CL_SMI_ERR_IO;
    /* the chunk loop only slices the same contiguous arrays (len &= ~3 never bites: 4 B/sample), so
     * the whole call is one pack launch; the FIFO then receives it in native-batch writes */
    /* the packed words go straight into the (pinned) TX FIFO, where the fd's write() side picks them up: the chunk loop of
     * caribou_smi.c:738-759 appends native-batch pieces of one contiguous array one after the other, i.e. the array */
    uint8_t *room = cl_smi_tx_reserve(dev, left);
    if (!room) return CL_SMI_ERR_IO;
    void *d_room = left <= cl_write_mapped_max() ? cl_fifo_device_ptr(&dev->tx, room) : NULL;
    if (h_buffer) {
        /* through a pinned buffer of ours: the runtime never sees the caller's pointer, so what it may remember of that
         * address (cached pinnings, released registrations) cannot matter; the memcpy is the one its staged path would do */
        if (cl_ensure((void **)&dev->h_txin, &dev->h_txin_cap, left + 64, 1, 1)) return CL_SMI_ERR_IO;
        memcpy(dev->h_txin, h_buffer, left);
        /* up to a few native batches the pack kernel itself reads the pinned samples and stores into the FIFO's room across
         * PCIe: one launch and one synchronisation, no copy-engine call on either side (cl_write_mapped_max: A/B) */
        void *d_in = d_room ? clhip_host_device_ptr(dev->h_txin) : NULL;
        if (d_in) d_src = (const int16_t *)d_in;
        else {
            if (clhip_memcpy_h2d(dev->d_iq, dev->h_txin, left, dev->stream)) { clhip_stream_sync(dev->stream); return CL_SMI_ERR_IO; }
            d_src = dev->d_iq;
        }
    }
    /* (every error exit behind a launch synchronises first: the kernels read dev->h_txin and store into the FIFO's reserved room,
     * both of which the next call may move or overwrite) */
    int bad;
    if (d_room) bad = clhip_smi_pack(dev->tx_mode, d_src, length_samples, (uint8_t *)d_room, dev->stream);
    else bad = clhip_smi_pack(dev->tx_mode, d_src, length_samples, dev->d_bytes, dev->stream) || clhip_memcpy_d2h(room, dev->d_bytes, left, dev->stream);
    if (clhip_stream_sync(dev->stream) || bad) return CL_SMI_ERR_IO;
    cl_smi_tx_commit(dev, left);                                /* len &= ~3 (:745) never bites: 4 bytes per sample */
    written_so_far = left / CL_BYTES_PER_SAMPLE;                /* :757 */
    dev->stat_written += written_so_far;
    return (int)written_so_far;
}

/* caribou_smi_write caribou_smi.c:720-762 */
int cl_smi_write(cl_smi *dev, int channel, cl_sample_complex_int16 *buffer, size_t length_samples)
{
    (void)channel;
    if (!dev || (!buffer && length_samples)) return CL_SMI_ERR_IO;
    return smi_write_core(dev, buffer, NULL, length_samples);
}

/* the same with the samples already in DEVICE memory (on the seam's stream, or complete) */
int cl_smi_write_from_device(cl_smi *dev, int channel, const int16_t *d_iq, size_t length_samples)
{
    (void)channel;
    if (!dev || (!d_iq && length_samples)) return CL_SMI_ERR_IO;
    return smi_write_core(dev, NULL, d_iq, length_samples);
}

/* ----------------------------------------------------- radio pass-through */
cl_radio *cl_radio_create(cl_smi *smi, int channel)
{
    if (!smi) return NULL;
    cl_radio *r = (cl_radio *)calloc(1, sizeof *r);
    if (r) { r->smi = smi; r->channel = channel; }
    return r;
}
void cl_radio_destroy(cl_radio *r) { free(r); }

/* cariboulite_radio.c:1258-1285 */
int cl_radio_read_samples(cl_radio *radio, cl_sample_complex_int16 *buffer, cl_sample_meta *metadata, size_t length)
{
    int ret = cl_smi_read(radio->smi, radio->channel, buffer, metadata, length);
    if (ret == CL_SMI_ERR_IO) fprintf(stderr, "SMI reading operation failed\n");
    else if (ret == CL_SMI_ERR_SYNC) fprintf(stderr, "SMI data synchronization failed\n");
    return ret;
}
/* cariboulite_radio.c:1288-1307 */
int cl_radio_write_samples(cl_radio *radio, cl_sample_complex_int16 *buffer, size_t length)
{
    int ret = cl_smi_write(radio->smi, radio->channel, buffer, length);
    if (ret < 0) fprintf(stderr, "SMI writing operation failed\n");
    return ret;
}
/* the trio's device-resident forms: the C++ API and the stream object keep their samples on the GPU */
int cl_radio_read_samples_device(cl_radio *radio, int16_t *d_iq, uint8_t *d_meta, size_t length)
{
    int ret = cl_smi_read_to_device(radio->smi, radio->channel, d_iq, d_meta, length);
    if (ret == CL_SMI_ERR_IO) fprintf(stderr, "SMI reading operation failed\n");
    else if (ret == CL_SMI_ERR_SYNC) fprintf(stderr, "SMI data synchronization failed\n");
    return ret;
}
int cl_radio_write_samples_device(cl_radio *radio, const int16_t *d_iq, size_t length)
{
    int ret = cl_smi_write_from_device(radio->smi, radio->channel, d_iq, length);
    if (ret < 0) fprintf(stderr, "SMI writing operation failed\n");
    return ret;
}
cl_smi *cl_radio_smi(cl_radio *radio) { return radio ? radio->smi : NULL; }
/* cariboulite_radio.c:1310-1315 */
size_t cl_radio_get_native_mtu_size_samples(cl_radio *radio) { return cl_smi_get_native_batch_samples(radio->smi); }
