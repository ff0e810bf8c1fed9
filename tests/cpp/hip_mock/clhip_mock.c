/* clhip_mock.c -- TEST INFRASTRUCTURE: a CPU model of the clhip_* layer (include/cariboulite_hip.h, layer 1) for the sanitizer
 * harnesses of the HOST code (cl_smi.c, cl_soapy.c, cl_group.c), built and run by tests/test_host_asan.py.  Never shipped, never
 * measured; the product links libcariboulite_hip.so.
 *
 * What it models is the ASYNCHRONY, which is what the host code can get wrong without a kernel being wrong: every HIP stream is
 * a worker thread with a queue of its own; copies and "kernels" are closures that run when the worker reaches them, concurrently
 * with the caller and with the other streams; events order streams the way hipEventRecord / hipStreamWaitEvent do.  Under
 * ThreadSanitizer a host write to memory that queued work still reads (a pinned staging buffer reused too early, a FIFO that
 * moves under a copy, an offsets table rewritten before its launch ran) is a reported race between the caller's thread and the
 * stream's thread; under AddressSanitizer a buffer freed under queued work is a reported use after free.
 *
 * The "kernels" compute with the oracle (oracle/cl_oracle.c: orc_find_buffer_offset, orc_rx_data_analyze, orc_cs16_to_*): the RX
 * unpack family exactly; the RX pipe as a two-tap stand-in with the real pipe's STATE contract (ping-pong history, per-stream
 * input counters, epochs of range runs, unrun) -- y[n] = i[n]/4096 + i[n-1]/8192, q[n]/4096 -- so that a run made twice, not at
 * all, or from the wrong history shows in the output; TX without a modulator (conversion + pack); the TX modulator pipe as a prefix-sum stand-in
 * with the real pipe's state contract (below).  IIR and the debug modes are not modelled (the calls fail loudly). */
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cariboulite_hip.h"
#include "cl_oracle.h"

/* --------------------------------------------------------------------------------------------- streams and events */
typedef struct job {
    struct job *next;
    void (*fn)(struct job *);
    void *p[6]; size_t z[8]; long i[6];
} job;

typedef struct mstream {
    pthread_t th; pthread_mutex_t mu; pthread_cond_t work, idle;
    job *head, *tail; int busy, stop;
    struct mstream *next_all;
} mstream;

typedef struct { pthread_mutex_t mu; pthread_cond_t cv; unsigned long recorded, done; } mevent;

static pthread_mutex_t g_all_mu = PTHREAD_MUTEX_INITIALIZER;
static mstream *g_all;
static __thread char g_err[256];

static void set_err(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
const char *clhip_last_error(void) { return g_err; }
const char *clhip_arch_name(void) { return "cpu-mock"; }
int clhip_device_count(void) { return 1; }
int clhip_set_device(int device) { return device == 0 ? 0 : -1; }

static void *worker(void *arg)
{
    mstream *s = (mstream *)arg;
    pthread_mutex_lock(&s->mu);
    for (;;) {
        while (!s->head && !s->stop) pthread_cond_wait(&s->work, &s->mu);
        if (!s->head) break;
        job *j = s->head;
        s->head = j->next; if (!s->head) s->tail = NULL;
        s->busy = 1;
        pthread_mutex_unlock(&s->mu);
        j->fn(j);
        free(j);
        pthread_mutex_lock(&s->mu);
        s->busy = 0;
        if (!s->head) pthread_cond_broadcast(&s->idle);
    }
    pthread_mutex_unlock(&s->mu);
    return NULL;
}

void *clhip_stream_create(void)
{
    mstream *s = (mstream *)calloc(1, sizeof *s);
    pthread_mutex_init(&s->mu, NULL); pthread_cond_init(&s->work, NULL); pthread_cond_init(&s->idle, NULL);
    pthread_create(&s->th, NULL, worker, s);
    pthread_mutex_lock(&g_all_mu); s->next_all = g_all; g_all = s; pthread_mutex_unlock(&g_all_mu);
    return s;
}

int clhip_stream_sync(void *stream)
{
    mstream *s = (mstream *)stream;
    if (!s) return 0;
    pthread_mutex_lock(&s->mu);
    while (s->head || s->busy) pthread_cond_wait(&s->idle, &s->mu);
    pthread_mutex_unlock(&s->mu);
    return 0;
}

static void sync_all(void)          /* what hipFree / hipHostFree do implicitly */
{
    pthread_mutex_lock(&g_all_mu);
    for (mstream *s = g_all; s; s = s->next_all) clhip_stream_sync(s);
    pthread_mutex_unlock(&g_all_mu);
}

void clhip_stream_destroy(void *stream)
{
    mstream *s = (mstream *)stream;
    if (!s) return;
    clhip_stream_sync(s);
    pthread_mutex_lock(&g_all_mu);
    for (mstream **pp = &g_all; *pp; pp = &(*pp)->next_all) if (*pp == s) { *pp = s->next_all; break; }
    pthread_mutex_unlock(&g_all_mu);
    pthread_mutex_lock(&s->mu); s->stop = 1; pthread_cond_broadcast(&s->work); pthread_mutex_unlock(&s->mu);
    pthread_join(s->th, NULL);
    pthread_mutex_destroy(&s->mu); pthread_cond_destroy(&s->work); pthread_cond_destroy(&s->idle);
    free(s);
}

static job *new_job(void (*fn)(job *)) { job *j = (job *)calloc(1, sizeof *j); j->fn = fn; return j; }

static int enqueue(void *stream, job *j)
{
    mstream *s = (mstream *)stream;
    if (!s) { j->fn(j); free(j); return 0; }            /* (the NULL stream: in place) */
    pthread_mutex_lock(&s->mu);
    if (s->tail) s->tail->next = j; else s->head = j;
    s->tail = j;
    pthread_cond_signal(&s->work);
    pthread_mutex_unlock(&s->mu);
    return 0;
}

void *clhip_event_create(void)
{
    mevent *e = (mevent *)calloc(1, sizeof *e);
    pthread_mutex_init(&e->mu, NULL); pthread_cond_init(&e->cv, NULL);
    return e;
}
void clhip_event_destroy(void *event)
{
    mevent *e = (mevent *)event;
    if (!e) return;
    pthread_mutex_destroy(&e->mu); pthread_cond_destroy(&e->cv); free(e);
}
static void job_event_done(job *j)
{
    mevent *e = (mevent *)j->p[0];
    pthread_mutex_lock(&e->mu);
    if (e->done < j->z[0]) e->done = j->z[0];
    pthread_cond_broadcast(&e->cv);
    pthread_mutex_unlock(&e->mu);
}
int clhip_event_record(void *event, void *stream)
{
    mevent *e = (mevent *)event;
    job *j = new_job(job_event_done);
    pthread_mutex_lock(&e->mu); j->z[0] = ++e->recorded; pthread_mutex_unlock(&e->mu);
    j->p[0] = e;
    return enqueue(stream, j);
}
static void wait_ticket(mevent *e, unsigned long t)
{
    pthread_mutex_lock(&e->mu);
    while (e->done < t) pthread_cond_wait(&e->cv, &e->mu);
    pthread_mutex_unlock(&e->mu);
}
int clhip_event_sync(void *event)
{
    mevent *e = (mevent *)event;
    pthread_mutex_lock(&e->mu); const unsigned long t = e->recorded; pthread_mutex_unlock(&e->mu);
    wait_ticket(e, t);
    return 0;
}
static void job_wait_event(job *j) { wait_ticket((mevent *)j->p[0], j->z[0]); }
int clhip_stream_wait_event(void *stream, void *event)
{
    mevent *e = (mevent *)event;
    job *j = new_job(job_wait_event);
    pthread_mutex_lock(&e->mu); j->z[0] = e->recorded; pthread_mutex_unlock(&e->mu);
    j->p[0] = e;
    return enqueue(stream, j);
}
float clhip_event_elapsed_ms(void *start, void *stop) { (void)start; clhip_event_sync(stop); return 0.f; }

/* --------------------------------------------------------------------------------------------- memory and copies */
void *clhip_malloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void  clhip_free(void *p) { if (p) { sync_all(); free(p); } }
void *clhip_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void  clhip_host_free(void *p) { if (p) { sync_all(); free(p); } }
void *clhip_host_device_ptr(void *h) { return h; }
void *clhip_host_register(void *h, size_t bytes) { (void)bytes; return h; }
void  clhip_host_unregister(void *h) { (void)h; }
size_t clhip_debug_ops(clhip_op_record *out, size_t max) { (void)out; (void)max; return 0; }
void clhip_debug_ops_dump(int fd) { (void)fd; }
void clhip_debug_copy_counters(uint64_t out[4]) { memset(out, 0, 4 * sizeof(uint64_t)); }

static void job_copy(job *j) { memcpy(j->p[0], j->p[1], j->z[0]); }
static int copy(void *dst, const void *src, size_t n, void *stream)
{
    job *j = new_job(job_copy); j->p[0] = dst; j->p[1] = (void *)src; j->z[0] = n;
    return enqueue(stream, j);
}
int clhip_memcpy_h2d(void *d, const void *h, size_t n, void *s) { return copy(d, h, n, s); }
int clhip_memcpy_d2h(void *h, const void *d, size_t n, void *s) { return copy(h, d, n, s); }
int clhip_memcpy_d2d(void *d, const void *src, size_t n, void *s) { return copy(d, src, n, s); }
static void job_copy2d(job *j)
{
    for (size_t r = 0; r < j->z[3]; r++) memcpy((uint8_t *)j->p[0] + r * j->z[0], (const uint8_t *)j->p[1] + r * j->z[1], j->z[2]);
}
int clhip_memcpy2d_h2d(void *d, size_t d_pitch, const void *h, size_t h_pitch, size_t width, size_t height, void *s)
{
    job *j = new_job(job_copy2d); j->p[0] = d; j->p[1] = (void *)h; j->z[0] = d_pitch; j->z[1] = h_pitch; j->z[2] = width; j->z[3] = height;
    return enqueue(s, j);
}
static void job_memset(job *j) { memset(j->p[0], (int)j->i[0], j->z[0]); }
int clhip_memset(void *d, int value, size_t n, void *s)
{
    job *j = new_job(job_memset); j->p[0] = d; j->i[0] = value; j->z[0] = n;
    return enqueue(s, j);
}

/* --------------------------------------------------------------------------------------------- the RX unpack family */
static size_t fmt_bytes(int fmt) { return fmt == CL_FORMAT_CF32 ? 8 : fmt == CL_FORMAT_CF64 ? 16 : fmt == CL_FORMAT_CS8 ? 2 : 4; }
static void put_format(const int16_t *iq, size_t n, int fmt, void *out)
{
    if (fmt == CL_FORMAT_CF32) orc_cs16_to_cf32(iq, (float *)out, n);
    else if (fmt == CL_FORMAT_CF64) orc_cs16_to_cf64(iq, (double *)out, n);
    else if (fmt == CL_FORMAT_CS8) orc_cs16_to_cs8(iq, (int8_t *)out, n);
    else memcpy(out, iq, 4 * n);
}

static void job_find_offsets(job *j)
{
    const uint8_t *b = (const uint8_t *)j->p[0]; int32_t *offs = (int32_t *)j->p[1];
    const size_t total = j->z[0], stride = j->z[1], len = j->z[2];
    for (long c = 0; c < j->i[0]; c++) {
        const size_t at = (size_t)c * stride, lc = at >= total ? 0 : (total - at < len ? total - at : len);
        offs[c] = orc_find_buffer_offset(b + at, lc);
    }
}
int clhip_smi_find_offsets(const uint8_t *d_bytes, size_t total, size_t stride, size_t len, int n_chunks, int32_t *d_offs, void *s)
{
    job *j = new_job(job_find_offsets); j->p[0] = (void *)d_bytes; j->p[1] = d_offs; j->z[0] = total; j->z[1] = stride; j->z[2] = len; j->i[0] = n_chunks;
    return enqueue(s, j);
}

static void job_unpack(job *j)
{
    const uint8_t *b = (const uint8_t *)j->p[0]; const int32_t *offs = (const int32_t *)j->p[1];
    uint8_t *out = (uint8_t *)j->p[2], *meta = (uint8_t *)j->p[3];
    const size_t total = j->z[0], stride = j->z[1], len = j->z[2];
    const int channel = (int)j->i[1], fmt = (int)j->i[2];
    for (long c = 0; c < j->i[0]; c++) {
        const size_t at = (size_t)c * stride, lc = at >= total ? 0 : (total - at < len ? total - at : len);
        const int32_t o = offs[c];                                 /* (read when the launch RUNS, like the kernel does) */
        if (o < 0 || lc < 4) continue;
        int16_t *iq = (int16_t *)malloc(lc + 16); uint8_t *mt = (uint8_t *)malloc(lc / 4 + 4);
        const int got = orc_rx_data_analyze(channel == 1 ? ORC_CH_HIF : 0, b + at, lc, iq, mt);
        if (got != o) { fprintf(stderr, "clhip_mock: unpack given offset %d where the search finds %d\n", o, got); abort(); }
        const size_t shortening = o > 0 ? (size_t)o / 4 + 1 : 0, n = (lc - 4 * shortening) / 4, written = n + (shortening && n >= 2 ? 1 : 0);
        const size_t slot0 = at / 4;
        put_format(iq, written, fmt, out + slot0 * fmt_bytes(fmt));
        if (meta) memcpy(meta + slot0, mt, n);
        free(iq); free(mt);
    }
}
int clhip_smi_unpack(int channel, const uint8_t *d_bytes, size_t total, size_t stride, size_t len, int n_chunks, const int32_t *d_offs, int format,
                     void *d_out, uint8_t *d_meta, void *s)
{
    job *j = new_job(job_unpack);
    j->p[0] = (void *)d_bytes; j->p[1] = (void *)d_offs; j->p[2] = d_out; j->p[3] = d_meta;
    j->z[0] = total; j->z[1] = stride; j->z[2] = len; j->i[0] = n_chunks; j->i[1] = channel; j->i[2] = format;
    return enqueue(s, j);
}

static void decode_aligned(int channel, const uint8_t *b, size_t n, int16_t *iq)
{
    for (size_t k = 0; k < n; k++) {
        uint32_t w; memcpy(&w, b + 4 * k, 4);
        int lo = (int)((w >> 1) & 0x1FFF), hi = (int)((w >> 17) & 0x1FFF);
        if (lo >= 0x1000) lo -= 0x2000;
        if (hi >= 0x1000) hi -= 0x2000;
        iq[2 * k] = (int16_t)(channel == 1 ? lo : hi); iq[2 * k + 1] = (int16_t)(channel == 1 ? hi : lo);     /* caribou_smi.c:342-378 */
    }
}
static void job_unpack_aligned(job *j)
{
    const size_t n = j->z[0] / 4;
    int16_t *iq = (int16_t *)malloc(4 * n + 4);
    decode_aligned((int)j->i[0], (const uint8_t *)j->p[0], n, iq);
    if (j->p[2]) memcpy(j->p[2], iq, 4 * n);
    if (j->p[1]) put_format(iq, n, (int)j->i[1], j->p[1]);
    free(iq);
}
int clhip_smi_unpack_aligned(int channel, const uint8_t *d_bytes, size_t n_bytes, int format, void *out, int16_t *d_cs16, void *s)
{
    job *j = new_job(job_unpack_aligned);
    j->p[0] = (void *)d_bytes; j->p[1] = out; j->p[2] = d_cs16; j->z[0] = n_bytes; j->i[0] = channel; j->i[1] = format;
    return enqueue(s, j);
}
static void job_convert(job *j) { put_format((const int16_t *)j->p[0], j->z[0], (int)j->i[0], j->p[1]); }
int clhip_convert_from_cs16(const int16_t *d_iq, size_t n, int format, void *d_out, void *s)
{
    job *j = new_job(job_convert); j->p[0] = (void *)d_iq; j->p[1] = d_out; j->z[0] = n; j->i[0] = format;
    return enqueue(s, j);
}

/* --------------------------------------------------------------------------------------------- the RX pipe's state contract */
struct clhip_rx_pipe {
    int n_streams, channel, cur, epoch_open, out_mode;
    float *hist[2];                                  /* ping-pong: the last input's I of every stream */
    unsigned long long *nt_s; uint8_t *ran;
    void *last_stream;
};

clhip_rx_pipe *clhip_rx_pipe_create(int n_streams, int channel, const float *fir, int n_fir, const float *rs, int n_rs, int up, int down, int out_mode)
{
    (void)fir; (void)n_fir; (void)rs; (void)n_rs;
    if (up != down || out_mode != CL_PIPE_OUT_IQ) { set_err("clhip_mock: the stand-in pipe is 1:1, complex out"); return NULL; }
    clhip_rx_pipe *p = (clhip_rx_pipe *)calloc(1, sizeof *p);
    p->n_streams = n_streams; p->channel = channel; p->out_mode = out_mode;
    p->hist[0] = (float *)calloc((size_t)n_streams, sizeof(float)); p->hist[1] = (float *)calloc((size_t)n_streams, sizeof(float));
    p->nt_s = (unsigned long long *)calloc((size_t)n_streams, sizeof *p->nt_s); p->ran = (uint8_t *)calloc((size_t)n_streams, 1);
    return p;
}
void clhip_rx_pipe_destroy(clhip_rx_pipe *p)
{
    if (!p) return;
    sync_all();
    free(p->hist[0]); free(p->hist[1]); free(p->nt_s); free(p->ran); free(p);
}
size_t clhip_rx_pipe_out_count(const clhip_rx_pipe *p, size_t n_in) { (void)p; return n_in; }
size_t clhip_rx_pipe_out_count_stream(const clhip_rx_pipe *p, int s, size_t n_in) { (void)p; (void)s; return n_in; }
unsigned long long clhip_rx_pipe_stream_total(const clhip_rx_pipe *p, int s) { return p && s >= 0 && s < p->n_streams ? p->nt_s[s] : 0; }
int clhip_rx_pipe_rollback(clhip_rx_pipe *p) { (void)p; set_err("clhip_mock: rollback is not modelled"); return -1; }

static void job_pipe_run(job *j)
{
    const int first = (int)j->i[0], count = (int)j->i[1], kind = (int)j->i[2], channel = (int)j->i[3];
    const float *hin = (const float *)j->p[2]; float *hout = (float *)j->p[3];
    const size_t in_stride = j->z[0], n = j->z[1], out_stride = j->z[2];
    for (int s = 0; s < count; s++) {
        int16_t *iq = (int16_t *)malloc(4 * n + 4);
        if (kind == CL_PIPE_IN_SMI_WORDS) decode_aligned(channel, (const uint8_t *)j->p[0] + 4 * (size_t)s * in_stride, n, iq);
        else memcpy(iq, (const int16_t *)j->p[0] + 2 * (size_t)s * in_stride, 4 * n);
        float *out = (float *)j->p[1] + 2 * (size_t)s * out_stride;
        float prev = hin[first + s];
        for (size_t k = 0; k < n; k++) {
            const float x = (float)iq[2 * k] / 4096.0f;
            out[2 * k] = x + 0.5f * prev; out[2 * k + 1] = (float)iq[2 * k + 1] / 4096.0f;
            prev = x;
        }
        hout[first + s] = prev;
        free(iq);
    }
}
long clhip_rx_pipe_run_range(clhip_rx_pipe *p, int first, int count, int in_kind, const void *d_in, size_t in_stride, size_t n_in, void *d_out,
                             size_t out_stride, void *stream)
{
    if (!p || !p->epoch_open || first < 0 || count < 1 || first + count > p->n_streams || in_kind == CL_PIPE_IN_CF32) { set_err("clhip_mock: run_range: bad arguments"); return -1; }
    for (int i = first; i < first + count; i++)
        if (p->ran[i]) { set_err("clhip_rx_pipe_run_range: stream %d already ran in this epoch", i); return -1; }
    job *j = new_job(job_pipe_run);
    j->p[0] = (void *)d_in; j->p[1] = d_out; j->p[2] = p->hist[p->cur]; j->p[3] = p->hist[p->cur ^ 1];
    j->z[0] = in_stride; j->z[1] = n_in; j->z[2] = out_stride; j->i[0] = first; j->i[1] = count; j->i[2] = in_kind; j->i[3] = p->channel;
    enqueue(stream, j);
    for (int i = first; i < first + count; i++) { p->ran[i] = 1; p->nt_s[i] += n_in; }
    return (long)n_in;
}
long clhip_rx_pipe_run(clhip_rx_pipe *p, int in_kind, const void *d_in, size_t in_stride, size_t n_in, void *d_out, size_t out_stride, void *stream)
{
    if (!p || p->epoch_open) { set_err("clhip_mock: run: an epoch is open"); return -1; }
    p->epoch_open = 1; memset(p->ran, 0, (size_t)p->n_streams);
    const long r = clhip_rx_pipe_run_range(p, 0, p->n_streams, in_kind, d_in, in_stride, n_in, d_out, out_stride, stream);
    p->epoch_open = 0; p->cur ^= 1;
    return r;
}
int clhip_rx_pipe_epoch_begin(clhip_rx_pipe *p)
{
    if (!p || p->epoch_open) { set_err("clhip_rx_pipe_epoch_begin: bad pipe or epoch already open"); return -1; }
    memset(p->ran, 0, (size_t)p->n_streams);
    p->epoch_open = 1;
    return 0;
}
int clhip_rx_pipe_epoch_end(clhip_rx_pipe *p, void *stream)
{
    if (!p || !p->epoch_open) { set_err("clhip_rx_pipe_epoch_end: no epoch open"); return -1; }
    for (int i = 0; i < p->n_streams; i++)
        if (!p->ran[i]) copy(p->hist[p->cur ^ 1] + i, p->hist[p->cur] + i, sizeof(float), stream);
    p->cur ^= 1; p->epoch_open = 0;
    return 0;
}
int clhip_rx_pipe_unrun_stream(clhip_rx_pipe *p, int s, size_t n_in)
{
    if (!p || !p->epoch_open || s < 0 || s >= p->n_streams || !p->ran[s] || p->nt_s[s] < n_in) { set_err("clhip_rx_pipe_unrun_stream: nothing to take back"); return -1; }
    p->ran[s] = 0; p->nt_s[s] -= n_in;
    return 0;
}

/* --------------------------------------------------------------------------------------------- not modelled */
struct clhip_iir { int dummy; };
clhip_iir *clhip_iir_create(const double *sos, int n_stages, int n_streams) { (void)sos; (void)n_stages; (void)n_streams; return (clhip_iir *)calloc(1, sizeof(clhip_iir)); }
void clhip_iir_destroy(clhip_iir *f) { free(f); }
int clhip_iir_run(clhip_iir *f, const int16_t *a, int16_t *b, size_t c, size_t d, void *s) { (void)f; (void)a; (void)b; (void)c; (void)d; (void)s; set_err("clhip_mock: no IIR"); return -1; }
int clhip_iir_run_smi(clhip_iir *f, int ch, const uint8_t *w, int16_t *o, size_t a, size_t b, void *s) { (void)f; (void)ch; (void)w; (void)o; (void)a; (void)b; (void)s; set_err("clhip_mock: no IIR"); return -1; }
int clhip_iir_status(clhip_iir *f) { (void)f; return 0; }
int clhip_iir_unrun(clhip_iir *f) { (void)f; return 0; }
int clhip_iir_set_state(clhip_iir *f, const double *h) { (void)f; (void)h; return 0; }
int clhip_iir_get_state(clhip_iir *f, double *h) { (void)f; memset(h, 0, 16 * sizeof(double)); return 0; }
void clhip_iir_set_poll_bound(clhip_iir *f, int polls) { (void)f; (void)polls; }
int clhip_smi_debug_analyze(int mode, const uint8_t *b, size_t n, uint32_t last, int32_t *res, void *s) { (void)mode; (void)b; (void)n; (void)last; (void)res; (void)s; set_err("clhip_mock: no debug modes"); return -1; }
/* TX without a modulator: the conversion loop (CaribouliteStream.cpp:199-244) and caribou_smi_generate_data (caribou_smi.c:684-717) */
static void job_pack(job *j)
{
    const size_t n = j->z[0]; const int fmt = (int)j->i[0], mode = (int)j->i[1];
    int16_t *iq = (int16_t *)malloc(4 * n + 4);
    if (fmt == CL_FORMAT_CF32) orc_cf32_to_cs16((const float *)j->p[0], iq, n);
    else if (fmt == CL_FORMAT_CF64) orc_cf64_to_cs16((const double *)j->p[0], iq, n);
    else if (fmt == CL_FORMAT_CS8) orc_cs8_to_cs16((const int8_t *)j->p[0], iq, n);
    else memcpy(iq, j->p[0], 4 * n);
    orc_generate_data(mode, iq, n, (uint8_t *)j->p[1]);
    free(iq);
}
int clhip_smi_pack(int mode, const int16_t *iq, size_t n, uint8_t *b, void *s)
{
    job *j = new_job(job_pack); j->p[0] = (void *)iq; j->p[1] = b; j->z[0] = n; j->i[0] = CL_FORMAT_CS16; j->i[1] = mode;
    return enqueue(s, j);
}
int clhip_convert_pack_rows(const void *const *in_rows, int fmt, size_t n, int n_rows, int mode, uint8_t *const *out_rows, void *s)
{
    if (n_rows < 0 || n_rows > CLHIP_PACK_ROWS) { set_err("clhip_convert_pack_rows: 1 .. 8 rows"); return -1; }
    for (int r = 0; r < n_rows; r++) {                    /* (the addresses are read at the call; the rows run one after the other on the stream) */
        job *j = new_job(job_pack); j->p[0] = (void *)in_rows[r]; j->p[1] = out_rows[r]; j->z[0] = n; j->i[0] = fmt; j->i[1] = mode;
        enqueue(s, j);
    }
    return 0;
}
int clhip_convert_pack(const void *in, int fmt, size_t n, int mode, uint8_t *b, void *s)
{
    job *j = new_job(job_pack); j->p[0] = (void *)in; j->p[1] = b; j->z[0] = n; j->i[0] = fmt; j->i[1] = mode;
    return enqueue(s, j);
}
/* ---- the TX modulator pipe: a stand-in with the real pipe's STATE contract (per-stream carried state in ping-pong halves flipped by the
 * host at the call, a verdict asked after the run, roll-back, streams moving between pipes; only up == down == 1).  Per message m:
 * v = lrintf(m * 4096); acc += v (the "phase": a prefix sum over the whole stream); word = pack(I = acc mod 8191 - 4095, Q = clamp(v + v_prev))
 * -- so that a run made twice, not at all, or from the wrong state shows in every later word.  clhip_mock_tx_fail_every(k): every k-th
 * run of a pipe "gives up" once (its words are garbage, clhip_tx_pipe_status says so and puts the state back). */
#define MOCK_TX_MAXS 16
struct clhip_tx_pipe {
    int n, mode, fm;
    long long acc[2][MOCK_TX_MAXS]; int prev[2][MOCK_TX_MAXS]; int cur;
    unsigned long long n_total, undo_n_total; int undo_cur, can_undo;
    int lb_err;                                      /* written by the run's job, read after the stream has been synchronised */
    long runs; int forced;
};
static int g_tx_fail_every;
void clhip_mock_tx_fail_every(int k) { g_tx_fail_every = k; }

static void job_take_i(job *j)
{
    for (size_t k = 0; k < j->z[0]; k++) ((float *)j->p[1])[k] = ((const float *)j->p[0])[2 * k];
}
int clhip_take_i_rail(const float *in, size_t n, float *out, void *s)
{
    job *j = new_job(job_take_i); j->p[0] = (void *)in; j->p[1] = out; j->z[0] = n;
    return enqueue(s, j);
}
static void job_words_rows(job *j)
{
    uint8_t **dst = (uint8_t **)j->p[1];
    for (size_t r = 0; r < j->z[2]; r++) memcpy(dst[r], (const uint8_t *)j->p[0] + r * j->z[0], 4 * j->z[1]);
    free(dst);
}
static void job_rows_rows(job *j)
{
    void **t = (void **)j->p[0];                       /* [src x n][dst x n][bytes x n] */
    const size_t n = j->z[0];
    for (size_t r = 0; r < n; r++) memcpy(t[n + r], t[r], (size_t)(uintptr_t)t[2 * n + r]);
    free(t);
}
int clhip_rows_to_rows(const void *const *src, void *const *dst, const size_t *bytes, int rows, void *s)
{
    if (rows < 0 || rows > CLHIP_PACK_ROWS) { set_err("clhip_rows_to_rows: 1 .. 8 rows"); return -1; }
    if (!rows) return 0;
    void **t = (void **)malloc(sizeof(void *) * 3 * (size_t)rows);       /* (the addresses are read at the call) */
    for (int r = 0; r < rows; r++) { t[r] = (void *)src[r]; t[rows + r] = dst[r]; t[2 * rows + r] = (void *)(uintptr_t)bytes[r]; }
    job *j = new_job(job_rows_rows); j->p[0] = t; j->z[0] = (size_t)rows;
    return enqueue(s, j);
}
int clhip_words_to_rows(const uint8_t *w, size_t is, size_t n, int rows, uint8_t *const *dst, void *s)
{
    if (rows < 0 || rows > CLHIP_PACK_ROWS) { set_err("clhip_words_to_rows: 1 .. 8 rows"); return -1; }
    uint8_t **d = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)(rows ? rows : 1));       /* (the addresses are read at the call) */
    memcpy(d, dst, sizeof(uint8_t *) * (size_t)rows);
    job *j = new_job(job_words_rows); j->p[0] = (void *)w; j->p[1] = d; j->z[0] = is; j->z[1] = n; j->z[2] = (size_t)rows;
    return enqueue(s, j);
}
clhip_tx_pipe *clhip_tx_pipe_create(int n_streams, double kf, double fs, const float *rs, int n_rs, int up, int down, int mode)
{
    (void)fs; (void)rs; (void)n_rs;
    if (n_streams < 1 || n_streams > MOCK_TX_MAXS || up != 1 || down != 1) { set_err("clhip_mock: TX pipes of 1 .. %d streams without a resampler", MOCK_TX_MAXS); return NULL; }
    clhip_tx_pipe *p = (clhip_tx_pipe *)calloc(1, sizeof *p);
    p->n = n_streams; p->mode = mode; p->fm = kf != 0.0;
    return p;
}
void clhip_tx_pipe_destroy(clhip_tx_pipe *p) { if (p) { sync_all(); free(p); } }
static void job_tx_run(job *j)
{
    clhip_tx_pipe *p = (clhip_tx_pipe *)j->p[0];
    const int from = (int)j->i[0], kind = (int)j->i[1], fail = (int)j->i[2];
    const size_t n = j->z[0], is = j->z[1], ws = j->z[2];
    int16_t *iq = (int16_t *)malloc(4 * (n ? n : 1));
    for (int s = 0; s < p->n; s++) {
        long long acc = p->acc[from][s]; int prev = p->prev[from][s];
        for (size_t k = 0; k < n; k++) {
            const float m = kind == CL_TXPIPE_IN_FM_MESSAGE ? ((const float *)j->p[1])[(size_t)s * is + k] : ((const float *)j->p[1])[2 * ((size_t)s * is + k)];
            const int v = (int)lrintf(m * 4096.0f);
            acc += v;
            int q = v + prev; if (q > 4095) q = 4095; if (q < -4096) q = -4096;
            iq[2 * k] = (int16_t)((int)(((acc % 8191) + 8191) % 8191) - 4095); iq[2 * k + 1] = (int16_t)q;
            prev = v;
        }
        p->acc[from ^ 1][s] = acc; p->prev[from ^ 1][s] = prev;
        if (fail) memset((uint8_t *)j->p[2] + (size_t)s * ws, 0xEE, 4 * n);
        else orc_generate_data(p->mode == CL_TX_AS_WRITTEN ? ORC_TX_AS_WRITTEN : ORC_TX_DOCUMENTED, iq, n, (uint8_t *)j->p[2] + (size_t)s * ws);
    }
    free(iq);
    if (fail) __atomic_store_n(&p->lb_err, 1, __ATOMIC_RELEASE);
}
long clhip_tx_pipe_run(clhip_tx_pipe *p, int kind, const void *in, size_t is, size_t n, uint8_t *w, size_t ws, float *iq, size_t iqs, void *s)
{
    (void)iq; (void)iqs;
    if (!p || !in || !w) { set_err("clhip_tx_pipe_run: bad arguments"); return -1; }
    if (!n) return 0;
    p->undo_n_total = p->n_total; p->undo_cur = p->cur; p->can_undo = 1;
    p->runs++;
    const int fail = g_tx_fail_every && !p->forced && p->runs % g_tx_fail_every == 0;
    job *j = new_job(job_tx_run); j->p[0] = p; j->p[1] = (void *)in; j->p[2] = w; j->z[0] = n; j->z[1] = is; j->z[2] = ws; j->i[0] = p->cur; j->i[1] = kind; j->i[2] = fail;
    if (enqueue(s, j)) return -1;
    p->cur ^= 1; p->n_total += n;
    return (long)n;
}
int clhip_tx_pipe_status(clhip_tx_pipe *p)
{
    if (!p) return -1;
    if (!__atomic_load_n(&p->lb_err, __ATOMIC_ACQUIRE)) return 0;
    p->lb_err = 0; p->forced = 1;
    if (p->can_undo) { p->cur = p->undo_cur; p->n_total = p->undo_n_total; p->can_undo = 0; }
    set_err("clhip_tx_pipe_status: look-back poll overran (mock)");
    return -1;
}
void clhip_tx_pipe_set_poll_bound(clhip_tx_pipe *p, int polls) { (void)p; (void)polls; }
unsigned long long clhip_tx_pipe_position(const clhip_tx_pipe *p) { return p ? p->n_total : 0; }
int clhip_tx_pipe_pack_mode(const clhip_tx_pipe *p) { return p ? p->mode : -1; }
int clhip_tx_pipe_set_position(clhip_tx_pipe *p, unsigned long long n) { if (!p) return -1; p->n_total = n; p->can_undo = 0; return 0; }
int clhip_tx_pipe_move_stream(clhip_tx_pipe *d, int ds, clhip_tx_pipe *sp, int ss, void *st)
{
    if (!d || !sp || ds < 0 || ss < 0 || ds >= d->n || ss >= sp->n) { set_err("clhip_tx_pipe_move_stream: bad arguments"); return -1; }
    clhip_stream_sync(st);
    /* (both pipes idle: a run in flight on either would race with these plain accesses -- which is what ThreadSanitizer is here to see) */
    d->acc[d->cur][ds] = sp->acc[sp->cur][ss]; d->prev[d->cur][ds] = sp->prev[sp->cur][ss];
    d->can_undo = 0; sp->can_undo = 0;
    return 0;
}
