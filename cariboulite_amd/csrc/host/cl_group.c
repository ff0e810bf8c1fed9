/* cl_group.c -- a STREAM GROUP at the Soapy boundary: N devices of one GPU read in one call.
 *
 * The reference's unit is one SoapySDR device per channel (soapy_api/SoapyCariboulite.cpp:46-69: every board enumerates
 * an S1G and a HiF device), each readStream a caribou_smi_read chunk loop of its own (caribou_smi/caribou_smi.c:632-682,
 * soapy_api/CaribouliteStreamFunctions.cpp:239-254).  A client with many boards calls readStream once per device and pays,
 * on a GPU, one launch chain and one synchronisation per device for 512 KiB of input.  cl_group_readStream is those N
 * calls as one: per-stream state, re-sync and "-3" semantics are exactly those of N single cl_readStream calls (a stream
 * that cannot take the batched route takes its own device's single-stream route, here, inside the call), while the
 * streams that are in sync -- the normal case -- share launches and cross PCIe as a pipeline:
 *
 *     sub-batch b (SUB streams):  FIFO bytes --copy engine--> d_in rows        streams s_in[row mod K], events in[b][k]
 *                                 one launch over the rows (fused pipe / unpack) stream s_k,   event k[b]
 *                                 d_out rows --copy engine--> pinned mirror      stream s_out, event out[b]
 *                                 mirror rows --memcpy pool--> the clients' (pageable) buffers
 *
 * so that sub-batch b + 1 comes in and sub-batch b - 1 leaves while b computes, and the last hop (which one thread cannot
 * do at PCIe rate: profiles/r04/pcie_duplex.json) is spread over a few threads with non-temporal stores.
 *
 * Calls overlap as well (READAHEAD): before a call waits for its own results it stages the members' NEXT batches in their
 * FIFOs, copies them in and (READAHEAD=2) launches over them into the second of two mirrors -- the next call finds its results
 * computed or on their way and the GPU never waits for the host between two calls.  Bytes read ahead stay the members' (staged,
 * counted as pending, given back to any other reader of the seam: cl_smi_foreign_cancel); a run made ahead is taken back when
 * its client goes another way (clhip_rx_pipe_unrun_stream).
 *
 * State: a lane of the group (members with one channel type and one stream configuration) owns ONE n-stream RX pipe whose
 * streams advance independently (clhip_rx_pipe_epoch_begin / _run_range / _epoch_end); formats without extension stages
 * are stateless behind the unpack.  What the single-stream route keeps for the reference's "untouched slots" (the
 * persistent native buffer) is rebuilt lazily from the previous call's raw words when a member leaves the batched route. */
#include <immintrin.h>
#include <time.h>
#include <unistd.h>

#include "cl_internal.h"

enum { ROUTE_PIPE = 1, ROUTE_PLAIN = 2, ROUTE_TX_PLAIN = 3, ROUTE_TX_SINGLE = 4 };

typedef struct { uint8_t *dst; const uint8_t *src; size_t bytes; } copy_job;

typedef struct {
    pthread_t *th; int n_threads;
    pthread_mutex_t mu; pthread_cond_t work, idle;
    copy_job *q; size_t q_cap, q_head, q_len;
    size_t in_flight;                     /* queued + being copied */
    int stop;
} copy_pool;

typedef struct {
    int channel, route, format;
    size_t elem_bytes;                    /* bytes per output element */
    int up, down;
    int n; int *member;                   /* indices into the group's device table */
    int sub;                              /* streams per sub-batch */
    clhip_rx_pipe *pipe;                  /* ROUTE_PIPE */
    size_t in_stride;                     /* bytes per row of d_in */
    uint8_t *d_in[3]; int cur_in, prev_in, next_in;   /* raw words of this call / of the call before it / read ahead for the next one (rotating) */
    size_t *primed; unsigned *primed_epoch;   /* per row: bytes of the NEXT call's batch already staged in the member's FIFO and copied to d_in[next_in]
                                               * (0: none), and the member seam's foreign_epoch then -- a reader of the seam's own that came in between bumps it */
    void *ev_primed;                      /* behind the read-ahead's copies */
    size_t out_stride;                    /* elements per row of d_out / h_out */
    uint8_t *d_out;
    uint8_t *h_out[2], *m_out[2]; int cur_m;   /* two pinned mirrors (this call's results / the next call's, computed ahead) and the device's addresses of
                                               * them (mapped pinned): kernels may store into a mirror themselves */
    int32_t *h_offs[4]; int32_t *d_offs[4];    /* ROUTE_PLAIN: per row 0 (unpack) / -1 (skip), mapped pinned: one table per (event set, launched ahead | in the
                                               * call) -- a launch reads its table when it RUNS */
    uint8_t *done_ahead; long *ahead_got;      /* per row: the previous call launched over the batch it read ahead, results in h_out[cur_m ^ 1] then; elements */
    uint8_t *direct;                           /* per call and row: the copy engine wrote the client's registered buffer */
    cl_read_ctx *ctx;                          /* per call and row: a one-by-one member's read in flight (lanes without extension stages) */
    /* the reference's low-pass over whole sub-batches (lanes without extension stages): one multi-stream filter object per (filter,
     * sub-batch), made when first needed; a member's carried state lives EITHER in its stream's own objects or here (iir_own) */
    clhip_iir **giir; int n_subs; uint8_t *iir_own; int16_t *d_f; uint8_t *sub_ft; uint8_t *how;
    uint8_t *ahead_ft;                         /* per sub-batch: the filter of a filter launch made AHEAD (its results in the other mirror; 0: none) */
    uint8_t *sub_verdict;                      /* per call and sub-batch: 0 = its filter launch has not been asked yet; 1 = good; 2 = gave up twice (nothing to deliver); 3 = runtime error */
    int epoch_open;                            /* the pipe's epoch of the NEXT call was opened by the read-ahead */
    int set;                                   /* this call's event set (0 / 1) */
    size_t sub0; int queued;                   /* the lane's first sub-batch among the group's; sub-batches queued in this call */
    uint8_t *fast; size_t *len; long *got;   /* per call */
    uint8_t *ahead_mark; size_t want;      /* per call: rows staged for the read-ahead; the call's bytes per batch */
    uint8_t **src;                        /* per call and row: where the staged batch lies in the member's pinned FIFO */
    cl_dsp_cfg dsp;
    /* a TX group's lane (cl_group_writeStream): the clients' samples, row by row, in pinned memory and on the device */
    uint8_t *tx_h_in, *tx_d_in; size_t tx_row;      /* two sets of rows each: this call's and the previous call's (still in flight) */
    uint8_t *tx_pend; int tx_pend_set; size_t tx_pend_want;   /* per row: words launched over and not committed yet (write-behind) */
} lane_t;

struct cl_group {
    int device;
    int dir;                              /* CL_SOAPY_SDR_RX: a group to read through; CL_SOAPY_SDR_TX: to write through (boards are half duplex: one stream per device) */
    size_t n; cl_device **dev;
    int *lane_of, *row_of;                /* member -> lane / row */
    int n_lanes; lane_t *lane;
    int sub;                              /* kwarg SUBBATCH: streams per sub-batch for every lane (0: by the lane's output size, cl_group_make) */
    int readahead;                        /* kwarg READAHEAD: 0 = none; 1 = before a call waits for its results the NEXT call's batches are staged and copied in;
                                           * 2 (default) = ... and launched over, into the second mirror */
    size_t n_sub;                         /* sub-batches over all lanes */
    int iir_polls; int iir_polls_set;     /* test hook (cl_group_set_iir_poll_bound): the poll bound of the group's filter objects */
    pthread_mutex_t tx_mu; int tx_mu_ok; int tx_pending;      /* a TX group: the call and the members' seams' call-backs (any thread) */
    int stale;                            /* work made ahead has just been given up: it is waited for before anything takes its place (settle) */
    int sink_mapped;                      /* kwarg SINK: "mapped" (default) = the sub-batch's kernel stores into the mapped pinned mirror itself; "copy" = device buffer + copy engine */
#define GRP_MAX_IN 8
    void *s_in[GRP_MAX_IN], *s_k, *s_out; /* ingest streams taken in turn by the members' copies: a copy's fixed cost (~10 us between two copies of one
                                           * stream, rocprofv3 trace: profiles/r04/group_call_timeline_cs16.txt) overlaps the transfer of its neighbour's */
    int n_in;                             /* kwarg INGEST_STREAMS: 1 .. 8 (default 2) */
    int ev_per;                           /* events per sub-batch: n_in (in) + 1 (launched) + 1 (out) */
    void **ev; size_t n_ev;               /* ev_per per sub-batch and set */
    copy_pool pool;
    uint8_t **reg_base; size_t *reg_len; size_t n_reg;   /* page ranges of client buffers registered with the GPU (cl_group_register_buffers), merged where they touch */
    uint8_t *has_reg;                                     /* per member: it has a registered buffer */
    uint8_t *slab; size_t slab_slice;     /* ONE pinned allocation the members' byte FIFOs live in, a slice each, lane after lane in row order: batches
                                           * of neighbouring rows that lie at the same offset of their slices are `slab_slice` apart -- one 2-D copy */
    cl_group_stats stats;
    char err[256];
};

/* ------------------------------------------------------------------ the last hop: pinned mirror -> client memory */
__attribute__((target("avx2"))) static void copy_stream_avx2(uint8_t *dst, const uint8_t *src, size_t n)
{
    /* non-temporal stores: the client's buffer is written once and read later, by someone else; going around the cache
     * saves the read-for-ownership of every destination line (tools/microbench/pcie_duplex.hip: 44 against 31 GB/s on
     * one thread, 120 against 90 on four) */
    const size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
    if (head > n) { memcpy(dst, src, n); return; }
    if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32)),
                      c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a); _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c); _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    _mm_sfence();
    if (i < n) memcpy(dst + i, src + i, n - i);
}

static void copy_out(uint8_t *dst, const uint8_t *src, size_t n)
{
    static int avx2_known = -1;                             /* (several threads may find out at once: the same answer, stored atomically) */
    int avx2 = __atomic_load_n(&avx2_known, __ATOMIC_RELAXED);
    if (avx2 < 0) { avx2 = __builtin_cpu_supports("avx2") ? 1 : 0; __atomic_store_n(&avx2_known, avx2, __ATOMIC_RELAXED); }
    if (avx2 && n >= 4096) copy_stream_avx2(dst, src, n); else memcpy(dst, src, n);
}

static void *pool_thread(void *arg)
{
    copy_pool *p = (copy_pool *)arg;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->q_len && !p->stop) pthread_cond_wait(&p->work, &p->mu);
        if (!p->q_len && p->stop) break;
        const copy_job j = p->q[p->q_head];
        p->q_head = (p->q_head + 1) % p->q_cap; p->q_len--;
        pthread_mutex_unlock(&p->mu);
        copy_out(j.dst, j.src, j.bytes);
        pthread_mutex_lock(&p->mu);
        if (--p->in_flight == 0) pthread_cond_broadcast(&p->idle);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

static int pool_start(copy_pool *p, int n_threads, size_t q_cap)
{
    memset(p, 0, sizeof *p);
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->work, NULL); pthread_cond_init(&p->idle, NULL);
    p->q = (copy_job *)calloc(q_cap, sizeof *p->q); p->q_cap = q_cap;
    p->th = (pthread_t *)calloc((size_t)(n_threads > 0 ? n_threads : 1), sizeof *p->th);
    if (!p->q || !p->th) return -1;
    for (int i = 0; i < n_threads; i++) {
        if (pthread_create(&p->th[i], NULL, pool_thread, p)) break;
        p->n_threads++;
    }
    return 0;
}

static void pool_stop(copy_pool *p)
{
    pthread_mutex_lock(&p->mu);
    p->stop = 1;
    pthread_cond_broadcast(&p->work);
    pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->n_threads; i++) pthread_join(p->th[i], NULL);
    free(p->th); free(p->q);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->work); pthread_cond_destroy(&p->idle);
    memset(p, 0, sizeof *p);
}

/* queue [src, src + bytes) -> dst in pieces; without worker threads the caller copies */
static void pool_submit(copy_pool *p, uint8_t *dst, const uint8_t *src, size_t bytes)
{
    const size_t piece = (size_t)512 << 10;
    if (!p->n_threads) { copy_out(dst, src, bytes); return; }
    pthread_mutex_lock(&p->mu);
    for (size_t o = 0; o < bytes; o += piece) {
        const size_t n = bytes - o < piece ? bytes - o : piece;
        if (p->q_len == p->q_cap) {                     /* full: do this piece here */
            pthread_mutex_unlock(&p->mu);
            copy_out(dst + o, src + o, n);
            pthread_mutex_lock(&p->mu);
            continue;
        }
        p->q[(p->q_head + p->q_len) % p->q_cap] = (copy_job){dst + o, src + o, n};
        p->q_len++; p->in_flight++;
    }
    pthread_cond_broadcast(&p->work);
    pthread_mutex_unlock(&p->mu);
}

/* the caller helps until the queue is empty, then waits for the pieces still being copied */
static void pool_drain(copy_pool *p)
{
    if (!p->n_threads) return;
    pthread_mutex_lock(&p->mu);
    while (p->q_len) {
        const copy_job j = p->q[p->q_head];
        p->q_head = (p->q_head + 1) % p->q_cap; p->q_len--;
        pthread_mutex_unlock(&p->mu);
        copy_out(j.dst, j.src, j.bytes);
        pthread_mutex_lock(&p->mu);
        if (--p->in_flight == 0) pthread_cond_broadcast(&p->idle);
    }
    while (p->in_flight) pthread_cond_wait(&p->idle, &p->mu);
    pthread_mutex_unlock(&p->mu);
}

/* ------------------------------------------------------------------------------------------- make / unmake */
static const char *kwget(const char *const *keys, const char *const *vals, size_t n, const char *key)
{
    for (size_t i = 0; i < n; i++)
        if (keys && vals && keys[i] && vals[i] && !strcmp(keys[i], key)) return vals[i];
    return NULL;
}

static size_t fmt_bytes(int fmt) { return fmt == CL_FORMAT_CF32 ? 8 : fmt == CL_FORMAT_CF64 ? 16 : fmt == CL_FORMAT_CS8 ? 2 : 4; }

static int same_dsp(const cl_dsp_cfg *a, const cl_dsp_cfg *b)
{
    return a->enabled == b->enabled && a->n_fir == b->n_fir && a->up == b->up && a->down == b->down && a->n_rs == b->n_rs &&
           a->demod_fm == b->demod_fm && !memcmp(a->fir, b->fir, sizeof(float) * (size_t)a->n_fir) &&
           !memcmp(a->rs, b->rs, sizeof(float) * (size_t)a->n_rs);
}

static void ahead_cancel_all(cl_group *g);
static void tx_settle_hook(void *ctx, int member);
static void tx_finish(cl_group *g);
static void settle(cl_group *g);
static void giir_ahead_drop(cl_group *g, lane_t *l, int sb);
static void **ev_of(const cl_group *g, const lane_t *l, int set, int a);
static void iir_home(void *ctx, int member);

static void lane_free(lane_t *l)
{
    if (l->pipe) clhip_rx_pipe_destroy(l->pipe);
    clhip_free(l->d_in[0]); clhip_free(l->d_in[1]); clhip_free(l->d_in[2]); clhip_free(l->d_out);
    clhip_event_destroy(l->ev_primed); free(l->primed); free(l->primed_epoch);
    clhip_host_free(l->h_out[0]); clhip_host_free(l->h_out[1]); clhip_host_free(l->h_offs[0]);
    clhip_host_free(l->tx_h_in); clhip_free(l->tx_d_in); free(l->tx_pend);
    free(l->done_ahead); free(l->ahead_got); free(l->direct); free(l->ctx);
    for (int i = 0; l->giir && i < 3 * l->n_subs; i++) clhip_iir_destroy(l->giir[i]);
    free(l->giir); free(l->iir_own); free(l->sub_ft); free(l->ahead_ft); free(l->sub_verdict); free(l->how); clhip_free(l->d_f);
    free(l->member); free(l->fast); free(l->len); free(l->got); free(l->src); free(l->ahead_mark);
    memset(l, 0, sizeof *l);
}

void cl_group_unmake(cl_group *g)
{
    if (!g) return;
    clhip_set_device(g->device);
    if (g->tx_mu_ok) {                                         /* a TX group: what is in flight lands, the members' seams stop calling back */
        pthread_mutex_lock(&g->tx_mu);
        tx_finish(g);
        for (size_t i = 0; g->dev && i < g->n; i++)
            if (g->dev[i] && g->dev[i]->smi->tx_settle_ctx == g) { g->dev[i]->smi->tx_settle = NULL; g->dev[i]->smi->tx_settle_ctx = NULL; }
        pthread_mutex_unlock(&g->tx_mu);
    }
    for (int k = 0; k < GRP_MAX_IN; k++) if (g->s_in[k]) clhip_stream_sync(g->s_in[k]);
    if (g->s_k) clhip_stream_sync(g->s_k);
    if (g->s_out) clhip_stream_sync(g->s_out);
    pool_stop(&g->pool);
    cl_group_unregister_buffers(g);
    ahead_cancel_all(g);                                       /* what was read ahead is pending again */
    for (size_t i = 0; g->lane_of && i < g->n; i++) {          /* the filters' carried state goes back to the streams' own objects */
        cl_stream *st = g->dev[i]->stream;
        if (st->iir_home_ctx == g) { iir_home(g, (int)i); st->iir_home = NULL; st->iir_home_ctx = NULL; }
    }
    for (int k = 0; k < g->n_lanes; k++)                       /* raw words standing in for a seam's persistent buffer: unpacked now, the lane's buffers go */
        for (int r = 0; g->lane[k].d_in[0] && r < g->lane[k].n; r++) cl_smi_restore_prev_words(g->dev[g->lane[k].member[r]]->smi, g->lane[k].channel);
    if (g->slab) {                                             /* the members' FIFOs move out before the slab goes */
        for (size_t i = 0; i < g->n; i++) {
            cl_smi *smi = g->dev[i]->smi;
            pthread_mutex_lock(&smi->fifo_mu);
            /* (a producer that holds an uncommitted reservation is writing into the slice: its commit is waited for) */
            while (smi->rx.external && smi->rx.reserved) pthread_cond_wait(&smi->fifo_fed, &smi->fifo_mu);
            if (smi->rx.external && smi->rx.data >= g->slab && smi->rx.data < g->slab + g->n * g->slab_slice) cl_fifo_leave(&smi->rx);
            pthread_mutex_unlock(&smi->fifo_mu);
        }
        clhip_host_free(g->slab);
    }
    for (int i = 0; i < g->n_lanes; i++) lane_free(&g->lane[i]);
    for (size_t i = 0; i < g->n_ev; i++) clhip_event_destroy(g->ev[i]);
    for (int k = 0; k < GRP_MAX_IN; k++) clhip_stream_destroy(g->s_in[k]);
    clhip_stream_destroy(g->s_k); clhip_stream_destroy(g->s_out);
    free(g->ev); free(g->lane); free(g->dev); free(g->lane_of); free(g->row_of);
    free(g->reg_base); free(g->reg_len); free(g->has_reg);
    if (g->tx_mu_ok) pthread_mutex_destroy(&g->tx_mu);
    free(g);
}

static char g_make_err[256];
const char *cl_group_last_error(const cl_group *g) { return g ? g->err : g_make_err; }

cl_group *cl_group_make(cl_device *const *devs, size_t n, const char *const *keys, const char *const *vals, size_t n_kwargs)
{
    g_make_err[0] = 0;
    if (!devs || !n) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: no devices"); return NULL; }
    for (size_t i = 0; i < n; i++) {
        if (!devs[i] || !devs[i]->stream || devs[i]->smi->device != devs[0]->smi->device) {
            cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: device %zu is missing or on another GPU (a group lives on one GPU)", i);
            return NULL;
        }
        for (size_t j = 0; j < i; j++)
            if (devs[j] == devs[i]) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: device %zu appears twice", i); return NULL; }
        if (devs[i]->stream->native_dir != devs[0]->stream->native_dir) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: device %zu is set up for the other direction (a group reads or writes)", i); return NULL; }
    }
    cl_group *g = (cl_group *)calloc(1, sizeof *g);
    if (!g) return NULL;
    g->device = devs[0]->smi->device;
    g->dir = devs[0]->stream->native_dir;
    { pthread_mutexattr_t at; pthread_mutexattr_init(&at); pthread_mutexattr_settype(&at, PTHREAD_MUTEX_RECURSIVE); pthread_mutex_init(&g->tx_mu, &at); pthread_mutexattr_destroy(&at); g->tx_mu_ok = 1; }
    clhip_set_device(g->device);
    g->n = n;
    g->dev = (cl_device **)calloc(n, sizeof *g->dev);
    g->lane_of = (int *)calloc(n, sizeof(int)); g->row_of = (int *)calloc(n, sizeof(int));
    g->lane = (lane_t *)calloc(n, sizeof *g->lane);
    g->reg_base = (uint8_t **)calloc(n, sizeof *g->reg_base);
    g->reg_len = (size_t *)calloc(n, sizeof *g->reg_len); g->has_reg = (uint8_t *)calloc(n, 1);
    if (!g->dev || !g->lane_of || !g->row_of || !g->lane || !g->reg_base || !g->reg_len || !g->has_reg) { cl_group_unmake(g); return NULL; }
    memcpy(g->dev, devs, n * sizeof *g->dev);
    const char *sub = kwget(keys, vals, n_kwargs, "SUBBATCH"), *ct = kwget(keys, vals, n_kwargs, "COPY_THREADS");
    g->sub = sub && atoi(sub) > 0 ? atoi(sub) : 0;     /* 0: by the lane's route (below) */
    const char *is = kwget(keys, vals, n_kwargs, "INGEST_STREAMS");
    g->n_in = is && atoi(is) >= 1 && atoi(is) <= GRP_MAX_IN ? atoi(is) : 2;     /* tools/group_ab.py, medians of 9 interleaved reps: CS16 4989 / 6508 / 5443 Msamples/s at 1 / 2 / 4, FIR64 + 3/2 3116 / 3131 / 2854 */
    g->ev_per = g->n_in + 2;
    const char *sk = kwget(keys, vals, n_kwargs, "SINK");
    g->sink_mapped = !(sk && !strcmp(sk, "copy"));
    { const char *ra = kwget(keys, vals, n_kwargs, "READAHEAD"); g->readahead = ra ? atoi(ra) : 2; }
    if (g->readahead < 0 || g->readahead > 2) g->readahead = 2;
    if (g->readahead == 2 && !g->sink_mapped) g->readahead = 1;     /* (results ahead are stored into the second mirror by the launches themselves) */
    int threads = ct ? atoi(ct) : 2;                     /* tools/group_ab.py (interleaved medians, FIR64 + 3/2 x 32): 0 / 1 / 2 / 3 / 4 / 8 threads -> 2099 / 2556 / 3158 / 3034 / 2925 / 2783 Msamples/s */
    if (threads < 0) threads = 0;
    if (threads > 16) threads = 16;
    /* lanes: members of one channel type with one stream configuration, in the caller's order */
    for (size_t i = 0; i < n; i++) {
        const cl_stream *st = devs[i]->stream;
        int li = -1;
        for (int k = 0; k < g->n_lanes && li < 0; k++) {
            const cl_stream *s0 = devs[g->lane[k].member[0]]->stream;
            if (g->dir == CL_SOAPY_SDR_TX) { if (s0->format == st->format && !s0->dsp.enabled && !st->dsp.enabled) li = k; }      /* (the TX words do not depend on the channel type) */
            else if (g->lane[k].channel == devs[i]->channel && s0->format == st->format && same_dsp(&s0->dsp, &st->dsp)) li = k;
        }
        if (li < 0) {
            li = g->n_lanes++;
            lane_t *l = &g->lane[li];
            l->channel = devs[i]->channel; l->format = st->format; l->dsp = st->dsp;
            l->route = g->dir == CL_SOAPY_SDR_TX ? (st->dsp.enabled ? ROUTE_TX_SINGLE : ROUTE_TX_PLAIN) : st->dsp.enabled ? ROUTE_PIPE : ROUTE_PLAIN;
            l->member = (int *)calloc(n, sizeof(int));
            if (!l->member) { cl_group_unmake(g); return NULL; }
        }
        lane_t *l = &g->lane[li];
        g->lane_of[i] = li; g->row_of[i] = l->n;
        l->member[l->n++] = (int)i;
    }
    size_t n_sub = 0;
    for (int k = 0; k < g->n_lanes; k++) {
        lane_t *l = &g->lane[k];
        const size_t nb = CL_NATIVE_BATCH_LEN, mtu = CL_NATIVE_MTU_SAMPLES;
        if (g->dir == CL_SOAPY_SDR_TX) {
            l->sub = g->sub && g->sub <= CLHIP_PACK_ROWS ? g->sub : CLHIP_PACK_ROWS;
            l->sub0 = n_sub;
            n_sub += ((size_t)l->n + (size_t)l->sub - 1) / (size_t)l->sub;
            l->fast = (uint8_t *)calloc((size_t)l->n, 1); l->src = (uint8_t **)calloc((size_t)l->n, sizeof(uint8_t *));
            if (!l->fast || !l->src) { cl_group_unmake(g); return NULL; }
            if (l->route != ROUTE_TX_PLAIN) continue;
            l->elem_bytes = fmt_bytes(l->format);
            l->tx_row = mtu * l->elem_bytes + 256;
            l->tx_h_in = (uint8_t *)clhip_host_alloc(2 * (size_t)l->n * l->tx_row); l->tx_d_in = (uint8_t *)clhip_malloc(2 * (size_t)l->n * l->tx_row);
            l->tx_pend = (uint8_t *)calloc((size_t)l->n, 1);
            for (int r = 0; r < l->n; r++) {
                cl_smi *smi = g->dev[l->member[r]]->smi;
                smi->tx_settle = tx_settle_hook; smi->tx_settle_ctx = g; smi->tx_settle_member = l->member[r];
            }
            if (!l->tx_h_in || !l->tx_d_in || !l->tx_pend) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: buffers for %d TX streams could not be allocated", l->n); cl_group_unmake(g); return NULL; }
            continue;
        }
        l->in_stride = nb + 256;
        l->up = l->dsp.enabled ? l->dsp.up : 1; l->down = l->dsp.enabled ? l->dsp.down : 1;
        if (l->route == ROUTE_PIPE) {
            l->elem_bytes = l->dsp.demod_fm ? 4 : 8;
            l->pipe = clhip_rx_pipe_create(l->n, l->channel, l->dsp.fir, l->dsp.n_fir, l->dsp.n_rs ? l->dsp.rs : NULL, l->dsp.n_rs, l->dsp.up,
                                           l->dsp.down, l->dsp.demod_fm ? CL_PIPE_OUT_FM_DEMOD : CL_PIPE_OUT_IQ);
            if (!l->pipe) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: %s", clhip_last_error()); cl_group_unmake(g); return NULL; }
            l->out_stride = ((mtu * (size_t)l->up + (size_t)l->down - 1) / (size_t)l->down + 2 + 31) & ~(size_t)31;
        } else {
            l->elem_bytes = fmt_bytes(l->format);
            l->out_stride = l->in_stride / 4;            /* clhip_smi_unpack: sample k of chunk c lands in slot c * stride / 4 + k */
        }
        const size_t out_bytes = (size_t)l->n * l->out_stride * l->elem_bytes + 256;
        const size_t in_bytes = (size_t)l->n * l->in_stride + 256;
        l->d_in[0] = (uint8_t *)clhip_malloc(in_bytes); l->d_in[1] = (uint8_t *)clhip_malloc(in_bytes); l->d_in[2] = (uint8_t *)clhip_malloc(in_bytes);
        l->cur_in = 0; l->prev_in = 1; l->next_in = 2;
        l->primed = (size_t *)calloc((size_t)l->n, sizeof(size_t)); l->primed_epoch = (unsigned *)calloc((size_t)l->n, sizeof(unsigned));
        l->ev_primed = clhip_event_create();
        l->d_out = (uint8_t *)clhip_malloc(out_bytes);
        l->h_out[0] = (uint8_t *)clhip_host_alloc(out_bytes);
        l->h_out[1] = g->readahead == 2 ? (uint8_t *)clhip_host_alloc(out_bytes) : NULL;
        const size_t offs_each = ((size_t)l->n + 15) & ~(size_t)15;
        l->h_offs[0] = (int32_t *)clhip_host_alloc(sizeof(int32_t) * 4 * offs_each);
        for (int q = 0; q < 4; q++) {
            l->h_offs[q] = l->h_offs[0] ? l->h_offs[0] + (size_t)q * offs_each : NULL;
            l->d_offs[q] = l->h_offs[0] ? (int32_t *)clhip_host_device_ptr(l->h_offs[q]) : NULL;
        }
        for (int q = 0; q < 2; q++) l->m_out[q] = l->h_out[q] ? (uint8_t *)clhip_host_device_ptr(l->h_out[q]) : NULL;
        l->done_ahead = (uint8_t *)calloc((size_t)l->n, 1); l->ahead_got = (long *)calloc((size_t)l->n, sizeof(long));
        l->direct = (uint8_t *)calloc((size_t)l->n, 1);
        l->ctx = (cl_read_ctx *)calloc((size_t)l->n, sizeof(cl_read_ctx));
        l->how = (uint8_t *)calloc((size_t)l->n, 1);
        l->sub0 = n_sub;
        l->fast = (uint8_t *)calloc((size_t)l->n, 1); l->len = (size_t *)calloc((size_t)l->n, sizeof(size_t));
        l->got = (long *)calloc((size_t)l->n, sizeof(long));
        l->src = (uint8_t **)calloc((size_t)l->n, sizeof(uint8_t *));
        l->ahead_mark = (uint8_t *)calloc((size_t)l->n, 1);
        if (!l->d_in[0] || !l->d_in[1] || !l->d_in[2] || !l->primed || !l->primed_epoch || !l->ev_primed || !l->d_out || !l->h_out[0] || (g->readahead == 2 && !l->h_out[1]) || !l->h_offs[0] || !l->d_offs[0] || !l->m_out[0] || !l->done_ahead || !l->ahead_got || !l->direct || !l->ctx || !l->how || !l->fast || !l->len || !l->got || !l->src || !l->ahead_mark) {
            cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: buffers for %d streams could not be allocated", l->n);
            cl_group_unmake(g);
            return NULL;
        }
        /* streams per launch.  Interleaved A/B with results computed ahead (profiles/r04/group_ab_subbatch_results_ahead.txt), 4 / 8 / 16:
         * FIR64 + 3/2 (1.5 MiB out per stream) 3994 / 3872 / 3859 Msamples/s; CS16 (0.5 MiB) 9874 / 10396 / 10119, FIR64 + FM demod (0.5 MiB)
         * 9489 / 10269 / 10343, CF32 (1 MiB) 5773 / 6034 / 6050 -- short launches (43 us for four CS16 streams) lie 6 ... 10 us apart */
        l->sub = g->sub ? g->sub : l->out_stride * l->elem_bytes >= ((size_t)3 << 19) ? 4 : 8;
        n_sub += ((size_t)l->n + (size_t)l->sub - 1) / (size_t)l->sub;
        l->n_subs = (l->n + l->sub - 1) / l->sub;
        if (l->route == ROUTE_PLAIN) {
            l->giir = (clhip_iir **)calloc((size_t)3 * (size_t)l->n_subs, sizeof(clhip_iir *)); l->iir_own = (uint8_t *)calloc((size_t)3 * (size_t)l->n, 1);
            l->sub_ft = (uint8_t *)calloc((size_t)l->n_subs, 1); l->ahead_ft = (uint8_t *)calloc((size_t)l->n_subs, 1); l->sub_verdict = (uint8_t *)calloc((size_t)l->n_subs, 1);
            l->d_f = (int16_t *)clhip_malloc((size_t)l->n * l->out_stride * 4 + 256);
            if (!l->giir || !l->iir_own || !l->sub_ft || !l->ahead_ft || !l->sub_verdict || !l->d_f) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: filter buffers"); cl_group_unmake(g); return NULL; }
            for (int r = 0; r < l->n; r++) {
                cl_stream *st = g->dev[l->member[r]]->stream;
                st->iir_home = iir_home; st->iir_home_ctx = g; st->iir_home_member = l->member[r];
            }
        }
    }
    for (int k = 0; k < g->n_in; k++) g->s_in[k] = clhip_stream_create();
    g->s_k = clhip_stream_create(); g->s_out = clhip_stream_create();
    {
        /* kwarg SLAB_MB: MiB of pinned FIFO room per member (default 8 = sixteen native batches; 0 = the members keep their own
         * buffers and every batch comes in by a copy of its own) */
        const char *sm = kwget(keys, vals, n_kwargs, "SLAB_MB");
        const size_t mb = sm ? (size_t)atol(sm) : 8;
        g->slab_slice = mb << 20;
        g->slab = mb && g->dir == CL_SOAPY_SDR_RX ? (uint8_t *)clhip_host_alloc(g->n * g->slab_slice) : NULL;
        size_t slot = 0;
        for (int k = 0; g->slab && k < g->n_lanes; k++)
            for (int r = 0; r < g->lane[k].n; r++, slot++) {
                cl_smi *smi = g->dev[g->lane[k].member[r]]->smi;
                cl_smi_readahead_cancel(smi);
                pthread_mutex_lock(&smi->fifo_mu);
                if (!cl_fifo_front_len(&smi->rx)) cl_fifo_adopt(&smi->rx, g->slab + slot * g->slab_slice, g->slab_slice);   /* (too full: stays where it is) */
                pthread_mutex_unlock(&smi->fifo_mu);
            }
    }
    g->n_sub = n_sub;
    g->n_ev = 2 * (size_t)g->ev_per * n_sub;                  /* two sets: this call's, and the one the launches made ahead record into */
    g->ev = (void **)calloc(g->n_ev, sizeof(void *));
    int bad = !g->s_k || !g->s_out || !g->ev;
    for (int k = 0; k < g->n_in; k++) bad |= !g->s_in[k];
    for (size_t i = 0; !bad && i < g->n_ev; i++) bad = !(g->ev[i] = clhip_event_create());
    if (bad || pool_start(&g->pool, threads, 16 * n + 64)) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: streams / events / threads"); cl_group_unmake(g); return NULL; }
    return g;
}

size_t cl_group_size(const cl_group *g) { return g ? g->n : 0; }
void cl_group_getStats(const cl_group *g, cl_group_stats *out) { if (out) { if (g) *out = g->stats; else memset(out, 0, sizeof *out); } }

/* Work made ahead that nobody will use -- a copy into an input row, a launch that reads that row and an offsets table and writes a
 * mirror row -- may still be queued when its row is staged, copied and launched over afresh, or when the buffers come round again.
 * Giving up work made ahead is rare (the client went another way): the streams are simply drained before anything takes its place. */
static void settle(cl_group *g)
{
    if (!g->stale) return;
    for (int k = 0; k < g->n_in; k++) if (g->s_in[k]) clhip_stream_sync(g->s_in[k]);
    if (g->s_k) clhip_stream_sync(g->s_k);
    g->stale = 0;
}

/* everything read or computed ahead is given back: the bytes are pending in the members' FIFOs again, the runs are taken back */
static void ahead_cancel_all(cl_group *g)
{
    for (int k = 0; k < g->n_lanes; k++) {
        lane_t *l = &g->lane[k];
        for (int r = 0; r < l->n; r++) {
            if (!l->primed || !l->primed[r]) continue;
            cl_smi *smi = g->dev[l->member[r]]->smi;
            if (smi->foreign_ahead == l->primed[r] && smi->foreign_epoch == l->primed_epoch[r]) cl_smi_foreign_cancel(smi);
            if (l->done_ahead[r] && l->pipe) clhip_rx_pipe_unrun_stream(l->pipe, r, l->primed[r] / 4);
            l->primed[r] = 0; l->done_ahead[r] = 0;
            g->stale = 1;
        }
    }
    settle(g);
    for (int k = 0; k < g->n_lanes; k++)
        for (int sb = 0; g->lane[k].ahead_ft && sb < g->lane[k].n_subs; sb++) giir_ahead_drop(g, &g->lane[k], sb);
}

/* Client buffers the members' outputs may be written into by the copy engine directly (no pinned mirror, no memcpy): one
 * buffer per member, registered with the GPU HERE, explicitly, for as long as the registration stands -- the client keeps them
 * allocated until cl_group_unregister_buffers / cl_group_unmake.  A call whose buffs[i] lies inside member i's registered
 * range takes the direct route; any other pointer takes the mirror route. */
int cl_group_register_buffers(cl_group *g, void *const *buffs, size_t bytes_each)
{
    if (!g || !buffs || !bytes_each) return -1;
    clhip_set_device(g->device);
    ahead_cancel_all(g);                                       /* (results computed ahead lie in the mirror: the next call decides afresh) */
    cl_group_unregister_buffers(g);
    /* registrations are whole pages and neighbouring heap buffers share pages: ranges that touch are registered as one */
    const uintptr_t pg = 4096;
    for (size_t i = 0; i < g->n; i++) {
        uintptr_t lo = (uintptr_t)buffs[i] & ~(pg - 1), hi = ((uintptr_t)buffs[i] + bytes_each + pg - 1) & ~(pg - 1);
        for (size_t k = 0; k < g->n_reg; k++) {
            const uintptr_t b0 = (uintptr_t)g->reg_base[k], b1 = b0 + g->reg_len[k];
            if (lo > b1 || hi < b0) continue;                      /* (adjacent ranges merge too: fewer registrations) */
            if (b0 < lo) lo = b0;
            if (b1 > hi) hi = b1;
            g->reg_base[k] = g->reg_base[g->n_reg - 1]; g->reg_len[k] = g->reg_len[g->n_reg - 1]; g->n_reg--;
            k = (size_t)-1;
        }
        g->reg_base[g->n_reg] = (uint8_t *)lo; g->reg_len[g->n_reg] = hi - lo; g->n_reg++;
    }
    for (size_t k = 0; k < g->n_reg; k++)
        if (!clhip_host_register(g->reg_base[k], g->reg_len[k])) {
            cl_seterr(g->err, sizeof g->err, "cl_group_register_buffers: %zu bytes at %p could not be registered (%s)", g->reg_len[k], (void *)g->reg_base[k],
                      clhip_last_error());
            for (size_t q = 0; q < k; q++) clhip_host_unregister(g->reg_base[q]);
            g->n_reg = 0;
            return -1;
        }
    memset(g->has_reg, 1, g->n);
    return 0;
}

void cl_group_unregister_buffers(cl_group *g)
{
    if (!g || !g->reg_base || !g->n_reg) return;
    clhip_set_device(g->device);
    ahead_cancel_all(g);
    if (g->s_out) clhip_stream_sync(g->s_out);          /* nothing may still be writing them */
    if (g->s_k) clhip_stream_sync(g->s_k);
    for (size_t k = 0; k < g->n_reg; k++) clhip_host_unregister(g->reg_base[k]);
    g->n_reg = 0;
    memset(g->has_reg, 0, g->n);
}

static int registered(const cl_group *g, int m, const void *p, size_t bytes)
{
    const uint8_t *q = (const uint8_t *)p;
    if (!g->has_reg[m]) return 0;
    for (size_t k = 0; k < g->n_reg; k++)
        if (q >= g->reg_base[k] && q + bytes <= g->reg_base[k] + g->reg_len[k]) return 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------- the call */
/* Does member m's next read() qualify for the batched route?  Under its FIFO lock: `want` bytes pending as ONE in-place read()
 * whose head carries the sync pattern (caribou_smi_find_buffer_offset returns 0 exactly then, caribou_smi.c:235-292 -- the
 * bytes are in pinned host memory, so the host knows before the device has looked); they are staged and their copy to the
 * lane's row is queued while they cannot move.  Everything else -- short, ragged or slipped reads, bytes given back earlier,
 * reader threads, the IIR, debug modes -- is the single-stream route's business. */
/* ---- the reference's low-pass over whole sub-batches ----
 * Stream::ReadSamples(int16*) (CaribouliteStream.cpp:282-301) runs the selected Butterworth over every sample of the read on the
 * client's thread; a sub-batch whose members all have the SAME filter selected and are all in sync goes through ONE multi-stream filter
 * launch fed from the raw words (clhip_iir_run_smi).  The carried state of a member's three filters persists for the life of its
 * Stream and is not reset when the selection changes (:84-91,127-141): it is moved, not copied -- into the group's object when the
 * member first takes the batched filter route, back into the stream's own object the moment that object is about to be used
 * (cl_soapy.c: filter_source calls iir_home) or the group goes. */
static int giir_index(const lane_t *l, int row, int ft) { return (ft - 1) * l->n_subs + row / l->sub; }

/* a filter launch made ahead over sub-batch `sb` is given up: waited for, its object's state put back, its rows' results forgotten
 * (what was READ ahead stands: the rows are launched over in the call) */
static void giir_ahead_drop(cl_group *g, lane_t *l, int sb)
{
    if (!l->ahead_ft || !l->ahead_ft[sb]) return;
    g->stale = 1; settle(g);
    clhip_iir_unrun(l->giir[(l->ahead_ft[sb] - 1) * l->n_subs + sb]);
    l->ahead_ft[sb] = 0;
    const int a = sb * l->sub, e = a + l->sub < l->n ? a + l->sub : l->n;
    for (int r = a; r < e; r++) l->done_ahead[r] = 0;
}

static void iir_home(void *ctx, int member)
{
    cl_group *g = (cl_group *)ctx;
    lane_t *l = &g->lane[g->lane_of[member]];
    const int row = g->row_of[member];
    if (!l->iir_own) return;
    giir_ahead_drop(g, l, row / l->sub);                       /* (a filter launch made ahead has advanced the state: taken back first) */
    for (int ft = 1; ft <= 3; ft++) {
        if (!l->iir_own[3 * row + ft - 1]) continue;
        clhip_iir *obj = l->giir[giir_index(l, row, ft)];
        const int a = row / l->sub * l->sub, cnt = (a + l->sub < l->n ? a + l->sub : l->n) - a;
        double *st = (double *)malloc(sizeof(double) * 16 * (size_t)cnt);
        if (st && obj && clhip_iir_get_state(obj, st) == 0) clhip_iir_set_state(g->dev[member]->stream->iir[ft - 1], st + 16 * (row - a));
        free(st);
        l->iir_own[3 * row + ft - 1] = 0;
    }
}

/* the filter object of sub-batch [a, e) for filter `ft`, holding the carried state of all its members (NULL: a runtime error) */
static clhip_iir *giir_get(cl_group *g, lane_t *l, int a, int e, int ft)
{
    clhip_iir **slot = &l->giir[giir_index(l, a, ft)];
    if (!*slot) {
        *slot = clhip_iir_create(g->dev[l->member[a]]->stream->sos[ft - 1], 3, e - a);
        if (!*slot) return NULL;
        if (g->iir_polls_set) clhip_iir_set_poll_bound(*slot, g->iir_polls);
    }
    int need = 0;
    for (int r = a; r < e; r++) need |= !l->iir_own[3 * r + ft - 1];
    if (need) {
        double *st = (double *)malloc(sizeof(double) * 16 * (size_t)(e - a));
        if (!st || clhip_iir_get_state(*slot, st)) { free(st); return NULL; }
        for (int r = a; r < e; r++) {
            if (l->iir_own[3 * r + ft - 1]) continue;
            if (clhip_iir_get_state(g->dev[l->member[r]]->stream->iir[ft - 1], st + 16 * (r - a))) { free(st); return NULL; }
            l->iir_own[3 * r + ft - 1] = 1;
        }
        const int bad = clhip_iir_set_state(*slot, st);
        free(st);
        if (bad) return NULL;
    }
    return *slot;
}

void cl_group_set_iir_poll_bound(cl_group *g, int polls)
{
    if (!g) return;
    g->iir_polls = polls; g->iir_polls_set = 1;
    for (int k = 0; k < g->n_lanes; k++)
        for (int i = 0; g->lane[k].giir && i < 3 * g->lane[k].n_subs; i++) clhip_iir_set_poll_bound(g->lane[k].giir[i], polls);
}

/* the filter launch(es) of sub-batch [a, e): raw words in `in` rows -> filtered samples in the lane's format in `outb` rows (the mapped
 * mirror); redo = the scan path after an overrun (or an object that is on it): unpack, filter, convert */
static int giir_launch(cl_group *g, lane_t *l, clhip_iir *obj, int a, int e, size_t want, uint8_t *in, uint8_t *outb, int offs_table)
{
    const size_t n = want / 4, stride = l->out_stride;
    int16_t *mirror16 = (int16_t *)(outb + (size_t)a * stride * l->elem_bytes), *f16 = l->d_f + 2 * (size_t)a * stride;
    int16_t *dst = l->format == CL_FORMAT_CS16 ? mirror16 : f16;
    int rc = clhip_iir_run_smi(obj, l->channel, in + (size_t)a * l->in_stride, dst, stride, n, g->s_k);
    if (rc == -2) {
        for (int r = a; r < e; r++) l->h_offs[offs_table][r] = 0;
        rc = clhip_smi_unpack(l->channel, in + (size_t)a * l->in_stride, (size_t)(e - a - 1) * l->in_stride + want, l->in_stride, want, e - a,
                              l->d_offs[offs_table] + a, CL_FORMAT_CS16, f16, NULL, g->s_k) ||
             clhip_iir_run(obj, f16, dst, stride, n, g->s_k);
    }
    if (rc) return -1;
    g->stats.launches++;
    if (dst == f16) return clhip_convert_from_cs16(f16, (size_t)(e - a - 1) * stride + n, l->format, outb + (size_t)a * stride * l->elem_bytes, g->s_k);
    return 0;
}

/* The verdict of the filter launch over THIS call's batch of sub-batch [a, e) (clhip_iir_status behind its event): a single-pass launch
 * that gave up has its state back where it was and the object on its scan path -- the sub-batch is filtered again, once, here
 * (CaribouliteStream.cpp has no such case: its loop cannot fail; a second failure delivers 0 elements like any read error,
 * :266-276).  Asked once per call -- before the object is launched over the NEXT batch (one call in flight per object: a launch
 * behind one that gave up would start from its garbage), or when the sub-batch's rows are handed out.  1 / 2 / 3 as sub_verdict. */
static int giir_verdict(cl_group *g, lane_t *l, int a, int e)
{
    const int sb = a / l->sub;
    if (l->sub_verdict[sb]) return l->sub_verdict[sb];
    int v = 1;
    if (clhip_event_sync(ev_of(g, l, l->set, a)[g->n_in + 1])) v = 3;
    else {
        clhip_iir *obj = l->giir[giir_index(l, a, l->sub_ft[sb])];
        if (clhip_iir_status(obj)) {
            for (int r = a; r < e; r++) g->dev[l->member[r]]->stream->stats.iir_overruns++;
            if (giir_launch(g, l, obj, a, e, l->want, l->d_in[l->cur_in], l->m_out[l->cur_m], 2 * l->set + 1) || clhip_stream_sync(g->s_k)) v = 3;
            else if (clhip_iir_status(obj)) v = 2;
        }
    }
    return l->sub_verdict[sb] = (uint8_t)v;
}

/* may member `row` take the batched route at all in this call? */
static int qualifies(const cl_group *g, const lane_t *l, int row, size_t want, int allow_filter)
{
    const cl_device *dev = g->dev[l->member[row]];
    const cl_stream *st = dev->stream;
    const cl_smi *smi = dev->smi;
    if (st->use_async || (st->filter_type != CL_DIGFILT_NONE && !allow_filter) || smi->debug_mode != CL_SMI_DEBUG_NONE || st->native_dir != CL_SOAPY_SDR_RX) return 0;
    if (st->format != l->format || !want || (want & 15) || want > smi->native_batch_len || (smi->max_read && smi->max_read < want)) return 0;
    return 1;
}

/* stage member `row`'s next `want` pending bytes in place if they are one in-sync batch: l->src[row] says where they lie.  Returns 1
 * with the member's FIFO lock HELD -- the bytes must not move before the copy that reads them is queued (copies_queue releases it) */
static int stage_row(cl_group *g, lane_t *l, int row, size_t want, void *s_in, int slot)
{
    cl_smi *smi = g->dev[l->member[row]]->smi;
    uint8_t *src = NULL;
    l->src[row] = NULL;
    pthread_mutex_lock(&smi->fifo_mu);
    if (!cl_fifo_front_len(&smi->rx) && smi->rx.len >= want) {
        smi->rx.dma_stream[slot] = s_in;                       /* a feeder that has to move the buffer waits for this copy first */
        const size_t got = cl_fifo_stage(&smi->rx, want, &src);
        if (got == want && cl_smi_head_in_sync(src, got)) { l->src[row] = src; return 1; }
        if (got) cl_fifo_unstage(&smi->rx, got);
    }
    pthread_mutex_unlock(&smi->fifo_mu);
    return 0;
}

static int on_phase_0(const lane_t *l, int row)
{
    /* (off polyphase phase 0 the pipe runs its generic kernels: such streams go one by one) */
    return l->route != ROUTE_PIPE || clhip_rx_pipe_stream_total(l->pipe, row) % (2ull * (unsigned long long)l->down) == 0;
}

/* 3 = the previous call read this batch ahead AND launched over it: its results are in the current mirror (or on their way);
 * 2 = the previous call read it ahead (staged in the FIFO, copied to d_in[cur_in]): to be launched over;
 * 1 = staged now, its copy still to be queued (FIFO lock held); 0 = not on the batched route in this call */
static int try_stage(cl_group *g, lane_t *l, int row, size_t want, void *s_in, int allow_filter)
{
    cl_smi *smi = g->dev[l->member[row]]->smi;
    if (l->primed[row]) {
        const int intact = smi->foreign_ahead == l->primed[row] && smi->foreign_epoch == l->primed_epoch[row];
        const size_t had = l->primed[row];
        const int was_done = l->done_ahead[row];
        l->primed[row] = 0; l->done_ahead[row] = 0; l->src[row] = NULL;
        /* (a run made ahead has advanced the stream's counter already: its phase was checked when it was made) */
        if (intact && had == want && qualifies(g, l, row, want, allow_filter) && (was_done || on_phase_0(l, row))) {
            smi->foreign_ahead = 0;                            /* this call's batch now: staged, the oldest unconfirmed bytes */
            cl_smi_ahead_note(smi);
            if (was_done && allow_filter && !(l->ahead_ft && l->ahead_ft[row / l->sub])) { g->stale = 1; settle(g); return 2; }   /* (computed ahead WITHOUT the low-pass that has been selected since: only the input stands) */
            return was_done ? 3 : 2;
        }
        if (intact) cl_smi_foreign_cancel(smi);                 /* another length, or off the batched route: pending again, in order */
        if (was_done && l->pipe) clhip_rx_pipe_unrun_stream(l->pipe, row, had / 4);
        g->stale = 1;
        settle(g);
    }
    l->src[row] = NULL;
    if (!qualifies(g, l, row, want, allow_filter) || !on_phase_0(l, row)) return 0;
    cl_smi_readahead_cancel(smi);                              /* what a single-stream call staged ahead is pending again */
    return stage_row(g, l, row, want, s_in, 1);
}

/* Queue the copies in of the staged rows [a, e) and let their FIFOs go again.  Neighbouring rows whose batches lie one slab slice
 * apart (members that are fed and read in step, the normal case) travel as ONE 2-D copy; any other row by a copy of its own.  A row
 * whose copy cannot be queued is unstaged and leaves the batched route.  Returns 0, or -1 on a runtime error. */
static int copies_queue(cl_group *g, lane_t *l, uint8_t *d_buf, int a, int e, size_t want, void *s_in)
{
    int rc = 0, r = a;
    while (r < e) {
        if (!l->fast[r] || !l->src[r]) { r++; continue; }     /* (not batched, or read ahead by the call before) */
        int r1 = r + 1;
        while (g->slab && r1 < e && l->fast[r1] && l->src[r1] && l->src[r1] == l->src[r] + (size_t)(r1 - r) * g->slab_slice) r1++;
        uint8_t *dst = d_buf + (size_t)r * l->in_stride;
        const int bad = r1 - r > 1 ? clhip_memcpy2d_h2d(dst, l->in_stride, l->src[r], g->slab_slice, want, (size_t)(r1 - r), s_in)
                                   : clhip_memcpy_h2d(dst, l->src[r], want, s_in);
        if (r1 - r > 1) g->stats.copies_2d++;
        for (int q = r; q < r1; q++) {
            cl_smi *smi = g->dev[l->member[q]]->smi;
            if (bad) { cl_fifo_unstage(&smi->rx, want); l->fast[q] = 0; rc = -1; }
            pthread_mutex_unlock(&smi->fifo_mu);
        }
        r = r1;
    }
    return rc;
}

static void confirm_staged(cl_smi *smi, size_t n)
{
    pthread_mutex_lock(&smi->fifo_mu);
    cl_fifo_confirm(&smi->rx, n);
    pthread_mutex_unlock(&smi->fifo_mu);
}

static void count_read(cl_stream *st, int ret)
{
    st->stats.read_calls++;
    if (ret > 0) st->stats.elements_read += (uint64_t)ret; else if (ret == 0) st->stats.reads_empty++;
}

/* A member off the batched route: its own device's single-stream call, with the group's pipe slot standing where the
 * device's own pipe would.  (What a re-sync finds in the slots it leaves untouched -- the reference's interm_native_buffer,
 * caribou_smi.c:382-389, CaribouliteStream.cpp:304-367 -- is the seam's business: the batched route leaves it the raw words of
 * the member's last batch, pass 3.) */
static int single_member(cl_group *g, lane_t *l, int row, void *out, size_t numElems, long timeoutUs)
{
    const int m = l->member[row];
    cl_device *dev = g->dev[m];
    cl_stream *st = dev->stream;
    g->stats.single_reads++;
    if (l->route == ROUTE_PLAIN || st->format != l->format || st->native_dir != CL_SOAPY_SDR_RX) {
        /* queued only: the caller runs every such member's first half before the first second half (cl_stream_read_end) */
        cl_stream_read_begin(dev, st, out, numElems, timeoutUs, &l->ctx[row]);
        return -1000;
    }
    /* pipe lane: Stream::Read (+ the low-pass) leaves the native samples on the device, the group's pipe slot runs from them */
    if (numElems > st->mtu_size) numElems = st->mtu_size;                        /* CaribouliteStream.cpp:306,328,351 */
    const int16_t *d_iq = NULL;
    const int res = cl_stream_read_native(dev, st, numElems, timeoutUs, &d_iq);
    if (res <= 0 || !d_iq) return 0;
    uint8_t *d_row = l->d_out + (size_t)row * l->out_stride * l->elem_bytes, *h_row = l->h_out[l->cur_m] + (size_t)row * l->out_stride * l->elem_bytes;
    const long got = clhip_rx_pipe_run_range(l->pipe, row, 1, CL_PIPE_IN_CS16, d_iq, 0, (size_t)res, d_row, 0, g->s_k);
    if (got < 0) { cl_seterr(g->err, sizeof g->err, "cl_group_readStream: %s", clhip_last_error()); return 0; }
    if (got == 0) return 0;
    const size_t bytes = (size_t)got * l->elem_bytes;
    if (registered(g, m, out, bytes)) {
        if (clhip_memcpy_d2h(out, d_row, bytes, g->s_k) || clhip_stream_sync(g->s_k)) return 0;
    } else {
        if (clhip_memcpy_d2h(h_row, d_row, bytes, g->s_k) || clhip_stream_sync(g->s_k)) return 0;
        memcpy(out, h_row, bytes);
    }
    return (int)got;
}

static void **ev_of(const cl_group *g, const lane_t *l, int set, int a)
{
    return g->ev + ((size_t)set * g->n_sub + l->sub0 + (size_t)(a / l->sub)) * (size_t)g->ev_per;
}

/* The launches of one sub-batch over the rows [a, e) marked in `run` (maximal runs of neighbours: one fused launch each / one unpack
 * launch with the other rows masked), from the raw words in `in` into `outb` (both row-strided); got[r] = elements per marked row. */
static int launch_rows(cl_group *g, lane_t *l, int a, int e, const uint8_t *run, size_t want, uint8_t *in, uint8_t *outb, int offs_table, long *got)
{
    if (l->route == ROUTE_PIPE) {
        int r = a;
        while (r < e) {
            if (!run[r]) { r++; continue; }
            int r1 = r + 1;
            while (r1 < e && run[r1]) r1++;
            const long n = clhip_rx_pipe_run_range(l->pipe, r, r1 - r, CL_PIPE_IN_SMI_WORDS, in + (size_t)r * l->in_stride, l->in_stride / 4, want / 4,
                                                   outb + (size_t)r * l->out_stride * l->elem_bytes, l->out_stride, g->s_k);
            if (n < 0) return -1;
            for (int q = r; q < r1; q++) got[q] = n;
            g->stats.launches++;
            r = r1;
        }
        return 0;
    }
    /* caribou_smi_rx_data_analyze at offset 0 + the format conversion, one launch over the sub-batch's rows (a row that is not
     * marked has offset -1: the kernel writes nothing for it) */
    int any = 0;
    for (int r = a; r < e; r++) { l->h_offs[offs_table][r] = run[r] ? 0 : -1; if (run[r]) { got[r] = (long)(want / 4); any = 1; } }
    if (!any) return 0;
    g->stats.launches++;
    return clhip_smi_unpack(l->channel, in + (size_t)a * l->in_stride, (size_t)(e - a - 1) * l->in_stride + want, l->in_stride, want, e - a,
                            l->d_offs[offs_table] + a, l->format, outb + (size_t)a * l->out_stride * l->elem_bytes, NULL, g->s_k);
}

int cl_group_readStream(cl_group *g, void *const *buffs, size_t numElems, int *rets, long timeoutUs)
{
    if (!g || !buffs || !rets) return -1;
    if (g->dir != CL_SOAPY_SDR_RX) { cl_seterr(g->err, sizeof g->err, "cl_group_readStream: the group's devices are set up for TX"); return -1; }
    clhip_set_device(g->device);
    g->err[0] = 0;
    g->stats.calls++;
    for (size_t i = 0; i < g->n; i++) rets[i] = 0;
    if (!numElems) return 0;
    for (int k = 0; k < g->n_lanes; k++) { memset(g->lane[k].fast, 0, (size_t)g->lane[k].n); g->lane[k].queued = 0; }
    int hard = 0;
    struct timespec t0, t1, t2, t3;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    /* ---- pass 1: what the previous call did not do ahead -- stage, copy in, launch, copy out: everything queued, nothing waited for */
    for (int k = 0; k < g->n_lanes && !hard; k++) {
        lane_t *l = &g->lane[k];
        const size_t mtu = CL_NATIVE_MTU_SAMPLES;
        /* CS16 is not clamped to the MTU by the reference (CaribouliteStream.cpp:282-301): longer calls are chunk loops, one by one */
        const size_t n_el = l->format == CL_FORMAT_CS16 && l->route == ROUTE_PLAIN ? numElems : (numElems > mtu ? mtu : numElems);
        const size_t want = n_el <= mtu ? n_el * 4 : 0;
        { const int p = l->prev_in; l->prev_in = l->cur_in; l->cur_in = l->next_in; l->next_in = p; }   /* what was read ahead is this call's input */
        if (g->readahead == 2) { l->cur_m ^= 1; l->set ^= 1; }   /* ... and what was computed ahead went to this mirror, behind these events */
        l->want = want;
        if (l->pipe && !l->epoch_open && clhip_rx_pipe_epoch_begin(l->pipe)) { hard = 1; break; }
        l->epoch_open = 1;
        int waits_primed = 0;
        uint8_t *in = l->d_in[l->cur_in];
        for (int a = 0; a < l->n && !hard; a += l->sub) {
            const int e = a + l->sub < l->n ? a + l->sub : l->n;
            void **ev_in = ev_of(g, l, l->set, a), *ev_k = ev_in[g->n_in], *ev_out = ev_in[g->n_in + 1];
            const size_t b = l->sub0 + (size_t)(a / l->sub);
            void *s_in = g->s_in[b % (size_t)g->n_in];          /* the sub-batches take turns on the ingest streams */
            int from_ahead = 0, any_run = 0, any_copy = 0;
            l->queued++;
            /* the sub-batch's low-pass: the same one selected on every member (and no registered buffers: the mirror route) -> one filter
             * launch over the sub-batch; members with a filter in a mixed sub-batch go through their own devices */
            int ft = 0;
            if (l->route == ROUTE_PLAIN && g->sink_mapped) {
                ft = g->dev[l->member[a]]->stream->filter_type;
                for (int r = a; r < e; r++)
                    if (g->dev[l->member[r]]->stream->filter_type != ft || g->has_reg[l->member[r]]) ft = 0;
            }
            if (l->sub_ft) { l->sub_ft[a / l->sub] = 0; l->sub_verdict[a / l->sub] = 0; }
            for (int r = a; r < e; r++) {
                const int how = try_stage(g, l, r, want, s_in, ft > 0);
                l->how[r] = (uint8_t)how;
                l->fast[r] = (uint8_t)(how != 0);
                l->ahead_mark[r] = (uint8_t)(how == 1 || how == 2);      /* (here: rows to launch over in this call) */
                from_ahead |= how == 2;
                g->stats.ahead_reads += how >= 2;
                l->got[r] = how == 3 ? l->ahead_got[r] : 0;
                l->direct[r] = 0;
            }
            int filtered_ahead = 0;
            if (l->ahead_ft && l->ahead_ft[a / l->sub]) {
                /* the previous call filtered this sub-batch ahead: good for this call if the same filter is still selected on everybody and
                 * every row's batch is still the one it read ahead -- otherwise the launch is taken back (the state it advanced), the rows
                 * that are still theirs keep their input */
                int all3 = ft == l->ahead_ft[a / l->sub];
                for (int r = a; r < e; r++) all3 &= l->how[r] == 3;
                if (all3) { filtered_ahead = 1; l->ahead_ft[a / l->sub] = 0; }
                else {
                    giir_ahead_drop(g, l, a / l->sub);
                    for (int r = a; r < e; r++)
                        if (l->how[r] == 3) { l->how[r] = 2; l->ahead_mark[r] = 1; l->got[r] = 0; from_ahead = 1; }
                }
            }
            if (ft > 0 && !filtered_ahead) {                   /* all of them or none: a member that is short, out of sync or off the route sends everybody home */
                int all = 1;
                for (int r = a; r < e; r++) all &= l->how[r] == 1 || l->how[r] == 2;
                if (!all) {
                    for (int r = a; r < e; r++) {
                        if (!l->fast[r]) continue;
                        cl_smi *smi = g->dev[l->member[r]]->smi;
                        if (l->how[r] != 1) pthread_mutex_lock(&smi->fifo_mu);       /* (staged just now: the lock is still held) */
                        cl_fifo_unstage(&smi->rx, want);
                        pthread_mutex_unlock(&smi->fifo_mu);
                        l->fast[r] = 0; l->ahead_mark[r] = 0; l->src[r] = NULL;
                    }
                    ft = 0;
                }
            }
            if (copies_queue(g, l, in, a, e, want, s_in)) hard = 1;     /* (a row whose copy cannot be queued leaves the batched route: fast = 0) */
            if (ft > 0) for (int r = a; r < e; r++) if (!l->fast[r]) hard = 1;   /* (a copy that could not be queued inside a filter sub-batch: a runtime error) */
            if (filtered_ahead) l->sub_ft[a / l->sub] = (uint8_t)ft;                /* (nothing to launch; its verdict is asked with the others', pass 3) */
            for (int r = a; r < e; r++) {
                l->ahead_mark[r] = (uint8_t)(l->ahead_mark[r] && l->fast[r]);
                l->len[r] = l->fast[r] ? want : 0;
                any_run |= l->ahead_mark[r];
                any_copy |= l->ahead_mark[r] && l->src[r] != NULL;
            }
            if (!any_run) continue;                            /* nothing to launch: the results are there (behind the event the previous call recorded), or there are none */
            if (from_ahead && !waits_primed) { hard = hard || clhip_stream_wait_event(g->s_k, l->ev_primed); waits_primed = 1; }
            if (any_copy && !hard) hard = clhip_event_record(ev_in[0], s_in) || clhip_stream_wait_event(g->s_k, ev_in[0]);
            /* where the sub-batch's launch stores: the mapped pinned mirror itself (its stores cross PCIe as the kernel produces them --
             * no second hop, no copy-engine call: tools/microbench/pcie_duplex.hip) unless one of its rows has a registered client
             * buffer, which the copy engine fills from the device buffer */
            int mapped = g->sink_mapped;
            for (int r = a; r < e && mapped; r++)
                if (l->fast[r] && g->has_reg[l->member[r]]) mapped = 0;
            uint8_t *outb = mapped ? l->m_out[l->cur_m] : l->d_out;
            if (!hard && ft > 0) {
                clhip_iir *obj = giir_get(g, l, a, e, ft);
                if (!obj || giir_launch(g, l, obj, a, e, want, in, outb, 2 * l->set + 1)) hard = 1;
                for (int r = a; r < e; r++) l->got[r] = (long)(want / 4);
                l->sub_ft[a / l->sub] = (uint8_t)ft;
            } else
            if (!hard && launch_rows(g, l, a, e, l->ahead_mark, want, in, outb, 2 * l->set + 1, l->got)) hard = 1;
            if (mapped) { hard = hard || clhip_event_record(ev_out, g->s_k); continue; }     /* "arrived" = the launch has ended */
            hard = hard || clhip_event_record(ev_k, g->s_k) || clhip_stream_wait_event(g->s_out, ev_k);
            /* out: rows whose client buffer is registered leave for it directly; the others in blocks of neighbours into the mirror */
            for (int r = a; r < e && !hard; ) {
                if (!l->ahead_mark[r] || l->got[r] <= 0) { r++; continue; }
                const size_t bytes = (size_t)l->got[r] * l->elem_bytes;
                if (registered(g, l->member[r], buffs[l->member[r]], bytes)) {
                    hard = clhip_memcpy_d2h(buffs[l->member[r]], l->d_out + (size_t)r * l->out_stride * l->elem_bytes, bytes, g->s_out);
                    l->direct[r] = 1; r++;
                    continue;
                }
                int hi = r;
                while (hi + 1 < e && l->ahead_mark[hi + 1] && l->got[hi + 1] > 0 &&
                       !registered(g, l->member[hi + 1], buffs[l->member[hi + 1]], (size_t)l->got[hi + 1] * l->elem_bytes)) hi++;
                const size_t o = (size_t)r * l->out_stride * l->elem_bytes;
                hard = clhip_memcpy_d2h(l->h_out[l->cur_m] + o, l->d_out + o, (size_t)(hi - r) * l->out_stride * l->elem_bytes + (size_t)l->got[hi] * l->elem_bytes, g->s_out);
                r = hi + 1;
            }
            hard = hard || clhip_event_record(ev_out, g->s_out);
        }
    }
    /* ---- pass 2: the members off the batched route, one by one, through their own devices; then the pipes' epochs end */
    for (int k = 0; k < g->n_lanes; k++) {
        lane_t *l = &g->lane[k];
        if (!l->queued) continue;                              /* (a runtime error before this lane's turn: its state stands) */
        for (int r = 0; r < l->n; r++) {                      /* first halves: everything queued on the members' own streams, nothing waited for */
            if (l->fast[r]) continue;
            const int m = l->member[r];
            rets[m] = hard ? 0 : single_member(g, l, r, buffs[m], numElems, timeoutUs);
        }
        for (int r = 0; r < l->n; r++) {                      /* second halves: synchronise, verdicts, deliver */
            if (l->fast[r]) continue;
            const int m = l->member[r];
            if (rets[m] == -1000) rets[m] = cl_stream_read_end(g->dev[m], g->dev[m]->stream, &l->ctx[r]);
            count_read(g->dev[m]->stream, rets[m]);
        }
        if (l->pipe && l->epoch_open && clhip_rx_pipe_epoch_end(l->pipe, g->s_k)) hard = 1;
        l->epoch_open = 0;
    }
    /* ---- ahead: before this call waits for its own results, the NEXT call's batches of the rows on the batched route -- staged in the
     * members' FIFOs (the newest staged bytes: cl_smi_foreign_cancel gives them back if anybody else reads that seam first), copied to
     * d_in[next_in], whole lanes at a time where the members are in step, and (READAHEAD=2) launched over into the other mirror: the
     * GPU goes on while the host hands this call's results out, and the next call finds its own computed. */
    for (int k = 0; g->readahead && k < g->n_lanes && !hard; k++) {
        lane_t *l = &g->lane[k];
        void *s_p = g->s_in[(size_t)k % (size_t)g->n_in];
        int any = 0;
        if (!l->queued || !l->want) continue;
        uint8_t *keep_fast = l->fast;                          /* (copies_queue walks l->fast: the rows staged ahead, for the moment) */
        l->fast = l->ahead_mark;
        for (int r = 0; r < l->n; r++) {
            l->ahead_mark[r] = keep_fast[r] && qualifies(g, l, r, l->want, l->route == ROUTE_PLAIN) && stage_row(g, l, r, l->want, s_p, 2) ? 1 : 0;
            any |= l->ahead_mark[r];
        }
        if (any && copies_queue(g, l, l->d_in[l->next_in], 0, l->n, l->want, s_p)) hard = 1;
        l->fast = keep_fast;
        if (!any) continue;
        for (int r = 0; r < l->n; r++) {
            if (!l->ahead_mark[r]) continue;                  /* (a row whose copy could not be queued was unstaged and unmarked) */
            cl_smi *smi = g->dev[l->member[r]]->smi;
            smi->foreign_ahead = l->want; cl_smi_ahead_note(smi);
            l->primed[r] = l->want; l->primed_epoch[r] = smi->foreign_epoch;
        }
        hard = hard || clhip_event_record(l->ev_primed, s_p);
        if (g->readahead < 2 || hard) continue;
        /* the launches: sub-batch by sub-batch like the call's own, into the OTHER mirror, behind the OTHER event set; sub-batches
         * with a registered client buffer among their members wait for the call (the copy engine needs the client's pointer) */
        if (l->pipe) { if (clhip_rx_pipe_epoch_begin(l->pipe)) { hard = 1; break; } l->epoch_open = 1; }
        int waited = 0;
        for (int a = 0; a < l->n && !hard; a += l->sub) {
            const int e = a + l->sub < l->n ? a + l->sub : l->n;
            int run = 0, ft = l->route == ROUTE_PLAIN ? g->dev[l->member[a]]->stream->filter_type : 0, whole = 1;
            for (int r = a; r < e; r++) {
                cl_stream *st = g->dev[l->member[r]]->stream;
                if (g->has_reg[l->member[r]] || st->filter_type != ft || (l->route != ROUTE_PLAIN && st->filter_type != CL_DIGFILT_NONE)) { run = 0; whole = 0; break; }   /* (the copy engine needs the client's pointer; a mixed sub-batch waits for the call) */
                l->done_ahead[r] = (uint8_t)(l->ahead_mark[r] && on_phase_0(l, r));
                run |= l->done_ahead[r]; whole &= l->done_ahead[r];
            }
            if (ft > 0 && !whole) run = 0;                     /* (a filter launch is over the whole sub-batch or not at all) */
            if (!run) { for (int r = a; r < e; r++) l->done_ahead[r] = 0; continue; }
            if (!waited) { hard = clhip_stream_wait_event(g->s_k, l->ev_primed); waited = 1; }
            if (ft > 0) {
                if (l->sub_ft[a / l->sub] && giir_verdict(g, l, a, e) == 3) hard = 1;      /* (this call's launch over the same object, first) */
                clhip_iir *obj = hard ? NULL : giir_get(g, l, a, e, ft);
                if (!obj || giir_launch(g, l, obj, a, e, l->want, l->d_in[l->next_in], l->m_out[l->cur_m ^ 1], 2 * (l->set ^ 1))) hard = 1;
                for (int r = a; r < e; r++) l->ahead_got[r] = (long)(l->want / 4);
                l->ahead_ft[a / l->sub] = (uint8_t)ft;
                hard = hard || clhip_event_record(ev_of(g, l, l->set ^ 1, a)[g->n_in + 1], g->s_k);
                continue;
            }
            if (!hard && launch_rows(g, l, a, e, l->done_ahead, l->want, l->d_in[l->next_in], l->m_out[l->cur_m ^ 1], 2 * (l->set ^ 1), l->ahead_got)) hard = 1;
            hard = hard || clhip_event_record(ev_of(g, l, l->set ^ 1, a)[g->n_in + 1], g->s_k);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    /* ---- pass 3: as the sub-batches arrive, their bytes are consumed for good and their rows go to the clients */
    for (int k = 0; k < g->n_lanes; k++) {
        lane_t *l = &g->lane[k];
        int sb = 0;
        for (int a = 0; a < l->n; a += l->sub, sb++) {
            const int e = a + l->sub < l->n ? a + l->sub : l->n;
            int any = 0;
            for (int r = a; r < e; r++) any |= l->fast[r];
            if (!any) continue;
            int arrived, filter_failed = 0;
            if (l->sub_ft && l->sub_ft[sb] && sb < l->queued && !hard) {
                const int v = giir_verdict(g, l, a, e);
                arrived = v != 3; filter_failed = v == 2;
            } else
                arrived = sb < l->queued && !hard && clhip_event_sync(ev_of(g, l, l->set, a)[g->n_in + 1]) == 0;
            if (!arrived) hard = 1;
            for (int r = a; r < e; r++) {
                if (!l->fast[r]) continue;
                const int m = l->member[r];
                cl_device *dev = g->dev[m];
                if (filter_failed) l->got[r] = 0;                  /* (consumed, nothing delivered) */
                if (!arrived) {                                    /* a runtime error: nothing is delivered, nothing is consumed */
                    for (int q = 0; q < g->n_in; q++) clhip_stream_sync(g->s_in[q]);
                    if (l->primed[r]) {                            /* (the newest staged bytes first) */
                        cl_smi_foreign_cancel(dev->smi);
                        if (l->done_ahead[r] && l->pipe) clhip_rx_pipe_unrun_stream(l->pipe, r, l->primed[r] / 4);
                        l->primed[r] = 0; l->done_ahead[r] = 0;
                        g->stale = 1;
                    }
                    pthread_mutex_lock(&dev->smi->fifo_mu);
                    cl_fifo_unstage(&dev->smi->rx, l->len[r]);
                    pthread_mutex_unlock(&dev->smi->fifo_mu);
                    l->fast[r] = 0;
                    count_read(dev->stream, 0);
                    continue;
                }
                confirm_staged(dev->smi, l->len[r]);
                dev->smi->stat_samples += (uint64_t)(l->len[r] / 4);
                /* the seam's persistent int16 buffer (the slots a re-synchronised read() leaves untouched keep what the call before
                 * left there, caribou_smi.c:382-389) was not written: this call's raw words stand in for it, where they lie -- the lane
                 * rotates three input buffers, a row is only written again in a call that reads it again and moves this pointer -- and
                 * the seam unpacks them the first time it needs the samples (cl_smi_restore_prev_words); a read through the member's
                 * own device in between overrides them like any other read */
                if (cl_smi_set_prev_words(dev->smi, l->channel, l->d_in[l->cur_in] + (size_t)r * l->in_stride, l->len[r])) hard = 1;
                const size_t bytes = (size_t)l->got[r] * l->elem_bytes;
                if (!l->direct[r])
                    pool_submit(&g->pool, (uint8_t *)buffs[m], l->h_out[l->cur_m] + (size_t)r * l->out_stride * l->elem_bytes, bytes);
                else g->stats.direct_reads++;
                rets[m] = (int)l->got[r];
                g->stats.batched_reads++;
                count_read(dev->stream, rets[m]);
            }
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t2);
    pool_drain(&g->pool);
    if (hard) {
        for (int k = 0; k < g->n_in; k++) clhip_stream_sync(g->s_in[k]);
        clhip_stream_sync(g->s_k); clhip_stream_sync(g->s_out);
        if (!g->err[0]) cl_seterr(g->err, sizeof g->err, "cl_group_readStream: %s", clhip_last_error());
        ahead_cancel_all(g);
        g->stale = 0;                                      /* (everything was drained above) */
        for (int k = 0; k < g->n_lanes; k++)
            if (g->lane[k].pipe && g->lane[k].epoch_open) { clhip_rx_pipe_epoch_end(g->lane[k].pipe, g->s_k); g->lane[k].epoch_open = 0; }
        g->stats.errors++;
        return -1;
    }
    clock_gettime(CLOCK_MONOTONIC, &t3);
    g->stats.last_queue_us = (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000);
    g->stats.last_arrive_us = (uint64_t)((t2.tv_sec - t0.tv_sec) * 1000000L + (t2.tv_nsec - t0.tv_nsec) / 1000);
    g->stats.last_total_us = (uint64_t)((t3.tv_sec - t0.tv_sec) * 1000000L + (t3.tv_nsec - t0.tv_nsec) / 1000);
    int delivered = 0;
    for (size_t i = 0; i < g->n; i++) delivered += rets[i] > 0;
    return delivered;
}

/* ------------------------------------------------------------------------------------------- the write call
 * N writeStream calls as one (Stream::WriteSamplesGen, CaribouliteStream.cpp:199-258, over caribou_smi_write, caribou_smi.c:720-762):
 * rets[i] is what cl_writeStream(devs[i], ..., &buffs[i], numElems) returns, and the packed words of every member land in its TX FIFO
 * behind what was there.  Members without a modulator share launches, up to eight streams each:
 *
 *     A  sub-batch b:  clients' samples --copy threads--> pinned rows --copy engine--> device rows           stream s_in[b mod K]
 *     B  the PREVIOUS call's launches are waited for, its words committed to the members' FIFOs
 *     C  room is reserved in every member's pinned TX FIFO; one launch per sub-batch converts and packs (caribou_smi_generate_data)
 *        every row and stores the words straight into the rooms                                               stream s_k
 *
 * The call returns with C queued (WRITE-BEHIND by one call: this call's copies in cross PCIe while the previous call's launches store
 * out, and the next call's host copies run while this one's launches do).  Nobody can tell: the members' seams call back
 * (tx_settle) before anything of theirs looks at or adds to a TX FIFO -- cl_smi_drain_bytes / _drain_to_fd from any thread, a
 * write through a member's own device -- and the group finishes what it has in flight first; so does cl_group_unmake.  (What a
 * client cannot be told any more is a runtime error of launches it was already told about: the next call reports it.)
 * A member with a modulator (MOD=FM, RESAMP), a CS16 call above one MTU (the reference does not clamp those) or a member whose pack
 * mode differs from its sub-batch's takes its own device's writeStream, here, inside the call. */
static void tx_finish(cl_group *g)
{
    if (!g->tx_pending) return;
    g->tx_pending = 0;
    for (int k = 0; k < g->n_lanes; k++) {
        lane_t *l = &g->lane[k];
        if (l->route != ROUTE_TX_PLAIN) continue;
        for (int a = 0; a < l->n; a += l->sub) {
            const int e = a + l->sub < l->n ? a + l->sub : l->n;
            int any = 0;
            for (int r = a; r < e; r++) any |= l->tx_pend[r];
            if (!any) continue;
            const int ok = clhip_event_sync(ev_of(g, l, l->tx_pend_set, a)[g->n_in + 1]) == 0;
            for (int r = a; r < e; r++) {
                if (!l->tx_pend[r]) continue;
                l->tx_pend[r] = 0;
                if (ok) cl_smi_tx_commit(g->dev[l->member[r]]->smi, 4 * l->tx_pend_want);
            }
            if (!ok) { cl_seterr(g->err, sizeof g->err, "cl_group_writeStream: launches of the previous call failed (%s): their words are lost", clhip_last_error()); g->stats.errors++; }
        }
    }
}

/* a member's seam is about to look at or add to its TX FIFO (any thread) */
static void tx_settle_hook(void *ctx, int member)
{
    cl_group *g = (cl_group *)ctx;
    (void)member;
    pthread_mutex_lock(&g->tx_mu);
    tx_finish(g);
    pthread_mutex_unlock(&g->tx_mu);
}

int cl_group_writeStream(cl_group *g, const void *const *buffs, size_t numElems, int *rets, long timeoutUs)
{
    if (!g || !buffs || !rets) return -1;
    if (g->dir != CL_SOAPY_SDR_TX) { cl_seterr(g->err, sizeof g->err, "cl_group_writeStream: the group's devices are set up for RX"); return -1; }
    clhip_set_device(g->device);
    pthread_mutex_lock(&g->tx_mu);
    const uint64_t errors_before = g->stats.errors;
    g->err[0] = 0;
    g->stats.calls++;
    for (size_t i = 0; i < g->n; i++) rets[i] = 0;
    if (!numElems) { pthread_mutex_unlock(&g->tx_mu); return 0; }
    const size_t mtu = CL_NATIVE_MTU_SAMPLES;
    int hard = 0;
    struct timespec t0, t1, t2, t3;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    /* ---- A: the clients' samples to the device (the previous call's launches are storing their words meanwhile) */
    for (int k = 0; k < g->n_lanes && !hard; k++) {
        lane_t *l = &g->lane[k];
        memset(l->fast, 0, (size_t)l->n);
        l->queued = 0;
        if (l->route != ROUTE_TX_PLAIN) continue;
        const size_t n_el = l->format != CL_FORMAT_CS16 && numElems > mtu ? mtu : numElems;      /* :201,217,234; CS16 is not clamped (:182-196) */
        if (n_el > mtu) continue;                              /* (a chunk loop of its own, member by member) */
        l->want = n_el;
        l->set ^= 1;                                           /* pinned rows, device rows and events of this call: the other set is the previous call's, still in flight */
        uint8_t *h_in = l->tx_h_in + (size_t)l->set * l->n * l->tx_row, *d_in = l->tx_d_in + (size_t)l->set * l->n * l->tx_row;
        for (int a = 0; a < l->n && !hard; a += l->sub) {
            const int e = a + l->sub < l->n ? a + l->sub : l->n;
            void **ev = ev_of(g, l, l->set, a);
            const size_t b = l->sub0 + (size_t)(a / l->sub);
            void *s_in = g->s_in[b % (size_t)g->n_in];
            int mode = -1, lo = -1, hi = -1;
            l->queued++;
            for (int r = a; r < e; r++) {
                cl_device *dev = g->dev[l->member[r]];
                const cl_stream *st = dev->stream;
                if (st->native_dir != CL_SOAPY_SDR_TX || st->format != l->format || st->tx_pipe || !buffs[l->member[r]]) continue;
                if (mode < 0) mode = dev->smi->tx_mode;
                if (dev->smi->tx_mode != mode) continue;
                l->fast[r] = 1;
                pool_submit(&g->pool, h_in + (size_t)r * l->tx_row, (const uint8_t *)buffs[l->member[r]], n_el * l->elem_bytes);
                if (lo < 0) lo = r;
                hi = r;
            }
            if (lo < 0) continue;
            pool_drain(&g->pool);                              /* (the previous sub-batch is on its way meanwhile) */
            hard = clhip_memcpy_h2d(d_in + (size_t)lo * l->tx_row, h_in + (size_t)lo * l->tx_row, (size_t)(hi - lo) * l->tx_row + n_el * l->elem_bytes, s_in) ||
                   clhip_event_record(ev[0], s_in);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    /* ---- B: the previous call's words are the FIFOs' */
    tx_finish(g);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    /* ---- C: room in the FIFOs behind them, the launches */
    for (int k = 0; k < g->n_lanes && !hard; k++) {
        lane_t *l = &g->lane[k];
        if (l->route != ROUTE_TX_PLAIN || !l->queued) continue;
        uint8_t *d_in = l->tx_d_in + (size_t)l->set * l->n * l->tx_row;
        for (int a = 0; a < l->n && !hard; a += l->sub) {
            const int e = a + l->sub < l->n ? a + l->sub : l->n;
            void **ev = ev_of(g, l, l->set, a);
            const void *in_rows[CLHIP_PACK_ROWS]; uint8_t *out_rows[CLHIP_PACK_ROWS];
            int n_rows = 0, mode = -1;
            for (int r = a; r < e; r++) {
                if (!l->fast[r]) continue;
                cl_smi *smi = g->dev[l->member[r]]->smi;
                /* caribou_smi_write's chunk loop appends native-batch pieces of one contiguous array (caribou_smi.c:738-759): the array,
                 * packed into the room behind what the FIFO holds, committed once the launch is known to have run */
                uint8_t *room = cl_smi_tx_reserve_raw(smi, 4 * l->want + 64);
                uint8_t *d_room = room ? (uint8_t *)cl_fifo_device_ptr(&smi->tx, room) : NULL;
                if (!d_room) { l->fast[r] = 0; continue; }        /* (through its own device, below) */
                mode = smi->tx_mode;
                in_rows[n_rows] = d_in + (size_t)r * l->tx_row; out_rows[n_rows] = d_room; n_rows++;
            }
            if (!n_rows) continue;
            hard = clhip_stream_wait_event(g->s_k, ev[0]) ||
                   clhip_convert_pack_rows(in_rows, l->format, l->want, n_rows, mode, out_rows, g->s_k) ||
                   clhip_event_record(ev[g->n_in + 1], g->s_k);
            g->stats.launches++;
        }
        if (hard) break;
        l->tx_pend_set = l->set; l->tx_pend_want = l->want;
        for (int r = 0; r < l->n; r++) {
            if (!l->fast[r]) continue;
            cl_device *dev = g->dev[l->member[r]];
            cl_stream *st = dev->stream;
            l->tx_pend[r] = 1; g->tx_pending = 1;
            rets[l->member[r]] = (int)l->want;
            st->stats.write_calls++; st->stats.elements_written += l->want;
            if (l->format == CL_FORMAT_CS16) dev->smi->stat_written += l->want;          /* (caribou_smi_write counts; the conversions' callers do not) */
            g->stats.batched_reads++;
        }
    }
    if (hard) {
        for (int k = 0; k < g->n_in; k++) clhip_stream_sync(g->s_in[k]);
        clhip_stream_sync(g->s_k);
        tx_finish(g);                                          /* (what was launched and told is waited for; what was not is not committed) */
        cl_seterr(g->err, sizeof g->err, "cl_group_writeStream: %s", clhip_last_error());
        g->stats.errors++;
        pthread_mutex_unlock(&g->tx_mu);
        return -1;
    }
    /* ---- the members off the batched route, through their own devices (their seams settle the group first: order in the FIFOs) */
    for (int k = 0; k < g->n_lanes; k++) {
        lane_t *l = &g->lane[k];
        for (int r = 0; r < l->n; r++) {
            if (l->fast[r]) continue;
            const int m = l->member[r];
            const void *const b1[1] = {buffs[m]};
            rets[m] = buffs[m] ? cl_writeStream(g->dev[m], g->dev[m]->stream, b1, numElems, NULL, 0, timeoutUs) : 0;
            g->stats.single_reads++;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t3);
    g->stats.last_queue_us = (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000);
    g->stats.last_arrive_us = (uint64_t)((t2.tv_sec - t0.tv_sec) * 1000000L + (t2.tv_nsec - t0.tv_nsec) / 1000);
    g->stats.last_total_us = (uint64_t)((t3.tv_sec - t0.tv_sec) * 1000000L + (t3.tv_nsec - t0.tv_nsec) / 1000);
    int delivered = 0;
    for (size_t i = 0; i < g->n; i++) delivered += rets[i] > 0;
    const int failed_before = g->stats.errors != errors_before;      /* (the previous call's launches: reported now) */
    pthread_mutex_unlock(&g->tx_mu);
    return failed_before ? -1 : delivered;
}
