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
 * persistent native buffer) is rebuilt lazily from the previous call's raw words when a member leaves the batched route.
 *
 * This file: types, the copy threads, make / unmake, registered buffers.  cl_group_rx.inc: cl_group_readStream.  cl_group_tx.inc:
 * cl_group_writeStream (a group of TX devices).  One translation unit. */
#include <immintrin.h>
#include <time.h>
#include <unistd.h>

#include "cl_internal.h"

enum { ROUTE_PIPE = 1, ROUTE_PLAIN = 2, ROUTE_TX_PLAIN = 3, ROUTE_TX_SINGLE = 4, ROUTE_TX_PIPE = 5 };

typedef struct { uint8_t *dst; const uint8_t *src; size_t bytes; int take_i; } copy_job;     /* take_i: `bytes` of dst from every other float of src (below) */

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
    uint8_t *d_out_alt; uint8_t *ahead_dev;    /* READAHEAD=2: the device rows the NEXT call's results are computed into when a sub-batch has registered client buffers among
                                               * its members (no mirror to store to ahead of the client's pointer): the two swap at every call; per sub-batch: its rows made
                                               * ahead lie there */
    uint8_t *h_out[2], *m_out[2]; int cur_m;   /* two pinned mirrors (this call's results / the next call's, computed ahead) and the device's addresses of
                                               * them (mapped pinned): kernels may store into a mirror themselves */
    int32_t *h_offs[4]; int32_t *d_offs[4];    /* ROUTE_PLAIN: per row 0 (unpack) / -1 (skip), mapped pinned: one table per (event set, launched ahead | in the
                                               * call) -- a launch reads its table when it RUNS */
    uint8_t *done_ahead; long *ahead_got;      /* per row: the previous call launched over the batch it read ahead, results in h_out[cur_m ^ 1] then; elements */
    uint8_t *direct;                           /* per call and row: a launch stores the row into the client's registered buffer */
    cl_read_ctx *ctx;                          /* per call and row: a one-by-one member's read in flight (lanes without extension stages) */
    /* the reference's low-pass over whole sub-batches (lanes without extension stages): one multi-stream filter object per (filter,
     * sub-batch), made when first needed; a member's carried state lives EITHER in its stream's own objects or here (iir_own) */
    clhip_iir **giir; int n_subs; uint8_t *iir_own; int16_t *d_f[2]; uint8_t *sub_ft; uint8_t *how;   /* (d_f: filtered int16 rows of this call's / the next call's set) */
    size_t f_stride;                            /* int16 pairs per row of d_f (= the input rows' stride in words) */
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
    /* a modulator lane (ROUTE_TX_PIPE: members with MOD=FM / RESAMP of one configuration): ONE multi-stream TX pipe per sub-batch, made when
     * first needed; a member's carried state (modulator phase, resampler history) lives EITHER in its stream's own pipe or here (tx_own) */
    clhip_tx_pipe **tx_gp; uint8_t *tx_own;
    float *tx_d_msg; size_t tx_msg_stride;          /* FM: the I rails as dense messages, two sets of rows */
    uint8_t *tx_d_words; size_t tx_words_row;       /* the packed words on the device, two sets of rows: they leave for the members' FIFOs by a copy each */
    uint8_t **tx_room; long *tx_packed;             /* per row: the room reserved in the member's FIFO for the call in flight; per sub-batch: its words per row */
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
    size_t tx_copy_bytes;                 /* kwarg TX_COPY_MB (default 8): a TX group's copies in carry neighbouring sub-batches until they are this long (0: one copy per sub-batch) */
    int tx_polls; int tx_polls_set;       /* test hook (cl_group_set_tx_poll_bound): the look-back poll bound of the group's modulator pipes */
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
    uint8_t **reg_dev;                                    /* ... and the device's addresses of them (a launch stores into them across PCIe) */
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

/* The clients' samples on their way into pinned memory for a modulator lane with MOD=FM: the modulator reads the I rail only (SURVEY.md
 * a13: "if given I/Q, use I"), so only the I rail is copied -- half the bytes cross PCIe, and the device needs no launch to pick them
 * out.  n floats of dst from src[0], src[2], src[4], ... */
__attribute__((target("avx2"))) static void take_i_avx2(float *dst, const float *src, size_t n)
{
    size_t i = 0;
    while (i < n && ((uintptr_t)(dst + i) & 31)) { dst[i] = src[2 * i]; i++; }
    for (; i + 8 <= n; i += 8) {
        const __m256 a = _mm256_loadu_ps(src + 2 * i), b = _mm256_loadu_ps(src + 2 * i + 8);
        const __m256 s = _mm256_shuffle_ps(a, b, 0x88);                                   /* a0 a2 b0 b2 | a4 a6 b4 b6 */
        _mm256_stream_ps(dst + i, _mm256_castpd_ps(_mm256_permute4x64_pd(_mm256_castps_pd(s), 0xD8)));   /* a0 a2 a4 a6 b0 b2 b4 b6 */
    }
    _mm_sfence();
    for (; i < n; i++) dst[i] = src[2 * i];
}

static void take_i(uint8_t *dst, const uint8_t *src, size_t bytes)
{
    float *d = (float *)dst; const float *s = (const float *)src;
    const size_t n = bytes / 4;
    if (__builtin_cpu_supports("avx2")) take_i_avx2(d, s, n);
    else for (size_t i = 0; i < n; i++) d[i] = s[2 * i];
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
        if (j.take_i) take_i(j.dst, j.src, j.bytes); else copy_out(j.dst, j.src, j.bytes);
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

/* queue [src, src + bytes) -> dst in pieces; without worker threads the caller copies.  take: `bytes` of dst from twice as many of src
 * (take_i) */
static void pool_submit_kind(copy_pool *p, uint8_t *dst, const uint8_t *src, size_t bytes, int take)
{
    const size_t piece = (size_t)(take ? 256 : 512) << 10;
    const size_t sm = take ? 2 : 1;                     /* source bytes per destination byte */
    if (!p->n_threads) { if (take) take_i(dst, src, bytes); else copy_out(dst, src, bytes); return; }
    pthread_mutex_lock(&p->mu);
    for (size_t o = 0; o < bytes; o += piece) {
        const size_t n = bytes - o < piece ? bytes - o : piece;
        if (p->q_len == p->q_cap) {                     /* full: do this piece here */
            pthread_mutex_unlock(&p->mu);
            if (take) take_i(dst + o, src + sm * o, n); else copy_out(dst + o, src + o, n);
            pthread_mutex_lock(&p->mu);
            continue;
        }
        p->q[(p->q_head + p->q_len) % p->q_cap] = (copy_job){dst + o, src + sm * o, n, take};
        p->q_len++; p->in_flight++;
    }
    pthread_cond_broadcast(&p->work);
    pthread_mutex_unlock(&p->mu);
}

static void pool_submit(copy_pool *p, uint8_t *dst, const uint8_t *src, size_t bytes) { pool_submit_kind(p, dst, src, bytes, 0); }

/* the caller helps until the queue is empty, then waits for the pieces still being copied */
static void pool_drain(copy_pool *p)
{
    if (!p->n_threads) return;
    pthread_mutex_lock(&p->mu);
    while (p->q_len) {
        const copy_job j = p->q[p->q_head];
        p->q_head = (p->q_head + 1) % p->q_cap; p->q_len--;
        pthread_mutex_unlock(&p->mu);
        if (j.take_i) take_i(j.dst, j.src, j.bytes); else copy_out(j.dst, j.src, j.bytes);
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

/* TX: one modulator configuration (the words do not depend on the channel type) */
static int same_tx_dsp(const cl_dsp_cfg *a, const cl_dsp_cfg *b)
{
    return a->enabled && b->enabled && a->up == b->up && a->down == b->down && a->n_rs == b->n_rs && a->mod_fm == b->mod_fm &&
           (!a->mod_fm || a->mod_kf == b->mod_kf) && !memcmp(a->rs, b->rs, sizeof(float) * (size_t)a->n_rs);
}

static void ahead_cancel_all(cl_group *g);
static void tx_settle_hook(void *ctx, int member);
static void tx_finish(cl_group *g);
static void settle(cl_group *g);
static void giir_ahead_drop(cl_group *g, lane_t *l, int sb);
static void **ev_of(const cl_group *g, const lane_t *l, int set, int a);
static void iir_home(void *ctx, int member);
static void tx_home(void *ctx, int member);

static void lane_free(lane_t *l)
{
    if (l->pipe) clhip_rx_pipe_destroy(l->pipe);
    clhip_free(l->d_in[0]); clhip_free(l->d_in[1]); clhip_free(l->d_in[2]); clhip_free(l->d_out); clhip_free(l->d_out_alt); free(l->ahead_dev);
    clhip_event_destroy(l->ev_primed); free(l->primed); free(l->primed_epoch);
    clhip_host_free(l->h_out[0]); clhip_host_free(l->h_out[1]); clhip_host_free(l->h_offs[0]);
    clhip_host_free(l->tx_h_in); clhip_free(l->tx_d_in); free(l->tx_pend);
    for (int i = 0; l->tx_gp && i < l->n_subs; i++) clhip_tx_pipe_destroy(l->tx_gp[i]);
    free(l->tx_gp); free(l->tx_own); clhip_free(l->tx_d_msg); clhip_free(l->tx_d_words); free(l->tx_room); free(l->tx_packed);
    free(l->done_ahead); free(l->ahead_got); free(l->direct); free(l->ctx);
    for (int i = 0; l->giir && i < 3 * l->n_subs; i++) clhip_iir_destroy(l->giir[i]);
    free(l->giir); free(l->iir_own); free(l->sub_ft); free(l->ahead_ft); free(l->sub_verdict); free(l->how); clhip_free(l->d_f[0]); clhip_free(l->d_f[1]);
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
        for (size_t i = 0; g->dev && g->lane_of && i < g->n; i++) {   /* the modulators' carried state goes back to the streams' own pipes */
            cl_stream *st = g->dev[i] ? g->dev[i]->stream : NULL;
            if (st && st->tx_home_ctx == g) { tx_home(g, (int)i); st->tx_home = NULL; st->tx_home_ctx = NULL; }
        }
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
    free(g->reg_base); free(g->reg_len); free(g->has_reg); free(g->reg_dev);
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
    g->reg_dev = (uint8_t **)calloc(n, sizeof *g->reg_dev);
    if (!g->dev || !g->lane_of || !g->row_of || !g->lane || !g->reg_base || !g->reg_len || !g->has_reg || !g->reg_dev) { cl_group_unmake(g); return NULL; }
    memcpy(g->dev, devs, n * sizeof *g->dev);
    const char *sub = kwget(keys, vals, n_kwargs, "SUBBATCH"), *ct = kwget(keys, vals, n_kwargs, "COPY_THREADS");
    g->sub = sub && atoi(sub) > 0 ? atoi(sub) : 0;     /* 0: by the lane's route (below) */
    const char *is = kwget(keys, vals, n_kwargs, "INGEST_STREAMS");
    g->n_in = is && atoi(is) >= 1 && atoi(is) <= GRP_MAX_IN ? atoi(is) : 2;     /* tools/group_ab.py, medians of 9 interleaved reps: CS16 4989 / 6508 / 5443 Msamples/s at 1 / 2 / 4, FIR64 + 3/2 3116 / 3131 / 2854 */
    g->ev_per = g->n_in + 2;
    const char *sk = kwget(keys, vals, n_kwargs, "SINK");
    g->sink_mapped = !(sk && !strcmp(sk, "copy"));
    { const char *ra = kwget(keys, vals, n_kwargs, "READAHEAD"); g->readahead = ra ? atoi(ra) : 2; }
    { const char *cm = kwget(keys, vals, n_kwargs, "TX_COPY_MB"); g->tx_copy_bytes = (size_t)(cm && atoi(cm) >= 0 ? atoi(cm) : 8) << 20; }
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
            if (g->dir == CL_SOAPY_SDR_TX) {                       /* (the TX words do not depend on the channel type) */
                if (s0->format == st->format && ((!s0->dsp.enabled && !st->dsp.enabled) || (s0->tx_pipe && st->tx_pipe && same_tx_dsp(&s0->dsp, &st->dsp)))) li = k;
            }
            else if (g->lane[k].channel == devs[i]->channel && s0->format == st->format && same_dsp(&s0->dsp, &st->dsp)) li = k;
        }
        if (li < 0) {
            li = g->n_lanes++;
            lane_t *l = &g->lane[li];
            l->channel = devs[i]->channel; l->format = st->format; l->dsp = st->dsp;
            l->route = g->dir == CL_SOAPY_SDR_TX ? (st->dsp.enabled ? (st->tx_pipe ? ROUTE_TX_PIPE : ROUTE_TX_SINGLE) : ROUTE_TX_PLAIN) : st->dsp.enabled ? ROUTE_PIPE : ROUTE_PLAIN;
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
            if (l->route == ROUTE_TX_PIPE && l->n < 2) l->route = ROUTE_TX_SINGLE;      /* (one modulator alone gains nothing from a pipe of the group's) */
            if (l->route != ROUTE_TX_PLAIN && l->route != ROUTE_TX_PIPE) continue;
            l->n_subs = (l->n + l->sub - 1) / l->sub;
            l->elem_bytes = fmt_bytes(l->format);
            l->tx_row = mtu * l->elem_bytes + 256;
            l->tx_h_in = (uint8_t *)clhip_host_alloc(2 * (size_t)l->n * l->tx_row); l->tx_d_in = (uint8_t *)clhip_malloc(2 * (size_t)l->n * l->tx_row);
            l->tx_pend = (uint8_t *)calloc((size_t)l->n, 1);
            for (int r = 0; r < l->n; r++) {
                cl_smi *smi = g->dev[l->member[r]]->smi;
                smi->tx_settle = tx_settle_hook; smi->tx_settle_ctx = g; smi->tx_settle_member = l->member[r];
            }
            if (!l->tx_h_in || !l->tx_d_in || !l->tx_pend) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: buffers for %d TX streams could not be allocated", l->n); cl_group_unmake(g); return NULL; }
            if (l->route == ROUTE_TX_PIPE) {
                l->tx_msg_stride = mtu + 64;
                l->tx_words_row = 4 * (mtu * (size_t)l->dsp.up + 64);                /* (the most a call can produce: every message `up` words) */
                l->tx_gp = (clhip_tx_pipe **)calloc((size_t)l->n_subs, sizeof(clhip_tx_pipe *)); l->tx_own = (uint8_t *)calloc((size_t)l->n, 1);
                l->tx_room = (uint8_t **)calloc((size_t)l->n, sizeof(uint8_t *)); l->tx_packed = (long *)calloc((size_t)l->n_subs, sizeof(long));
                l->tx_d_msg = l->dsp.mod_fm ? (float *)clhip_malloc(sizeof(float) * 2 * (size_t)l->n * l->tx_msg_stride) : NULL;
                l->tx_d_words = (uint8_t *)clhip_malloc(2 * (size_t)l->n * l->tx_words_row);
                if (!l->tx_gp || !l->tx_own || !l->tx_room || !l->tx_packed || (l->dsp.mod_fm && !l->tx_d_msg) || !l->tx_d_words) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: buffers for %d modulator streams could not be allocated", l->n); cl_group_unmake(g); return NULL; }
                for (int r = 0; r < l->n; r++) {
                    cl_stream *st = g->dev[l->member[r]]->stream;
                    st->tx_home = tx_home; st->tx_home_ctx = g; st->tx_home_member = l->member[r];
                }
            }
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
        l->d_out_alt = g->readahead == 2 ? (uint8_t *)clhip_malloc(out_bytes) : NULL;
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
        if (!l->d_in[0] || !l->d_in[1] || !l->d_in[2] || !l->primed || !l->primed_epoch || !l->ev_primed || !l->d_out || !l->h_out[0] || (g->readahead == 2 && (!l->h_out[1] || !l->d_out_alt)) || !l->h_offs[0] || !l->d_offs[0] || !l->m_out[0] || !l->done_ahead || !l->ahead_got || !l->direct || !l->ctx || !l->how || !l->fast || !l->len || !l->got || !l->src || !l->ahead_mark) {
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
        l->ahead_dev = (uint8_t *)calloc((size_t)l->n_subs, 1);
        if (!l->ahead_dev) { cl_group_unmake(g); return NULL; }
        if (l->route == ROUTE_PLAIN || l->route == ROUTE_PIPE) {
            l->f_stride = l->in_stride / 4;
            l->giir = (clhip_iir **)calloc((size_t)3 * (size_t)l->n_subs, sizeof(clhip_iir *)); l->iir_own = (uint8_t *)calloc((size_t)3 * (size_t)l->n, 1);
            l->sub_ft = (uint8_t *)calloc((size_t)l->n_subs, 1); l->ahead_ft = (uint8_t *)calloc((size_t)l->n_subs, 1); l->sub_verdict = (uint8_t *)calloc((size_t)l->n_subs, 1);
            l->d_f[0] = (int16_t *)clhip_malloc((size_t)l->n * l->f_stride * 4 + 256); l->d_f[1] = (int16_t *)clhip_malloc((size_t)l->n * l->f_stride * 4 + 256);
            if (!l->giir || !l->iir_own || !l->sub_ft || !l->ahead_ft || !l->sub_verdict || !l->d_f[0] || !l->d_f[1]) { cl_seterr(g_make_err, sizeof g_make_err, "cl_group_make: filter buffers"); cl_group_unmake(g); return NULL; }
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
        if (l->ahead_dev) memset(l->ahead_dev, 0, (size_t)l->n_subs);
    }
    settle(g);
    for (int k = 0; k < g->n_lanes; k++)
        for (int sb = 0; g->lane[k].ahead_ft && sb < g->lane[k].n_subs; sb++) giir_ahead_drop(g, &g->lane[k], sb);
}

/* Client buffers the members' outputs are stored into directly, across PCIe, by a launch (no pinned mirror, no memcpy): one
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
        if (!(g->reg_dev[k] = (uint8_t *)clhip_host_register(g->reg_base[k], g->reg_len[k]))) {
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

/* the device's address of a byte of a registered client buffer (registered() said yes) */
static uint8_t *registered_dev(const cl_group *g, const void *p)
{
    const uint8_t *q = (const uint8_t *)p;
    for (size_t k = 0; k < g->n_reg; k++)
        if (q >= g->reg_base[k] && q < g->reg_base[k] + g->reg_len[k]) return g->reg_dev[k] + (q - g->reg_base[k]);
    return NULL;
}

/* the two calls (same translation unit: everything above is file-local) */
#include "cl_group_rx.inc"
#include "cl_group_tx.inc"
