/* CPU-only sanitizer harness for the single-stream host paths (cariboulite_amd/csrc/host/cl_soapy.c, cl_smi.c, cl_ring.c) over the
 * threaded model of the HIP layer (tests/cpp/hip_mock/clhip_mock.c: every HIP stream a thread).  Built twice by
 * tests/test_host_asan.py (-fsanitize=thread; -fsanitize=address,undefined).  No GPU.  Three parts, each with a producer thread
 * racing the client:
 *   A  readStream with ASYNC=1 (reader thread -> device ring -> client; CaribouliteStream.cpp:16-49,262-263), CS16 and CF32;
 *   B  readStream on the calling thread with the seam's read-ahead, CF32, plain and into a registered buffer (ZEROCOPY=1), with a
 *      flush in between;
 *   C  writeStream CS16 / CF32 (conversion + caribou_smi_generate_data into the pinned TX FIFO) against a drainer thread;
 *   D  cl_group_writeStream over eleven TX devices, a drainer thread per member -- and the same through a cl_node that cuts them into three
 *      groups, each call's three group calls running at once on threads of their own;
 *   E  the same with a modulator on every member (the group's multi-stream TX pipes: carried state moving between the members' own
 *      pipes and the group's, a launch that gives up and is repeated at commit time), a drainer thread per member.
 * Sample n of a stream carries n in its 24 payload bits, so every delivered block names its own place in the stream: blocks must be
 * contiguous inside and strictly ascending across calls (an overwrite-oldest ring may skip, never repeat or reorder). */
#include <assert.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "cariboulite_hip.h"
#include "cl_oracle.h"

#define MTU 131072u
#define NB (4u * MTU)

static void sample_of(uint64_t g, int *I, int *Q) { *I = (int)(g & 0xFFF) - 2048; *Q = (int)((g >> 12) & 0xFFF) - 2048; }
static uint64_t index_of(int I, int Q) { return (uint64_t)(I + 2048) | ((uint64_t)(Q + 2048) << 12); }

static void words_of(uint64_t g0, size_t n, uint8_t *out)                /* S1G layout: I in the high field (caribou_smi.c:338-359) */
{
    for (size_t k = 0; k < n; k++) {
        int I, Q; sample_of(g0 + k, &I, &Q);
        const uint32_t w = 0x80004000u | (((uint32_t)I & 0x1FFF) << 17) | (((uint32_t)Q & 0x1FFF) << 1);
        memcpy(out + 4 * k, &w, 4);
    }
}

typedef struct { cl_smi *smi; uint64_t total; size_t max_pending; volatile int stop; } feed_arg;
static void *feeder(void *arg)
{
    feed_arg *a = (feed_arg *)arg;
    uint8_t *piece = (uint8_t *)malloc(NB);
    uint32_t r = 4242u;
    uint64_t g = 0;
    while (g < a->total && !a->stop) {
        if (cl_smi_pending_bytes(a->smi) >= a->max_pending) { sched_yield(); usleep(20); continue; }
        r = r * 1664525u + 1013904223u;
        size_t n = MTU >> ((r >> 20) % 3);
        if (g + n > a->total) n = (size_t)(a->total - g);
        words_of(g, n, piece);
        if (cl_smi_feed_bytes(a->smi, piece, 4 * n)) abort();
        g += n;
    }
    free(piece);
    return NULL;
}

/* a delivered block: contiguous, starting at or behind `*next`; returns 0 and moves *next behind it */
static int block_ok(const void *buf, int is_f32, size_t n, uint64_t *next, int may_skip)
{
    uint64_t g0 = 0;
    for (size_t k = 0; k < n; k++) {
        int I, Q;
        if (is_f32) { I = (int)(((const float *)buf)[2 * k] * 4096.0f); Q = (int)(((const float *)buf)[2 * k + 1] * 4096.0f); }
        else { I = ((const int16_t *)buf)[2 * k]; Q = ((const int16_t *)buf)[2 * k + 1]; }
        const uint64_t g = index_of(I, Q);
        if (!k) { g0 = g; if (g < *next || (!may_skip && g != *next)) { fprintf(stderr, "block starts at %llu, expected %s%llu\n", (unsigned long long)g, may_skip ? ">= " : "", (unsigned long long)*next); return -1; } }
        else if (g != g0 + k) { fprintf(stderr, "sample %zu of the block is %llu, not %llu\n", k, (unsigned long long)g, (unsigned long long)(g0 + k)); return -1; }
    }
    if (n) *next = g0 + n;
    return 0;
}

static cl_device *make_device(const char *fmt, int dir, const char *const *sk, const char *const *sv, size_t nk, cl_stream **st)
{
    const char *dk[] = {"driver", "channel"}, *dv[] = {"Cariboulite", "S1G"};
    cl_device *d = cl_device_make(dk, dv, 2);
    assert(d);
    *st = cl_setupStream(d, dir, fmt, NULL, 0, sk, sv, nk);
    if (!*st) { fprintf(stderr, "setupStream: %s\n", cl_device_last_error(d)); abort(); }
    assert(cl_activateStream(d, *st, 0, 0, 0) == 0);
    return d;
}

static int part_a(const char *fmt, int n_mtus)
{
    cl_stream *st; const char *sk[] = {"ASYNC"}, *sv[] = {"1"};
    cl_device *d = make_device(fmt, CL_SOAPY_SDR_RX, sk, sv, 1, &st);
    feed_arg fa = {cl_device_smi(d), (uint64_t)n_mtus * MTU, 2 * NB, 0};
    pthread_t th; pthread_create(&th, NULL, feeder, &fa);
    const int f32 = !strcmp(fmt, "CF32");
    void *buf = malloc((MTU + 16) * 8);
    uint64_t next = 0, delivered = 0; long calls = 0, empty = 0;
    while (next < fa.total && calls < 100000) {
        void *b[1] = {buf};
        const size_t num = calls % 5 == 4 ? MTU / 2 : MTU;
        const int r = cl_readStream(d, st, b, num, NULL, NULL, 20000);
        calls++;
        if (r < 0) { fprintf(stderr, "readStream: %d\n", r); return -1; }
        if (!r) { empty++; continue; }
        if (block_ok(buf, f32, (size_t)r, &next, 1)) return -1;
        delivered += (uint64_t)r;
    }
    fa.stop = 1; pthread_join(th, NULL);
    cl_device_unmake(d); free(buf);
    printf("A %s: %llu of %llu samples delivered in %ld calls (%ld empty)\n", fmt, (unsigned long long)delivered, (unsigned long long)fa.total, calls, empty);
    return next == fa.total && delivered * 10 >= fa.total * 5 ? 0 : -1;      /* (the ring overwrites its oldest when the client is late: at least half must arrive) */
}

static int part_b(int n_mtus)
{
    cl_stream *st; const char *sk[] = {"ZEROCOPY"}, *sv[] = {"1"};
    cl_device *d = make_device("CF32", CL_SOAPY_SDR_RX, sk, sv, 1, &st);
    cl_smi *smi = cl_device_smi(d);
    feed_arg fa = {smi, (uint64_t)n_mtus * MTU, 3 * NB, 0};
    pthread_t th; pthread_create(&th, NULL, feeder, &fa);
    float *reg = (float *)malloc((MTU + 16) * 8), *plain = (float *)malloc((MTU + 16) * 8);
    assert(cl_stream_register_buffer(d, st, reg, (MTU + 16) * 8) == 0);
    uint64_t next = 0; long calls = 0, flushed = 0;
    while (next < fa.total && calls < 100000) {
        void *b[1] = {calls & 1 ? (void *)reg : (void *)plain};
        const size_t num = calls % 7 == 6 ? MTU / 4 : MTU;
        const int r = cl_readStream(d, st, b, num, NULL, NULL, 1000);
        calls++;
        assert(r >= 0);
        if (block_ok(b[0], 1, (size_t)r, &next, flushed > 0)) return -1;
        if (calls == 9) { cl_smi_flush_fifo(smi); flushed = 1; }         /* what is pending (and what was staged ahead) is dropped: the stream skips */
        if (!r) { sched_yield(); usleep(50); }
    }
    fa.stop = 1; pthread_join(th, NULL);
    cl_stream_stats ss; cl_getStreamStats(d, st, &ss);
    cl_device_unmake(d); free(reg); free(plain);
    printf("B: %ld calls, %llu zero-copy reads\n", calls, (unsigned long long)ss.zero_copy_reads);
    return next == fa.total && ss.zero_copy_reads > 2 ? 0 : -1;
}

typedef struct { cl_smi *smi; volatile uint64_t total; volatile int bad; } drain_arg;
static void *drainer(void *arg)
{
    drain_arg *a = (drain_arg *)arg;
    uint8_t *got = (uint8_t *)malloc(NB), *want = (uint8_t *)malloc(NB);
    int16_t *iq = (int16_t *)malloc(NB);
    uint64_t g = 0;
    while (g < a->total && !a->bad) {
        const size_t n = cl_smi_drain_bytes(a->smi, got, NB);
        if (!n) { sched_yield(); usleep(20); continue; }
        assert(n % 4 == 0);
        for (size_t k = 0; k < n / 4; k++) { int I, Q; sample_of(g + k, &I, &Q); iq[2 * k] = (int16_t)I; iq[2 * k + 1] = (int16_t)Q; }
        orc_generate_data(ORC_TX_DOCUMENTED, iq, n / 4, want);
        if (memcmp(got, want, n)) { fprintf(stderr, "TX bytes differ behind sample %llu\n", (unsigned long long)g); a->bad = 1; }
        g += n / 4;
    }
    free(got); free(want); free(iq);
    return NULL;
}

static int part_c(const char *fmt, int n_mtus)
{
    cl_stream *st;
    cl_device *d = make_device(fmt, CL_SOAPY_SDR_TX, NULL, NULL, 0, &st);
    cl_smi_set_tx_mode(cl_device_smi(d), CL_TX_DOCUMENTED);
    drain_arg da = {cl_device_smi(d), (uint64_t)n_mtus * MTU, 0};
    pthread_t th; pthread_create(&th, NULL, drainer, &da);
    const int f32 = !strcmp(fmt, "CF32");
    void *buf = malloc(MTU * 8);
    uint64_t g = 0; long calls = 0;
    while (g < da.total && !da.bad) {
        const size_t num = calls % 3 == 2 ? MTU / 2 : MTU;
        for (size_t k = 0; k < num; k++) {
            int I, Q; sample_of(g + k, &I, &Q);
            if (f32) { ((float *)buf)[2 * k] = (float)I / 4096.0f; ((float *)buf)[2 * k + 1] = (float)Q / 4096.0f; }
            else { ((int16_t *)buf)[2 * k] = (int16_t)I; ((int16_t *)buf)[2 * k + 1] = (int16_t)Q; }
        }
        const void *b[1] = {buf};
        const int r = cl_writeStream(d, st, b, num, NULL, 0, 1000);
        calls++;
        if (r != (int)num) { fprintf(stderr, "writeStream: %d (%s)\n", r, cl_device_last_error(d)); da.bad = 1; break; }
        g += (uint64_t)r;
    }
    pthread_join(th, NULL);
    cl_device_unmake(d); free(buf);
    printf("C %s: %ld calls\n", fmt, calls);
    return da.bad ? -1 : 0;
}

/* D: a group of TX devices (cl_group_writeStream: copy threads -> pinned rows -> device rows -> one launch per sub-batch that stores into
 * the room reserved in every member's TX FIFO) against one drainer thread per member */
#define N_TX 11
static int part_d(const char *fmt, int n_mtus, int shards)       /* shards > 0: through a cl_node that cuts the members into that many groups, one thread each */
{
    cl_device *d[N_TX]; cl_stream *st[N_TX]; drain_arg da[N_TX]; pthread_t th[N_TX];
    for (int i = 0; i < N_TX; i++) {
        d[i] = make_device(fmt, CL_SOAPY_SDR_TX, NULL, NULL, 0, &st[i]);
        cl_smi_set_tx_mode(cl_device_smi(d[i]), CL_TX_DOCUMENTED);
        da[i] = (drain_arg){cl_device_smi(d[i]), (uint64_t)n_mtus * MTU, 0};
    }
    const char *gk[] = {"COPY_THREADS", "SHARDS"}, *gv[] = {"2", shards == 3 ? "3" : "2"};
    cl_group *grp = shards ? NULL : cl_group_make(d, N_TX, gk, gv, 1);
    cl_node *node = shards ? cl_node_make(d, N_TX, gk, gv, 2) : NULL;
    if (!grp && !node) { fprintf(stderr, "cl_group_make / cl_node_make: %s / %s\n", cl_group_last_error(NULL), cl_node_last_error(NULL)); return -1; }
    if (node && (int)cl_node_shards(node) != shards) { fprintf(stderr, "cl_node_shards: %zu\n", cl_node_shards(node)); return -1; }
    for (int i = 0; i < N_TX; i++) pthread_create(&th[i], NULL, drainer, &da[i]);
    const int f32 = !strcmp(fmt, "CF32");
    void *buf = malloc(MTU * 8);                                    /* (every member is written the same samples: one buffer) */
    const void *bufs[N_TX]; int rets[N_TX];
    for (int i = 0; i < N_TX; i++) bufs[i] = buf;
    uint64_t g = 0; long calls = 0; int bad = 0;
    while (g < da[0].total && !bad) {
        const size_t num = calls % 3 == 2 ? MTU / 2 : MTU;
        for (size_t k = 0; k < num; k++) {
            int I, Q; sample_of(g + k, &I, &Q);
            if (f32) { ((float *)buf)[2 * k] = (float)I / 4096.0f; ((float *)buf)[2 * k + 1] = (float)Q / 4096.0f; }
            else { ((int16_t *)buf)[2 * k] = (int16_t)I; ((int16_t *)buf)[2 * k + 1] = (int16_t)Q; }
        }
        const int nd = node ? cl_node_writeStream(node, bufs, num, rets, 1000) : cl_group_writeStream(grp, bufs, num, rets, 1000);
        calls++;
        if (nd != N_TX) { fprintf(stderr, "cl_group_writeStream: %d (%s)\n", nd, node ? cl_node_last_error(node) : cl_group_last_error(grp)); bad = 1; break; }
        for (int i = 0; i < N_TX; i++) if (rets[i] != (int)num) bad = 1;
        for (int i = 0; i < N_TX; i++) bad |= da[i].bad;
        g += num;
    }
    if (bad) for (int i = 0; i < N_TX; i++) da[i].total = 0;         /* (let the drainers go) */
    for (int i = 0; i < N_TX; i++) { pthread_join(th[i], NULL); bad |= da[i].bad; }
    cl_group_stats gs; memset(&gs, 0, sizeof gs);
    for (int s = 0; s < (node ? shards : 1); s++) {
        cl_group_stats one; cl_group_getStats(node ? cl_node_group(node, (size_t)s) : grp, &one);
        gs.batched_reads += one.batched_reads; gs.launches += one.launches; gs.errors += one.errors; gs.single_reads += one.single_reads;
    }
    cl_group_unmake(grp); cl_node_unmake(node);
    for (int i = 0; i < N_TX; i++) cl_device_unmake(d[i]);
    free(buf);
    printf("D %s (%d shards): %ld calls, %llu batched writes, %llu launches\n", fmt, shards, calls, (unsigned long long)gs.batched_reads, (unsigned long long)gs.launches);
    return bad || gs.errors || gs.single_reads ? -1 : 0;
}

/* E: a group of TX devices WITH A MODULATOR (cl_group_writeStream's modulator lanes: one multi-stream TX pipe per sub-batch; the mock's
 * pipe is a prefix sum over the whole stream, so every word names the state it was made from) against one drainer thread per member.
 * Every member is written the same samples; every fifth call one member is written through its own device instead and left out of the
 * group's call (buffs[i] = NULL) -- its carried state goes home and comes back, its sub-batch goes one by one for that call; every
 * seventh run of a pipe "gives up" once (clhip_mock_tx_fail_every): the group repeats the sub-batch when it commits its words. */
void clhip_mock_tx_fail_every(int k);
typedef struct { cl_smi *smi; volatile uint64_t total; volatile int bad; } mdrain_arg;
static void *mod_drainer(void *arg)
{
    mdrain_arg *a = (mdrain_arg *)arg;
    uint8_t *got = (uint8_t *)malloc(NB), *want = (uint8_t *)malloc(NB);
    int16_t *iq = (int16_t *)malloc(NB);
    uint64_t g = 0; long long acc = 0; int prev = 0;
    while (g < a->total && !a->bad) {
        const size_t n = cl_smi_drain_bytes(a->smi, got, NB);
        if (!n) { sched_yield(); usleep(20); continue; }
        assert(n % 4 == 0);
        for (size_t k = 0; k < n / 4; k++) {
            int I, Q; sample_of(g + k, &I, &Q);
            acc += I;
            int q = I + prev; if (q > 4095) q = 4095; if (q < -4096) q = -4096;
            iq[2 * k] = (int16_t)((int)(((acc % 8191) + 8191) % 8191) - 4095); iq[2 * k + 1] = (int16_t)q;
            prev = I;
        }
        orc_generate_data(ORC_TX_DOCUMENTED, iq, n / 4, want);
        if (memcmp(got, want, n)) { fprintf(stderr, "modulated TX bytes differ behind message %llu\n", (unsigned long long)g); a->bad = 1; }
        g += n / 4;
    }
    free(got); free(want); free(iq);
    return NULL;
}

static int part_e(int n_mtus)
{
    cl_device *d[N_TX]; cl_stream *st[N_TX]; mdrain_arg da[N_TX]; pthread_t th[N_TX];
    const char *sk[] = {"MOD"}, *sv[] = {"FM:75000"};
    for (int i = 0; i < N_TX; i++) {
        cl_smi *smi0;
        {   /* (the pack mode is the pipe's, taken when the stream is set up) */
            const char *dk[] = {"driver", "channel"}, *dv[] = {"Cariboulite", "S1G"};
            d[i] = cl_device_make(dk, dv, 2); assert(d[i]);
            smi0 = cl_device_smi(d[i]);
            cl_smi_set_tx_mode(smi0, CL_TX_DOCUMENTED);
            st[i] = cl_setupStream(d[i], CL_SOAPY_SDR_TX, "CF32", NULL, 0, sk, sv, 1);
            if (!st[i]) { fprintf(stderr, "setupStream: %s\n", cl_device_last_error(d[i])); abort(); }
            assert(cl_activateStream(d[i], st[i], 0, 0, 0) == 0);
        }
        da[i] = (mdrain_arg){smi0, (uint64_t)n_mtus * MTU, 0};
    }
    const char *gk[] = {"COPY_THREADS"}, *gv[] = {"2"};
    cl_group *grp = cl_group_make(d, N_TX, gk, gv, 1);
    if (!grp) { fprintf(stderr, "cl_group_make: %s\n", cl_group_last_error(NULL)); return -1; }
    clhip_mock_tx_fail_every(7);
    for (int i = 0; i < N_TX; i++) pthread_create(&th[i], NULL, mod_drainer, &da[i]);
    float *buf = (float *)malloc(MTU * 8);                          /* (every member is written the same samples: one buffer) */
    const void *bufs[N_TX]; int rets[N_TX];
    uint64_t g = 0; long calls = 0, lone = 0; int bad = 0;
    while (g < da[0].total && !bad) {
        const size_t num = calls % 3 == 2 ? MTU / 2 : MTU;
        for (size_t k = 0; k < num; k++) { int I, Q; sample_of(g + k, &I, &Q); buf[2 * k] = (float)I / 4096.0f; buf[2 * k + 1] = (float)Q / 4096.0f; }
        const int out = calls % 5 == 4 ? (int)(calls / 5) % N_TX : -1;
        for (int i = 0; i < N_TX; i++) bufs[i] = i == out ? NULL : buf;
        if (out >= 0) {
            const void *b[1] = {buf};
            if (cl_writeStream(d[out], st[out], b, num, NULL, 0, 1000) != (int)num) { fprintf(stderr, "writeStream: %s\n", cl_device_last_error(d[out])); bad = 1; break; }
            lone++;
        }
        const int nd = cl_group_writeStream(grp, bufs, num, rets, 1000);
        calls++;
        if (nd != N_TX - (out >= 0)) { fprintf(stderr, "cl_group_writeStream: %d (%s)\n", nd, cl_group_last_error(grp)); bad = 1; break; }
        for (int i = 0; i < N_TX; i++) if (rets[i] != (i == out ? 0 : (int)num)) bad = 1;
        for (int i = 0; i < N_TX; i++) bad |= da[i].bad;
        g += num;
    }
    if (bad) for (int i = 0; i < N_TX; i++) da[i].total = 0;         /* (let the drainers go) */
    for (int i = 0; i < N_TX; i++) { pthread_join(th[i], NULL); bad |= da[i].bad; }
    cl_group_stats gs; cl_group_getStats(grp, &gs);
    uint64_t overruns = 0;
    for (int i = 0; i < N_TX; i++) { cl_stream_stats ss; cl_getStreamStats(d[i], st[i], &ss); overruns += ss.tx_overruns; }
    cl_group_unmake(grp);
    clhip_mock_tx_fail_every(0);
    for (int i = 0; i < N_TX; i++) cl_device_unmake(d[i]);
    free(buf);
    printf("E: %ld calls, %llu batched writes, %llu one by one, %ld lone, %llu launches, %llu repeated member-calls\n", calls, (unsigned long long)gs.batched_reads,
           (unsigned long long)gs.single_reads, lone, (unsigned long long)gs.launches, (unsigned long long)overruns);
    return bad || gs.errors || !gs.batched_reads || !gs.single_reads || !overruns || !lone ? -1 : 0;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 24;
    if (part_a("CS16", n) || part_a("CF32", n) || part_b(n) || part_c("CS16", n) || part_c("CF32", n) || part_d("CS16", n / 2, 0) || part_d("CF32", n / 2, 0) || part_d("CS16", n / 2, 3) || part_e(n)) {
        fprintf(stderr, "stream mock harness FAILED\n"); return 1;
    }
    printf("stream mock harness ok\n");
    return 0;
}
