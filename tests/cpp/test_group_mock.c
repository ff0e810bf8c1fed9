/* CPU-only sanitizer harness for the HOST side of the stream group (cariboulite_amd/csrc/host/cl_group.c with cl_smi.c, cl_soapy.c,
 * cl_ring.c) over tests/cpp/hip_mock/clhip_mock.c, where every HIP stream is a thread: what the group queues -- copies in out of the
 * members' FIFOs, launches that read offset tables and write mirrors, work made AHEAD of the client -- runs concurrently with the
 * caller, with one feeder thread per member and with the group's copy threads.  Built twice by tests/test_host_asan.py:
 * -fsanitize=thread (a host write under queued work is a race) and -fsanitize=address,undefined (a buffer freed or overrun under
 * queued work).  No GPU.
 *
 * Eight devices (+ one, below), two lanes (CS16 on S1G without stages; CF32 on HiF behind the mock's stand-in pipe, whose output depends on the
 * pipe's history and therefore on every run being made exactly once and in order).  The feeders push pieces of a quarter, half or
 * whole batch and let the FIFOs run between empty and four batches deep, so calls find anything from nothing to several batches
 * pending: full batches (batched route, read and computed ahead), short reads (the members' own devices).  Between group calls the
 * client reads a member through its own device, asks for half batches, registers and releases its buffers.  Every sample every
 * stream delivers is checked against the sequence that was fed: nothing lost, nothing twice, nothing out of order.  A ninth stream (in
 * the plain lane) is fed damage -- batches that start six junk bytes late, batches without any sync word -- so that the re-sync and
 * "-3" paths of its own device run inside the group's calls, next to the others; its samples are not checked, its neighbours' are. */
#include <assert.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "cariboulite_hip.h"

#define N_DEV 9
#define N_GOOD 8          /* streams 0 .. 7 are checked sample by sample; stream 8 is fed DAMAGE */
#define MTU 131072u
#define NB (4u * MTU)

static int n_batches = 20;
static cl_device *dev[N_DEV]; static cl_stream *st[N_DEV]; static cl_smi *smi[N_DEV];
static int chan_of(int i) { return i < N_GOOD / 2 || i == N_GOOD ? CL_CHANNEL_S1G : CL_CHANNEL_HIF; }
static int piped(int i) { return i >= N_GOOD / 2 && i < N_GOOD; }

static void sample_of(int i, uint64_t g, int *I, int *Q)
{
    *I = (int)((g * 7 + (uint64_t)i * 13) % 8191) - 4095;
    *Q = (int)((g * 3 + (uint64_t)i) % 8191) - 4095;
}

static void words_of(int i, uint64_t g0, size_t n, uint8_t *out)
{
    for (size_t k = 0; k < n; k++) {
        int I, Q; sample_of(i, g0 + k, &I, &Q);
        const uint32_t A = (uint32_t)(chan_of(i) == CL_CHANNEL_HIF ? Q : I) & 0x1FFF, B = (uint32_t)(chan_of(i) == CL_CHANNEL_HIF ? I : Q) & 0x1FFF;
        const uint32_t w = 0x80004000u | (A << 17) | (B << 1);          /* caribou_smi.c:338-340 */
        memcpy(out + 4 * k, &w, 4);
    }
}

/* what stream i delivers for its samples [g0, g0 + n): int16 pairs (plain lane) or the stand-in pipe's floats */
static int check(int i, uint64_t g0, size_t n, const void *got)
{
    for (size_t k = 0; k < n; k++) {
        int I, Q; sample_of(i, g0 + k, &I, &Q);
        if (!piped(i)) {
            const int16_t *p = (const int16_t *)got + 2 * k;
            if (p[0] != I || p[1] != Q) { fprintf(stderr, "stream %d sample %llu: got (%d, %d), fed (%d, %d)\n", i, (unsigned long long)(g0 + k), p[0], p[1], I, Q); return -1; }
        } else {
            int Ip = 0, Qp = 0;
            if (g0 + k) sample_of(i, g0 + k - 1, &Ip, &Qp);
            const float x = (float)I / 4096.0f, want_re = x + 0.5f * ((float)Ip / 4096.0f), want_im = (float)Q / 4096.0f;
            const float *p = (const float *)got + 2 * k;
            if (p[0] != want_re || p[1] != want_im) { fprintf(stderr, "stream %d sample %llu: got (%g, %g), want (%g, %g)\n", i, (unsigned long long)(g0 + k), p[0], p[1], want_re, want_im); return -1; }
        }
    }
    return 0;
}

static volatile int g_stop;
static void *feeder(void *arg)
{
    const int i = (int)(intptr_t)arg;
    uint8_t *piece = (uint8_t *)malloc(NB);
    uint32_t r = 777u + (uint32_t)i * 31u;
    uint64_t g = 0;
    const uint64_t total = (uint64_t)n_batches * MTU;
    while (g < total && !g_stop) {
        r = r * 1664525u + 1013904223u;
        const size_t depth = 1 + (r >> 28) % 4;                        /* let the FIFO run this deep before the next piece */
        if (cl_smi_pending_bytes(smi[i]) >= depth * NB) { sched_yield(); usleep(20); continue; }
        size_t n = MTU >> ((r >> 20) % 3);                             /* a whole, half or quarter batch */
        if (g + n > total) n = (size_t)(total - g);
        words_of(i, g, n, piece);
        if (i == N_GOOD && n == MTU) {                                  /* the damaged stream: every third whole batch slipped, every fifth lost */
            const uint64_t b = g / MTU;
            if (b % 3 == 1) { memmove(piece + 6, piece, 4 * n - 6); memset(piece, 0x11, 6); }
            else if (b % 5 == 2) memset(piece, 0, 4 * n);
        }
        if (cl_smi_feed_bytes(smi[i], piece, 4 * n)) { fprintf(stderr, "feed failed\n"); abort(); }
        g += n;
    }
    free(piece);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc > 1) n_batches = atoi(argv[1]);
    for (int i = 0; i < N_DEV; i++) {
        const char *dk[] = {"driver", "channel"}, *dv[] = {"Cariboulite", chan_of(i) == CL_CHANNEL_S1G ? "S1G" : "HiF"};
        dev[i] = cl_device_make(dk, dv, 2);
        assert(dev[i]);
        const char *sk[] = {"FIR"}, *sv[] = {"64:1000000"};
        st[i] = cl_setupStream(dev[i], CL_SOAPY_SDR_RX, piped(i) ? "CF32" : "CS16", NULL, 0, sk, sv, piped(i) ? 1 : 0);
        if (!st[i]) { fprintf(stderr, "setupStream: %s\n", cl_device_last_error(dev[i])); return 1; }
        assert(cl_activateStream(dev[i], st[i], 0, 0, 0) == 0);
        smi[i] = cl_device_smi(dev[i]);
    }
    const char *gk[] = {"SUBBATCH", "SLAB_MB", "COPY_THREADS"}, *gv[] = {"2", "2", "2"};       /* 2 MiB slices: four batches -- compaction and outgrowing happen */
    cl_group *grp = cl_group_make(dev, N_DEV, gk, gv, 3);
    if (!grp) { fprintf(stderr, "cl_group_make: %s\n", cl_group_last_error(NULL)); return 1; }

    pthread_t th[N_DEV];
    for (int i = 0; i < N_DEV; i++) pthread_create(&th[i], NULL, feeder, (void *)(intptr_t)i);

    void *bufs[N_DEV]; int rets[N_DEV]; uint64_t pos[N_DEV] = {0};
    for (int i = 0; i < N_DEV; i++) bufs[i] = malloc((MTU + 16) * 8);
    const uint64_t total = (uint64_t)n_batches * MTU;
    long iter = 0, lone_reads = 0, half_calls = 0, registered_calls = 0;
    int is_registered = 0;
    for (;; iter++) {
        int done = 1;
        for (int i = 0; i < N_GOOD; i++) done &= pos[i] == total;
        if (done) break;
        if (iter > 200000) { fprintf(stderr, "no progress\n"); g_stop = 1; return 1; }
        if (iter % 5 == 2) {                                           /* a member read through its own device between two group calls */
            const int i = (int)(iter / 5) % (N_GOOD / 2);
            void *b[1] = {bufs[i]};
            const int r = cl_readStream(dev[i], st[i], b, MTU, NULL, NULL, 1000);
            assert(r >= 0);
            if (check(i, pos[i], (size_t)r, bufs[i])) return 1;
            pos[i] += (uint64_t)r; lone_reads += r > 0;
        }
        if (iter % 11 == 5 && !is_registered) { assert(cl_group_register_buffers(grp, bufs, (MTU + 16) * 8) == 0); is_registered = 1; }
        else if (iter % 11 == 8 && is_registered) { cl_group_unregister_buffers(grp); is_registered = 0; }
        const size_t num = iter % 7 == 3 ? MTU / 2 : MTU;
        half_calls += num != MTU; registered_calls += is_registered;
        const int nd = cl_group_readStream(grp, bufs, num, rets, 1000);
        if (nd < 0) { fprintf(stderr, "cl_group_readStream: %s\n", cl_group_last_error(grp)); g_stop = 1; return 1; }
        for (int i = 0; i < N_DEV; i++) {
            assert(rets[i] >= 0 && (size_t)rets[i] <= num);
            if (i < N_GOOD && check(i, pos[i], (size_t)rets[i], bufs[i])) { g_stop = 1; return 1; }
            pos[i] += (uint64_t)rets[i];
        }
        if (!nd) { sched_yield(); usleep(50); }
    }
    g_stop = 1;                                                        /* (the damaged stream's feeder may still be waiting for room) */
    for (int i = 0; i < N_DEV; i++) pthread_join(th[i], NULL);
    cl_smi_flush_fifo(smi[N_GOOD]);
    if (is_registered) cl_group_unregister_buffers(grp);
    cl_group_stats gs; cl_group_getStats(grp, &gs);

    /* the group goes with batches read and computed ahead: they are the devices' again, pending, in order */
    uint8_t *three = (uint8_t *)malloc(3 * NB);
    for (int i = 0; i < N_GOOD; i++) { words_of(i, pos[i], 3 * MTU, three); assert(cl_smi_feed_bytes(smi[i], three, 3 * NB) == 0); }
    free(three);
    assert(cl_group_readStream(grp, bufs, MTU, rets, 1000) == N_GOOD);
    for (int i = 0; i < N_GOOD; i++) { assert(rets[i] == (int)MTU && check(i, pos[i], MTU, bufs[i]) == 0); pos[i] += MTU; }
    for (int i = 0; i < N_GOOD; i++) assert(cl_smi_pending_bytes(smi[i]) == 2 * NB);
    cl_group_unmake(grp);
    for (int i = 0; i < N_GOOD; i++) {
        assert(cl_smi_pending_bytes(smi[i]) == 2 * NB);
        if (piped(i)) continue;                                        /* (a lone device's pipe starts from rest: not the group's history) */
        for (int c = 0; c < 2; c++) {
            void *b[1] = {bufs[i]};
            assert(cl_readStream(dev[i], st[i], b, MTU, NULL, NULL, 1000) == (int)MTU && check(i, pos[i], MTU, bufs[i]) == 0);
            pos[i] += MTU;
        }
    }
    for (int i = 0; i < N_DEV; i++) { cl_device_unmake(dev[i]); free(bufs[i]); }
    printf("calls %llu batched %llu single %llu ahead %llu direct %llu launches %llu errors %llu; lone reads %ld, half-batch calls %ld, calls with registered buffers %ld\n",
           (unsigned long long)gs.calls, (unsigned long long)gs.batched_reads, (unsigned long long)gs.single_reads, (unsigned long long)gs.ahead_reads,
           (unsigned long long)gs.direct_reads, (unsigned long long)gs.launches, (unsigned long long)gs.errors, lone_reads, half_calls, registered_calls);
    if (gs.errors || gs.ahead_reads < 20 || gs.single_reads < 5 || gs.direct_reads < 5 || lone_reads < 3) { fprintf(stderr, "the run did not exercise what it is for\n"); return 1; }
    printf("group mock harness ok\n");
    return 0;
}
