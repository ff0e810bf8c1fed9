// CPU-box test of clfan.cpp's world > 1 branch against the in-process RCCL model (tests/cpp/rccl_mock): every rank is a
// thread, every "device" buffer host memory.  Checks, for several (world, streams, root, row length, strides):
//   scatter  every rank ends up with exactly its streams (s mod world == rank, increasing s), byte for byte, and nothing
//            outside the rows (the stride gaps, the guard zone behind the last row) is written;
//   gather   the inverse puts every stream's (modified) row back where the root's layout wants it, gaps untouched;
//   schedule all transfers of a call sit in ONE group (all links busy at once), the root posts one transfer per remote
//            stream, a peer one per stream it owns, and nobody sends to itself.
// A schedule that would hang the real library (a send nobody receives, a receive nobody feeds) fails here by timeout
// with the operation named.  Usage: test_clfan_mock  -> "OK clfan mock" and exit code 0.
#include <pthread.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/cariboulite_fanout.h"

struct Case { int world, n_streams, root; size_t bytes, root_stride, mine_stride; };

struct Shared {
    Case cs;
    uint8_t id[CLFAN_ID_BYTES];
    pthread_barrier_t bar;
    std::vector<uint8_t> root_in, root_out;      // the root's layouts (scatter source, gather target)
    int failures;
    pthread_mutex_t mu;
};

struct RankArg { Shared *sh; int rank; };

static uint8_t pat(int stream, size_t k) { return (uint8_t)(stream * 131 + k * 7 + (k >> 8) * 3 + 1); }
static const uint8_t GUARD = 0xEE;

static void fail(Shared *sh, const char *what, int rank, long a, long b)
{
    pthread_mutex_lock(&sh->mu);
    if (sh->failures++ < 8)
        fprintf(stderr, "FAIL world %d streams %d root %d bytes %zu: rank %d: %s (%ld, %ld) [clfan: %s] [mock: %s]\n", sh->cs.world,
                sh->cs.n_streams, sh->cs.root, sh->cs.bytes, rank, what, a, b, clfan_last_error(), rccl_mock_last_error());
    pthread_mutex_unlock(&sh->mu);
}

static void *rank_main(void *p)
{
    RankArg *ra = (RankArg *)p;
    Shared *sh = ra->sh;
    const Case &cs = sh->cs;
    const int me = ra->rank, W = cs.world;
    clfan_comm *c = clfan_create(sh->id, W, me);
    if (!c) { fail(sh, "clfan_create", me, 0, 0); pthread_barrier_wait(&sh->bar); pthread_barrier_wait(&sh->bar); return NULL; }
    const int n_mine = clfan_local_count(cs.n_streams, W, me);
    std::vector<uint8_t> mine((size_t)n_mine * cs.mine_stride + 64, GUARD);
    uint8_t *d_mine = n_mine ? mine.data() : NULL;

    // ---- scatter
    int rc = clfan_scatter_streams(c, cs.root, me == cs.root ? sh->root_in.data() : NULL, cs.root_stride, cs.bytes, cs.n_streams,
                                   d_mine, cs.mine_stride, NULL);
    if (rc) fail(sh, "clfan_scatter_streams returned", me, rc, 0);
    for (int j = 0; j < n_mine && !rc; j++) {
        const int st = me + j * W;
        const uint8_t *row = mine.data() + (size_t)j * cs.mine_stride;
        for (size_t k = 0; k < cs.bytes; k++)
            if (row[k] != pat(st, k)) { fail(sh, "scattered row differs at (stream, byte)", me, st, (long)k); break; }
        for (size_t k = cs.bytes; k < cs.mine_stride; k++)
            if (row[k] != GUARD) { fail(sh, "scatter wrote into a stride gap at (stream, byte)", me, st, (long)k); break; }
    }
    for (size_t k = (size_t)n_mine * cs.mine_stride; k < mine.size(); k++)
        if (mine[k] != GUARD) { fail(sh, "scatter wrote behind the last row", me, (long)k, 0); break; }
    unsigned long groups = 0, sends = 0, recvs = 0, maxops = 0;
    if (W > 1 && cs.n_streams && cs.bytes) {
        rccl_mock_counters(&groups, &sends, &recvs, &maxops);
        const unsigned long remote = (unsigned long)(cs.n_streams - clfan_local_count(cs.n_streams, W, cs.root));
        const unsigned long want_s = me == cs.root ? remote : 0, want_r = me == cs.root ? 0 : (unsigned long)n_mine;
        if (sends != want_s || recvs != want_r) fail(sh, "scatter posted (sends, recvs)", me, (long)sends, (long)recvs);
        if ((want_s + want_r) && (groups != 1 || maxops != want_s + want_r)) fail(sh, "scatter is not ONE group: (groups, largest)", me, (long)groups, (long)maxops);
    }
    pthread_barrier_wait(&sh->bar);

    // ---- every owner marks its rows, then gather
    for (int j = 0; j < n_mine; j++)
        for (size_t k = 0; k < cs.bytes; k++) mine[(size_t)j * cs.mine_stride + k] ^= 0x5A;
    rc = clfan_gather_streams(c, cs.root, d_mine, cs.mine_stride, cs.bytes, cs.n_streams, me == cs.root ? sh->root_out.data() : NULL,
                              cs.root_stride, NULL);
    if (rc) fail(sh, "clfan_gather_streams returned", me, rc, 0);
    pthread_barrier_wait(&sh->bar);
    if (me == cs.root && !rc) {
        for (int st = 0; st < cs.n_streams; st++) {
            const uint8_t *row = sh->root_out.data() + (size_t)st * cs.root_stride;
            for (size_t k = 0; k < cs.bytes; k++)
                if (row[k] != (uint8_t)(pat(st, k) ^ 0x5A)) { fail(sh, "gathered row differs at (stream, byte)", me, st, (long)k); break; }
            for (size_t k = cs.bytes; k < cs.root_stride; k++)
                if (row[k] != GUARD) { fail(sh, "gather wrote into a stride gap at (stream, byte)", me, st, (long)k); break; }
        }
        for (size_t k = (size_t)cs.n_streams * cs.root_stride; k < sh->root_out.size(); k++)
            if (sh->root_out[k] != GUARD) { fail(sh, "gather wrote behind the last row", me, (long)k, 0); break; }
    }
    clfan_destroy(c);
    return NULL;
}

static int run_case(const Case &cs)
{
    Shared sh;
    sh.cs = cs;
    sh.failures = 0;
    pthread_mutex_init(&sh.mu, NULL);
    pthread_barrier_init(&sh.bar, NULL, (unsigned)cs.world);
    if (clfan_unique_id(sh.id)) { fprintf(stderr, "clfan_unique_id failed\n"); return 1; }
    sh.root_in.assign((size_t)cs.n_streams * cs.root_stride + 64, GUARD);
    sh.root_out.assign((size_t)cs.n_streams * cs.root_stride + 64, GUARD);
    for (int st = 0; st < cs.n_streams; st++)
        for (size_t k = 0; k < cs.bytes; k++) sh.root_in[(size_t)st * cs.root_stride + k] = pat(st, k);
    std::vector<pthread_t> th((size_t)cs.world);
    std::vector<RankArg> args((size_t)cs.world);
    for (int r = 0; r < cs.world; r++) { args[r].sh = &sh; args[r].rank = r; pthread_create(&th[r], NULL, rank_main, &args[r]); }
    for (int r = 0; r < cs.world; r++) pthread_join(th[r], NULL);
    pthread_barrier_destroy(&sh.bar);
    return sh.failures;
}

// a deliberately broken exchange straight on the model: the harness must be able to SEE a deadlock
static void *lonely_sender(void *p)
{
    ncclUniqueId *id = (ncclUniqueId *)p;
    ncclComm_t c;
    if (ncclCommInitRank(&c, 2, *id, 0) != ncclSuccess) return (void *)1;
    char b[8] = {0};
    const ncclResult_t r = ncclSend(b, 8, ncclUint8, 1, c, NULL);
    ncclCommDestroy(c);
    return (void *)(intptr_t)(r == ncclSuccess);      // success would mean the model lets a hang through
}
static void *silent_peer(void *p)
{
    ncclUniqueId *id = (ncclUniqueId *)p;
    ncclComm_t c;
    if (ncclCommInitRank(&c, 2, *id, 1) != ncclSuccess) return (void *)1;
    ncclCommDestroy(c);
    return NULL;
}

int main(void)
{
    int bad = 0;
    {
        rccl_mock_set_timeout_ms(300);
        ncclUniqueId id;
        ncclGetUniqueId(&id);
        pthread_t a, b;
        void *ra, *rb;
        pthread_create(&a, NULL, lonely_sender, &id);
        pthread_create(&b, NULL, silent_peer, &id);
        pthread_join(a, &ra); pthread_join(b, &rb);
        if (ra || rb) { fprintf(stderr, "FAIL: the model did not report an unmatched send\n"); bad++; }
        rccl_mock_set_timeout_ms(20000);
    }
    const int worlds[] = {1, 2, 3, 8};
    const int streams[] = {0, 1, 2, 7, 8, 9, 37, 256};
    const size_t lens[] = {1, 1000, 4099};
    int n_cases = 0;
    for (int W : worlds)
        for (int ns : streams)
            for (size_t len : lens) {
                if (ns == 256 && len != 1000) continue;
                const int roots[2] = {0, W - 1};
                for (int ri = 0; ri < (W > 1 ? 2 : 1); ri++) {
                    const Case cs = {W, ns, roots[ri], len, len + 24 + (size_t)(ns % 3), len + 8 + (size_t)(W % 5)};
                    bad += run_case(cs);
                    n_cases++;
                }
            }
    if (bad) { fprintf(stderr, "%d failure(s) in %d cases\n", bad, n_cases); return 1; }
    printf("OK clfan mock: %d cases (worlds 1, 2, 3, 8; ragged strides; both root positions)\n", n_cases);
    return 0;
}
