/* test_ring_tsan.c -- the sample ring's span calls with ONE producer and ONE consumer thread, host storage, under
 * ThreadSanitizer (tests/test_host_asan.py builds it with -fsanitize=thread).  The ring is not locked between _begin and
 * _end: an open put owns the slots behind the newest element, an open get the oldest ones, and a put that has to displace
 * the oldest elements of a full ring waits for an open get to end -- so the two threads' accesses to the storage must never
 * overlap (TSan reports it if they do), and the consumer must see strictly increasing values (the producer writes a counter;
 * an overwrite-oldest ring may skip values, never repeat or reorder them). */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "cariboulite_hip.h"

typedef struct { cl_ring *r; int rounds; unsigned seed; int fail; uint64_t got; } ctx_t;

static unsigned rnd(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static void *producer(void *a)
{
    ctx_t *c = (ctx_t *)a;
    uint32_t *st = (uint32_t *)cl_ring_storage(c->r);
    uint32_t next = 1;
    for (int i = 0; i < c->rounds; i++) {
        const size_t want = 1 + rnd(&c->seed) % 700;
        cl_ring_span sp;
        const size_t n = cl_ring_put_begin(c->r, want, &sp);
        const int mode = rnd(&c->seed) % 16;                       /* mostly publish; sometimes give up untouched / touched */
        if (mode == 0) { cl_ring_put_cancel(c->r); continue; }
        uint32_t v = next;
        for (int k = 0; k < 2; k++)
            for (size_t j = 0; j < sp.len[k]; j++) st[sp.pos[k] + j] = v++;
        if ((rnd(&c->seed) & 7) == 0) usleep(rnd(&c->seed) % 50);
        if (mode == 1) { cl_ring_put_abandon(c->r); continue; }    /* nothing published: the counter does not advance */
        cl_ring_put_end(c->r, n);
        next += (uint32_t)n;
    }
    return NULL;
}

static void *consumer(void *a)
{
    ctx_t *c = (ctx_t *)a;
    const uint32_t *st = (const uint32_t *)cl_ring_storage(c->r);
    uint32_t last = 0;
    for (int i = 0; i < c->rounds; i++) {
        cl_ring_span sp;
        const size_t n = cl_ring_get_begin(c->r, 1 + rnd(&c->seed) % 900, 200, &sp);
        if (!n) continue;
        for (int k = 0; k < 2; k++)
            for (size_t j = 0; j < sp.len[k]; j++) {
                const uint32_t v = st[sp.pos[k] + j];
                if (v <= last) c->fail = 1;
                last = v;
            }
        if ((rnd(&c->seed) & 7) == 0) usleep(rnd(&c->seed) % 50);
        cl_ring_get_end(c->r, n);
        c->got += n;
    }
    return NULL;
}

int main(void)
{
    for (int cfg = 0; cfg < 4; cfg++) {
        const int override_write = cfg & 1, block_read = (cfg >> 1) & 1;
        cl_ring *r = cl_ring_create(1000, sizeof(uint32_t), override_write, block_read);
        if (!r) { printf("ring create failed\n"); return 1; }
        ctx_t p = {r, 20000, 12345u + (unsigned)cfg, 0, 0}, q = {r, 20000, 999u + (unsigned)cfg, 0, 0};
        pthread_t tp, tq;
        pthread_create(&tp, NULL, producer, &p);
        pthread_create(&tq, NULL, consumer, &q);
        pthread_join(tp, NULL);
        pthread_join(tq, NULL);
        if (q.fail) { printf("cfg %d: values repeated or reordered\n", cfg); return 1; }
        printf("cfg %d (override %d, whole requests %d): %llu elements consumed in order\n", cfg, override_write, block_read, (unsigned long long)q.got);
        cl_ring_destroy(r);
    }
    printf("ring tsan harness ok\n");
    return 0;
}
