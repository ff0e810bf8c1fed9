/* CPU-only sanitizer harness (gcc -fsanitize=address,undefined): the host-side bookkeeping of the ingest path --
 * the three-cursor byte FIFO (stage in place / confirm / unstage / un-pop, moves of the buffer) and the sample ring's
 * span calls on host storage -- driven by random operation sequences against a plain model.  No GPU call is made
 * (host-memory FIFO and ring); built and run by tests/test_host_asan.py. */
#include <assert.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cl_internal.h"

static uint32_t rng_state = 12345;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

/* model: a byte vector; `staged` bytes at its front have been handed out in place (oldest first) */
static uint8_t *m_base, *model; static size_t m_len, m_cap, m_staged, m_off;     /* model = m_base + m_off */
static void m_push(const uint8_t *p, size_t n)
{
    if (m_off + m_len + n > m_cap) {
        if (m_off) { memmove(m_base, m_base + m_off, m_len); m_off = 0; }
        if (m_len + n > m_cap) { m_cap = (m_len + n) * 2 + 64; m_base = (uint8_t *)realloc(m_base, m_cap); }
    }
    model = m_base + m_off;
    memcpy(model + m_len, p, n); m_len += n;
}
static void m_drop_front(size_t n) { m_off += n; m_len -= n; model = m_base + m_off; }

static void fifo_fuzz(void)
{
    cl_fifo f; memset(&f, 0, sizeof f);
    uint8_t counter = 0, tmp[5000];
    uint8_t *last_stage = NULL; size_t last_stage_n = 0;
    uint8_t *slices[4096]; int n_slices = 0, adoptions = 0, outgrown = 0;     /* memory lent to the FIFO (a stream group's slab): the FIFO never frees it */
    for (int it = 0; it < 200000; it++) {
        const uint32_t op = rnd() % 100;
        if (it % 97 == 0 && n_slices < 4096) {
            /* the FIFO moves into a slice somebody else owns (cl_group.c: the group's pinned slab) -- with staged bytes, pending bytes, a
             * front stash, whatever it holds -- or back out into a buffer of its own; either way not a byte changes */
            const int was_external = f.external;
            if (was_external && (rnd() & 1)) { assert(cl_fifo_leave(&f) == 0 && !f.external); }
            else {
                const size_t live = f.head - f.keep + f.len, cap = live + rnd() % 6000;
                uint8_t *sl = (uint8_t *)malloc(cap ? cap : 1);
                slices[n_slices++] = sl;
                assert(cl_fifo_adopt(&f, sl, cap) == 0 && f.external && f.data == sl);
                adoptions++;
            }
            last_stage = NULL;
            assert(cl_fifo_pending(&f) == m_len - m_staged && f.head - f.keep == m_staged);
            if (m_staged) assert(memcmp(f.data + f.keep, model, m_staged) == 0);
            if (f.len) assert(memcmp(f.data + f.head, model + m_staged + cl_fifo_front_len(&f), f.len) == 0);
        }
        const int ext_before = f.external;
        if (op < 35) {                                          /* push / reserve+commit */
            size_t n = rnd() % (m_len > 200000 ? 50 : 3000);     /* keep the backlog bounded */
            for (size_t k = 0; k < n; k++) tmp[k] = counter++;
            if (op & 1) { assert(cl_fifo_push(&f, tmp, n) == 0); }
            else { uint8_t *p = cl_fifo_reserve(&f, n + 17); assert(p); memcpy(p, tmp, n); cl_fifo_commit(&f, n); }
            m_push(tmp, n);
            last_stage = NULL;                                  /* the buffer may have moved */
        } else if (op < 55) {                                   /* stage in place */
            size_t want = rnd() % 2500; uint8_t *where = NULL;
            size_t got = cl_fifo_stage(&f, want, &where);
            size_t pend = m_len - m_staged;
            if (cl_fifo_front_len(&f)) assert(got == 0);        /* older bytes in the front stash: the in-place readers step aside */
            else assert(got == (want < pend ? want : pend));
            assert(got == 0 || memcmp(where, model + m_staged, got) == 0);
            m_staged += got; last_stage = where; last_stage_n = got;
        } else if (op < 70) {                                   /* confirm the oldest staged bytes */
            size_t n = m_staged ? rnd() % (m_staged + 1) : 0;
            cl_fifo_confirm(&f, n);
            m_drop_front(n); m_staged -= n;
        } else if (op < 80) {                                   /* unstage the newest staged bytes */
            size_t n = m_staged ? rnd() % (m_staged + 1) : 0;
            cl_fifo_unstage(&f, n);
            m_staged -= n;
            if (n) last_stage = NULL;
        } else if (op < 95) {                                   /* pop with a copy (only legal with nothing staged) */
            if (m_staged) { cl_fifo_unstage(&f, m_staged); m_staged = 0; }
            size_t want = rnd() % 4000;
            size_t got = cl_fifo_pop(&f, (op & 1) ? tmp : NULL, want);
            assert(got == (want < m_len ? want : m_len));
            if (op & 1) assert(memcmp(tmp, model, got) == 0);
            m_drop_front(got);
        } else if (op < 97) {                                   /* staged bytes are still readable in place */
            if (last_stage && last_stage_n && m_staged >= last_stage_n)
                assert(memcmp(last_stage, model + m_staged - last_stage_n, last_stage_n) == 0);
        } else {
            /* a batched reader pops k bytes with a copy and gives the last j back (the "-3" exit, caribou_smi.c:665-668)
             * WHILE a producer holds a reservation it is still writing into: the buffer must not move under it */
            if (m_staged) { cl_fifo_unstage(&f, m_staged); m_staged = 0; }
            const size_t rn = rnd() % 3000;
            uint8_t *res = cl_fifo_reserve(&f, rn + 1);         /* producer: reserve, lock dropped, read(fd) in progress */
            assert(res);
            const size_t k = rnd() % 4000;
            const size_t got = cl_fifo_pop(&f, tmp, k);
            assert(got == (k < m_len ? k : m_len) && memcmp(tmp, model, got) == 0);
            const size_t j = got ? rnd() % (got + 1) : 0;
            assert(cl_fifo_unpop(&f, tmp + got - j, j) == 0);   /* consumer: give the tail back */
            m_drop_front(got - j);
            uint8_t fill[3000];
            for (size_t q = 0; q < rn; q++) fill[q] = counter++;
            memcpy(res, fill, rn);                              /* producer: the read() lands where the reservation pointed */
            cl_fifo_commit(&f, rn);
            m_push(fill, rn);
            last_stage = NULL;
        }
        assert(cl_fifo_pending(&f) == m_len - m_staged && f.head - f.keep == m_staged && f.head + f.len <= f.cap);
        assert(!(m_staged && cl_fifo_front_len(&f)));           /* staged bytes and a front stash never coexist */
        if (ext_before && !f.external) outgrown++;              /* a push found the lent slice too small: the FIFO has a buffer of its own again */
    }
    assert(adoptions > 500 && outgrown > 50);
    cl_fifo_free(&f);                                           /* (must not free a slice it was lent) */
    for (int k = 0; k < n_slices; k++) free(slices[k]);
    free(m_base); m_base = model = NULL; m_len = m_cap = m_staged = m_off = 0;
}

static void ring_fuzz(int override_write, int block_read)
{
    const size_t cap = 256;
    cl_ring *r = cl_ring_create(200, 4, override_write, block_read);
    assert(r && cl_ring_capacity(r) == cap && !cl_ring_on_device(r));
    uint32_t *store = (uint32_t *)cl_ring_storage(r);
    uint64_t w = 0, rd = 0;                                     /* model: elements w-1.. hold their own index */
    for (int it = 0; it < 100000; it++) {
        cl_ring_span sp;
        if (rnd() & 1) {
            size_t n = rnd() % 300;
            const uint64_t rd_before = rd;
            size_t take = cl_ring_put_begin(r, n, &sp);
            size_t held = (size_t)(w - rd);
            if (override_write && n > cap - held) { size_t drop = n - (cap - held); if (drop > held) drop = held; rd += drop; held -= drop; }
            assert(take == (n < cap - held ? n : cap - held));
            assert(sp.len[0] + sp.len[1] == take && sp.pos[0] == (size_t)(w & (cap - 1)) && (sp.len[1] == 0 || sp.pos[0] + sp.len[0] == cap));
            if (rnd() % 8 == 0) {                               /* cancel: nothing published, nothing displaced */
                cl_ring_put_cancel(r);
                rd = rd_before;
            } else {
                uint64_t v = w;
                for (int k = 0; k < 2; k++) for (size_t i = 0; i < sp.len[k]; i++) store[sp.pos[k] + i] = (uint32_t)v++;
                cl_ring_put_end(r, take); w += take;
            }
        } else {
            size_t n = rnd() % 300;
            size_t held = (size_t)(w - rd);
            size_t got = cl_ring_get_begin(r, n, 0, &sp);
            size_t want = block_read ? (held >= n ? n : 0) : (n < held ? n : held);
            assert(got == want);
            if (got) {
                uint64_t v = rd;
                for (int k = 0; k < 2; k++) for (size_t i = 0; i < sp.len[k]; i++) assert(store[sp.pos[k] + i] == (uint32_t)v++);
                cl_ring_get_end(r, got); rd += got;
            }
        }
        assert(cl_ring_size(r) == (size_t)(w - rd));
    }
    cl_ring_destroy(r);
}

int main(void)
{
    fifo_fuzz();
    for (int ov = 0; ov < 2; ov++) for (int blk = 0; blk < 2; blk++) ring_fuzz(ov, blk);
    printf("fifo + ring sanitizer harness ok\n");
    return 0;
}
