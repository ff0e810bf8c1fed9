/* CPU-only ThreadSanitizer harness for the stream group's copy pool (cariboulite_amd/csrc/host/cl_group.c: pool_start / pool_submit /
 * pool_drain / pool_stop are static there, so the file is compiled INTO this one).  What it hammers: a caller that submits rows of
 * different lengths (pieces of 512 KiB, the queue overflowing into copies on the calling thread), helps draining and waits for the
 * pieces still in flight, call after call, with 0 / 1 / 3 worker threads; every destination must equal its source after every drain.
 * No GPU call is made.  Built and run by tests/test_host_asan.py. */
#include "../../cariboulite_amd/csrc/host/cl_group.c"

#include <assert.h>

int main(void)
{
    const size_t rows = 12, row_bytes = (size_t)1536 << 10;
    uint8_t *src = (uint8_t *)malloc(rows * row_bytes), *dst = (uint8_t *)malloc(rows * row_bytes + 64);
    assert(src && dst);
    for (int threads = 0; threads <= 3; threads += threads ? 2 : 1) {
        copy_pool p;
        assert(pool_start(&p, threads, 8) == 0 && p.n_threads == threads);       /* a queue of 8 pieces: 12 rows x 3 pieces overflow it */
        for (int call = 0; call < 40; call++) {
            for (size_t k = 0; k < rows * row_bytes; k += 4099) src[k] = (uint8_t)(k + call + threads);
            memset(dst, 0xEE, rows * row_bytes + 64);
            for (size_t r = 0; r < rows; r++) {
                const size_t n = row_bytes - (r * 4093 + call) % 70000, off = (r & 1) ? 3 : 0;      /* ragged lengths, odd destinations */
                pool_submit(&p, dst + r * row_bytes + off, src + r * row_bytes, n - off);
            }
            pool_drain(&p);
            for (size_t r = 0; r < rows; r++) {
                const size_t n = row_bytes - (r * 4093 + call) % 70000, off = (r & 1) ? 3 : 0;
                assert(memcmp(dst + r * row_bytes + off, src + r * row_bytes, n - off) == 0);
                assert(dst[r * row_bytes + n] == 0xEE);                          /* nothing behind the row */
            }
            assert(p.in_flight == 0 && p.q_len == 0);
        }
        pool_stop(&p);
    }
    free(src); free(dst);
    printf("group pool tsan harness ok\n");
    return 0;
}
