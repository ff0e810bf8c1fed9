// ref_ring_harness.cpp -- exported C wrappers around the REFERENCE's own circular_buffer<T>
// (software/libcariboulite/src/datatypes/circular_buffer.h), #included from where it lies under
// /root/reference at build time by oracle/Makefile (`make ref`), output oracle/_ref/libref_ring.so.
// TEST INFRASTRUCTURE ONLY; contains no reference code.
#include <cstdint>

#include "datatypes/circular_buffer.h"   // -I$(REF)/software/libcariboulite/src

typedef circular_buffer<uint32_t> ring_t;

extern "C" {
void *ref_ring_new(size_t size, int override_write, int block_read) { return new ring_t(size, override_write != 0, block_read != 0); }
void ref_ring_free(void *r) { delete (ring_t *)r; }
size_t ref_ring_put(void *r, const uint32_t *data, size_t n) { return ((ring_t *)r)->put(data, n); }
size_t ref_ring_get(void *r, uint32_t *data, size_t n, int timeout_us) { return ((ring_t *)r)->get(data, n, timeout_us); }
size_t ref_ring_size(void *r) { return ((ring_t *)r)->size(); }
size_t ref_ring_capacity(void *r) { return ((ring_t *)r)->capacity(); }
void ref_ring_reset(void *r) { ((ring_t *)r)->reset(); }
}
