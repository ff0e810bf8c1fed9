"""circular_buffer<T> semantics: the oracle restatement and the product's
cl_ring (host C, no GPU needed) against op sequences recorded from the reference's own template."""
import threading
import time

import numpy as np
import pytest

from conftest import load_golden


def _names():
    return [str(n) for n in load_golden("ring_cases.npz")["names"]]


def _replay(ring, g, name, get_kw):
    ops, rets, sizes, popped = g[f"{name}__ops"], g[f"{name}__rets"], g[f"{name}__sizes"], g[f"{name}__popped"]
    ctr, pp = 0, 0
    for (kind, n), want, sz in zip(ops.tolist(), rets.tolist(), sizes.tolist()):
        if kind == 0:
            d = np.arange(ctr, ctr + n, dtype=np.uint32); ctr += n
            assert ring.put(d) == want
        else:
            k, d = ring.get(n, **get_kw)
            assert k == want and np.array_equal(d, popped[pp:pp + k])
            pp += k
        assert ring.size() == sz
    assert pp == popped.size


@pytest.mark.parametrize("name", _names())
def test_oracle_ring_vs_reference_fixture(orc, name):
    g = load_golden("ring_cases.npz")
    size, ov, blk, cap = [int(v) for v in g[f"{name}__cfg"]]
    r = orc.Ring(size, ov, blk)
    assert r.capacity() == cap
    _replay(r, g, name, {})


@pytest.mark.parametrize("name", _names())
def test_product_ring_vs_reference_fixture(name):
    from cariboulite_amd import soapy
    g = load_golden("ring_cases.npz")
    size, ov, blk, cap = [int(v) for v in g[f"{name}__cfg"]]
    r = soapy.Ring(size, ov, blk)
    assert r.capacity() == cap
    _replay(r, g, name, {"timeout_us": 100})


def test_product_ring_blocking_get():
    """block_read: get() waits up to timeout_us for the WHOLE request, else returns 0 (circular_buffer.h:64-82)."""
    from cariboulite_amd import soapy
    r = soapy.Ring(1024, True, True)
    t0 = time.perf_counter()
    k, _ = r.get(10, timeout_us=50000)
    assert k == 0 and 0.03 < time.perf_counter() - t0 < 0.5
    r.put(np.arange(6))
    assert r.get(10, timeout_us=1000)[0] == 0 and r.size() == 6          # partial data stays queued
    th = threading.Thread(target=lambda: (time.sleep(0.05), r.put(np.arange(6, 10))))
    th.start()
    k, d = r.get(10, timeout_us=2_000_000)
    th.join()
    assert k == 10 and d.tolist() == list(range(10))
    # overwrite-oldest: the newest `capacity` items survive
    r.put(np.arange(3000, dtype=np.uint32)[:1000]); r.put(np.arange(1000, 1900, dtype=np.uint32))
    assert r.size() == 1024
    k, d = r.get(1024, timeout_us=1000)
    assert d[-1] == 1899 and d[0] == 1900 - 1024


def test_span_calls_on_host_storage_and_put_cancel():
    """The two-phase calls: _begin names at most two linear pieces of the storage, the caller moves the data, _end
    publishes / releases; put_cancel undoes a put nobody has seen, including the elements it would have displaced."""
    import ctypes as C
    from cariboulite_amd import soapy
    L = soapy.lib()

    class Span(C.Structure):
        _fields_ = [("pos", C.c_size_t * 2), ("len", C.c_size_t * 2)]

    r = soapy.Ring(100, True, True)                     # capacity 128 uint32, host storage
    assert L.cl_ring_on_device(r.h) == 0
    base = L.cl_ring_storage(r.h)
    store = (C.c_uint32 * 128).from_address(base)
    sp = Span()
    assert r.put(np.arange(100, dtype=np.uint32)) == 100 and r.get(90, timeout_us=100)[0] == 90       # written 100, released 90
    n = L.cl_ring_put_begin(r.h, 60, C.byref(sp))
    assert n == 60 and (sp.pos[0], sp.len[0], sp.pos[1], sp.len[1]) == (100, 28, 0, 32)             # wraps at 128
    for k in range(28):
        store[100 + k] = 1000 + k
    for k in range(32):
        store[k] = 1028 + k
    L.cl_ring_put_end(r.h, n)
    assert r.size() == 70
    # a put that must displace the oldest elements, then thinks better of it: nothing changed
    n = L.cl_ring_put_begin(r.h, 100, C.byref(sp))
    assert n == 100
    L.cl_ring_put_cancel(r.h)
    assert r.size() == 70
    k = L.cl_ring_get_begin(r.h, 70, 1000, C.byref(sp))
    assert k == 70 and (sp.pos[0], sp.len[0], sp.pos[1], sp.len[1]) == (90, 38, 0, 32)
    got = [store[90 + i] for i in range(38)] + [store[i] for i in range(32)]
    L.cl_ring_get_end(r.h, k)
    assert got == list(range(90, 100)) + list(range(1000, 1060))
    assert r.size() == 0 and L.cl_ring_get_begin(r.h, 1, 1000, C.byref(sp)) == 0                   # whole-request rule: times out unlocked
    assert r.put(np.arange(5, dtype=np.uint32)) == 5                                               # (so the ring is not left locked)
