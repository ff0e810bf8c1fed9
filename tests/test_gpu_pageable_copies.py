"""-m gpu: the library never hands a pageable range of caller-owned memory above 512 KiB to ONE runtime copy -- with the HIP
runtime's own configuration left at its DEFAULT (GPU_PINNED_MIN_XFER_SIZE unset: what a C client such as the Soapy module or
the C++ API runs with).  From 1 MiB up the runtime would pin such a range in place and let the copy engine into the caller's heap
-- the mechanism behind round 3's GPU page faults on host heap addresses -- so every copy of memory the
library does not own goes in pieces through the runtime's staging buffers (clhip_memcpy_h2d / _d2h), or not through the runtime
at all (pinned mirror + memcpy).  Deterministic, one process, run once: drives only library paths above 1 MiB and reads the
library's own counters and operation ring."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
assert "GPU_PINNED_MIN_XFER_SIZE" not in os.environ
from cariboulite_amd import soapy as S, hip, synth
MTU, NB = 131072, 524288
b, i, q = synth.smi_stream_bytes(8 * MTU, 0, stream=5)
iq = np.stack([i, q], 1)
res = {}
sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
# caribou_smi_read of four native batches: 2 MiB of samples + 512 KiB of meta into pageable numpy buffers
sdr.feedSmiBytes(b[: 4 * NB])
ret, got, meta = sdr.smiRead(0, 4 * MTU)
res["smi_read"] = [int(ret), bool(np.array_equal(got[: 4 * MTU], iq[: 4 * MTU]))]
# readStream CS16, 4 MTU in one call (not clamped): 2 MiB out
rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
sdr.feedSmiBytes(b[: 4 * NB])
big = np.zeros((4 * MTU, 2), np.int16)
res["read_cs16_4mtu"] = [int(sdr.readStream(rx, [big], 4 * MTU).ret), bool(np.array_equal(big, iq[: 4 * MTU]))]
# readStream CF32 through FIR64 + 3/2: 1.5 MiB out per call
rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"FIR": "64:1000000", "RESAMP": "3/2"})
sdr.feedSmiBytes(b[: 2 * NB])
out = np.zeros((MTU * 3 // 2, 2), np.float32)
res["read_fir_rs"] = [int(sdr.readStream(rx, [out], MTU).ret), int(sdr.readStream(rx, [out], MTU).ret)]
# readStream CF64: 2 MiB out per call
rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF64)
sdr.feedSmiBytes(b[:NB])
o64 = np.zeros((MTU, 2), np.float64)
res["read_cf64"] = int(sdr.readStream(rx, [o64], MTU).ret)
sdr.close()
# caribou_smi_write of four native batches (2 MiB in), writeStream CF64 (2 MiB in), drained to a pageable buffer (2 MiB out)
tx = S.Device(dict(driver="Cariboulite", channel="S1G"))
res["smi_write"] = int(tx.smiWrite(0, iq[: 4 * MTU].astype(np.int16)))
res["drained"] = int(tx.drainSmiBytes(1 << 22).size)
st = tx.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CF64)
res["write_cf64"] = int(tx.writeStream(st, [np.full((MTU, 2), 0.25, np.float64)], MTU).ret)
tx.close()
ctr = (C.c_uint64 * 4)()
hip.lib().clhip_debug_copy_counters(ctr)
res["counters"] = [int(v) for v in ctr]
class Rec(C.Structure):
    _fields_ = [("seq", C.c_uint64), ("op", C.c_uint32), ("pad", C.c_uint32), ("base", C.c_uint64), ("len", C.c_uint64)]
recs = (Rec * 256)()
n = hip.lib().clhip_debug_ops(recs, 256)
res["ops"] = [[int(r.op), int(r.len)] for r in recs[:n]]
print("RESULT " + json.dumps(res))
'''


def test_no_pageable_range_above_one_piece_reaches_a_runtime_copy(tmp_path):
    env = {k: v for k, v in os.environ.items() if k != "GPU_PINNED_MIN_XFER_SIZE"}
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    MTU = 131072
    assert res["smi_read"] == [4 * MTU, True] and res["read_cs16_4mtu"] == [4 * MTU, True]
    assert res["read_fir_rs"] == [MTU * 3 // 2] * 2 and res["read_cf64"] == MTU
    assert res["smi_write"] == 4 * MTU and res["drained"] == 16 * MTU and res["write_cf64"] == MTU
    in_pieces, locked_whole, largest_pageable, pageable_bytes = res["counters"]
    assert largest_pageable <= 512 << 10, res["counters"]                  # THE assertion
    assert in_pieces >= 2 and pageable_bytes >= 4 << 20                    # the pageable copies above 1 MiB did happen (smi_read, read of 4 MTU) ...
    assert locked_whole >= 1                                               # ... and the pinned FIFO / mirror routes took theirs whole (where a kernel does not store across PCIe itself)
    kinds = {op for op, _ in res["ops"]}
    assert kinds <= {4, 5, 6, 7}, kinds                                    # no registration of anybody's memory along the way
    assert all(ln > 512 << 10 for _, ln in res["ops"])                     # (the ring notes copies above one piece only)


@pytest.mark.gpu
def test_a_pageable_buffer_that_starts_in_a_registered_page_is_copied_in_pieces():
    """Registrations are whole pages: a client's registered buffer that ends in the middle of a page pins the head of whatever lies
    behind it.  A copy to or from that neighbour -- pageable, 2 MiB -- used to be taken for page-locked by its first byte and handed
    to the runtime whole, which refused it (found by the stream group's random walk: the lone devices' buffers lie right behind the
    registered ones).  The WHOLE range decides now: such a copy goes in pieces, both directions, bytes intact."""
    import numpy as np
    import torch
    from cariboulite_amd import hip
    assert torch.cuda.is_available()
    page = 4096
    raw = np.zeros(8 * (1 << 20) + 2 * page, np.uint8)
    base = (-raw.ctypes.data) % page                              # a page-aligned carve
    a = raw[base: base + (1 << 20) + 100]                         # ends 100 bytes into a page ...
    b = raw[base + (1 << 20) + 100: base + (1 << 20) + 100 + (2 << 20)]      # ... in which the pageable neighbour starts
    assert hip.lib().clhip_host_register(a.ctypes.data, a.size)
    try:
        c0 = np.zeros(4, np.uint64); hip.lib().clhip_debug_copy_counters(c0.ctypes.data)
        b[:] = np.arange(b.size, dtype=np.uint32).astype(np.uint8)
        d = torch.zeros(b.size, dtype=torch.uint8, device="cuda:0")
        s = hip.current_stream()
        assert hip.lib().clhip_memcpy_h2d(d.data_ptr(), b.ctypes.data, b.size, s) == 0
        back = np.zeros_like(b)
        assert hip.lib().clhip_memcpy_d2h(back.ctypes.data, d.data_ptr(), b.size, s) == 0
        torch.cuda.synchronize()
        assert np.array_equal(back, b) and np.array_equal(d.cpu().numpy(), b)
        c1 = np.zeros(4, np.uint64); hip.lib().clhip_debug_copy_counters(c1.ctypes.data)
        assert c1[0] - c0[0] >= 1 and c1[2] <= 512 << 10            # (in pieces; nothing above one piece went to the runtime in one go)
        # ... and a copy below the piece size, which the library hands to the runtime without asking what the memory is: the runtime
        # refuses a range that starts page-locked and ends pageable; the library then walks it page by page
        small = b[: 300000]
        back2 = np.zeros_like(small)
        assert hip.lib().clhip_memcpy_d2h(back2.ctypes.data, d.data_ptr(), small.size, s) == 0
        assert hip.lib().clhip_memcpy_h2d(d.data_ptr() + (1 << 20), small.ctypes.data, small.size, s) == 0
        torch.cuda.synchronize()
        assert np.array_equal(back2, small) and np.array_equal(d[1 << 20: (1 << 20) + small.size].cpu().numpy(), small)
        assert hip.lib().clhip_debug_sticky_error() == 0            # nothing handled is left pending for the next launch to trip over
    finally:
        torch.cuda.synchronize()
        hip.lib().clhip_host_unregister(a.ctypes.data)
