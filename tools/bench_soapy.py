#!/usr/bin/env python3
"""PCIe-inclusive rate of the SoapySDR-shaped boundary: host SMI bytes in (cl_smi_feed_bytes), host samples out
(readStream, one MTU = 131072 samples per call), one stream.  Never bench.py's `value`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = ["cs16", "cf32", "cs16_iir", "cf32_fir64_rs_3_2", "cs16_async_ring", "zc_cs16", "zc_cf32", "zc_cs16_iir", "zc_cf32_fir64_rs_3_2",
         "tx_cs16", "tx_cf32", "tx_cf32_fm_rs_2_3"]
if not os.environ.get("BENCH_SOAPY_ONLY"):
    # one fresh process per case (this one never touches the GPU): what a call costs must not depend on what the process did
    # before -- a write from heap pages that an earlier ZEROCOPY session of the same process had registered and released took
    # 290-390 us instead of 58 (the CPU's own reads of those pages are slow; nothing of the library's is involved)
    import subprocess
    res = {}
    for c in CASES:
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, BENCH_SOAPY_ONLY=c), capture_output=True, text=True, check=True).stdout
        res.update(json.loads(out[out.index("{"):]))
    print(json.dumps(res, indent=1))
    sys.exit(0)

import numpy as np
from cariboulite_amd import soapy as S, synth

MTU, NB = 131072, 524288
K = 64
b, _, _ = synth.smi_stream_bytes(K * MTU, 0, stream=1)
res = {}
for name, fmt, dt, width, args, bw in (
        ("cs16", S.SOAPY_SDR_CS16, np.int16, 2, None, None),
        ("cf32", S.SOAPY_SDR_CF32, np.float32, 2, None, None),
        ("cs16_iir", S.SOAPY_SDR_CS16, np.int16, 2, None, 100e3),
        ("cf32_fir64_rs_3_2", S.SOAPY_SDR_CF32, np.float32, 2, {"FIR": "64:1000000", "RESAMP": "3/2"}, None),
        ("cs16_async_ring", S.SOAPY_SDR_CS16, np.int16, 2, {"ASYNC": "1"}, None),
        # ZEROCOPY=1 + registerStreamBuffer: the client's buffer is registered with the GPU, the last kernel of the read stores into it
        ("zc_cs16", S.SOAPY_SDR_CS16, np.int16, 2, {"ZEROCOPY": "1"}, None),
        ("zc_cf32", S.SOAPY_SDR_CF32, np.float32, 2, {"ZEROCOPY": "1"}, None),
        ("zc_cs16_iir", S.SOAPY_SDR_CS16, np.int16, 2, {"ZEROCOPY": "1"}, 100e3),
        ("zc_cf32_fir64_rs_3_2", S.SOAPY_SDR_CF32, np.float32, 2, {"FIR": "64:1000000", "RESAMP": "3/2", "ZEROCOPY": "1"}, None)):
    if os.environ.get("BENCH_SOAPY_ONLY") and name not in os.environ["BENCH_SOAPY_ONLY"].split(","):
        continue
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, fmt, args=args)
    if bw:
        sdr.setBandwidth(S.SOAPY_SDR_RX, 0, bw)
    sdr.activateStream(rx)
    buf = np.zeros((2 * MTU, width), dt)
    if args and args.get("ZEROCOPY") == "1":
        sdr.registerStreamBuffer(rx, buf)          # explicit since round 4: nothing is registered behind the client's back
    t_feed = t_read = 0.0
    got = 0
    for rep in range(9 if (args and "ASYNC" in args) else 3):
        t0 = time.perf_counter()
        if args and "ASYNC" in args:
            sdr.feedSmiBytes(b[: 8 * NB])          # stay inside the ring: 8 of its 16 MTUs
        else:
            sdr.feedSmiBytes(b)
        t1 = time.perf_counter()
        n = 0
        if args and "ASYNC" in args:
            # the reader thread drains the FIFO into the ring (10 MTUs, overwrite-oldest): feed and read in lock step
            n = 0
            for k in range(8):
                r = sdr.readStream(rx, [buf], MTU, timeoutUs=2_000_000).ret
                if r <= 0:
                    break
                n += r
        else:
            while True:
                r = sdr.readStream(rx, [buf], MTU).ret
                if r <= 0:
                    break
                n += r
        t2 = time.perf_counter()
        if rep:                       # first repetition warms up
            t_feed += t1 - t0; t_read += t2 - t1; got += n
    res[name] = dict(msps_in=got / t_read / 1e6 if (args and "ASYNC" in args) else 2 * K * MTU / t_read / 1e6,
                     ms_per_mtu_call=t_read / max(got // MTU, 1) * 1e3 if (args and "ASYNC" in args) else t_read / (2 * K) * 1e3,
                     feed_gbps=2 * b.size / t_feed / 1e9, out_elems=got)
    sdr.close()
# TX: host samples in (writeStream, one MTU per call), host SMI bytes out (cl_smi_drain_bytes between the timed regions)
tx = {}
rng = np.random.default_rng(3)
for name, fmt, mk, args in (
        ("tx_cs16", S.SOAPY_SDR_CS16, lambda: rng.integers(-4096, 4096, (MTU, 2)).astype(np.int16), None),
        ("tx_cf32", S.SOAPY_SDR_CF32, lambda: (rng.random((MTU, 2), dtype=np.float32) - 0.5), None),
        ("tx_cf32_fm_rs_2_3", S.SOAPY_SDR_CF32, lambda: (rng.random((MTU, 2), dtype=np.float32) - 0.5) * 0.6, {"MOD": "FM:75000", "RESAMP": "2/3"})):
    if os.environ.get("BENCH_SOAPY_ONLY") and name not in os.environ["BENCH_SOAPY_ONLY"].split(","):
        continue
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    st = sdr.setupStream(S.SOAPY_SDR_TX, fmt, args=args)
    sdr.activateStream(st)
    buf = mk()
    t_write, calls = 0.0, 0
    for rep in range(4):
        t0 = time.perf_counter()
        for k in range(32):
            assert sdr.writeStream(st, [buf], MTU).ret == MTU
        t1 = time.perf_counter()
        sdr.drainSmiBytes()
        if rep:
            t_write += t1 - t0; calls += 32
    tx[name] = dict(msps_in=calls * MTU / t_write / 1e6, ms_per_mtu_call=t_write / calls * 1e3)
    sdr.close()
res.update(tx)
print(json.dumps(res, indent=1))
