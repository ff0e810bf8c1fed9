#!/usr/bin/env python3
"""A/B of cl_group kwargs that run-to-run noise (+-10 % between processes on this pool) would bury: every configuration lives in
ONE process, the configurations take turns rep after rep, the MEDIAN over the reps is reported.
usage: group_ab.py <case> <reps> <calls> KEY=a,b,c [KEY2=...]   e.g.  group_ab.py cs16 9 12 INGEST_STREAMS=1,2,4"""
import itertools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cariboulite_amd import soapy as S, synth
MTU = 131072
case, reps, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[4:]]
fmt, dt, width, args = {"cs16": ("CS16", np.int16, 2, None), "cf32": ("CF32", np.float32, 2, None),
                        "c2": ("CF32", np.float32, 2, {"FIR": "64:1000000", "RESAMP": "3/2"}),
                        "fm": ("CF32", np.float32, 1, {"FIR": "64:100000", "DEMOD": "FM"}),
                        "cs16_iir": ("CS16", np.int16, 2, None), "cf32_iir": ("CF32", np.float32, 2, None)}[case]      # (_iir: the reference's 100 kHz low-pass on every stream)
n = 32
words = [synth.smi_stream_bytes(K * MTU, i % 2, stream=i)[0] for i in range(4)]
cfgs = []
for combo in itertools.product(*[v for _, v in axes]):
    kw = {k: v for (k, _), v in zip(axes, combo)}
    devs = []
    for i in range(n):
        d = S.Device(dict(driver="Cariboulite", channel="S1G" if i % 2 == 0 else "HiF"))
        d.activateStream(d.setupStream(S.SOAPY_SDR_RX, fmt, args=args))
        if case.endswith("_iir"):
            d.setBandwidth(S.SOAPY_SDR_RX, 0, 100e3)
        devs.append(d)
    cfgs.append((kw, devs, S.Group(devs, kw), [np.zeros((MTU * 3 // 2 + 8, width) if width > 1 else (MTU * 3 // 2 + 8,), dt) for _ in range(n)], []))
for rep in range(reps + 1):
    for kw, devs, grp, bufs, times in cfgs:
        for i, d in enumerate(devs):
            d.feedSmiBytes(words[i % 4])
        t0 = time.perf_counter()
        for k in range(K):
            nd, _ = grp.readStream(bufs, MTU)
            assert nd == n
        if rep:
            times.append((time.perf_counter() - t0) / K)
out = []
for kw, devs, grp, bufs, times in cfgs:
    t = np.array(times)
    out.append({"kwargs": kw, "msps_median": round(n * MTU / float(np.median(t)) / 1e6, 1), "msps_min_max": [round(n * MTU / float(t.max()) / 1e6, 1), round(n * MTU / float(t.min()) / 1e6, 1)],
                "ms_per_call_median": round(float(np.median(t)) * 1e3, 4)})
    grp.close()
    for d in devs:
        d.close()
print(json.dumps({"case": case, "reps": reps, "calls_per_rep": K, "results": out}, indent=1))
