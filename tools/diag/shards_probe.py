"""Probe: N Soapy RX devices of one GPU read through ONE group, or through k groups of N / k driven by k threads at once.
python tools/diag/shards_probe.py [cs16|c2] [streams] [calls]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", str(1 << 40))
from cariboulite_amd import soapy as S, synth

MTU = 131072; NB = 4 * MTU
case = sys.argv[1] if len(sys.argv) > 1 else "cs16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K = int(sys.argv[3]) if len(sys.argv) > 3 else 24
fmt, args, out_el, dt, w = ("CS16", None, 1.0, np.int16, 2) if case == "cs16" else ("CF32", {"FIR": "64:1000000", "RESAMP": "3/2"}, 1.5, np.float32, 2)
words = [synth.smi_stream_bytes(K * MTU, i % 2, stream=i)[0] for i in range(4)]
for shards in (1, 2, 4, 1, 2, 4):
    devs = []
    for i in range(n):
        d = S.Device(dict(driver="Cariboulite", channel="S1G" if i % 2 == 0 else "HiF"))
        d.activateStream(d.setupStream(S.SOAPY_SDR_RX, fmt, args=args))
        devs.append(d)
    per = n // shards
    groups = [S.Group(devs[s * per:(s + 1) * per], {"SLAB_MB": str((K * NB >> 20) + 1)}) for s in range(shards)]
    bufs = [np.zeros((int(MTU * out_el) + 8, w), dt) for _ in range(n)]
    best = None
    for rep in range(3):
        for i, d in enumerate(devs):
            d.feedSmiBytes(words[i % 4])
        def run(s):
            for k in range(K):
                nd, rets = groups[s].readStream(bufs[s * per:(s + 1) * per], MTU)
                assert nd == per, (nd, groups[s].lastError())
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(s,)) for s in range(1, shards)]
        for t in th: t.start()
        run(0)
        for t in th: t.join()
        dt_s = time.perf_counter() - t0
        if rep and (best is None or dt_s < best): best = dt_s
    print(f"{case} streams {n} shards {shards}: {n * K * MTU / best / 1e6:.1f} Msamples/s", flush=True)
    for g in groups: g.close()
    for d in devs: d.close()
