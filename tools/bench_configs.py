#!/usr/bin/env python3
"""Secondary configs of BASELINE.json (parity-test cases, not bench lines) timed on one GPU
for the DESIGN.md table: C1 unpack->CF32, C3 2 ch FIR64+FM demod, C4 per-GPU share
(32 streams x 2^24, FIR128 + 5/4), C5 TX pipe, and the IIR."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cariboulite_amd import hip, synth

dev = torch.device("cuda", 0)
taps = np.load(os.path.join(ROOT, "tests", "golden", "taps.npz"))
L = hip.lib()


def timeit(fn, warm=15, reps=40):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


res = {}
n = 1 << 28
words = synth.torch_smi_words(n, dev, 0, 0)
# C1: sync check + unpack -> CF32 (12 B/sample)
nch = n // 131072
offs = torch.zeros(nch, dtype=torch.int32, device=dev)
out = torch.empty((n, 2), dtype=torch.float32, device=dev)
def c1():
    hip.smi_find_offsets(words, 4 * n, 524288, 524288, nch, offs)
    hip.smi_unpack(0, words, 4 * n, 524288, 524288, nch, offs, hip.FORMAT_CF32, out)
t = timeit(c1)
res["c1_unpack_cf32"] = dict(ms=t * 1e3, gsps=n / t / 1e9, hbm_frac=12 * n / t / 8e12)
meta = torch.empty(n, dtype=torch.uint8, device=dev)
iq = out.view(torch.int16).reshape(-1)[: 2 * n].view(n, 2)
def c1b():
    hip.smi_unpack(0, words, 4 * n, 524288, 524288, nch, offs, hip.FORMAT_CS16, iq, meta)
t = timeit(c1b)
res["unpack_cs16_meta"] = dict(ms=t * 1e3, gsps=n / t / 1e9, hbm_frac=9 * n / t / 8e12)
# IIR on CS16 (in place, fp64)
f = hip.IIR(np.array([[1e-4, 2e-4, 1e-4, -1.9, 0.9025]] * 3))
n_i = 1 << 26
def iir():
    f.run(iq, n_i)
t = timeit(iir, 3, 10)
res["iir_cs16"] = dict(ms=t * 1e3, gsps=n_i / t / 1e9)
del out, iq, meta
# C3: two channels (S1G + HiF), FIR64 + FM demod, float out (8 B/sample)
n3 = 1 << 27
w2 = torch.stack([synth.torch_smi_words(n3, dev, 0, 1), synth.torch_smi_words(n3, dev, 1, 2)])
p3 = [hip.RxPipe(1, ch, taps["fir64_c3"], None, 1, 1, hip.PIPE_OUT_FM_DEMOD) for ch in (0, 1)]
o3 = torch.empty((2, n3), dtype=torch.float32, device=dev)
def c3():
    for ch in (0, 1):
        p3[ch].run(hip.PIPE_IN_SMI_WORDS, w2[ch], 0, n3, o3[ch], 0)
t = timeit(c3)
res["c3_2ch_fir64_fm"] = dict(ms=t * 1e3, gsps=2 * n3 / t / 1e9, hbm_frac=8 * 2 * n3 / t / 8e12)
del w2, o3
# C4 share of one GPU: 32 streams x 2^24, FIR128 + 5/4 (14 B/sample)
ns, n4 = 32, 1 << 24
del words
w4 = torch.stack([synth.torch_smi_words(n4, dev, 0, 100 + k) for k in range(ns)])
p4 = hip.RxPipe(ns, 0, taps["fir128_c4"], taps["rs_5_4"], 5, 4, hip.PIPE_OUT_IQ)
no4 = p4.out_count(n4)
o4 = torch.empty((ns, no4, 2), dtype=torch.float32, device=dev)
def c4():
    p4.run(hip.PIPE_IN_SMI_WORDS, w4, n4, n4, o4, no4)
t = timeit(c4, 8, 20)
res["c4_32streams_fir128_5_4"] = dict(ms=t * 1e3, gsps=ns * n4 / t / 1e9, hbm_frac=14 * ns * n4 / t / 8e12,
                                      valu_frac=552 * ns * n4 / t / 157.3e12)
del o4
# C5: TX pipe, 2^27 messages -> FM -> 2/3 -> pack
n5 = 1 << 27
msg = torch.randn(n5, device=dev) * 0.3
p5 = hip.TxPipe(1, 75e3, 4e6, taps["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
no5 = p5.out_count(n5)
by = torch.empty(4 * (no5 + 4), dtype=torch.uint8, device=dev)
def c5():
    p5.run(hip.TXPIPE_IN_FM_MESSAGE, msg, 0, n5, by, 4 * (no5 + 4))
t = timeit(c5, 3, 10)
res["c5_tx_fm_2_3_pack"] = dict(ms=t * 1e3, gsps=n5 / t / 1e9, hbm_frac=(4 + 8 / 3) * n5 / t / 8e12)
print(json.dumps(res, indent=1))
