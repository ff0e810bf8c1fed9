#!/usr/bin/env python3
"""Fused vs generic RX pipe on config 2's shape, and the generic path on shapes with no fused instantiation."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from scipy import signal
from cariboulite_amd import hip, synth
dev = torch.device("cuda", 0)
taps = np.load(os.path.join(ROOT, "tests", "golden", "taps.npz"))
n = 1 << 26
words = synth.torch_smi_words(n, dev, 0, 0)
def timeit(fn, warm=3, reps=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
res = {}
for name, fir, rs, L, M, force in (("c2_fused", taps["fir64_c2"], taps["rs_3_2"], 3, 2, False),
                                   ("c2_generic", taps["fir64_c2"], taps["rs_3_2"], 3, 2, True),
                                   ("fir96_4_3", signal.firwin(96, 0.4).astype(np.float32), (4 * signal.firwin(32, 0.2)).astype(np.float32), 4, 3, False),
                                   ("fir64_1_4", taps["fir64_c2"], signal.firwin(8, 0.2).astype(np.float32), 1, 4, False)):
    p = hip.RxPipe(1, 0, fir, rs, L, M, hip.PIPE_OUT_IQ)
    if force: p.force_generic(True)
    out = torch.empty((p.out_count(n) + 8, 2), dtype=torch.float32, device=dev)
    t = timeit(lambda: p.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0))
    res[name] = dict(ms=t * 1e3, gsps=n / t / 1e9, fused=bool(p.uses_fused(n, hip.PIPE_IN_SMI_WORDS)) and not force)
print(json.dumps(res, indent=1))
