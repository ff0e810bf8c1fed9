#!/usr/bin/env python3
"""IIR (Butterworth-6 on CS16, fp64) alone: 2^26 samples in place, per-kernel times via rocprofv3 if wrapped."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cariboulite_amd import hip
dev = torch.device("cuda", 0)
n = 1 << 26
iq = torch.randint(-4096, 4096, (n, 2), dtype=torch.int16, device=dev)
f = hip.IIR(np.array([[1e-4, 2e-4, 1e-4, -1.9, 0.9025]] * 3))
for _ in range(3): f.run(iq, n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f.run(iq, n)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print(json.dumps(dict(ms=t * 1e3, gsps=n / t / 1e9, hbm_frac=8 * n / t / 8e12)))
