#!/usr/bin/env python3
"""Config 5 (TX pipe) alone: 2^27 messages -> FM -> 2/3 -> quantise -> pack, timed with HIP events."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cariboulite_amd import hip
dev = torch.device("cuda", 0)
taps = np.load(os.path.join(ROOT, "tests", "golden", "taps.npz"))
n5 = 1 << int(os.environ.get("BENCH_TX_LOG2", "27"))
msg = torch.randn(n5, device=dev) * 0.3
p5 = hip.TxPipe(1, 75e3, 4e6, taps["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
no5 = p5.out_count(n5)
by = torch.empty(4 * (no5 + 4), dtype=torch.uint8, device=dev)
def c5():
    p5.reset_counts() if hasattr(p5, "reset_counts") else None
    p5.run(hip.TXPIPE_IN_FM_MESSAGE, msg, 0, n5, by, 4 * (no5 + 4))
for _ in range(int(os.environ.get("BENCH_TX_WARM", "5"))): c5()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K = int(os.environ.get("BENCH_TX_STEPS", "20"))
for _ in range(K): c5()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / K * 1e-3
print(json.dumps(dict(ms=t * 1e3, gsps=n5 / t / 1e9, hbm_frac=(4 + 8 / 3) * n5 / t / 8e12)))
