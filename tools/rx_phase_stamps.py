#!/usr/bin/env python3
"""Where does a wave of the fused kernel spend its cycles?  Runs bench.py's workload through the
s_memtime-stamped diagnostic build (never the shipped kernel) and prints the per-phase shares."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cariboulite_amd import hip, synth
dev = torch.device("cuda", 0)
taps = np.load(os.path.join(ROOT, "tests", "golden", "taps.npz"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28     # e.g. 4000000: one second of one stream, a one-round launch
words = synth.torch_smi_words(n, dev, 0, 0)
pipe = hip.RxPipe(1, 0, taps["fir64_c2"], taps["rs_3_2"], 3, 2, 0)
out = torch.empty((pipe.out_count(n), 2), dtype=torch.float32, device=dev)
for _ in range(40):
    pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0)
buf = torch.zeros((256 * 16 + 8) * 4 * 8, dtype=torch.int64, device=dev)
pipe.set_diag(buf)
for _ in range(3):
    pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0)
torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(-1, 8)
d = d[d[:, 6] > 0]
names = ["stage(convert+ds_write)", "barrier0", "prefetch+FIR", "second stage(+barrier1)", "store", "barrier2"]
tot = d[:, :6].sum()
print("waves:", d.shape[0], "tiles/wave avg: %.1f" % d[:, 6].mean(), "cycles/tile/wave: %.0f" % (tot / d[:, 6].sum()))
for i, nme in enumerate(names):
    print("%-26s %6.1f %%   %8.0f cycles/tile" % (nme, 100 * d[:, i].sum() / tot, d[:, i].sum() / d[:, 6].sum()))
dt = (d[:, 7].astype(np.uint64) >> np.uint64(32)).astype(np.float64)
dr = (d[:, 7].astype(np.uint64) & np.uint64(0xffffffff)).astype(np.float64)
clk = dt / np.maximum(dr, 1) * 100e6
print("in-kernel shader clock (median over waves): %.3f GHz   wave lifetime median %.1f us (p10 %.1f, p90 %.1f, max %.1f)" % (np.median(clk) / 1e9, np.median(dr) / 100.0, np.percentile(dr, 10) / 100.0, np.percentile(dr, 90) / 100.0, dr.max() / 100.0))
print("stamped phases per tile: %.1f us at that clock" % (tot / d[:, 6].sum() / np.median(clk) * 1e6))
