#!/usr/bin/env python3
"""Do the four waves of a SIMD finish together?  Start / end stamps and the hardware identity (HW_ID, XCC_ID) of every wave
of one single-pass IIR launch (CLHIP_IIR_STAMPS=1), grouped by SIMD.  usage: iir_wave_balance.py [log2 n] [fc]"""
import os, sys, json
os.environ["CLHIP_IIR_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cariboulite_amd import hip, soapy as S
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
fc = float(sys.argv[2]) if len(sys.argv) > 2 else 50e3
iq = torch.randint(-4096, 4096, (n, 2), dtype=torch.int16, device="cuda:0")
f = hip.IIR(S.design_butter_lowpass(6, 4e6, fc))
for _ in range(5):
    f.run(iq, n)
torch.cuda.synchronize()
_, wv = f.debug_stamps()
wv = wv[wv[:, 0] > 0]
t0 = int(wv[:, 0].min())
start = (wv[:, 0].astype(np.int64) - t0) * 10e-3
end = (wv[:, 1].astype(np.int64) - t0) * 10e-3
steps = (wv[:, 2] & 0xFFFF).astype(int)
hw = ((wv[:, 2] >> 16) & 0xFFFFFFFF).astype(np.int64)
xcc = ((wv[:, 2] >> 48) & 0xF).astype(int)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
groups = {}
for k, e, s_, st in zip(key, end, start, steps):
    groups.setdefault(int(k), []).append((e, s_, st))
sizes = np.array([len(g) for g in groups.values()])
spread = np.array([max(x[0] for x in g) - min(x[0] for x in g) for g in groups.values()])
last = np.array([max(x[0] for x in g) for g in groups.values()])
first = np.array([min(x[0] for x in g) for g in groups.values()])
q = lambda a: [round(float(np.percentile(a, p)), 1) for p in (0, 10, 50, 90, 100)]
cuk = key // 4
cul = {}
for k, e in zip(cuk, end):
    cul[int(k)] = max(cul.get(int(k), 0), e)
print(json.dumps({"n": n, "waves": int(len(wv)), "simds": len(groups), "waves_per_simd_min_med_max": [int(sizes.min()), float(np.median(sizes)), int(sizes.max())],
                  "steps_per_wave_min_max": [int(steps.min()), int(steps.max())],
                  "end_us_all_waves": q(end), "first_end_per_simd": q(first), "last_end_per_simd": q(last), "end_spread_within_simd": q(spread),
                  "last_end_per_cu": q(np.array(list(cul.values()))), "xcc_seen": sorted(set(int(x) for x in xcc)),
                  "last_end_by_xcc": {int(x): round(float(end[xcc == x].max()), 1) for x in sorted(set(xcc))}}, indent=1))
