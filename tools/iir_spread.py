#!/usr/bin/env python3
"""Why do the waves of one single-pass IIR launch (2^26 samples: 4 096 waves, one chunk of 8 tiles each) end between 154 and 204 us?
Per wave (CLHIP_IIR_STAMPS=1): start, end, where it ran (HW_ID / XCC_ID), the ticks it spent polling for its chunk's first-tile
aggregates, its chunk.  Prints the slowest decile against the median: which XCD / SIMD, how late it started, how long it polled, how
its SIMD's other waves did -- so that the spread has a cause with a name.   usage: iir_spread.py [log2 n] [fc]  -> JSON"""
import json, os, sys
os.environ["CLHIP_IIR_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cariboulite_amd import hip, soapy as S
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
fc = float(sys.argv[2]) if len(sys.argv) > 2 else 50e3
iq = torch.randint(-4096, 4096, (n, 2), dtype=torch.int16, device="cuda:0")
f = hip.IIR(S.design_butter_lowpass(6, 4e6, fc))
for _ in range(6):
    f.run(iq, n)
torch.cuda.synchronize()
_, wv = f.debug_stamps()
wv = wv[wv[:, 0] > 0]
t0 = int(wv[:, 0].min())
us = lambda a: (a.astype(np.int64) - t0) * 10e-3                       # s_memrealtime: 100 MHz
start, end = us(wv[:, 0]), us(wv[:, 1])
life = end - start
poll = wv[:, 3].astype(np.int64) * 10e-3
steps = (wv[:, 2] & 0xFFFF).astype(int)
hw = ((wv[:, 2] >> 16) & 0xFFFFFFFF).astype(np.int64)
xcc = ((wv[:, 2] >> 48) & 0xF).astype(int)
chunk = (wv[:, 4] & 0xFFFFFFFF).astype(np.int64)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd                    # one SIMD of the chip
order = np.argsort(end)
n_w = len(end)
dec = order[-(n_w // 10):]                                              # the slowest decile by end time
mid = order[int(0.45 * n_w): int(0.55 * n_w)]
# how did a wave's SIMD do as a whole?  (waves that share its SIMD, itself included)
simd_last, simd_mean, simd_n = {}, {}, {}
for k in np.unique(key):
    m = key == k
    simd_last[int(k)], simd_mean[int(k)], simd_n[int(k)] = float(end[m].max()), float(end[m].mean()), int(m.sum())
sib_mean = np.array([simd_mean[int(k)] for k in key]); sib_n = np.array([simd_n[int(k)] for k in key])
rank_in_simd = np.array([int((end[key == k] < e).sum()) for k, e in zip(key, end)])   # 0 = first of its SIMD to end
q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (10, 50, 90)]
def grp(ix):
    return {"waves": int(len(ix)), "end_us_p10_p50_p90": q(end[ix]), "start_us": q(start[ix]), "lifetime_us": q(life[ix]),
            "polled_us": q(poll[ix]), "polled_over_1us_frac": round(float((poll[ix] > 1.0).mean()), 3),
            "tiles_done": q(steps[ix]), "waves_on_its_simd": q(sib_n[ix]), "mean_end_of_its_simd_us": q(sib_mean[ix]),
            "rank_within_its_simd_0_first": q(rank_in_simd[ix]),
            "share_per_xcc": {int(x): round(float((xcc[ix] == x).mean()), 3) for x in sorted(set(xcc))},
            "chunk_index_p10_p50_p90": q(chunk[ix])}
corr = lambda a, b: round(float(np.corrcoef(a, b)[0, 1]), 3)
res = {"n": n, "fc": fc, "waves": int(n_w), "simds_used": int(len(simd_n)), "end_us_min_p10_p50_p90_max": [round(float(end.min()), 1)] + q(end) + [round(float(end.max()), 1)],
       "slowest_decile": grp(dec), "median_band_45_55": grp(mid),
       "correlation_of_end_time_with": {"start": corr(end, start), "polled": corr(end, poll), "lifetime": corr(end, life),
                                        "mean_end_of_its_simd": corr(end, sib_mean), "waves_on_its_simd": corr(end, sib_n), "chunk_index": corr(end, chunk)},
       "last_end_by_xcc_us": {int(x): round(float(end[xcc == x].max()), 1) for x in sorted(set(xcc))},
       "mean_end_by_xcc_us": {int(x): round(float(end[xcc == x].mean()), 1) for x in sorted(set(xcc))},
       "end_spread_within_simd_us_p10_p50_p90": q(np.array([simd_last[k] - float(end[key == k].min()) for k in simd_last])),
       "between_simd_spread_of_mean_end_us_p10_p50_p90": q(np.array(list(simd_mean.values())))}
print(json.dumps(res, indent=1))
