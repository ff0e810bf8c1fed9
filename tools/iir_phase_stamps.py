#!/usr/bin/env python3
"""Where a wave of the single-pass IIR kernel spends a tile: s_memrealtime stamps at the phase boundaries of the first
64 waves' first 16 tiles (CLHIP_IIR_STAMPS=1), averaged.  usage: phase_stamps.py [log2 n] [fc]"""
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
st, wv = f.debug_stamps()
st = st.astype(np.int64); wv = wv.astype(np.int64)
wv = wv[wv[:, 0] > 0]
t0 = wv[:, 0].min()
start, end = (wv[:, 0] - t0) * 10e-3, (wv[:, 1] - t0) * 10e-3
q = lambda a: [round(float(np.percentile(a, p)), 1) for p in (0, 10, 50, 90, 100)]
waves = {"waves": int(len(wv)), "start_us_p0_10_50_90_100": q(start), "end_us_p0_10_50_90_100": q(end),
         "life_us": q(end - start), "tiles_per_wave_min_mean_max": [int(wv[:, 2].min()), round(float(wv[:, 2].mean()), 2), int(wv[:, 2].max())]}
names = ["load wait + commit", "segment FIR", "scan", "publish (+P^m issue)", "wait for predecessors", "fold aggregates",
         "start states", "ticket + prefetch issue", "recursion", "store", "(loop)"]
d = np.diff(st, axis=2)[:, :, :10] * 10e-3          # us per phase (100 MHz)
done = (st[:, :, 0] > 0) & (st[:, :, 10] > 0)
ok = done & (d[:, :, 8] > 0)                        # full steps (a prologue step stops after the scan: no recursion)
pro = done & ~ok
res = {}
for k in range(10):
    v = d[:, :, k][ok]
    res[names[k]] = (round(float(v.mean()), 2), round(float(np.median(v)), 2), round(float(v.max()), 2))
tile = (st[:, :, 10] - st[:, :, 0])[ok] * 10e-3
gap = (st[:, 1:, 0] - st[:, :-1, 10])[done[:, 1:] & done[:, :-1]] * 10e-3
protime = (st[:, :, 10] - st[:, :, 0])[pro] * 10e-3
print(json.dumps({"n": n, "fc": fc, "waves": waves, "tiles_sampled": int(ok.sum()), "prologue_steps_sampled": int(pro.sum()),
                  "prologue_step_us_mean": round(float(protime.mean()), 2) if protime.size else None, "us_per_phase_mean_median_max": res,
                  "tile_us_mean": round(float(tile.mean()), 2), "between_tiles_us_mean": round(float(gap.mean()), 2) if gap.size else None,
                  "env": {k: v for k, v in os.environ.items() if k.startswith("CLHIP_")}}, indent=1))
