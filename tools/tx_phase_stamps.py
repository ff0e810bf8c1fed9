#!/usr/bin/env python3
"""Config 5 chain kernel, where a workgroup's time goes (diagnostic build only).

Build here:  python -c "from cariboulite_amd import _build; _build.build_hip_variant('tx_stamps', ['TXQ_STAMPS=1'], source='clhip_tx.hip')"
Run on the GPU box:  CLHIP_LIB=abl/tx_stamps/libcariboulite_hip.so python tools/tx_phase_stamps.py

Thread 0 of every workgroup stamps s_memrealtime (100 MHz) at: 0 entry, 1 sum pass done (loads landed, workgroup
reduced), 2 sum published, 3 first sub-block's arithmetic done, 4 look-back done, 5 first sub-block emitted,
5 + sb sub-block sb done; 24..27 inside sub-block 3 (messages landed, after the scan's barrier, after the phasors' barrier,
words packed).  Printed: the mean length of each phase over the workgroups of the steady state, and how many
workgroups are alive at once."""
import os, sys, ctypes as C
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
for _ in range(30):
    p5.run(hip.TXPIPE_IN_FM_MESSAGE, msg, 0, n5, by, 4 * (no5 + 4))
torch.cuda.synchronize()
L = hip.lib()
L.clhip_tx_debug_stamps.restype = C.c_int
L.clhip_tx_debug_stamps.argtypes = [C.c_void_p]
L.clhip_tx_debug_nsub.restype = C.c_int
nw = L.clhip_tx_debug_stamps(None)
if not nw:
    sys.exit("this build carries no stamps (TXQ_STAMPS=1)")
st = np.zeros(nw, dtype=np.uint64)
L.clhip_tx_debug_stamps(st.ctypes.data)
NSUB = L.clhip_tx_debug_nsub()
st = st.reshape(-1, 32).astype(np.int64)
SB = 256 * 12 * NSUB
n_super = min(st.shape[0], (n5 + SB - 1) // SB)
inner = st[:n_super, 24:28]
st = st[:n_super, :6 + NSUB - 1]
LAST = 5 + NSUB - 1
t0 = st[:, 0].min()
us = (st - t0) / 100.0
print(f"superblocks {n_super}; kernel span {us.max():.1f} us")
names = ["sum pass (loads + reduce)", "publish", "sub-block 0 arithmetic", "look-back", "sub-block 0 emit"] + [f"sub-block {k}" for k in range(1, NSUB)]
lo, hi = n_super // 4, 3 * n_super // 4
d = np.diff(us[lo:hi], axis=1)
for k, nm in enumerate(names):
    print(f"  {nm:28s} mean {d[:, k].mean():6.2f} us   p10 {np.percentile(d[:, k], 10):6.2f}   p90 {np.percentile(d[:, k], 90):6.2f}")
if inner.any():                                   # inside sub-block 3 (stamp 7 = its start, stamp 8 = its end)
    iu = (inner - t0) / 100.0
    seq = np.concatenate([us[:, 7:8], iu, us[:, 8:9]], axis=1)[lo:hi]     # stamp 7 = end of sub-block 2, 8 = end of sub-block 3
    dd = np.diff(seq, axis=1)
    for k, nm in enumerate(["messages landed", "prefix + wave scan + barrier", "phasors -> LDS + barrier", "window reads + resample + pack",
                            "stores + tail hand-over + barrier"]):
        print(f"    sub-block 3: {nm:34s} mean {dd[:, k].mean():6.2f} us   p10 {np.percentile(dd[:, k], 10):6.2f}   p90 {np.percentile(dd[:, k], 90):6.2f}")
life = us[lo:hi, LAST] - us[lo:hi, 0]
print(f"  {'workgroup lifetime':28s} mean {life.mean():6.2f} us   p10 {np.percentile(life, 10):6.2f}   p90 {np.percentile(life, 90):6.2f}")
# residency: workgroups alive at the middle of the kernel
mid = us.max() / 2
alive = np.sum((us[:, 0] <= mid) & (us[:, LAST] >= mid))
print(f"  alive at mid-kernel: {alive} workgroups ({alive / 256:.2f} per CU)")
# which phase are the alive workgroups in, at 9 instants across the steady state
for frac in (0.3, 0.5, 0.7):
    T = us.max() * frac
    al = (us[:, 0] <= T) & (us[:, LAST] >= T)
    ph = np.array([np.searchsorted(r, T, side="right") - 1 for r in us[al]])
    cnt = np.bincount(ph, minlength=len(names))[:len(names)]
    print(f"  at {frac:.0%} of the kernel: in " + ", ".join(f"{nm.split(' (')[0]}: {c}" for nm, c in zip(names, cnt)))
