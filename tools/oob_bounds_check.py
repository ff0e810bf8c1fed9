#!/usr/bin/env python3
"""Diagnostic: the fused RX kernel built with -DCLHIP_RX_BOUNDS=1 (abl/rx_bounds, _build.build_hip_variant) compares every
global access with the extent of the call's buffers, counts the ones outside and does not perform them.  Replays the
sync-recovery call sequences, the BASELINE-size configuration and a handful of ragged sizes; prints the counters.
usage: CLHIP_LIB=abl/rx_bounds/libcariboulite_hip.so python tools/oob_bounds_check.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cariboulite_amd import hip, synth
import test_gpu_sync_recovery as T
L = hip.lib()
assert hasattr(L, "clhip_rx_debug_bounds"), "load the diagnostic build: CLHIP_LIB=abl/rx_bounds/libcariboulite_hip.so"
dev = torch.device("cuda:0")
t = np.load(os.path.join(ROOT, "tests", "golden", "taps.npz"))
total = 0


def report(tag):
    global total
    torch.cuda.synchronize()
    h = (C.c_ulonglong * 16)()
    L.clhip_rx_debug_bounds(h)
    v = list(h)
    if v[0]:
        total += v[0]
        print(f"  !! {tag}: {v[0]} accesses outside the buffers; first: site {v[1]} address {v[2]:#x} bounds [{v[3]:#x}, {v[4]:#x}) "
              f"(= end {'+' if v[2] >= v[4] else '-'}{abs(v[2] - v[4]) if v[2] >= v[4] else v[3] - v[2]}); per site {v[8:16]}")
    L.clhip_rx_debug_bounds_reset()


L.clhip_rx_debug_bounds_reset()
for channel in (0, 1):
    for chunk_len, n_chunks in ((4 * 8192, 5), (524288, 3)):
        calls = T.build_calls(channel, chunk_len, n_chunks, (1, 3, 6), seed=40 + channel)
        pipe = hip.RxPipe(1, channel, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
        for ci, (b, kind) in enumerate(calls):
            n = b.size // 4
            nch = -(-b.size // chunk_len)
            d = torch.from_numpy(b.copy()).to(dev)                     # exactly b.size bytes: no slack
            offs = torch.full((nch,), 77, dtype=torch.int32, device=dev)
            cs16 = torch.full((1, n + 2, 2), -21846, dtype=torch.int16, device=dev)
            no = pipe.out_count(n)
            out = torch.full((1, no, 2), float("nan"), dtype=torch.float32, device=dev)
            h_offs = np.full(nch, 99, dtype=np.int32)
            rc = pipe.run_smi(d, 0, b.size, chunk_len, offs, cs16, out, no, h_offs=h_offs)
            report(f"sync-recovery ch {channel} chunk {chunk_len} call {ci} {kind} (rc {rc})")
print("sync-recovery sequences done")
for cfg, (fir, rs, Lr, M, mode) in {"c2": ("fir64_c2", "rs_3_2", 3, 2, 0), "c3": ("fir64_c3", None, 1, 1, 1), "c4": ("fir128_c4", "rs_5_4", 5, 4, 0)}.items():
    for n in (4064 * 3 + 100, 131072, 393216, 1 << 22, (1 << 22) + 4, 1000, 4064, 4063, 8128):
        if cfg == "c4": n = n // 4 * 4
        if cfg == "c2": n = n // 2 * 2
        pipe = hip.RxPipe(1, 0, t[fir], t[rs] if rs else None, Lr, M, mode)
        words = synth.torch_smi_words(n, dev, 0, 3)
        no = pipe.out_count(n)
        out = torch.empty((no, 1 if mode else 2), dtype=torch.float32, device=dev)
        for kind, src in ((hip.PIPE_IN_SMI_WORDS, words),):
            assert pipe.run(kind, src, 0, n, out, 0) == no
            report(f"{cfg} n {n}")
        assert pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0) == no      # a second call: carried history
        report(f"{cfg} n {n} (second call)")
print("violations in total:", total)
