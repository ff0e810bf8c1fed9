#!/usr/bin/env python3
"""IIR (Butterworth-6 on CS16, fp64) alone: [log2 n] samples in place (default 2^26), HIP-event timed.
CLHIP_LIB=<old build> still works: a library with the round-2 entry points is driven through them."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from scipy import signal
dev = torch.device("cuda", 0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
fc = float(sys.argv[3]) if len(sys.argv) > 3 else 50e3
_s = signal.butter(6, fc, "low", fs=4e6, output="sos")          # (the host helper's design to rounding; this tool only times)
sos = np.ascontiguousarray(np.concatenate([_s[:, :3], _s[:, 4:]], 1))
iq = torch.randint(-4096, 4096, (n, 2), dtype=torch.int16, device=dev)
lib_path = os.environ.get("CLHIP_LIB")
legacy = False
if lib_path:
    L = C.CDLL(lib_path)
    legacy = hasattr(L, "clhip_iir_cs16_batch")
if legacy:
    L.clhip_iir_workspace_bytes.restype = C.c_size_t
    L.clhip_iir_workspace_bytes.argtypes = [C.c_size_t, C.c_int]
    L.clhip_iir_cs16_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    ws = torch.empty(L.clhip_iir_workspace_bytes(n, 3), dtype=torch.uint8, device=dev)
    st = torch.zeros(16, dtype=torch.float64, device=dev)
    sos = np.ascontiguousarray(sos)
    def run():
        L.clhip_iir_cs16_batch(sos.ctypes.data, 3, st.data_ptr(), iq.data_ptr(), n, n, 1, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
else:
    from cariboulite_amd import hip
    f = hip.IIR(sos)
    def run():
        f.run(iq, n)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e-3
ok = True if legacy else (f.status() == 0 and not f.on_scan_path())
print(json.dumps(dict(n=n, fc=fc, ms=round(t * 1e3, 4), gsps=round(n / t / 1e9, 2), hbm_frac=round(8 * n / t / 8e12, 4), legacy=legacy, ok=ok,
                      env={k: v for k, v in os.environ.items() if k.startswith("CLHIP_")})))
