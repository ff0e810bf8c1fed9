import numpy as np, torch, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cariboulite_amd import hip, synth
from conftest import load_golden
t = load_golden("taps.npz")
DEV = "cuda:0"
n = 131072
b = synth.smi_stream_bytes(n + 64, 0, stream=5)[0]
w = torch.from_numpy(b.view(np.int32).copy()).to(DEV)
def run():
    p = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    o = torch.zeros((n * 3 // 2 + 64, 2), device=DEV)
    g = p.run(hip.PIPE_IN_SMI_WORDS, w, 0, n, o, 0)
    torch.cuda.synchronize()
    return o[:g].cpu().numpy()
ref = run()
for k in range(6):
    # dirty the LDS / registers of every CU with other kernels: sorts, matmuls, another pipe configuration
    x = torch.randn(1 << 22, device=DEV); x.sort(); a = torch.randn(2048, 2048, device=DEV); (a @ a).sum().item()
    p2 = hip.RxPipe(4, 1, t["fir64_c3"], None, 1, 1, hip.PIPE_OUT_FM_DEMOD)
    w4 = torch.stack([w[:65536 + 64]] * 4).contiguous(); o4 = torch.zeros((4, 65536 + 64), device=DEV)
    p2.run(hip.PIPE_IN_SMI_WORDS, w4, w4.shape[1], 65536, o4, o4.shape[1]); torch.cuda.synchronize()
    y = run()
    d = np.flatnonzero((y != ref).any(axis=1))
    print("round", k, "rows differing from the first run:", d.size, d[-4:] if d.size else "", float(np.abs(y - ref).max()))
