import numpy as np, torch, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cariboulite_amd import hip, synth
from conftest import load_golden
t = load_golden("taps.npz")
DEV = "cuda:0"
for n in (131072, 65536, 1000, 1001, 131069):
    b = synth.smi_stream_bytes(n + 64, 0, stream=5)[0]
    wa = torch.from_numpy(b.view(np.int32).copy()).to(DEV)                 # valid words beyond n
    wb = wa.clone(); wb[n:] = 0                                           # zeros beyond n
    outs = []
    for w in (wa, wb):
        p = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
        o = torch.zeros((n * 3 // 2 + 64, 2), device=DEV)
        g = p.run(hip.PIPE_IN_SMI_WORDS, w, 0, n, o, 0)
        torch.cuda.synchronize()
        outs.append(o[:g].cpu().numpy())
    d = np.flatnonzero((outs[0] != outs[1]).any(axis=1))
    print("n", n, "outs", outs[0].shape[0], "rows that depend on what lies BEHIND the input:", d.size, d[:5], d[-3:] if d.size else "", float(np.abs(outs[0] - outs[1]).max()))
