import numpy as np, torch, sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from cariboulite_amd import hip, synth
from conftest import load_golden
t = load_golden("taps.npz")
n = 131072
DEV = "cuda:0"
w = torch.from_numpy(np.concatenate([synth.smi_stream_bytes(n, 0, stream=5)[0], np.zeros(256, np.uint8)]).view(np.int32).copy()).to(DEV)
def cs16_of(wr, count):
    o = torch.zeros((count + 2, 2), dtype=torch.int16, device=DEV)
    offs = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.smi_unpack(0, wr, 4 * count, 4 * count, 4 * count, 1, offs, hip.FORMAT_CS16, o)
    return o
for pre in (0, 65536, 131072):
    pa = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    pb = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    so = n * 3 // 2 + 32
    oa = torch.zeros((so, 2), device=DEV); ob = torch.zeros((so, 2), device=DEV)
    if pre:
        pa.run(hip.PIPE_IN_SMI_WORDS, w, 0, pre, oa, 0); pb.run(hip.PIPE_IN_SMI_WORDS, w, 0, pre, ob, 0)
    ga = pa.run(hip.PIPE_IN_SMI_WORDS, w, 0, n, oa, 0)
    gb = pb.run(hip.PIPE_IN_CS16, cs16_of(w, n), 0, n, ob, 0)
    torch.cuda.synchronize()
    d = (oa[:ga] != ob[:gb]).any(dim=1).nonzero().flatten()
    print("pre", pre, "outs", ga, gb, "rows differing", d.numel(), "first", int(d[0]) if d.numel() else None, "last", int(d[-1]) if d.numel() else None,
          "max abs", float((oa[:ga] - ob[:gb]).abs().max()))
# one run of n against two runs of n/2, and against runs of n/4 + 3n/4
for split in ((n // 2, n // 2), (n // 4, 3 * n // 4), (n - 4096, 4096)):
    pa = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    pb = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    so = n * 3 // 2 + 32
    oa = torch.zeros((so, 2), device=DEV); ob = torch.zeros((so, 2), device=DEV)
    ga = pa.run(hip.PIPE_IN_SMI_WORDS, w, 0, n, oa, 0)
    g1 = pb.run(hip.PIPE_IN_SMI_WORDS, w, 0, split[0], ob, 0)
    g2 = pb.run(hip.PIPE_IN_SMI_WORDS, w[split[0]:], 0, split[1], ob[g1:], 0)
    torch.cuda.synchronize()
    d = (oa[:ga] != ob[:g1 + g2]).any(dim=1).nonzero().flatten()
    print("split", split, "outs", ga, g1, g2, "rows differing", d.numel(), "first", int(d[0]) if d.numel() else None, "last", int(d[-1]) if d.numel() else None,
          "max abs", float((oa[:ga] - ob[:g1 + g2]).abs().max()), "fused?", pb.uses_fused(split[1], hip.PIPE_IN_SMI_WORDS) if hasattr(pb, "uses_fused") else "?")
