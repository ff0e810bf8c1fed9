import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from cariboulite_amd import hip, synth
dev = torch.device("cuda", 0)
taps = np.load("tests/golden/taps.npz")
n = 1 << 28
words = synth.torch_smi_words(n, dev, 0, 0)
pipe = hip.RxPipe(1, 0, taps["fir64_c2"], taps["rs_3_2"], 3, 2, 0)
out = torch.empty((pipe.out_count(n), 2), dtype=torch.float32, device=dev)
def run(k):
    evs = []
    for _ in range(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    return [round(a.elapsed_time(b), 3) for a, b in evs]
print("cold 60:", run(60)[::4])
print("after sync, 12:", run(12))
time.sleep(0.05)
print("after 50ms idle, 12:", run(12))
time.sleep(1.0)
print("after 1s idle, 12:", run(12))
# a filler kernel stream keeping the GPU busy across the sync? (not allowed in bench; just to learn)
