#!/usr/bin/env python3
"""One 2^30-sample stream (4 GiB in, 12 GiB out) through the fused pipe: outputs far beyond 2^31 bytes are checked
against a second pipe that is placed there by seek + halo (64-bit indexing, tile queue over 262 144 tiles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cariboulite_amd import hip, shard, synth
dev = torch.device("cuda", 0)
t = np.load(os.path.join(ROOT, "tests", "golden", "taps.npz"))
n = 1 << 30
chunk = 1 << 26
words = torch.empty(n, dtype=torch.int32, device=dev)
for k in range(n // chunk):                      # generate in pieces: the generator keeps several temporaries
    words[k * chunk:(k + 1) * chunk] = synth.torch_smi_words(chunk, dev, 0, 50 + k).view(torch.int32)
pipe = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
n_out = pipe.out_count(n)
out = torch.empty((n_out, 2), dtype=torch.float32, device=dev)
assert pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0) == n_out
torch.cuda.synchronize()
scratch = torch.empty((1024, 2), dtype=torch.float32, device=dev)
for start in (0, (1 << 29) + 2, (1 << 30) - (1 << 20)):
    ln = 1 << 20
    p2 = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    o2 = torch.zeros((ln * 3 // 2 + 8, 2), dtype=torch.float32, device=dev)
    k = shard.run_time_slice(p2, hip.PIPE_IN_SMI_WORDS, words, start, start + ln, o2, None, scratch)
    a = start * 3 // 2
    assert torch.equal(out[a:a + k], o2[:k]), start
    assert float(out[a:a + k].abs().max()) > 0.1
print("big index check ok: 2^30 samples, slices at 0, 2^29+2, 2^30-2^20 identical")
