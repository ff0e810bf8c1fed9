"""diagnostic: replay tests/test_gpu_group.py::test_random_walk_against_lone_devices for one seed and say, at the first difference, which
side (the group or the lone device) left the truth: a fresh single-stream pipe over the bytes that stream was fed, call by call"""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cariboulite_amd import hip, soapy as S, synth
from conftest import load_golden
import test_gpu_group as T
seed, staged = int(sys.argv[1]), sys.argv[2] == "1"
MTU, NB = T.MTU, T.NB
rng = np.random.default_rng(seed)
n = 8
args = {"FIR": "64:1000000", "RESAMP": "3/2"} if staged else None
full = MTU * 3 // 2 if staged else MTU
chan = lambda i: "S1G" if i % 2 else "HiF"
gdevs, gsts = T.make_devices(S, n, S.SOAPY_SDR_CF32, args, chan)
sdevs, ssts = T.make_devices(S, n, S.SOAPY_SDR_CF32, args, chan)
grp = S.Group(gdevs, {"SUBBATCH": "2", "SLAB_MB": "4"})
gb, sb = T.sentinel_buffers(n, (full + 8, 2), np.float32), T.sentinel_buffers(n, (full + 8, 2), np.float32)
fed = [0] * n
stream_bytes = [[] for _ in range(n)]
consumed = [0] * n            # samples each stream has delivered so far (valid while nothing was damaged / flushed for it)
hist = [[] for _ in range(n)]
def feed(i, how):
    ch = 0 if chan(i) == "S1G" else 1
    b = T.batch_bytes(i, fed[i], ch); fed[i] += 1
    hist[i].append(str(how))
    if how == "slip": b = T.slipped(b, int(rng.integers(1, 9)))
    elif how == "lost": b[:] = 0
    elif how == "half": b = b[: NB // 2]
    stream_bytes[i].append(b.copy())
    gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)
registered = False
for step in range(36):
    for i in range(n):
        while gdevs[i].pendingSmiBytes() < int(rng.integers(1, 4)) * NB:
            feed(i, rng.choice(["good"] * 12 + ["slip", "lost", "half"]))
    op = rng.choice(["call"] * 5 + ["half", "lone", "flush", "filter", "register"])
    for x in gb + sb: x[...] = np.nan
    if op == "lone" and not staged:
        i = int(rng.integers(0, n))
        gdevs[i].readStream(gsts[i], [gb[i]], MTU); sdevs[i].readStream(ssts[i], [sb[i]], MTU)
    elif op == "flush":
        i = int(rng.integers(0, n)); gdevs[i].flushSmiFifo(); sdevs[i].flushSmiFifo(); hist[i].append("flush")
    elif op == "filter":
        i = int(rng.integers(0, n)); bw = float(rng.choice([100e3, 1e6]))
        gdevs[i].setBandwidth(S.SOAPY_SDR_RX, 0, bw); sdevs[i].setBandwidth(S.SOAPY_SDR_RX, 0, bw); hist[i].append(f"bw{bw:g}")
    elif op == "register":
        grp.unregisterBuffers() if registered else grp.registerBuffers(gb)
        registered = not registered
    num = MTU // 2 if op == "half" else MTU
    for x in gb + sb: x[...] = np.nan
    st0 = grp.stats()
    pend = [d.pendingSmiBytes() for d in gdevs]
    _, rets = grp.readStream(gb, num)
    srets = [sdevs[i].readStream(ssts[i], [sb[i]], num).ret for i in range(n)]
    st1 = grp.stats()
    for i in range(n): hist[i].append(f"<{op}:{num}:{rets[i]}>")
    print(f"step {step} {op} num {num} ahead+{st1['ahead_reads'] - st0['ahead_reads']} single+{st1['single_reads'] - st0['single_reads']} rets {rets}")
    for i in range(n):
        if not T.same(gb[i], sb[i]):
            a, b = gb[i].reshape(-1), sb[i].reshape(-1)
            d = np.flatnonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))
            print(f"DIFF stream {i}: {d.size} values, first {d[0]} last {d[-1]}, max abs {np.nanmax(np.abs(a - b))}, pending before the call {pend[i]}")
            print("history", hist[i])
            print("group seam", gdevs[i].smiStats(), "\nlone seam ", sdevs[i].smiStats())
            print("pending now", gdevs[i].pendingSmiBytes(), sdevs[i].pendingSmiBytes(), "bytes fed", sum(x.size for x in stream_bytes[i]))
            rows = np.flatnonzero((gb[i] != sb[i]).any(axis=1) & ~np.isnan(gb[i]).any(axis=1))
            print("rows differing", rows.size, rows[:4], rows[-4:], "group", gb[i][rows[0]], "lone", sb[i][rows[0]])
            # the truth: a fresh pipe over every byte this stream was fed (valid if none of it was damaged), in this stream's call partition
            if staged and all(h in ("good", "half") or h.startswith("<") for h in hist[i]):
                t = load_golden("taps.npz")
                allb = np.concatenate(stream_bytes[i])
                w = torch.from_numpy(np.concatenate([allb, np.zeros(256, np.uint8)]).view(np.int32).copy()).to("cuda:0")
                pipe = hip.RxPipe(1, 0 if chan(i) == "S1G" else 1, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
                pos = 0
                out = torch.zeros((full + 32, 2), device="cuda:0")
                for h in hist[i]:
                    if h.startswith("<"):
                        cnt = int(h[1:-1].split(":")[2]) * 2 // 3
                        if cnt:
                            out.zero_()
                            g_ = pipe.run(hip.PIPE_IN_SMI_WORDS, w[pos:], 0, cnt, out, 0); pos += cnt
                torch.cuda.synchronize()
                truth = out[:g_].cpu().numpy()
                print("group == truth:", np.array_equal(gb[i][:g_], truth), " lone == truth:", np.array_equal(sb[i][:g_], truth),
                      " max |group - truth|", float(np.abs(gb[i][:g_] - truth).max()), " max |lone - truth|", float(np.abs(sb[i][:g_] - truth).max()))
            sys.exit(1)
print("no difference")
