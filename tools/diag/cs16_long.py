import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cariboulite_amd import hip, soapy as S, synth
import test_gpu_group as T
MTU, NB = T.MTU, T.NB
devs, sts = T.make_devices(S, 4, S.SOAPY_SDR_CS16, None, lambda i: "S1G" if i % 2 else "HiF")
bufs = [np.zeros((2 * MTU + 8, 2), np.int16) for _ in range(4)]
for rep in range(3):
    for i, d in enumerate(devs):
        for c in range(3):
            d.feedSmiBytes(T.batch_bytes(i, 3 * rep + c, 0 if i % 2 else 1))
    for i, d in enumerate(devs):
        for num in (2 * MTU, MTU):
            r = d.readStream(sts[i], [bufs[i]], num).ret
            print("rep", rep, "dev", i, "num", num, "ret", r, "pending", d.pendingSmiBytes(), "| dev err:", d.lastError() if hasattr(d, "lastError") else "", "| hip err:", hip.last_error() if r == 0 else "")
