import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cariboulite_amd import hip, soapy as S
import test_gpu_group as T
MTU, NB, SENT = T.MTU, T.NB, T.SENT
n = 8
chan = lambda i: "S1G" if i % 2 else "HiF"
se = lambda what: print(what, "-> sticky error", hip.lib().clhip_debug_sticky_error())
gdevs, gsts = T.make_devices(S, n, S.SOAPY_SDR_CS16, None, chan)
sdevs, ssts = T.make_devices(S, n, S.SOAPY_SDR_CS16, None, chan)
grp = S.Group(gdevs, {"SUBBATCH": "2", "SLAB_MB": "4"}); se("group make")
rows = 2 * MTU + 8
gb, sb = T.sentinel_buffers(n, (rows, 2), np.int16), T.sentinel_buffers(n, (rows, 2), np.int16)
print("gb", [hex(x.ctypes.data) for x in gb]); print("sb", [hex(x.ctypes.data) for x in sb])
for i in range(n):
    for c in range(3):
        b = T.batch_bytes(i, c, 0 if chan(i) == "S1G" else 1)
        gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)
grp.registerBuffers(gb); se("registerBuffers")
num = MTU + 4096
print(grp.readStream(gb, num)); se("group call above one MTU, registered")
for i in range(n):
    r = sdevs[i].readStream(ssts[i], [sb[i]], num).ret
    print("lone", i, r, hip.last_error() if r != num else ""); se("  lone read")
grp.close()
