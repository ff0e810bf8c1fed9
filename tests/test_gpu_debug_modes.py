"""-m gpu: link-integrity (debug) modes through the C-ABI vs the reference fixtures."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import torch
    from cariboulite_amd import hip, soapy
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return soapy


def _names():
    return [str(n) for n in load_golden("smi_debug_cases.npz")["names"]]


@pytest.mark.parametrize("name", _names())
def test_debug_read_vs_reference_fixture(S, name):
    g = load_golden("smi_debug_cases.npz")
    mode, n_calls, length_samples, nb = [int(v) for v in g[f"{name}__args"]]
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    sdr.setSmiDebugMode(mode)
    sdr.setMaxRead(nb)                      # the fixture's native batch
    sdr.feedSmiBytes(g[f"{name}__bytes"])
    for k in range(n_calls):
        ret, _, _ = sdr.smiRead(0, length_samples)
        assert ret == int(g[f"{name}__rets"][k])            # -2 debug mode, -3 no sync
        d = sdr.smiDebugData()
        assert list(d[:3]) == g[f"{name}__states"][k].tolist()
        assert d[3] == g[f"{name}__rates"][k]                # same fp64 EMA arithmetic
    sdr.close()


def test_debug_kernel_large_random_vs_oracle(S, orc):
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(4)
    n = 524288
    lf = orc.lfsr_stream(n, seed=0x31)
    bad = lf.copy()
    idx = rng.choice(n, 500, replace=False)
    bad[idx] ^= rng.integers(1, 256, 500).astype(np.uint8)
    for mode, buf in ((1, bad), (2, np.where(rng.random(n // 4) < 0.01, rng.integers(0, 2 ** 32, n // 4, dtype=np.uint64),
                                              0xABCDEF01).astype(np.uint32).view(np.uint8))):
        st = orc.DebugState()
        offs = st.analyze(mode, buf)
        d = torch.from_numpy(buf.copy()).to("cuda:0")
        res = torch.zeros(4, dtype=torch.int32, device="cuda:0")
        assert hip.lib().clhip_smi_debug_analyze(mode, d.data_ptr(), buf.size, 0, res.data_ptr(), hip.current_stream()) == 0
        torch.cuda.synchronize()
        r = res.cpu().numpy()
        assert r[0] == offs and r[1] == st.tuple()[1]
        if mode == 1:
            assert r[3] == st.tuple()[2]
    # through Soapy the debug return is squashed to 0 samples (CaribouliteStream.cpp:266-276)
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    sdr.setSmiDebugMode(1)
    sdr.feedSmiBytes(lf)
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    buf = np.zeros((1000, 2), np.int16)
    assert sdr.readStream(rx, [buf], 1000).ret == 0
    sdr.close()
