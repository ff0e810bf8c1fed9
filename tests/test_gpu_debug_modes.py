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


def test_debug_bitrate_ema_on_a_replayed_clock(S, orc):
    """caribou_smi.c:210: one smi_calculate_performance per analysed chunk with the chunk's analysed length.  With the clock
    readings replayed (cl_smi_set_debug_clock) the running Mbit/s figure equals the restatement that the compiled reference
    pins (tests/test_oracle_golden.py::test_bitrate_ema_vs_the_reference_itself), chunk after chunk -- first call from {0, 0}."""
    g = load_golden("smi_debug_cases.npz")
    name = _names()[0]
    mode, n_calls, length_samples, nb = [int(v) for v in g[f"{name}__args"]]
    stream = g[f"{name}__bytes"]
    clock = [(1_760_000_000 + k // 3, (137 + 400_003 * k) % 1_000_000) for k in range(n_calls)]
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    sdr.setSmiDebugMode(mode)
    sdr.setMaxRead(nb)
    sdr.setSmiDebugClock(clock)
    sdr.feedSmiBytes(stream)
    st = orc.DebugState()
    want, old, pos, analysed = 0.0, (0, 0), 0, 0
    for k in range(n_calls):
        chunk = stream[pos:pos + min(nb, 4 * length_samples)]
        pos += chunk.size
        ret, _, _ = sdr.smiRead(0, length_samples)
        offs = st.analyze(mode, chunk)
        if offs < 0:
            assert ret == -3
            continue                                          # no analysis, no clock reading (caribou_smi.c:665-668)
        alen = chunk.size - 4 * ((offs // 4 + 1) if offs > 0 else 0)
        want = orc.bitrate_ema(alen, old, clock[analysed], want)
        old = clock[analysed]; analysed += 1
        got, last = sdr.smiDebugBitrate()
        assert got == want and last == old
    assert analysed > 1
    sdr.close()
