"""-m gpu: the SoapySDR stream calls and the SMI seam of the host C layer, driven like the
reference's own clients (examples/python/read_test.py), checked against the oracle."""
import os
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
NB = 524288          # native batch bytes (caribou_smi.c:78)
MTU = 131072
SENT = -21846


@pytest.fixture(scope="module")
def S():
    import torch
    from cariboulite_amd import hip, soapy
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return soapy


def words(n, ch=0, seed=0):
    from cariboulite_amd import synth
    rng = np.random.default_rng(seed)
    return synth.iq_to_words(rng.integers(-4096, 4096, n), rng.integers(-4096, 4096, n), ch,
                             rng.integers(0, 2, n)).view(np.uint8)


def test_device_and_stream_api_surface(S):
    with pytest.raises(RuntimeError):
        S.Device(dict(driver="Cariboulite", channel="XYZ"))          # Cariboulite.cpp:25-28 throws
    with pytest.raises(RuntimeError):
        S.Device(dict(driver="Cariboulite"))
    sdr = S.Device(dict(driver="Cariboulite", channel="HiF", device_id="0", label="x"))
    assert sdr.getStreamFormats(S.SOAPY_SDR_RX, 0) == ["CS16", "CS8", "CF32", "CF64"]
    assert sdr.getNativeStreamFormat(S.SOAPY_SDR_RX, 0) == ("CS16", 4095.0)
    with pytest.raises(RuntimeError, match="invalid format"):
        sdr.setupStream(S.SOAPY_SDR_RX, "CS12")                       # CaribouliteStreamFunctions.cpp:111-115
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16, args={"CW": "0"})
    assert sdr.getStreamMTU(rx) == MTU
    assert sdr.activateStream(rx) == 0
    buf = np.zeros((16, 2), np.int16)
    assert sdr.readStream(rx, [buf], 16).ret == 0                     # nothing pending: timeout -> 0 samples
    assert sdr.writeStream(rx, [buf], 16).ret == S.SOAPY_SDR_NOT_SUPPORTED
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CS16)
    assert tx == rx                                                   # THE preallocated stream (:105,138)
    assert sdr.readStream(tx, [buf], 16).ret == S.SOAPY_SDR_NOT_SUPPORTED
    assert sdr.deactivateStream(tx) == 0
    sdr.closeStream(tx)
    sdr.close()


@pytest.mark.parametrize("ch,name", [(0, "S1G"), (1, "HiF")])
def test_read_cs16_like_read_test_py(S, orc, ch, name):
    sdr = S.Device(dict(driver="Cariboulite", channel=name))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    sdr.activateStream(rx)
    n = 3 * MTU + 1000
    b = words(n, ch, seed=1)
    sdr.feedSmiBytes(b)
    buf = np.full((n + 2, 2), SENT, np.int16)
    sr = sdr.readStream(rx, [buf], n, timeoutUs=int(5e6))             # CS16 is NOT clamped to the MTU
    ret, want, _ = orc.smi_read(ch, b, n, NB)
    assert sr.ret == ret == n and np.array_equal(buf, want)
    assert sdr.pendingSmiBytes() == 0
    sdr.close()


def test_read_converted_formats_clamp_to_mtu(S, orc):
    for fmt, dt, conv in (("CF32", np.float32, orc.cs16_to_cf32), ("CF64", np.float64, orc.cs16_to_cf64),
                          ("CS8", np.int8, orc.cs16_to_cs8)):
        sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
        rx = sdr.setupStream(S.SOAPY_SDR_RX, fmt)
        b = words(MTU + 5000, 0, seed=2)
        sdr.feedSmiBytes(b)
        buf = np.zeros((200000, 2), dt)
        sr = sdr.readStream(rx, [buf], 200000)
        assert sr.ret == MTU                                          # CaribouliteStream.cpp:306
        _, iq, _ = orc.rx_data_analyze(0, b[:NB])
        assert np.array_equal(buf[:MTU], conv(iq[:MTU])) and not buf[MTU:].any()
        assert sdr.readStream(rx, [buf], 200000).ret == 5000          # the rest of the FIFO, then drained
        sdr.close()


def test_smi_seam_return_codes_and_untouched_slots(S, orc):
    g = load_golden("smi_read_cases.npz")
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    # the reference's own outputs, at the native batch size
    for seed, mis in ((3, 0), (4, 2), (5, 6), (6, 4097)):
        n = 2 * MTU
        body = words(n, 0, seed=seed)
        stream = np.concatenate([body[:NB], np.zeros(mis, np.uint8), body[NB:]])[: 4 * n]
        sdr.feedSmiBytes(stream)
        ret, iq, meta = sdr.smiRead(0, n)
        wret, wiq, wmeta = orc.smi_read(0, stream, n, NB)
        assert ret == wret == n
        assert np.array_equal(iq, wiq) and np.array_equal(meta, wmeta)   # sentinel slots stay untouched
    # sync failure in the second chunk: -3, first chunk delivered, later bytes still pending
    n = 3 * MTU
    bad = words(n, 0, seed=7).copy(); bad[NB:2 * NB] = 0
    sdr.feedSmiBytes(bad)
    ret, iq, meta = sdr.smiRead(0, n)
    wret, wiq, wmeta = orc.smi_read(0, bad, n, NB)
    assert ret == wret == S.SMI_ERR_SYNC
    assert np.array_equal(iq, wiq) and np.array_equal(meta, wmeta)
    assert sdr.pendingSmiBytes() == NB                                 # the reference stops reading at the bad chunk
    ret, iq2, _ = sdr.smiRead(0, MTU)
    assert ret == MTU and np.array_equal(iq2[:MTU], orc.rx_data_analyze(0, bad[2 * NB:])[1][:MTU])
    # through Soapy the same failure is squashed to 0 samples (CaribouliteStream.cpp:266-276)
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    sdr.feedSmiBytes(bad[NB:2 * NB])
    buf = np.zeros((MTU, 2), np.int16)
    assert sdr.readStream(rx, [buf], MTU).ret == 0
    # short reads of the fd (max_read) and a drained source (timeout -> partial count)
    for name in ("five_chunks_aligned", "eof_timeout", "partial_last_chunk", "zero_length"):
        ch, nn, batch = [int(v) for v in g[f"{name}__args"]]
        sdr.setMaxRead(batch)                                          # emulate the fixture's 4 KiB native batch
        sdr.feedSmiBytes(g[f"{name}__bytes"])
        ret, iq, meta = sdr.smiRead(ch, nn)
        assert ret == int(g[f"{name}__ret"])
        assert np.array_equal(iq, g[f"{name}__iq"]) and np.array_equal(meta, g[f"{name}__meta"])
        sdr.drainSmiBytes(); sdr.smiRead(0, 1 << 16)                   # flush leftovers
        sdr.setMaxRead(0)
    sdr.close()


def test_iir_via_set_bandwidth_state_persists(S, orc):
    """Cariboulite.cpp:395-417 -> CaribouliteStream.cpp:291-298; states are never reset on a swap."""
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    from cariboulite_amd import synth
    b, i, q = synth.smi_stream_bytes(3 * MTU, 0, stream=9)
    iq = np.stack([i, q], 1).astype(np.int16)
    f20, f100 = orc.IIR(6, 4e6, 10e3), orc.IIR(6, 4e6, 50e3)
    sdr.feedSmiBytes(b)
    buf = np.zeros((MTU, 2), np.int16)
    for bw, filt, want_type, k in ((15e3, f20, 1, 0), (90e3, f100, 3, 1), (20e3, f20, 1, 2)):
        sdr.setBandwidth(S.SOAPY_SDR_RX, 0, bw)
        assert sdr.getDigitalFilter() == want_type
        assert sdr.readStream(rx, [buf], MTU).ret == MTU
        want = filt.apply_cs16(iq[k * MTU:(k + 1) * MTU])              # f20 keeps its state from call 0 to call 2
        d = np.abs(buf.astype(int) - want.astype(int))
        assert d.max() <= 1 and np.mean(d != 0) < 1e-4
    sdr.setBandwidth(S.SOAPY_SDR_RX, 0, 200e3)
    assert sdr.getDigitalFilter() == 0
    sdr.setBandwidth(S.SOAPY_SDR_RX, 0, 120e3)
    assert sdr.getDigitalFilter() == 0                                 # 100k < bw < 160k -> none (:404)
    # IIR + CF32: filter on int16, truncate, then /4096
    sdr.setBandwidth(S.SOAPY_SDR_RX, 0, 50e3)
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32)
    b2, i2, q2 = synth.smi_stream_bytes(MTU, 0, stream=10)
    sdr.feedSmiBytes(b2)
    fb = np.zeros((MTU, 2), np.float32)
    assert sdr.readStream(rx, [fb], MTU).ret == MTU
    want = orc.cs16_to_cf32(orc.IIR(6, 4e6, 25e3).apply_cs16(np.stack([i2, q2], 1)))
    assert np.max(np.abs(fb - want)) <= 1.0 / 4096 + 1e-9 and np.mean(fb != want) < 1e-4
    sdr.close()


def test_iir_overrun_is_repaired_inside_the_read(S, orc):
    """A bounded poll of the single-pass IIR kernel that gives up (forced here with a negative bound on this stream's
    filters) must not reach the client as samples, and must not poison the filter either: the reference's filter state
    lives as long as the Stream (CaribouliteStream.cpp:84-91).  The read repeats the call on the scan path from the
    unfiltered samples it still holds (the filter runs out of place) and delivers; the counter says it happened; the
    following reads continue from the right state."""
    if os.environ.get("CLHIP_IIR_ONEPASS") == "0":
        pytest.skip("the A/B switch in force replaces the single-pass kernel whose polls this test forces to give up")
    from cariboulite_amd import synth
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    sdr.setBandwidth(S.SOAPY_SDR_RX, 0, 90e3)
    K = 16                                                        # CS16 reads are not clamped to the MTU
    b, i, q = synth.smi_stream_bytes((K + 3) * MTU, 0, stream=12)
    want = orc.IIR(6, 4e6, 50e3).apply_cs16(np.stack([i, q], 1))
    sdr.feedSmiBytes(b)
    got = []
    buf = np.zeros((MTU, 2), np.int16)
    assert sdr.readStream(rx, [buf], MTU).ret == MTU              # a normal read first: the state is not at rest
    got.append(buf.copy())
    assert sdr.streamIirOverruns(rx) == 0
    sdr.setStreamIirPollBound(rx, -1)
    big = np.zeros((K * MTU, 2), np.int16)
    assert sdr.readStream(rx, [big], K * MTU).ret == K * MTU      # overran, rolled back, repeated: delivered
    got.append(big.copy())
    assert sdr.streamIirOverruns(rx) == 1 and "gave up" in sdr.lastError()
    for _ in range(2):                                            # the following batches continue from the right state
        assert sdr.readStream(rx, [buf], MTU).ret == MTU
        got.append(buf.copy())
    assert sdr.streamIirOverruns(rx) == 1                         # the filter stays on the scan path: nothing to give up
    g = np.concatenate(got)
    d = np.abs(g.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 1e-4, (d.max(), np.mean(d != 0))
    sdr.close()


def test_two_filtered_streams_one_overruns_the_other_does_not_notice(S, orc):
    """The board's two channels on one GPU, both with the IIR selected, one thread each (SoapyCariboulite.cpp:46-69).
    HiF's filters are forced to overrun on every read; S1G's are not.  S1G's samples and carried state must be exact
    (it never sees HiF's verdicts: the overrun word belongs to the filter object), and HiF's reads -- the overrun one and
    the ones after it -- equal the oracle too (rollback + repeat on the scan path)."""
    if os.environ.get("CLHIP_IIR_ONEPASS") == "0":
        pytest.skip("the A/B switch in force replaces the single-pass kernel whose polls this test forces to give up")
    import threading
    from cariboulite_amd import synth
    n_batches = 8
    devs = {ch: S.Device(dict(driver="Cariboulite", channel=name)) for ch, name in ((0, "S1G"), (1, "HiF"))}
    data = {ch: synth.smi_stream_bytes(n_batches * MTU, ch, stream=70 + ch) for ch in (0, 1)}
    rx, fmt = {}, {0: S.SOAPY_SDR_CS16, 1: S.SOAPY_SDR_CF32}
    for ch in (0, 1):
        rx[ch] = devs[ch].setupStream(S.SOAPY_SDR_RX, fmt[ch])
        devs[ch].setBandwidth(S.SOAPY_SDR_RX, 0, 45e3)            # fc = 25 kHz
        devs[ch].feedSmiBytes(data[ch][0])
    devs[1].setStreamIirPollBound(rx[1], -1)
    got = {0: [], 1: []}
    errs = []

    def reader(ch):
        try:
            for _ in range(n_batches):
                buf = np.zeros((MTU, 2), np.int16 if ch == 0 else np.float32)
                r = devs[ch].readStream(rx[ch], [buf], MTU).ret
                got[ch].append(buf[:r].copy())
        except Exception as e:                                   # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=reader, args=(ch,)) for ch in (0, 1)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=120)
    assert not errs and all(len(got[ch]) == n_batches for ch in (0, 1))
    assert devs[0].streamIirOverruns(rx[0]) == 0
    assert devs[1].streamIirOverruns(rx[1]) == 1                  # the first read overran; from then on HiF's filter is on the scan path
    for ch in (0, 1):
        want = orc.IIR(6, 4e6, 25e3).apply_cs16(np.stack([data[ch][1], data[ch][2]], 1))
        g = np.concatenate(got[ch])
        if ch == 1:
            assert np.array_equal(g * 4096.0, np.round(g * 4096.0))
            g = (g * 4096.0).astype(np.int16)
        d = np.abs(g.astype(np.int32) - want.astype(np.int32))
        assert g.shape == want.shape and d.max() <= 1 and np.mean(d != 0) < 1e-4, (ch, d.max(), np.mean(d != 0))
    for ch in (0, 1):
        devs[ch].close()


def test_rx_extension_stages_via_kwargs(S, orc):
    """FIR / RESAMP / DEMOD kwargs (SURVEY.md section 5 'Config / flags'), default = reference behaviour."""
    from cariboulite_amd import synth
    t = load_golden("taps.npz")
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"FIR": "64:1000000", "RESAMP": "3/2"})
    b, i, q = synth.smi_stream_bytes(2 * MTU, 0, stream=12)
    sdr.feedSmiBytes(b)
    x = orc.cs16_to_cf32(np.stack([i, q], 1))
    fir, rs = orc.FIR(t["fir64_c2"]), orc.Resampler(t["rs_3_2"], 3, 2)
    for k in range(2):
        out = np.zeros((MTU * 3 // 2 + 8, 2), np.float32)
        sr = sdr.readStream(rx, [out], MTU)
        assert sr.ret == MTU * 3 // 2
        want = rs.f64(fir.f64(x[k * MTU:(k + 1) * MTU]))
        assert np.max(np.abs(out[:sr.ret] - want)) <= 1e-5 * np.max(np.abs(want))
    # misaligned chunk -> CS16 route through the same stages (re-sync + extrapolated sample)
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"FIR": "64:100000", "DEMOD": "FM"})
    mis = np.concatenate([np.zeros(3, np.uint8), b])[:NB]
    sdr.smiRead(0, 1 << 20)
    sdr.feedSmiBytes(mis)
    outf = np.zeros(MTU + 8, np.float32)
    sr = sdr.readStream(rx, [outf], MTU)
    assert sr.ret == MTU
    _, iqm, _ = orc.rx_data_analyze(0, mis)
    xm = orc.cs16_to_cf32(iqm[:MTU - 1])            # slot MTU-1 is an untouched (stale) slot: compare the rest
    y = orc.FIR(t["fir64_c3"]).f64(xm)
    want = orc.fm_demod_f64(y)[0]
    mag = np.hypot(y[:, 0], y[:, 1]); ok = mag > 0.02; ok[1:] &= ok[:-1]
    d = np.abs(outf[:MTU - 1] - want); d = np.minimum(d, 2 * np.pi - d)
    assert np.max(d[ok]) <= 1e-4 and ok.mean() > 0.5
    with pytest.raises(RuntimeError, match="need format CF32"):
        sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16, args={"FIR": "64:1000000"})
    with pytest.raises(RuntimeError, match="invalid RESAMP"):
        sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"RESAMP": "x"})
    sdr.close()


def test_read_stream_cs16_batches_with_a_lost_and_a_slipped_batch(S, orc):
    """readStream(CS16), one MTU per call (the read-ahead reader: the next batch is staged while this one is analysed):
    a batch without sync yields 0 (Stream::Read squashes -3) and consumes exactly that batch -- what had been staged
    ahead goes back to the FIFO --, a slipped batch is re-synchronised, the batches around them are untouched; then
    flush drops what is pending (caribou_smi_flush_fifo)."""
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    b = words(5 * MTU, 0, seed=71).copy()
    b[NB:2 * NB] = 0                                                             # batch 1: no sync
    b[3 * NB:4 * NB] = np.concatenate([np.full(3, 0x15, np.uint8), b[3 * NB:4 * NB - 3]])   # batch 3: 3 bytes late
    sdr.feedSmiBytes(b)
    buf = np.full((MTU, 2), -21846, np.int16)
    for k in range(5):
        buf[:] = -21846
        sr = sdr.readStream(rx, [buf], MTU)
        ret, iq, _ = orc.smi_read(0, b[k * NB:(k + 1) * NB], MTU, NB)
        if k == 1:
            assert ret == -3 and sr.ret == 0
            assert (buf == -21846).all()                                         # a failed read leaves the client's buffer alone
            assert sdr.pendingSmiBytes() == 3 * NB                               # batches 2..4 still queued (staged-ahead bytes count)
            continue
        assert sr.ret == ret == MTU
        assert np.array_equal(buf, iq[:MTU]), k                                  # slot for slot (sentinel where the reference writes nothing)
    assert sdr.readStream(rx, [buf], MTU).ret == 0
    sdr.feedSmiBytes(b[:2 * NB])
    assert S.lib().cl_smi_flush_fifo(sdr.smi) == 0 and sdr.pendingSmiBytes() == 0
    assert sdr.readStream(rx, [buf], MTU).ret == 0
    sdr.close()


def test_rx_pipe_stream_resync_and_sync_loss(S, orc):
    """readStream(CF32, FIR + 3/2) over a byte stream with a slipped batch and a batch without sync: the slipped one is
    redone with the reference's re-sync semantics (its untouched slot keeps the previous batch's sample, as the
    Stream's persistent interm_native_buffer does), the lost one yields 0 elements (Stream::Read squashes -3,
    CaribouliteStream.cpp:266-276) and leaves the float stages where they were -- the batches after it continue the
    oracle's smi_read -> FIR -> 3/2 chain."""
    from cariboulite_amd import synth
    t = load_golden("taps.npz")
    SENT = -21846
    sdr = S.Device(dict(driver="Cariboulite", channel="HiF"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"FIR": "64:1000000", "RESAMP": "3/2"})
    b = synth.smi_stream_bytes(5 * MTU, 1, stream=31)[0].copy()
    b[NB:2 * NB] = np.concatenate([np.full(6, 0x11, np.uint8), b[NB:2 * NB - 6]])      # batch 1: 6 bytes late
    b[3 * NB:4 * NB] = 0                                                               # batch 3: no sync
    sdr.feedSmiBytes(b)
    fir, rs = orc.FIR(t["fir64_c2"]), orc.Resampler(t["rs_3_2"], 3, 2)
    interm = np.zeros((MTU + 2, 2), np.int16)            # the Stream's interm_native_buffer2: untouched slots persist
    for k in range(5):
        out = np.zeros((MTU * 3 // 2 + 8, 2), np.float32)
        sr = sdr.readStream(rx, [out], MTU)
        ret, iq, _ = orc.smi_read(1, b[k * NB:(k + 1) * NB], MTU, NB, fill=SENT)
        if k == 3:
            assert ret == -3 and sr.ret == 0
            assert not out.any()                              # nothing delivered, nothing written
            continue
        assert ret == MTU
        touched = (iq != SENT).any(axis=1)
        assert touched[:MTU].sum() == (MTU - 1 if k == 1 else MTU)          # offs 6: slot MTU-1 untouched
        interm[touched] = iq[touched]
        want = rs.f64(fir.f64(orc.cs16_to_cf32(interm[:MTU])))
        assert sr.ret == want.shape[0]
        assert np.max(np.abs(out[:sr.ret] - want)) <= 1e-5 * np.max(np.abs(want)), k
    sdr.close()


def test_write_stream_all_formats(S, orc):
    from cariboulite_amd import hip
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rng = np.random.default_rng(21)
    n = MTU + 777
    iq = rng.integers(-4096, 4096, (n, 2)).astype(np.int16)
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CS16)
    assert sdr.writeStream(tx, [iq], n).ret == n                        # CS16: no MTU clamp (:182-196)
    assert np.array_equal(sdr.drainSmiBytes(), orc.generate_data(iq, orc.TX_DOCUMENTED))
    f = (rng.standard_normal((n, 2)) * 0.4).astype(np.float32)
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CF32)
    assert sdr.writeStream(tx, [f], n).ret == MTU                       # clamped (:201)
    assert np.array_equal(sdr.drainSmiBytes(), orc.generate_data(orc.cf32_to_cs16(f[:MTU]), orc.TX_DOCUMENTED))
    d = f.astype(np.float64) * 1.00001
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CF64)
    assert sdr.writeStream(tx, [d], 5000).ret == 5000
    assert np.array_equal(sdr.drainSmiBytes(), orc.generate_data(orc.cf64_to_cs16(d[:5000]), orc.TX_DOCUMENTED))
    i8 = rng.integers(-128, 128, (5000, 2)).astype(np.int8)
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CS8)
    assert sdr.writeStream(tx, [i8], 5000).ret == 5000
    assert np.array_equal(sdr.drainSmiBytes(), orc.generate_data(orc.cs8_to_cs16(i8), orc.TX_DOCUMENTED))
    # compat mode: the packer exactly as shipped (caribou_smi.c:700-701)
    sdr.setTxMode(hip.TX_AS_WRITTEN)
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CS16)
    sdr.writeStream(tx, [iq], 256)
    g = load_golden("smi_tx_as_written.npz")
    assert np.array_equal(sdr.drainSmiBytes(), g["bytes"])
    sdr.setTxMode(hip.TX_DOCUMENTED)
    # lower seam
    assert sdr.smiWrite(0, iq[:1000]) == 1000
    assert np.array_equal(sdr.drainSmiBytes(), orc.generate_data(iq[:1000]))
    # config 5 through kwargs: FM modulate the I rail, 2/3 resample, pack
    t = load_golden("taps.npz"); g = load_golden("dsp_float.npz")
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CF32, args={"MOD": "FM:75000", "RESAMP": "2/3"})
    m = g["fm_msg"]
    msg_iq = np.stack([m, np.zeros_like(m)], 1)
    assert sdr.writeStream(tx, [msg_iq], m.size).ret == m.size
    by = sdr.drainSmiBytes()
    assert by.size == 4 * (-(-m.size * 2 // 3))
    w = orc.fpga_tx_parse(by)
    _, got, _ = orc.rx_data_analyze(0, (w & ~np.uint32(1 << 16)).view(np.uint8))
    want = g["fm_mod_rs_2_3"] * 4096.0
    wq = ((np.trunc(want).astype(np.int64) + 4096) & 0x1FFF) - 4096
    dd = np.abs(got[:wq.shape[0]].astype(np.int64) - wq)
    dd = np.minimum(dd, 8192 - dd)
    assert dd.max() <= 1                                               # float->int boundary: +-1 LSB
    sdr.close()


def test_write_stream_refuses_bytes_of_an_overrun_lookback(S, orc, monkeypatch):
    """cl_writeStream asks the modulator for its verdict before anything reaches the TX FIFO: with the look-back
    forced to give up, the same call returns 0 (errors are squashed, CaribouliteStream.cpp:185-194) and the FIFO
    stays empty; the stream then works normally again."""
    g = load_golden("dsp_float.npz")
    m = np.tile(g["fm_msg"], 8)[:60000]
    msg_iq = np.stack([m, np.zeros_like(m)], 1)
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    monkeypatch.setenv("CLHIP_TX_POLL_BOUND", "0")              # read when the TX pipe is created
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CF32, args={"MOD": "FM:75000", "RESAMP": "2/3"})
    monkeypatch.delenv("CLHIP_TX_POLL_BOUND")
    assert sdr.writeStream(tx, [msg_iq], m.size).ret == 0
    assert sdr.drainSmiBytes().size == 0
    assert "look-back" in sdr.lastError()
    assert sdr.writeStream(tx, [msg_iq], m.size).ret == 0       # still refused: same pipe, same bound
    assert sdr.drainSmiBytes().size == 0
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CF32, args={"MOD": "FM:75000", "RESAMP": "2/3"})
    assert sdr.writeStream(tx, [msg_iq], m.size).ret == m.size
    assert sdr.drainSmiBytes().size == 4 * (-(-m.size * 2 // 3))
    sdr.close()


def test_file_and_pipe_replay_front_end(S, orc, tmp_path):
    """recorded capture from a file, live feed through a pipe with short ragged
    reads, and the TX mirror into a file -- all through the reference's chunk semantics."""
    import os
    n = 2 * MTU + 3000
    b = words(n, 0, seed=33)
    # a capture with 5 junk bytes in front: first chunk re-synchronises (extrapolated sample, untouched slot)
    cap = np.concatenate([np.zeros(5, np.uint8), b])
    f = tmp_path / "capture.smi"
    cap.tofile(f)
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    assert sdr.feedSmiFile(f) == cap.size
    ret, iq, meta = sdr.smiRead(0, n)
    wret, wiq, wmeta = orc.smi_read(0, cap, n, NB)
    assert ret == wret and np.array_equal(iq, wiq) and np.array_equal(meta, wmeta)
    sdr.smiRead(0, 1 << 20)                                            # flush the tail
    # offset + length window of the same file
    assert sdr.feedSmiFile(f, offset=5, max_bytes=4 * MTU) == 4 * MTU
    ret, iq, _ = sdr.smiRead(0, MTU)
    assert ret == MTU and np.array_equal(iq[:MTU], orc.rx_data_analyze(0, b[:NB])[1][:MTU])
    # pipe: the writer hands over ragged pieces; whatever arrives is queued in order
    r, w = os.pipe()
    os.set_blocking(r, False)
    pieces = [1001, 7, 4093, 65536 - 5101, 20000]
    pos = 0
    for p in pieces:
        os.write(w, b[pos:pos + p].tobytes()); pos += p
        assert sdr.feedSmiFd(r) == p                                   # EAGAIN ends the pump, nothing lost
    os.close(w)
    assert sdr.feedSmiFd(r) == 0                                       # EOF
    os.close(r)
    nn = pos // 4
    ret, iq, _ = sdr.smiRead(0, nn)
    assert ret == nn and np.array_equal(iq[:nn], orc.rx_data_analyze(0, b[:4 * nn])[1][:nn])
    # TX mirror
    rng = np.random.default_rng(2)
    tx_iq = rng.integers(-4096, 4096, (MTU + 50, 2)).astype(np.int16)
    assert sdr.smiWrite(0, tx_iq) == MTU + 50
    out = tmp_path / "tx.smi"
    fd = os.open(out, os.O_WRONLY | os.O_CREAT)
    assert sdr.drainSmiToFd(fd) == 4 * (MTU + 50)
    os.close(fd)
    assert np.array_equal(np.fromfile(out, np.uint8), orc.generate_data(tx_iq))
    sdr.close()


def test_async_stream_reader_thread_and_ring(S, orc):
    """ASYNC=1: the reference's compiled-out USE_ASYNC path (CaribouliteStream.cpp:16-49,70-75) --
    a reader thread fills a 10-MTU overwrite-oldest ring, readStream pops whole requests with its timeout."""
    import time
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16, args={"ASYNC": "1"})
    buf = np.zeros((MTU, 2), np.int16)
    t0 = time.perf_counter()
    assert sdr.readStream(rx, [buf], MTU, timeoutUs=30000).ret == 0        # inactive: nothing arrives, times out
    assert time.perf_counter() - t0 >= 0.025
    sdr.activateStream(rx)
    b = words(4 * MTU, 0, seed=41)
    sdr.feedSmiBytes(b)
    want = orc.smi_read(0, b, 4 * MTU, NB)[1]
    for k in range(4):
        sr = sdr.readStream(rx, [buf], MTU, timeoutUs=2_000_000)
        assert sr.ret == MTU and np.array_equal(buf, want[k * MTU:(k + 1) * MTU])
    assert sdr.readStream(rx, [buf], MTU, timeoutUs=20000).ret == 0        # drained: whole-request rule
    # partial request sizes pop across chunk boundaries in order
    sdr.feedSmiBytes(b[: 4 * 2 * MTU])
    time.sleep(0.3)
    small = np.zeros((50000, 2), np.int16)
    got = []
    for _ in range(5):
        assert sdr.readStream(rx, [small], 50000, timeoutUs=1_000_000).ret == 50000
        got.append(small.copy())
    assert np.array_equal(np.concatenate(got), want[:250000])
    # CF32 + IIR on the consumer side of the ring
    sdr.deactivateStream(rx)
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"ASYNC": "1"})
    sdr.setBandwidth(S.SOAPY_SDR_RX, 0, 90e3)
    sdr.activateStream(rx)
    from cariboulite_amd import synth
    b2, i2, q2 = synth.smi_stream_bytes(MTU, 0, stream=3)
    sdr.feedSmiBytes(b2)
    fb = np.zeros((MTU, 2), np.float32)
    assert sdr.readStream(rx, [fb], MTU, timeoutUs=2_000_000).ret == MTU
    wf = orc.cs16_to_cf32(orc.IIR(6, 4e6, 50e3).apply_cs16(np.stack([i2, q2], 1)))
    assert np.max(np.abs(fb - wf)) <= 1.0 / 4096 + 1e-9 and np.mean(fb != wf) < 1e-4
    # overwrite-oldest: feed 12 MTUs without reading; the ring keeps the newest 10 MTUs worth (capacity 2^21)
    sdr.setBandwidth(S.SOAPY_SDR_RX, 0, 200e3)
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16, args={"ASYNC": "1"})
    sdr.activateStream(rx)
    big = words(12 * MTU, 0, seed=43)
    for k in range(12):
        sdr.feedSmiBytes(big[k * NB:(k + 1) * NB])
    time.sleep(1.0)
    wantb = orc.smi_read(0, big, 12 * MTU, NB)[1][:12 * MTU]
    assert sdr.readStream(rx, [buf], MTU, timeoutUs=1_000_000).ret == MTU
    # 12 MTU put into a 16-MTU ring (capacity = next power of two of 10 MTU): nothing dropped yet
    assert np.array_equal(buf, wantb[:MTU])
    sdr.close()


@pytest.mark.parametrize("name", [str(n) for n in load_golden("ring_cases.npz")["names"]])
def test_device_ring_replays_the_reference_fixtures(S, name):
    """The ASYNC mode's ring with its storage in DEVICE memory: counts, order and fill level of the op sequences
    recorded from the reference's circular_buffer<uint32_t> (tests/golden/ring_cases.npz)."""
    g = load_golden("ring_cases.npz")
    size, ov, blk, cap = [int(v) for v in g[f"{name}__cfg"]]
    r = S.Ring(size, ov, blk, device=0)
    assert r.capacity() == cap and S.lib().cl_ring_on_device(r.h) == 1
    ops, rets, sizes, popped = g[f"{name}__ops"], g[f"{name}__rets"], g[f"{name}__sizes"], g[f"{name}__popped"]
    ctr, pp = 0, 0
    for (kind, n), want, sz in zip(ops.tolist(), rets.tolist(), sizes.tolist()):
        if kind == 0:
            d = np.arange(ctr, ctr + n, dtype=np.uint32); ctr += n
            assert r.put(d) == want
        else:
            k, d = r.get(n, timeout_us=100)
            assert k == want and np.array_equal(d, popped[pp:pp + k])
            pp += k
        assert r.size() == sz
    assert pp == popped.size


def test_device_ring_spans_move_data_device_to_device(S):
    """Span calls: the caller moves the data itself (here torch D2D copies) between _begin and _end; wrap-around gives
    two pieces; overwrite-oldest releases exactly what the new elements need."""
    import ctypes as C
    import torch
    r = S.Ring(1000, True, True, device=0)                    # capacity 1024 uint32
    base = S.lib().cl_ring_storage(r.h)

    class Span(C.Structure):
        _fields_ = [("pos", C.c_size_t * 2), ("len", C.c_size_t * 2)]

    from cariboulite_amd import hip
    sp = Span()

    def put(src):              # src: device int32 tensor
        n = S.lib().cl_ring_put_begin(r.h, src.numel(), C.byref(sp))
        off = 0
        for k in range(2):
            if sp.len[k]:
                hip.lib().clhip_memcpy_d2d(base + 4 * sp.pos[k], src.data_ptr() + 4 * off, 4 * sp.len[k], None)
                off += sp.len[k]
        torch.cuda.synchronize(); hip.lib().clhip_stream_sync(None)
        S.lib().cl_ring_put_end(r.h, n)
        return n, [(sp.pos[k], sp.len[k]) for k in range(2)]

    def get(n):
        dst = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        k = S.lib().cl_ring_get_begin(r.h, n, 1000, C.byref(sp))
        if not k:
            return 0, dst, None
        off = 0
        for j in range(2):
            if sp.len[j]:
                hip.lib().clhip_memcpy_d2d(dst.data_ptr() + 4 * off, base + 4 * sp.pos[j], 4 * sp.len[j], None)
                off += sp.len[j]
        hip.lib().clhip_stream_sync(None)
        S.lib().cl_ring_get_end(r.h, k)
        return k, dst, [(sp.pos[j], sp.len[j]) for j in range(2)]

    a = torch.arange(0, 700, dtype=torch.int32, device="cuda:0")
    assert put(a)[0] == 700
    k, d, _ = get(600)
    assert k == 600 and torch.equal(d, a[:600])
    b = torch.arange(700, 1300, dtype=torch.int32, device="cuda:0")
    n, pieces = put(b)                                         # 100 held; 600 more wrap past slot 1023
    assert n == 600 and pieces == [(700, 324), (0, 276)]
    c = torch.arange(1300, 1800, dtype=torch.int32, device="cuda:0")
    assert put(c)[0] == 500 and r.size() == 1024              # 700 + 500 > 1024: the oldest 176 are released
    k, d, pieces = get(1024)
    assert k == 1024 and d[0].item() == 1800 - 1024 and d[-1].item() == 1799
    assert torch.equal(d, torch.arange(776, 1800, dtype=torch.int32, device="cuda:0"))
    assert get(1)[0] == 0                                      # empty: whole-request rule, times out


def test_fanout_c_abi_single_rank_and_strides(S):
    """clfan_* (include/cariboulite_fanout.h) on the one GPU of this box: world 1 -- every stream is the root's own,
    so scatter / gather are the device-to-device legs of the same code that posts ncclSend / ncclRecv for peers --
    with row strides larger than the rows; and through shard.fanout_streams / gather_streams (dist=None)."""
    import torch
    from cariboulite_amd import fanout, shard
    ident = fanout.Comm.unique_id()                              # ncclGetUniqueId through the C ABI, torch's RCCL in the same process
    assert len(ident) == fanout.ID_BYTES and any(ident)
    c = fanout.Comm(1, 0, ident)
    assert fanout.lib().clfan_world(c.h) == 1 and fanout.lib().clfan_rank(c.h) == 0
    ns, n, pad = 5, 3000, 40
    root = torch.arange(ns * (n + pad), dtype=torch.int32, device="cuda:0").reshape(ns, n + pad)
    mine = torch.full((ns, n + 8), -1, dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    c.scatter(0, root.data_ptr(), 4 * (n + pad), 4 * n, ns, mine.data_ptr(), 4 * (n + 8), st)
    torch.cuda.synchronize()
    assert torch.equal(mine[:, :n], root[:, :n]) and bool((mine[:, n:] == -1).all())
    back = torch.zeros_like(root)
    c.gather(0, mine.data_ptr(), 4 * (n + 8), 4 * n, ns, back.data_ptr(), 4 * (n + pad), st)
    torch.cuda.synchronize()
    assert torch.equal(back[:, :n], root[:, :n]) and bool((back[:, n:] == 0).all())
    loc = shard.fanout_streams(root[:, :n].contiguous(), ns, None, 1, 0, root=0)
    assert torch.equal(loc, root[:, :n])
    assert torch.equal(shard.gather_streams(loc, ns, None, 1, 0), root[:, :n])
    c.close()


def test_two_devices_two_threads_concurrently(S, orc):
    """SURVEY 8(b) threading: one thread per stream, many streams at once.  The board's two channels are two Soapy
    devices (SoapyCariboulite.cpp:46-69); here each has its own SMI seam, HIP streams and pipes, and two client threads
    drive them at the same time: S1G reads CF32 through FIR64 + 3/2 with the IIR-free fused pipe, HiF reads CS16
    with the 100 kHz IIR selected.  Every batch must equal what the same calls give single-threaded (the oracle chain)."""
    import threading
    from cariboulite_amd import synth
    t = load_golden("taps.npz")
    n_batches = 6
    devs = {ch: S.Device(dict(driver="Cariboulite", channel=name)) for ch, name in ((0, "S1G"), (1, "HiF"))}
    data = {ch: synth.smi_stream_bytes(n_batches * MTU, ch, stream=90 + ch) for ch in (0, 1)}
    rx0 = devs[0].setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"FIR": "64:1000000", "RESAMP": "3/2"})
    rx1 = devs[1].setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    devs[1].setBandwidth(S.SOAPY_SDR_RX, 0, 90e3)
    for ch in (0, 1):
        devs[ch].feedSmiBytes(data[ch][0])
    got = {0: [], 1: []}
    errs = []

    def reader(ch):
        try:
            for _ in range(n_batches):
                if ch == 0:
                    buf = np.zeros((MTU * 3 // 2 + 8, 2), np.float32)
                    r = devs[0].readStream(rx0, [buf], MTU).ret
                else:
                    buf = np.zeros((MTU, 2), np.int16)
                    r = devs[1].readStream(rx1, [buf], MTU).ret
                got[ch].append(buf[:r].copy())
        except Exception as e:                                   # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=reader, args=(ch,)) for ch in (0, 1)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=120)
    assert not errs and all(len(got[ch]) == n_batches for ch in (0, 1))
    x0 = orc.cs16_to_cf32(np.stack([data[0][1], data[0][2]], 1))
    fir, rs = orc.FIR(t["fir64_c2"]), orc.Resampler(t["rs_3_2"], 3, 2)
    want0 = rs.f64(fir.f64(x0))
    g0 = np.concatenate(got[0])
    assert g0.shape == want0.shape and np.max(np.abs(g0 - want0)) <= 1e-5 * np.max(np.abs(want0))
    want1 = orc.IIR(6, 4e6, 50e3).apply_cs16(np.stack([data[1][1], data[1][2]], 1))
    g1 = np.concatenate(got[1])
    d = np.abs(g1.astype(np.int32) - want1.astype(np.int32))
    assert g1.shape == want1.shape and d.max() <= 1 and np.mean(d != 0) < 1e-4
    for ch in (0, 1):
        devs[ch].close()


def test_feeder_thread_races_the_in_place_reads(S, orc):
    """The threading model fifo_mu exists for: a producer (the kernel FIFO's stand-in) feeds while the consumer reads.
    Every read here is ONE read() that EMPTIES the pinned FIFO and is copied to the device from where it lies; the feeder
    pushes the next batch the moment the FIFO runs dry -- into the very memory the copy may still be reading if the
    consumed bytes were released before the copy had run (an empty FIFO restarts at the front of its buffer).  Three
    read paths: caribou_smi_read's seam (cl_smi_read), readStream CF32 (the chunk-at-a-time reader) and the fused pipe
    straight from the staged words.  Every batch must equal the oracle's analysis of the bytes that were fed."""
    import threading
    import time
    from cariboulite_amd import synth
    t = load_golden("taps.npz")
    n_batches = 40
    for mode in ("smi_read", "cf32", "pipe"):
        sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
        b, i, q = synth.smi_stream_bytes(n_batches * MTU, 0, stream=30)
        raw = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b.view(np.uint8)
        batches = [np.ascontiguousarray(raw[4 * MTU * k: 4 * MTU * (k + 1)]) for k in range(n_batches)]
        stop = threading.Event()

        def feeder():
            k = 0
            while k < n_batches and not stop.is_set():
                if sdr.pendingSmiBytes() == 0:                  # the moment the FIFO runs dry
                    sdr.feedSmiBytes(batches[k]); k += 1
                else:
                    time.sleep(0)

        th = threading.Thread(target=feeder)
        got = []
        if mode == "cf32":
            rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32)
        elif mode == "pipe":
            rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"FIR": "64:1000000", "RESAMP": "3/2"})
        th.start()
        try:
            deadline = time.time() + 120
            while len(got) < n_batches and time.time() < deadline:
                if mode == "smi_read":
                    ret, iq, _ = sdr.smiRead(0, MTU, want_meta=False)
                    if ret > 0:
                        assert ret == MTU
                        got.append(iq[:MTU].copy())
                else:
                    buf = np.zeros((MTU * 3 // 2 + 8, 2), np.float32)
                    r = sdr.readStream(rx, [buf], MTU).ret
                    if r > 0:
                        got.append(buf[:r].copy())
        finally:
            stop.set(); th.join(timeout=30)
        assert len(got) == n_batches, (mode, len(got))
        g = np.concatenate(got)
        x = np.stack([i, q], 1)
        if mode == "smi_read":
            assert np.array_equal(g, x), mode
        elif mode == "cf32":
            assert np.array_equal(g, orc.cs16_to_cf32(x)), mode
        else:
            want = orc.Resampler(t["rs_3_2"], 3, 2).f64(orc.FIR(t["fir64_c2"]).f64(orc.cs16_to_cf32(x)))
            assert g.shape == want.shape and np.max(np.abs(g - want)) <= 1e-5 * np.max(np.abs(want)), mode
        sdr.close()


def test_seam_and_stream_counters(S, orc):
    """What the reference only prints (cariboulite_radio.c:1276-1283, caribou_smi.c:657-668, CaribouliteStream.cpp:266-276)
    can be read back: samples, re-syncs, -3 exits and time-outs at the seam; calls, elements and squashed-to-zero calls
    at the stream."""
    from cariboulite_amd import synth
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    buf = np.zeros((MTU, 2), np.int16)
    assert sdr.readStream(rx, [buf], MTU).ret == 0                       # nothing pending: "Reading timed-out"
    b, _, _ = synth.smi_stream_bytes(3 * MTU, 0, stream=3)
    shifted = np.concatenate([np.zeros(6, np.uint8), b[4 * MTU: 8 * MTU - 6]])       # a batch that re-syncs at byte 6
    garbage = np.zeros(4 * MTU, np.uint8)                                 # a batch without any sync word
    sdr.feedSmiBytes(np.concatenate([b[: 4 * MTU], shifted, garbage, b[8 * MTU:]]))
    rets = [sdr.readStream(rx, [buf], MTU).ret for _ in range(4)]
    assert rets == [MTU, MTU, 0, MTU]
    assert sdr.readStream(rx, [buf], MTU).ret == 0
    st, sm = sdr.streamStats(rx), sdr.smiStats()
    assert st["read_calls"] == 6 and st["elements_read"] == 3 * MTU and st["reads_empty"] == 3 and st["iir_overruns"] == 0
    assert sm["samples_read"] == 3 * MTU and sm["resyncs"] == 1 and sm["sync_losses"] == 1 and sm["timeouts"] == 2 and sm["io_errors"] == 0
    tx = S.Device(dict(driver="Cariboulite", channel="HiF"))
    ts = tx.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CS16)
    assert tx.writeStream(ts, [buf], 1000).ret == 1000
    assert tx.streamStats(ts)["write_calls"] == 1 and tx.streamStats(ts)["elements_written"] == 1000
    assert tx.smiStats()["samples_written"] == 1000
    sdr.close(); tx.close()


def test_zero_copy_streams_deliver_what_the_default_route_delivers(S, orc):
    """ZEROCOPY=1 + cl_stream_register_buffer: buffers the client has REGISTERED are written by the last kernel of the read.  Same
    samples as the default route, for every format, through the IIR, through the fused stages, with a re-synchronising and a lost
    batch in between (the client's buffer sees nothing of a failed read); a pointer that is not registered -- or not 16-byte
    aligned -- takes the default route; the table holds eight buffers and refuses a ninth (no eviction); nothing is registered
    unless the client asks."""
    from cariboulite_amd import synth
    t = load_golden("taps.npz")
    b, i, q = synth.smi_stream_bytes(4 * MTU, 0, stream=21)
    iq = np.stack([i, q], 1)
    for fmt, dt, conv in ((S.SOAPY_SDR_CS16, np.int16, lambda v: v), ("CF32", np.float32, orc.cs16_to_cf32),
                          ("CF64", np.float64, orc.cs16_to_cf64), ("CS8", np.int8, orc.cs16_to_cs8)):
        sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
        rx = sdr.setupStream(S.SOAPY_SDR_RX, fmt, args={"ZEROCOPY": "1"})
        sdr.activateStream(rx)
        shifted = np.concatenate([np.zeros(6, np.uint8), b[4 * MTU: 8 * MTU - 6]])
        garbage = np.zeros(4 * MTU, np.uint8)
        sdr.feedSmiBytes(np.concatenate([b[: 4 * MTU], shifted, garbage, b[8 * MTU:]]))
        bufs = [np.full((MTU + 4, 2), 77, dt) for _ in range(2)]
        for x in bufs:
            sdr.registerStreamBuffer(rx, x)
        assert sdr.readStream(rx, [bufs[0]], MTU).ret == MTU
        assert np.array_equal(bufs[0][:MTU], conv(iq[:MTU])) and (bufs[0][MTU:] == 77).all()
        assert sdr.readStream(rx, [bufs[1]], MTU).ret == MTU                # re-sync at byte 6: not a one-read() in-sync call
        _, want, _ = orc.rx_data_analyze(0, shifted)
        if dt == np.int16:
            assert np.array_equal(bufs[1][:MTU - 2], want[:MTU - 2])
        else:
            assert np.array_equal(bufs[1][:MTU - 2], conv(want[:MTU - 2]))
        before = bufs[0].copy()
        assert sdr.readStream(rx, [bufs[0]], MTU).ret == 0                  # the batch without sync: nothing delivered,
        assert np.array_equal(bufs[0], before)                              # nothing written
        for k in range(2):
            assert sdr.readStream(rx, [bufs[k]], MTU).ret == MTU
            assert np.array_equal(bufs[k][:MTU], conv(iq[(2 + k) * MTU:(3 + k) * MTU]))
        st = sdr.streamStats(rx)
        # CS16 re-synchronised reads copy slot by slot (only what the reference writes): not a kernel's store; the others are
        assert st["zero_copy_registrations"] == 2 and st["zero_copy_reads"] == (3 if dt == np.int16 else 4), st
        sdr.close()
    # nothing is registered behind the client's back; an odd address takes the default route; the table of eight refuses a ninth
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16, args={"ZEROCOPY": "1"})
    plain = np.zeros((MTU, 2), np.int16)
    sdr.feedSmiBytes(b[: 4 * MTU])
    assert sdr.readStream(rx, [plain], MTU).ret == MTU and np.array_equal(plain, iq[:MTU])
    assert sdr.streamStats(rx)["zero_copy_reads"] == 0 and sdr.streamStats(rx)["zero_copy_registrations"] == 0
    # ten page-aligned slots two pages apart inside ONE allocation: their page ranges never touch (ranges that do are registered
    # together, as one -- neighbouring heap buffers share pages), so the table's count is the number of buffers
    slot_b = 4 * MTU + 8192
    big = np.zeros(10 * slot_b + 4096, np.uint8)
    a0 = (-big.ctypes.data) % 4096
    slots = [big[a0 + k * slot_b: a0 + k * slot_b + 4 * MTU + 64] for k in range(10)]
    raw = slots[0]
    odd = raw[4: 4 + 4 * MTU].view(np.int16).reshape(-1, 2)              # 4 bytes off the 16-byte grid
    sdr.registerStreamBuffer(rx, raw)
    sdr.feedSmiBytes(b[: 4 * MTU])
    assert sdr.readStream(rx, [odd], MTU).ret == MTU and np.array_equal(odd, iq[:MTU])
    assert sdr.streamStats(rx)["zero_copy_reads"] == 0
    many = [x[: 4 * MTU].view(np.int16).reshape(-1, 2) for x in slots[1:]]
    for m in many[:7]:
        sdr.registerStreamBuffer(rx, m)
    sdr.registerStreamBuffer(rx, many[3])                                # inside a registered range already: nothing to do
    with pytest.raises(RuntimeError, match="8 buffers are registered already"):
        sdr.registerStreamBuffer(rx, many[8])
    for rep in range(2):
        for k, m in enumerate(many):
            sdr.feedSmiBytes(b[4 * MTU * (k % 4): 4 * MTU * (k % 4 + 1)])
            assert sdr.readStream(rx, [m], MTU).ret == MTU and np.array_equal(m, iq[(k % 4) * MTU:(k % 4 + 1) * MTU])
    st = sdr.streamStats(rx)
    assert st["zero_copy_reads"] == 14 and st["zero_copy_registrations"] == 8, st     # many[7], many[8]: the default route
    sdr.unregisterStreamBuffers(rx)
    sdr.feedSmiBytes(b[: 4 * MTU])
    assert sdr.readStream(rx, [many[0]], MTU).ret == MTU and sdr.streamStats(rx)["zero_copy_reads"] == 14
    with pytest.raises(RuntimeError, match="not set up with ZEROCOPY=1"):
        d2 = S.Device(dict(driver="Cariboulite", channel="S1G"))
        d2.registerStreamBuffer(d2.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16), many[0])
    # the IIR and the fused stages as last kernels
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16, args={"ZEROCOPY": "1"})
    ref = S.Device(dict(driver="Cariboulite", channel="S1G"))
    rr = ref.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    for d in (sdr, ref):
        d.setBandwidth(S.SOAPY_SDR_RX, 0, 100e3)
        d.feedSmiBytes(b)
    o1, o2 = np.zeros((MTU, 2), np.int16), np.zeros((MTU, 2), np.int16)
    sdr.registerStreamBuffer(rx, o1)
    zc0 = sdr.streamStats(rx)["zero_copy_reads"]
    for k in range(4):
        assert sdr.readStream(rx, [o1], MTU).ret == MTU and ref.readStream(rr, [o2], MTU).ret == MTU
        assert np.array_equal(o1, o2)
    assert sdr.streamStats(rx)["zero_copy_reads"] == zc0 + 4          # (the filter is the last kernel on either route)
    ref.close()
    rx = sdr.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, args={"FIR": "64:1000000", "RESAMP": "3/2", "ZEROCOPY": "1"})
    sdr.setBandwidth(S.SOAPY_SDR_RX, 0, 2.5e6)
    sdr.feedSmiBytes(b[: 8 * MTU])
    x = orc.cs16_to_cf32(iq)
    fir, rs = orc.FIR(t["fir64_c2"]), orc.Resampler(t["rs_3_2"], 3, 2)
    out = np.zeros((MTU * 3 // 2, 2), np.float32)
    sdr.registerStreamBuffer(rx, out)
    zc1 = sdr.streamStats(rx)["zero_copy_reads"]
    for k in range(2):
        assert sdr.readStream(rx, [out], MTU).ret == MTU * 3 // 2
        want = rs.f64(fir.f64(x[k * MTU:(k + 1) * MTU]))
        assert np.max(np.abs(out - want)) <= 1e-5 * np.max(np.abs(want))
    assert sdr.streamStats(rx)["zero_copy_reads"] == zc1 + 2
    sdr.close()


def test_write_stream_with_a_concurrent_drainer(S, orc):
    """One thread writes (writeStream packs on the GPU into reserved room of the pinned TX FIFO), another drains what has been
    committed (cl_smi_drain_bytes, where the fd's write() side stands): the drained bytes, in order, are the oracle's pack of
    everything written -- the FIFO grows and is emptied under the writer's open reservations many times."""
    import threading
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G"))
    tx = sdr.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CS16)
    sdr.activateStream(tx)
    rng = np.random.default_rng(31)
    calls = [rng.integers(-4096, 4096, (int(n), 2)).astype(np.int16) for n in rng.integers(1000, 131072, 120)]
    total = 4 * sum(c.shape[0] for c in calls)
    got, stop = [], threading.Event()

    def drain():
        n = 0
        while n < total and not stop.is_set():
            b = sdr.drainSmiBytes(1 << 20)
            if b.size:
                got.append(b); n += b.size

    t = threading.Thread(target=drain)
    t.start()
    try:
        for c in calls:
            assert sdr.writeStream(tx, [c], c.shape[0]).ret == c.shape[0]
    finally:
        t.join(timeout=30)
        stop.set()
        t.join()
    out = np.concatenate(got) if got else np.zeros(0, np.uint8)
    assert out.size == total
    assert np.array_equal(out, orc.generate_data(np.concatenate(calls), orc.TX_DOCUMENTED))
    sdr.close()
