"""-m gpu: one Soapy device against a MODEL of the reference over a random walk -- the lone device is what the stream group's tests
compare against, so it is itself held against the reference's semantics for SEQUENCES of calls over one byte stream, which the
scripted tests (one batch fed per call) cannot reach:

  * /dev/smi is a byte FIFO: a read() returns as many bytes as there are, up to the length asked for (caribou_smi.c:466-492,655), so
    where one read() ends decides where the next one looks for the sync pattern;
  * caribou_smi_read's chunk loop (caribou_smi.c:632-682: the oracle's orc_smi_read, run here over exactly the bytes that are
    pending at the call): re-syncs, the extrapolated sample, the slots it leaves untouched, "-3" with the failing read() consumed,
    short counts when the FIFO runs dry, ragged byte counts behind slipped batches;
  * Stream::ReadSamples (CaribouliteStream.cpp:282-367): CS16 goes straight into the client's buffer (only the slots the chunk loop
    writes change), every other format converts ALL n slots of the Stream's persistent intermediate buffer -- an overlay of every
    call so far -- and any error is 0 elements with the client's buffer untouched (:266-276).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NB, MTU, SENT = 524288, 131072, -21846


@pytest.fixture(scope="module")
def S():
    import torch
    from cariboulite_amd import hip, soapy
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return soapy


class RefDevice:
    """the reference's Stream over a byte FIFO (the model)"""

    def __init__(self, orc, ch, fmt):
        self.orc, self.ch, self.fmt = orc, ch, fmt
        self.fifo = np.zeros(0, np.uint8)
        self.interm = np.zeros((8 * MTU + 2, 2), np.int16)      # interm_native_buffer (the formats that convert); zero-initialised like ours
        self.max_read = 0                                       # a driver that hands out at most this many bytes per read() (0: no limit)

    def feed(self, b):
        self.fifo = np.concatenate([self.fifo, b])

    def pending(self):
        return self.fifo.size

    def flush(self):
        self.fifo = self.fifo[:0]

    def read(self, client, num):
        """Stream::ReadSamplesGen into `client` (a numpy buffer the caller keeps between calls); returns the element count"""
        n = num if self.fmt == "CS16" else min(num, MTU)
        if n == 0:
            return 0
        ret, iq, pos = self.orc.smi_read_pos(self.ch, self.fifo[: 4 * n], n, NB, max_read=self.max_read, fill=SENT)
        self.fifo = self.fifo[pos:]
        touched = (iq != SENT).any(axis=1)
        if self.fmt == "CS16":
            client[: n + 2][touched] = iq[touched]              # the chunk loop writes the client's buffer itself, failed calls included
            return ret if ret > 0 else 0
        self.interm[: n + 2][touched] = iq[touched]
        if ret <= 0:
            return 0
        conv = {"CF32": self.orc.cs16_to_cf32, "CF64": self.orc.cs16_to_cf64, "CS8": self.orc.cs16_to_cs8}[self.fmt]
        client[:ret] = conv(self.interm[:ret])                  # (:304-367: the count the chunk loop returned, stale slots included)
        return ret


@pytest.mark.parametrize("seed,fmt", [(1, "CS16"), (2, "CF32"), (3, "CS16"), (4, "CF32"), (5, "CS8"), (6, "CF64"), (7, "CF32"), (8, "CS16"),
                                      (9, "CS16"), (10, "CF32"), (11, "CS16"), (12, "CF32"), (13, "CS16"), (14, "CS8"), (15, "CF32"), (16, "CS16")])
def test_one_device_against_the_reference_model(S, orc, seed, fmt):
    from cariboulite_amd import synth
    rng = np.random.default_rng(100 + seed)
    ch = seed % 2
    zc = seed % 4 == 2                                             # (seeds 2, 6, 10, 14: ZEROCOPY=1 with the client's buffer registered -- the last kernel stores into it)
    dev = S.Device(dict(driver="Cariboulite", channel="S1G" if ch == 0 else "HiF"))
    st = dev.setupStream(S.SOAPY_SDR_RX, fmt, args={"ZEROCOPY": "1"} if zc else None)
    dev.activateStream(st)
    ref = RefDevice(orc, ch, fmt)
    dt = {"CS16": np.int16, "CF32": np.float32, "CF64": np.float64, "CS8": np.int8}[fmt]
    rows = 2 * MTU + 16
    got, want = np.zeros((rows, 2), dt), np.zeros((rows, 2), dt)        # the client's buffers persist across calls (CS16: untouched slots)
    if zc:
        dev.registerStreamBuffer(st, got)
    fed = 0
    stats = dict(resync=0, lost=0, short=0, ragged=0, multi=0)
    for step in range(60):
        while ref.pending() < int(rng.integers(0, 4)) * NB:
            how = rng.choice(["good"] * 10 + ["slip", "lost", "half", "quarter"])
            b = synth.smi_stream_bytes(MTU, ch, stream=300 + seed, n0=fed * MTU)[0].copy(); fed += 1
            if how == "slip":
                k = int(rng.integers(1, 9))
                b = np.concatenate([((np.arange(k, dtype=np.uint8) * 7 + 3) & 0x3F), b[: b.size - (k if rng.integers(0, 2) else 0)]])
                stats["resync"] += 1; stats["ragged"] += b.size % 4 != 0
            elif how == "lost":
                b[:] = 0; stats["lost"] += 1
            elif how == "half":
                b = b[: NB // 2]
            elif how == "quarter":
                b = b[: NB // 4]
            dev.feedSmiBytes(b); ref.feed(b)
        if rng.integers(0, 12) == 0:
            dev.flushSmiFifo(); ref.flush()
        if rng.integers(0, 8) == 0:                             # the driver's read() size changes: several read()s per call, ragged ones too
            ref.max_read = int(rng.choice([0, 0, NB // 2, 100000, 4098, 65536 + 2]))
            dev.setMaxRead(ref.max_read)
        choices = [MTU] * 6 + [MTU // 2, 1000, 4, MTU - 4]
        if fmt == "CS16":
            choices += [MTU + 4096, 2 * MTU]                    # CS16 is not clamped to the MTU: chunk loops of several read()s
        num = int(rng.choice(choices))
        r_dev = dev.readStream(st, [got], num).ret
        r_ref = ref.read(want, num)
        assert r_dev == r_ref, (step, num, r_dev, r_ref, dev.pendingSmiBytes(), ref.pending())
        assert dev.pendingSmiBytes() == ref.pending(), (step, num)
        if not np.array_equal(got, want):
            d = np.flatnonzero((got != want).any(axis=1))
            raise AssertionError(f"step {step} num {num} ret {r_dev}: {d.size} slots differ, first {d[0]} ({got[d[0]]} vs {want[d[0]]}), last {d[-1]}")
        stats["short"] += 0 < r_dev < min(num, MTU if fmt != "CS16" else num)
        stats["multi"] += num > MTU
    assert stats["resync"] + stats["lost"] >= 2, stats             # (the walk went where it is meant to go)
    if zc:
        assert dev.streamStats(st)["zero_copy_reads"] >= 10
    dev.close()


@pytest.mark.parametrize("seed,stages", [(21, "rs32"), (22, "fm"), (23, "rs32"), (24, "rs32"), (25, "fm"), (26, "rs32")])
def test_one_device_with_extension_stages_against_the_model(S, orc, seed, stages):
    """The same walk with the extension stages behind the read (FIR64 -> 3/2 polyphase, FIR64 -> FM demod):
    the model runs the oracle's float64 FIR / resampler / demodulator, state carried from call to call, over the n slots of the
    persistent buffer a successful call converts; a call that delivers nothing leaves the state alone.  Counts must be equal (the
    resampler's phase moves with every odd length), values within 1e-5 of the stage's peak (atan2 where the phasor is not small)."""
    from conftest import load_golden
    from cariboulite_amd import synth
    t = load_golden("taps.npz")
    rng = np.random.default_rng(seed)
    ch = seed % 2
    args = {"FIR": "64:1000000", "RESAMP": "3/2"} if stages == "rs32" else {"FIR": "64:100000", "DEMOD": "FM"}
    dev = S.Device(dict(driver="Cariboulite", channel="S1G" if ch == 0 else "HiF"))
    st = dev.setupStream(S.SOAPY_SDR_RX, "CF32", args=args)
    dev.activateStream(st)
    ref = RefDevice(orc, ch, "CF32")
    fir = orc.FIR(t["fir64_c2"] if stages == "rs32" else t["fir64_c3"])
    rs = orc.Resampler(t["rs_3_2"], 3, 2) if stages == "rs32" else None
    fm_prev = None
    n_rows = MTU * 3 // 2 + 16
    got = np.zeros((n_rows, 2) if stages == "rs32" else (n_rows,), np.float32)
    tmp = np.zeros((MTU + 16, 2), np.float32)
    fed, delivered = 0, 0
    for step in range(40):
        while ref.pending() < int(rng.integers(0, 4)) * NB:
            how = rng.choice(["good"] * 10 + ["slip", "lost", "half"])
            b = synth.smi_stream_bytes(MTU, ch, stream=400 + seed, n0=fed * MTU)[0].copy(); fed += 1
            if how == "slip":
                k = int(rng.integers(1, 9))
                b = np.concatenate([((np.arange(k, dtype=np.uint8) * 7 + 3) & 0x3F), b[: b.size - k]])
            elif how == "lost":
                b[:] = 0
            elif how == "half":
                b = b[: NB // 2]
            dev.feedSmiBytes(b); ref.feed(b)
        num = int(rng.choice([MTU] * 6 + [MTU // 2, 1000, 1001, 6, MTU - 3]))
        got[...] = np.nan
        r_dev = dev.readStream(st, [got], num).ret
        r = ref.read(tmp, num)                                      # CF32 of the persistent buffer's first r slots (0: nothing delivered)
        assert dev.pendingSmiBytes() == ref.pending(), (step, num)
        if r <= 0:
            assert r_dev == 0 and np.isnan(got).all(), (step, num, r_dev)
            continue
        z = fir.f64(tmp[:r])
        if stages == "rs32":
            want = rs.f64(z)
            assert r_dev == want.shape[0], (step, num, r, r_dev, want.shape)
            assert np.max(np.abs(got[:r_dev] - want)) <= 1e-5 * max(np.max(np.abs(want)), 1e-3), (step, num, r)
        else:
            want, fm_prev = orc.fm_demod_f64(z, fm_prev)
            assert r_dev == r, (step, num, r, r_dev)
            mag = np.hypot(z[:, 0], z[:, 1]); ok = mag > 0.02; ok[1:] &= ok[:-1]; ok[0] = False
            d = np.abs(got[:r] - want); d = np.minimum(d, 2 * np.pi - d)
            assert (not ok.any()) or np.max(d[ok]) <= 1e-4, (step, num, float(np.max(d[ok])))
        assert np.isnan(got[r_dev:]).all()
        delivered += 1
    assert delivered >= 20
    dev.close()


@pytest.mark.parametrize("seed", [31, 32, 33, 34, 35, 36])
def test_the_lower_seam_against_orc_smi_read(S, orc, seed):
    """caribou_smi_read itself (Binding B's seam, cariboulite_radio_read_samples is a pass-through: cariboulite_radio.c:1258-1285) with
    the metadata plane: fresh sentinel-filled buffers every call, so every slot the chunk loop writes or leaves alone shows --
    samples, the extrapolated one, meta[n] untouched behind a re-sync (caribou_smi.c:344-389) -- over lengths up to three MTUs,
    read() sizes that change, damage and flushes; the return code is the chunk loop's own (-3 included)."""
    from cariboulite_amd import synth
    rng = np.random.default_rng(seed)
    ch = seed % 2
    dev = S.Device(dict(driver="Cariboulite", channel="S1G" if ch == 0 else "HiF"))
    fifo = np.zeros(0, np.uint8)
    fed, max_read, codes = 0, 0, set()
    for step in range(60):
        while fifo.size < int(rng.integers(0, 5)) * NB:
            how = rng.choice(["good"] * 10 + ["slip", "lost", "half", "quarter"])
            b = synth.smi_stream_bytes(MTU, ch, stream=500 + seed, n0=fed * MTU)[0].copy(); fed += 1
            if how == "slip":
                k = int(rng.integers(1, 9))
                b = np.concatenate([((np.arange(k, dtype=np.uint8) * 7 + 3) & 0x3F), b[: b.size - (k if rng.integers(0, 2) else 0)]])
            elif how == "lost":
                b[:] = 0
            elif how == "half":
                b = b[: NB // 2]
            elif how == "quarter":
                b = b[: NB // 4]
            dev.feedSmiBytes(b); fifo = np.concatenate([fifo, b])
        if rng.integers(0, 8) == 0:
            max_read = int(rng.choice([0, 0, NB // 2, 100000, 4098]))
            dev.setMaxRead(max_read)
        if rng.integers(0, 15) == 0:
            dev.flushSmiFifo(); fifo = fifo[:0]
        n = int(rng.choice([MTU] * 4 + [MTU // 2, 1000, 5, 2 * MTU, 3 * MTU, MTU + 4096]))
        want_meta = bool(rng.integers(0, 2))
        ret, iq, meta = dev.smiRead(ch, n, want_meta=want_meta)
        r_ret, r_iq, r_meta = orc.smi_read(ch, fifo[: 4 * n], n, NB, max_read=max_read, want_meta=want_meta)
        _, _, pos = orc.smi_read_pos(ch, fifo[: 4 * n], n, NB, max_read=max_read)
        fifo = fifo[pos:]
        assert ret == r_ret, (step, n, ret, r_ret)
        assert dev.pendingSmiBytes() == fifo.size, (step, n)
        assert np.array_equal(iq, r_iq), (step, n, ret, np.flatnonzero((iq != r_iq).any(axis=1))[:4])
        if want_meta:
            assert np.array_equal(meta, r_meta), (step, n, ret, np.flatnonzero(meta != r_meta)[:4])
        codes.add(ret if ret < 0 else (0 if ret == 0 else 1))
    assert 1 in codes and -3 in codes, codes
    dev.close()


@pytest.mark.parametrize("seed,fmt", [(41, "CS16"), (42, "CF32"), (43, "CS16"), (44, "CF32")])
def test_the_reader_thread_and_its_ring_against_the_model(S, orc, seed, fmt):
    """ASYNC=1 (the reference's USE_ASYNC path, CaribouliteStream.cpp:16-49,70-75,260-279): a reader thread runs caribou_smi_read for one
    MTU into interm_native_buffer1 and puts what it got into a ring that overwrites its oldest elements when full; readStream pops
    whole requests.  Deterministic here because the walk feeds ONE piece (at most a native batch), waits until the reader thread has
    taken it (the ring holds what the model says), and only then goes on: every piece is one iteration of the reader's loop -- its
    re-syncs with interm_native_buffer1's stale slots, its "-3"s (nothing queued), short pieces -- and the ring (16 MTU: the next power
    of two above 10) is run over its capacity on purpose."""
    import time
    from cariboulite_amd import synth
    rng = np.random.default_rng(seed)
    ch = seed % 2
    dev = S.Device(dict(driver="Cariboulite", channel="S1G" if ch == 0 else "HiF"))
    st = dev.setupStream(S.SOAPY_SDR_RX, fmt, args={"ASYNC": "1"})
    dev.activateStream(st)
    ring = orc.Ring(10 * MTU, override_write=True, block_read=True)
    interm = np.zeros((MTU + 2, 2), np.int16)
    dt = np.int16 if fmt == "CS16" else np.float32
    got = np.zeros((MTU + 8, 2), dt)
    fed, overruns, reads = 0, 0, 0

    def wait_for(cond, what):
        t0 = time.time()
        while not cond():
            assert time.time() - t0 < 10, what
            time.sleep(0.0005)

    for step in range(70):
        # one piece for the reader thread
        how = rng.choice(["good"] * 8 + ["slip", "lost", "half", "quarter"])
        b = synth.smi_stream_bytes(MTU, ch, stream=600 + seed, n0=fed * MTU)[0].copy(); fed += 1
        if how == "slip":
            k = 4 * int(rng.integers(1, 3))                        # (whole words: the reader's next read() starts aligned again)
            b = np.concatenate([((np.arange(k, dtype=np.uint8) * 7 + 3) & 0x3F), b[: b.size - k]])
        elif how == "lost":
            b[:] = 0
        elif how == "half":
            b = b[: NB // 2]
        elif how == "quarter":
            b = b[: NB // 4]
        ret, iq, pos = orc.smi_read_pos(ch, b, MTU, NB, fill=SENT)
        assert pos == b.size
        touched = (iq != SENT).any(axis=1)
        interm[touched] = iq[touched]
        before = ring.size()
        if ret > 0:
            overruns += before + ret > ring.capacity() - 1
            ring.put(interm[:ret].copy().view(np.uint32).reshape(-1))
        dev.feedSmiBytes(b)
        wait_for(lambda: dev.pendingSmiBytes() == 0 and dev.streamQueueSize(st) == ring.size(), ("the reader thread", step, how, dev.streamQueueSize(st), ring.size()))
        # the client: now and then, whole requests only when the ring holds them
        while ring.size() >= MTU and rng.integers(0, 3) == 0:
            num = int(rng.choice([MTU, MTU // 2, 50000]))
            k, want = ring.get(num)
            assert k == num
            r = dev.readStream(st, [got], num, timeoutUs=1_000_000).ret
            assert r == num, (step, num, r)
            w16 = want.view(np.int16).reshape(-1, 2)
            exp = w16 if fmt == "CS16" else orc.cs16_to_cf32(w16)
            assert np.array_equal(got[:num], exp), (step, num, np.flatnonzero((got[:num] != exp).any(axis=1))[:4])
            assert dev.streamQueueSize(st) == ring.size()
            reads += 1
    assert overruns >= 1 and reads >= 5, (overruns, reads)
    dev.close()


@pytest.mark.parametrize("seed", [51, 52, 53])
def test_write_stream_with_the_modulator_against_the_oracle(S, orc, seed):
    """writeStream(CF32, MOD=FM:75000, RESAMP=2/3) over a walk of call lengths (whole MTUs, every residue mod 3, tiny, above the MTU:
    clamped): the oracle's fp64 chain -- phase accumulated from call to call, 2/3 polyphase with its history carried (SURVEY.md
    section 8 a13) -- quantised like Stream::WriteSamples (CaribouliteStream.cpp:199-214: (int16)(f * 4096)) and parsed back from the
    words the FPGA would see: every word within one LSB of the truncation boundary, the output count exact call by call."""
    from conftest import load_golden
    t = load_golden("taps.npz")
    rng = np.random.default_rng(seed)
    dev = S.Device(dict(driver="Cariboulite", channel="S1G"))
    st = dev.setupStream(S.SOAPY_SDR_TX, "CF32", args={"MOD": "FM:75000", "RESAMP": "2/3"})
    dev.activateStream(st)
    rs = orc.Resampler(t["rs_2_3"], 2, 3)
    phase, pos = 0.0, 0
    for step in range(25):
        num = int(rng.choice([MTU] * 3 + [MTU - 1, MTU - 2, 1000, 1001, 1002, 5, MTU + 3000]))
        took = min(num, MTU)
        msg = (0.4 * np.sin(2 * np.pi * 3e3 * (pos + np.arange(num)) / 4e6) + 0.3 * rng.standard_normal(num)).astype(np.float32)
        buf = np.stack([msg, rng.standard_normal(num).astype(np.float32)], 1)      # (the Q rail is ignored: "if given I/Q, use I")
        assert dev.writeStream(st, [buf], num).ret == took, (step, num)
        iq, phase = orc.fm_mod_f64(msg[:took], 75e3, 4e6, phase)
        want = rs.f64(iq) * 4096.0
        by = dev.drainSmiBytes()
        assert by.size == 4 * want.shape[0], (step, num, by.size, want.shape)
        if want.shape[0]:
            w = orc.fpga_tx_parse(by)
            _, got, _ = orc.rx_data_analyze(0, (w & ~np.uint32(1 << 16)).view(np.uint8))
            wq = ((np.trunc(want).astype(np.int64) + 4096) & 0x1FFF) - 4096
            dd = np.abs(got[: wq.shape[0]].astype(np.int64) - wq); dd = np.minimum(dd, 8192 - dd)
            assert dd.max() <= 1 and np.mean(dd != 0) < 2e-3, (step, num, int(dd.max()), float(np.mean(dd != 0)))
        pos += took
    dev.close()


@pytest.mark.parametrize("seed,fmt", [(61, "CF32"), (62, "CS16"), (63, "CF32"), (64, "CS16"), (65, "CS8"), (66, "CF32")])
def test_one_device_with_the_low_pass_against_the_model(S, orc, seed, fmt):
    """The walk with setBandwidth at random moments (Cariboulite.cpp:395-417: <= 20 / 50 / 100 kHz select one of three Butterworth-6
    filters, >= 160 kHz none).  Stream::ReadSamples(int16*) filters IN PLACE in the buffer it read into (CaribouliteStream.cpp:282-301)
    -- interm_native_buffer2 for the formats that convert, the client's own buffer for CS16 -- so the persistent buffer holds what
    was DELIVERED: the slots a later re-synchronised read() leaves untouched carry filtered samples, and go through the filter again.
    Each filter's state persists while another one (or none) is selected and is never reset (:84-91,127-141).  Compared up to the
    one-LSB crossings of the (int16) truncation (the oracle's and the kernel's fp64 sums differ in their last bits): never more than
    one LSB, a few in 10^4 samples."""
    from cariboulite_amd import synth
    rng = np.random.default_rng(seed)
    ch = seed % 2
    dev = S.Device(dict(driver="Cariboulite", channel="S1G" if ch == 0 else "HiF"))
    st = dev.setupStream(S.SOAPY_SDR_RX, fmt)
    dev.activateStream(st)
    ref = RefDevice(orc, ch, "CS16")                                # (the model below does the filtering and the conversion itself)
    filters = {bw: orc.IIR(6, 4e6, bw / 2) for bw in (20e3, 50e3, 100e3)}
    sel = None
    dt = {"CS16": np.int16, "CF32": np.float32, "CS8": np.int8}[fmt]
    conv = {"CS16": lambda v: v, "CF32": orc.cs16_to_cf32, "CS8": orc.cs16_to_cs8}[fmt]
    lsb = {"CS16": 1, "CF32": 1.0 / 4096, "CS8": 1}[fmt]
    rows = MTU + 16
    got, want = np.zeros((rows, 2), dt), np.zeros((rows, 2), dt)
    persist = np.zeros((rows, 2), np.int16)                        # what the reference's buffer holds: the int16 samples last delivered
    fed, filtered_calls, resynced_filtered = 0, 0, 0
    for step in range(60):
        while ref.pending() < int(rng.integers(0, 4)) * NB:
            how = rng.choice(["good"] * 9 + ["slip", "slip", "lost", "half"])
            b = synth.smi_stream_bytes(MTU, ch, stream=700 + seed, n0=fed * MTU)[0].copy(); fed += 1
            if how == "slip":
                k = int(rng.integers(1, 9))
                b = np.concatenate([((np.arange(k, dtype=np.uint8) * 7 + 3) & 0x3F), b[: b.size - k]])
            elif how == "lost":
                b[:] = 0
            elif how == "half":
                b = b[: NB // 2]
            dev.feedSmiBytes(b); ref.feed(b)
        if rng.integers(0, 5) == 0:
            bw = float(rng.choice([20e3, 50e3, 100e3, 1e6, 100e3]))
            dev.setBandwidth(S.SOAPY_SDR_RX, 0, bw)
            sel = bw if bw < 160e3 else None
        num = int(rng.choice([MTU] * 6 + [MTU // 2, 1000, MTU - 4]))
        r_dev = dev.readStream(st, [got], num).ret
        # the model: the chunk loop into the persistent buffer, then the filter over the returned count IN PLACE, then the conversion
        n = min(num, MTU)
        ret, iq, pos = orc.smi_read_pos(ch, ref.fifo[: 4 * n], n, NB, fill=SENT)
        ref.fifo = ref.fifo[pos:]
        touched = (iq != SENT).any(axis=1)
        persist[: n + 2][touched] = iq[touched]
        r_ref = ret if ret > 0 else 0
        if r_ref and sel:
            persist[:r_ref] = filters[sel].apply_cs16(persist[:r_ref])
            filtered_calls += 1; resynced_filtered += not touched[:r_ref].all()
        if r_ref:
            want[:r_ref] = conv(persist[:r_ref])
        assert r_dev == r_ref and dev.pendingSmiBytes() == ref.pending(), (step, num, r_dev, r_ref)
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        assert d.max() <= lsb * (1 + 1e-9) and np.mean(d != 0) < 1e-3, (step, num, sel, float(d.max()), int(np.count_nonzero(d)), np.flatnonzero(d.max(axis=1) > lsb * 1.000001)[:5])
        if fmt == "CS16":
            want[...] = got                                         # (the client's buffer IS the persistent one: what the device left there is what the next call finds)
            persist[:] = got[:rows]
    assert filtered_calls >= 8 and resynced_filtered >= 1, (filtered_calls, resynced_filtered)
    dev.close()
