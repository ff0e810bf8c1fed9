"""-m gpu: the stream group at the Soapy boundary (cl_group_readStream, cariboulite_amd/csrc/host/cl_group.c).

The contract: one group call IS N single cl_readStream calls -- the reference's unit is one Soapy device per channel
(soapy_api/SoapyCariboulite.cpp:46-69), each call its own caribou_smi_read chunk loop (caribou_smi/caribou_smi.c:632-682,
soapy_api/CaribouliteStreamFunctions.cpp:239-254) -- so every stream of a group must deliver, call after call, bit for bit
what its own device delivers when it is read alone with the same bytes: outputs, return values, untouched slots of the
client's buffer, pending bytes, counters; slipped, lost and short batches in some streams must not be seen by the others.
The single-stream route is itself checked against the oracle elsewhere (test_gpu_soapy.py, test_gpu_sync_recovery.py); a few
streams are checked against the oracle here as well."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
NB = 524288
MTU = 131072
SENT = -21846


@pytest.fixture(scope="module")
def S():
    import torch
    from cariboulite_amd import hip, soapy
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return soapy


def batch_bytes(stream, call, ch):
    from cariboulite_amd import synth
    return synth.smi_stream_bytes(MTU, ch, stream=200 + stream, n0=call * MTU)[0].copy()


def slipped(b, k):
    junk = (np.arange(k, dtype=np.uint8) * 7 + 3) & 0x3F          # never looks like a sync word
    return np.concatenate([junk, b[: b.size - k]])


def make_devices(S, n, fmt, args, channel_of):
    devs, sts = [], []
    for i in range(n):
        d = S.Device(dict(driver="Cariboulite", channel=channel_of(i)))
        sts.append(d.setupStream(S.SOAPY_SDR_RX, fmt, args=args))
        d.activateStream(sts[-1])
        devs.append(d)
    return devs, sts


def sentinel_buffers(n, shape, dtype):
    fill = np.nan if np.issubdtype(dtype, np.floating) else SENT if dtype == np.int16 else -86
    return [np.full(shape, fill, dtype) for _ in range(n)]


def same(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()       # NaN sentinels included


def same_behind_a_filter(a, b, max_count=4):
    """Two filter objects that reach the same sample by different launch shapes (a multi-stream launch of the group, a single-stream
    one of the lone device: which tile looks back and which carries its state along differs) hold fp64 states an ulp apart, and
    (int16) truncation turns that into a one-LSB difference about once in 10^10 samples (the time-slice test):
    identical, or a handful of single-LSB differences."""
    if same(a, b):
        return True
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    lsb = 1.0 / 4096 if np.issubdtype(a.dtype, np.floating) else 1
    d = np.abs(np.nan_to_num(a.astype(np.float64)) - np.nan_to_num(b.astype(np.float64)))
    return d.max() <= lsb * (1 + 1e-9) and np.count_nonzero(d) <= max_count


def run_script(S, n, fmt, args, dtype, out_shape, script, n_calls, channel_of=lambda i: "S1G" if i % 3 else "HiF", group_args=None,
               num_elems=MTU, register=False, deep=False):
    """`script[(call, stream)]` = how that stream's batch of that call is damaged: ("slip", k) | ("lost",) | ("short", bytes) | ("none",).
    Runs the same byte streams through a group and through n lone devices; returns per call (group rets, single rets)."""
    gdevs, _ = make_devices(S, n, fmt, args, channel_of)
    sdevs, ssts = make_devices(S, n, fmt, args, channel_of)
    grp = S.Group(gdevs, group_args)
    gb = sentinel_buffers(n, out_shape, dtype)
    sb = sentinel_buffers(n, out_shape, dtype)
    if register:
        grp.registerBuffers(gb)
    log = []
    for c in range(n_calls):
        for i in range(n if not (deep and c) else 0):      # (deep: every call's bytes are fed before the first call -- the group reads and computes ahead)
            for cc in (range(n_calls) if deep else (c,)):
                feed_one(S, gdevs, sdevs, i, cc, channel_of, script)
        for x in gb + sb:
            x[...] = np.nan if np.issubdtype(dtype, np.floating) else SENT if dtype == np.int16 else -86
        delivered, rets = grp.readStream(gb, num_elems)
        srets = [sdevs[i].readStream(ssts[i], [sb[i]], num_elems).ret for i in range(n)]
        assert delivered == sum(r > 0 for r in rets)
        assert rets == srets, (c, rets, srets)
        for i in range(n):
            assert same(gb[i], sb[i]), f"call {c} stream {i} ({script.get((c, i), ('good',))}): group and lone device differ"
            assert gdevs[i].pendingSmiBytes() == sdevs[i].pendingSmiBytes(), (c, i)
        log.append((rets, [x.copy() for x in gb]))
    return finish_script(grp, gdevs, sdevs, n, log)


def feed_one(S, gdevs, sdevs, i, c, channel_of, script):
    ch = 0 if channel_of(i) == "S1G" else 1
    b = batch_bytes(i, c, ch)
    what = script.get((c, i), ("good",))
    if what[0] == "slip":
        b = slipped(b, what[1])
    elif what[0] == "lost":
        b[:] = 0
    elif what[0] == "short":
        b = b[: what[1]]
    elif what[0] == "none":
        b = b[:0]
    if b.size:
        gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)


def finish_script(grp, gdevs, sdevs, n, log):
    gstats = [gdevs[i].smiStats() for i in range(n)]
    sstats = [sdevs[i].smiStats() for i in range(n)]
    for i in range(n):
        for key in ("samples_read", "resyncs", "sync_losses"):
            assert gstats[i][key] == sstats[i][key], (i, key, gstats[i], sstats[i])
    st = grp.stats()
    grp.close()
    for d in gdevs + sdevs:
        d.close()
    return log, st


def test_32_streams_fir64_resample_3_2_with_slipped_lost_and_short_batches(S, orc):
    """BASELINE config 2's stages at the boundary, 32 streams (S1G and HiF mixed: two lanes), sub-batches of 4: two streams with a
    slipped and one with a lost batch in call 1, a short read in call 3, everything in sync otherwise -- each stream equal to
    its own lone device bit for bit, and stream 3 (slipped in call 1) / 9 (lost) / 0 against the oracle chain."""
    n, calls = 32, 5
    script = {(1, 3): ("slip", 3), (1, 17): ("slip", 6), (1, 9): ("lost",), (3, 5): ("short", NB // 2), (3, 21): ("none",)}
    args = {"FIR": "64:1000000", "RESAMP": "3/2"}
    log, st = run_script(S, n, S.SOAPY_SDR_CF32, args, np.float32, (MTU * 3 // 2 + 8, 2), script, calls)
    assert st["calls"] == calls and st["errors"] == 0
    # a stream that took the single route at call c with an odd number of samples behind it stays there; here: 5 damaged reads + what follows a short one
    assert st["single_reads"] >= 5 and st["batched_reads"] >= n * calls - 8
    assert st["launches"] <= calls * 9 + 6                 # 11 + 21 streams in sub-batches of 4: 3 + 6 launches per call, runs split around the damaged streams
    t = load_golden("taps.npz")
    for i in (0, 3, 9):
        ch = 0 if i % 3 else 1
        fir, rs = orc.FIR(t["fir64_c2"]), orc.Resampler(t["rs_3_2"], 3, 2)
        interm = np.zeros((MTU + 2, 2), np.int16)
        for c in range(calls):
            b = batch_bytes(i, c, ch)
            what = script.get((c, i), ("good",))
            if what[0] == "slip":
                b = slipped(b, what[1])
            elif what[0] == "lost":
                b[:] = 0
            ret, iq, _ = orc.smi_read(ch, b, MTU, NB, fill=SENT)
            rets, bufs = log[c]
            if ret < 0:
                assert rets[i] == 0 and np.isnan(bufs[i]).all()
                continue
            touched = (iq != SENT).any(axis=1)
            interm[touched] = iq[touched]
            want = rs.f64(fir.f64(orc.cs16_to_cf32(interm[:MTU])))
            assert rets[i] == want.shape[0]
            assert np.max(np.abs(bufs[i][: rets[i]] - want)) <= 1e-5 * np.max(np.abs(want)), (i, c)
            assert np.isnan(bufs[i][rets[i]:]).all()


@pytest.mark.parametrize("fmt,dtype,cols", [("CS16", np.int16, 2), ("CF32", np.float32, 2), ("CS8", np.int8, 2), ("CF64", np.float64, 2)])
def test_plain_formats_equal_lone_devices(S, orc, fmt, dtype, cols):
    """the reference's own formats (no extension stages): the unpack + conversion of a whole sub-batch is one launch; slipped,
    lost and short batches as above.  CS16 also against the oracle: slot for slot, sentinels where the reference writes nothing."""
    n, calls = 11, 4
    script = {(1, 2): ("slip", 5), (1, 7): ("lost",), (2, 4): ("short", NB - 4096), (2, 2): ("slip", 1), (3, 10): ("none",)}
    log, st = run_script(S, n, fmt, None, dtype, (MTU + 2, cols), script, calls, group_args={"SUBBATCH": "4", "COPY_THREADS": "2"})
    assert st["errors"] == 0 and st["batched_reads"] >= n * calls - 7
    if fmt == "CS16":
        for i in (0, 2, 7):
            ch = 0 if i % 3 else 1
            for c in range(calls):
                b = batch_bytes(i, c, ch)
                what = script.get((c, i), ("good",))
                if what[0] == "slip":
                    b = slipped(b, what[1])
                elif what[0] == "lost":
                    b[:] = 0
                ret, iq, _ = orc.smi_read(ch, b, MTU, NB, fill=SENT)
                rets, bufs = log[c]
                assert rets[i] == max(ret, 0)
                if ret > 0:
                    assert np.array_equal(bufs[i][:MTU], iq[:MTU]), (i, c)
                else:
                    assert (bufs[i] == SENT).all()


def test_fm_demod_lane_and_copy_on_the_calling_thread(S):
    """config 3's stages (FIR64 + FM demod, real fp32 out) through a group without copy threads, sub-batches of 3"""
    script = {(0, 1): ("slip", 2), (2, 0): ("lost",)}
    args = {"FIR": "64:100000", "DEMOD": "FM"}
    log, st = run_script(S, 7, S.SOAPY_SDR_CF32, args, np.float32, (MTU + 8,), script, 3, group_args={"SUBBATCH": "3", "COPY_THREADS": "0"})
    assert st["errors"] == 0 and st["batched_reads"] == 7 * 3 - 2 - 0


def test_registered_client_buffers_take_the_direct_route(S):
    """cl_group_register_buffers: a launch stores the sub-batch's rows into the clients' buffers itself -- same bytes as the mirror
    route, which a pointer outside the registered range still takes"""
    script = {(1, 2): ("slip", 3)}
    args = {"FIR": "64:1000000", "RESAMP": "3/2"}
    log, st = run_script(S, 6, S.SOAPY_SDR_CF32, args, np.float32, (MTU * 3 // 2 + 8, 2), script, 3, register=True)
    assert st["direct_reads"] == 6 * 3 - 1 and st["errors"] == 0


def test_a_member_with_the_iir_selected_is_read_by_its_own_device(S):
    """setBandwidth(< 160 kHz) on one member (Cariboulite.cpp:395-417): its reads take the single-stream route inside the group
    call, filter state and all, the others stay batched"""
    n, calls = 5, 3
    gdevs, _ = make_devices(S, n, S.SOAPY_SDR_CS16, None, lambda i: "S1G")
    sdevs, ssts = make_devices(S, n, S.SOAPY_SDR_CS16, None, lambda i: "S1G")
    gdevs[2].setBandwidth(S.SOAPY_SDR_RX, 0, 100e3); sdevs[2].setBandwidth(S.SOAPY_SDR_RX, 0, 100e3)
    grp = S.Group(gdevs)
    gb = sentinel_buffers(n, (MTU + 2, 2), np.int16); sb = sentinel_buffers(n, (MTU + 2, 2), np.int16)
    for c in range(calls):
        for i in range(n):
            b = batch_bytes(i, c, 0)
            gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)
        _, rets = grp.readStream(gb, MTU)
        srets = [sdevs[i].readStream(ssts[i], [sb[i]], MTU).ret for i in range(n)]
        assert rets == srets == [MTU] * n
        for i in range(n):
            assert (same_behind_a_filter if i == 2 else same)(gb[i], sb[i]), (c, i)
    st = grp.stats()
    assert st["single_reads"] == calls and st["batched_reads"] == (n - 1) * calls
    grp.close()
    for d in gdevs + sdevs:
        d.close()


def test_group_make_refuses_what_it_cannot_read(S):
    d0 = S.Device(dict(driver="Cariboulite", channel="S1G"))
    d0.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CS16)
    d1 = S.Device(dict(driver="Cariboulite", channel="S1G"))
    d1.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16)
    with pytest.raises(RuntimeError, match="other direction"):     # (a group reads or writes: test_gpu_group_tx.py)
        S.Group([d1, d0])
    with pytest.raises(RuntimeError, match="twice"):
        S.Group([d1, d1])
    g = S.Group([d1])
    buf = np.zeros((MTU, 2), np.int16)
    assert g.readStream([buf], MTU) == (0, [0])           # nothing pending: every member times out like the lone device
    assert g.readStream([buf], 0) == (0, [0])
    g.close(); d0.close(); d1.close()


def test_a_member_closed_first_takes_its_group_with_it(S):
    """(the Python face) closing a member while its group stands closes the group first: the group holds the members' seams"""
    devs, sts = make_devices(S, 3, S.SOAPY_SDR_CS16, None, lambda i: "S1G")
    grp = S.Group(devs)
    for d in devs:
        d.feedSmiBytes(batch_bytes(0, 0, 0))
    bufs = sentinel_buffers(3, (MTU + 2, 2), np.int16)
    assert grp.readStream(bufs, MTU)[0] == 3
    devs[1].close()
    assert grp.h is None
    assert devs[0].readStream(sts[0], [bufs[0]], MTU).ret == 0        # an ordinary device again (nothing pending)
    for d in devs:
        d.close()


def test_the_groups_slab_members_in_step_out_of_step_outgrown_and_released(S, orc):
    """The members' byte FIFOs live in the group's ONE pinned slab (a slice each): batches of members fed and read in step lie one
    stride apart and come in as one 2-D copy per sub-batch; a member that is out of step (it was fed an extra half batch once) comes
    in by a copy of its own from then on; a FIFO fed more than its slice holds moves into a buffer of its own without losing a byte;
    and when the group goes the FIFOs move out of the slab with what they hold -- the lone device reads it."""
    n = 8
    devs, sts = make_devices(S, n, S.SOAPY_SDR_CS16, None, lambda i: "S1G")
    grp = S.Group(devs, {"SLAB_MB": "2", "SUBBATCH": "4"})
    bufs = sentinel_buffers(n, (MTU + 2, 2), np.int16)
    streams = [np.concatenate([batch_bytes(i, c, 0) for c in range(12)]) for i in range(n)]
    want = [orc.smi_read(0, streams[i], 12 * MTU, NB)[1] for i in range(n)]
    pos = [0] * n                      # bytes of each stream fed so far
    got = [0] * n                      # samples of each stream read so far

    def feed(i, nbytes):
        devs[i].feedSmiBytes(streams[i][pos[i]: pos[i] + nbytes]); pos[i] += nbytes

    def read_all(expect):
        delivered, rets = grp.readStream(bufs, MTU)
        assert rets == expect, (rets, expect)
        for i, r in enumerate(rets):
            assert np.array_equal(bufs[i][:r], want[i][got[i]: got[i] + r]), i
            got[i] += r
    # in step: two sub-batches of four, each ONE 2-D copy
    for c in range(2):
        for i in range(n):
            feed(i, NB)
        read_all([MTU] * n)
    assert grp.stats()["copies_2d"] == 4
    # member 5 is fed half a batch more: its batches no longer lie on the others' stride
    for i in range(n):
        feed(i, NB + (NB // 2 if i == 5 else 0))
    read_all([MTU] * n)
    c2d = grp.stats()["copies_2d"]
    assert c2d == 4 + 2                                    # rows 0-3 as one; 4 | 5 | 6-7: the pair 6-7 still as one
    for i in range(n):
        feed(i, NB)
    read_all([MTU] * n)
    # member 2 is fed 3 MiB at once: more than its 2 MiB slice holds -> a buffer of its own, nothing lost, still read by the group
    feed(2, 6 * NB)
    for i in range(n):
        if i != 2:
            feed(i, NB)
    read_all([MTU] * n)
    for k in range(2):
        for i in range(n):
            if i != 2:
                feed(i, NB)
        read_all([MTU] * n)
    assert grp.stats()["errors"] == 0 and grp.stats()["single_reads"] == 0
    # the group goes; what the FIFOs hold moves out of the slab with them
    for i in range(n):
        if i != 2:
            feed(i, NB)
    pend = [d.pendingSmiBytes() for d in devs]
    grp.close()
    assert [d.pendingSmiBytes() for d in devs] == pend
    for i in range(n):
        r = devs[i].readStream(sts[i], [bufs[i]], MTU).ret
        assert r == MTU and np.array_equal(bufs[i][:MTU], want[i][got[i]: got[i] + MTU]), i
    for d in devs:
        d.close()


def test_feeder_threads_race_the_group(S, orc):
    """One feeder thread per member pushes the next batch the moment its FIFO runs dry while the group reads: a batch is staged under
    the member's FIFO lock and its copy in is queued before the lock is released, a feeder that has to compact or move a FIFO waits
    for the ingest stream first -- every stream must come out as the oracle's analysis of the bytes that were fed, in order."""
    import threading
    import time
    n, n_batches = 6, 24
    devs, sts = make_devices(S, n, S.SOAPY_SDR_CF32, None, lambda i: "S1G" if i % 2 else "HiF")
    grp = S.Group(devs, {"SLAB_MB": "1", "SUBBATCH": "4"})      # 1 MiB slices: two batches -- compaction and outgrowing happen
    data = [np.concatenate([batch_bytes(i, c, 0 if i % 2 else 1) for c in range(n_batches)]) for i in range(n)]
    stop = threading.Event()

    def feeder(i):
        k = 0
        while k < n_batches and not stop.is_set():
            if devs[i].pendingSmiBytes() <= (NB if i % 3 == 0 else 0):      # some keep a batch in reserve: FIFOs at different fill levels
                devs[i].feedSmiBytes(data[i][k * NB:(k + 1) * NB]); k += 1
            else:
                time.sleep(0)

    ths = [threading.Thread(target=feeder, args=(i,)) for i in range(n)]
    for t in ths:
        t.start()
    got = [[] for _ in range(n)]
    bufs = [np.zeros((MTU, 2), np.float32) for _ in range(n)]
    try:
        deadline = time.time() + 120
        while min(len(g_) for g_ in got) < n_batches and time.time() < deadline:
            _, rets = grp.readStream(bufs, MTU)
            for i, r in enumerate(rets):
                if r > 0:
                    assert r == MTU
                    got[i].append(bufs[i].copy())
    finally:
        stop.set()
        for t in ths:
            t.join(timeout=30)
    for i in range(n):
        assert len(got[i]) == n_batches, (i, len(got[i]))
        ch = 0 if i % 2 else 1
        want = orc.cs16_to_cf32(orc.smi_read(ch, data[i], n_batches * MTU, NB)[1][: n_batches * MTU])
        assert np.array_equal(np.concatenate(got[i]), want), i
    assert grp.stats()["errors"] == 0
    grp.close()
    for d in devs:
        d.close()


@pytest.mark.parametrize("readahead,staged", [("2", True), ("2", False), ("1", True), ("1", False), ("0", True)])
def test_read_ahead_between_calls_is_invisible(S, orc, readahead, staged):
    """READAHEAD=1: before a call waits for its results the group stages the members' NEXT batches in their FIFOs and copies them in;
    READAHEAD=2 (the default): ... and launches over them into its second mirror, so with FIFOs that hold several batches a call
    finds its results computed.  Whatever happens between two calls -- another numElems, a member read through its own device, a
    member flushed, a low-pass selected on one, client buffers registered and released, a batch ahead that is out of sync, the
    group closed with batches read ahead -- every stream delivers what its lone device delivers from the same bytes, and counts
    the same bytes pending after every step."""
    n, depth = 6, 14
    args = {"FIR": "64:1000000", "RESAMP": "3/2"} if staged else None
    full = MTU * 3 // 2 if staged else MTU
    gdevs, gsts = make_devices(S, n, S.SOAPY_SDR_CF32, args, lambda i: "S1G" if i % 2 else "HiF")
    sdevs, ssts = make_devices(S, n, S.SOAPY_SDR_CF32, args, lambda i: "S1G" if i % 2 else "HiF")
    grp = S.Group(gdevs, {"READAHEAD": readahead, "SUBBATCH": "2"})
    shape = (MTU * 3 // 2 + 8, 2)
    gb, sb = sentinel_buffers(n, shape, np.float32), sentinel_buffers(n, shape, np.float32)
    for i in range(n):
        ch = 0 if i % 2 else 1
        parts = [batch_bytes(i, c, ch) for c in range(depth)]
        if i == 4:
            parts[5] = slipped(parts[5], 6)                # a batch that will be looked at ahead of time and is not in sync
        if i == 1:
            parts[6][:] = 0                                # ... and one with no sync at all
        b = np.concatenate(parts)
        gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)

    def reset():
        for x in gb + sb:
            x[...] = np.nan

    def both(num):
        reset()
        _, rets = grp.readStream(gb, num)
        srets = [sdevs[i].readStream(ssts[i], [sb[i]], num).ret for i in range(n)]
        assert rets == srets, (rets, srets)
        check()
        return rets

    def check():
        for i in range(n):
            assert same(gb[i], sb[i]), i
            assert gdevs[i].pendingSmiBytes() == sdevs[i].pendingSmiBytes(), i

    assert both(MTU) == [full] * n                          # call 0: staged and copied in the call; batch 1 read ahead behind it
    assert both(MTU) == [full] * n                          # call 1: from the read-ahead
    st = grp.stats()
    assert st["ahead_reads"] == (n if readahead != "0" else 0), st
    if readahead == "2":
        assert st["launches"] == 3 * 4                          # call 0's own, and what calls 0 and 1 launched ahead: two lanes of 3 rows = 4 sub-batches
    # member 2 is read through its own device between two group calls: it gets batch 2 (which the group had read ahead), the group
    # batch 3.  (Plain formats only: the FIR's and the resampler's history of a member live in the group's pipe, not in its device's.)
    if not staged:
        reset()
        assert gdevs[2].readStream(gsts[2], [gb[2]], MTU).ret == sdevs[2].readStream(ssts[2], [sb[2]], MTU).ret == MTU
        check()
    assert both(MTU)[0] == full
    # another length: what was read ahead for MTU is pending again, in order
    both(MTU // 2); both(MTU // 2)
    # member 3 is flushed (the read-ahead goes with it), member 5 gets the low-pass (off the batched route from now on)
    assert gdevs[3].flushSmiFifo() == sdevs[3].flushSmiFifo() == 0
    assert gdevs[3].pendingSmiBytes() == sdevs[3].pendingSmiBytes() == 0
    gdevs[5].setBandwidth(S.SOAPY_SDR_RX, 0, 100e3); sdevs[5].setBandwidth(S.SOAPY_SDR_RX, 0, 100e3)
    for k in range(4):
        both(MTU)                                           # ... over member 4's slipped and member 1's lost batch
    # client buffers are registered while results computed ahead lie in the mirror, and released again
    grp.registerBuffers(gb)
    both(MTU); both(MTU)
    assert grp.stats()["direct_reads"] > 0
    grp.unregisterBuffers()
    both(MTU); both(MTU)
    assert grp.stats()["errors"] == 0
    if readahead != "0":
        assert grp.stats()["ahead_reads"] > st["ahead_reads"] + 2 * n
    # closed with batches read ahead: they are pending on the devices, which read on alone
    pend = [d.pendingSmiBytes() for d in gdevs]
    assert pend == [d.pendingSmiBytes() for d in sdevs] and max(pend) > 0
    grp.close()
    assert [d.pendingSmiBytes() for d in gdevs] == pend
    reset()
    for i in range(n):
        assert gdevs[i].readStream(gsts[i], [gb[i]], MTU).ret == sdevs[i].readStream(ssts[i], [sb[i]], MTU).ret
        assert gdevs[i].pendingSmiBytes() == sdevs[i].pendingSmiBytes(), i
    if not staged:
        check()
    for d in gdevs + sdevs:
        d.close()


@pytest.mark.parametrize("fmt,dtype,args,shape", [("CF32", np.float32, {"FIR": "64:1000000", "RESAMP": "3/2"}, (MTU * 3 // 2 + 8, 2)),
                                                   ("CS16", np.int16, None, (MTU + 2, 2)),
                                                   ("CF32", np.float32, {"FIR": "64:100000", "DEMOD": "FM"}, (MTU + 8,))])
def test_32_streams_with_every_batch_pending_beforehand(S, fmt, dtype, args, shape):
    """The FIFOs hold ALL the calls' bytes before the first call (a client that reads slower than its boards deliver): the group reads
    and computes every next batch ahead -- over slipped and lost batches, a short one and a missing one in the middle of the streams
    (behind which the members' batches are out of step with their neighbours') -- and still every stream equals its lone device call
    for call, in output, return value, pending bytes and counters."""
    n, calls = 32, 7
    script = {(1, 3): ("slip", 3), (2, 17): ("slip", 6), (2, 9): ("lost",), (3, 5): ("short", NB // 2), (3, 21): ("none",), (4, 30): ("slip", 9)}
    log, st = run_script(S, n, getattr(S, "SOAPY_SDR_" + fmt), args, dtype, shape, script, calls, deep=True)
    assert st["errors"] == 0 and st["ahead_reads"] >= (calls - 3) * (n - 6), st


@pytest.mark.parametrize("seed,staged,filters", [(1, True, 0), (2, True, 0), (3, False, 0), (4, False, 0), (5, True, 0), (6, False, 0), (7, True, 0), (8, False, 0), (9, False, 0),
                                                 (10, True, 0), (11, False, 1), (12, False, 1), (13, False, 1), (14, True, 1), (15, False, 2), (16, False, 2)])
def test_random_walk_against_lone_devices(S, seed, staged, filters):
    """A seeded random walk over everything a client can do between and in group calls -- feed whole / half / damaged / no batches to
    some members, ask for a whole or half MTU, read a member through its own device (plain lanes), flush one, switch a low-pass on or
    off, register / release buffers -- with eight streams whose FIFOs run several batches deep (work is made ahead and given up all
    the time).  After every step every stream equals its lone device: output, return value, pending bytes."""
    rng = np.random.default_rng(seed)
    n = 8
    args = {"FIR": "64:1000000", "RESAMP": "3/2"} if staged else None
    full = MTU * 3 // 2 if staged else MTU
    chan = lambda i: "S1G" if i % 2 else "HiF"
    cs16 = filters == 2                                    # (seeds 15, 16: the reference's own format -- calls above one MTU are not clamped, the client's buffer is the persistent one)
    fmt, dtype = (S.SOAPY_SDR_CS16, np.int16) if cs16 else (S.SOAPY_SDR_CF32, np.float32)
    gdevs, gsts = make_devices(S, n, fmt, args, chan)
    sdevs, ssts = make_devices(S, n, fmt, args, chan)
    grp = S.Group(gdevs, {"SUBBATCH": "2", "SLAB_MB": "4"})
    rows = 2 * MTU + 8 if cs16 else full + 8
    gb, sb = sentinel_buffers(n, (rows, 2), dtype), sentinel_buffers(n, (rows, 2), dtype)
    fed = [0] * n                                          # batches fed per stream so far

    def feed(i, how):
        ch = 0 if chan(i) == "S1G" else 1
        b = batch_bytes(i, fed[i], ch); fed[i] += 1
        hist[i].append(str(how))
        if how == "slip":
            b = slipped(b, int(rng.integers(1, 9)))
        elif how == "lost":
            b[:] = 0
        elif how == "half":
            b = b[: NB // 2]
        gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)

    hist = [[] for _ in range(n)]                          # what was fed to / done with every stream, for the failure message
    filtered = [False] * n                                 # a low-pass is selected on the stream (outputs compared up to a handful of LSBs)

    def compare(where):
        for i in range(n):
            # (behind the extension stages one differing input sample shows in every output its FIR window reaches)
            if not (same(gb[i], sb[i]) or (filtered[i] and same_behind_a_filter(gb[i], sb[i], 1200 if staged else 4))):
                a, b = gb[i].reshape(-1), sb[i].reshape(-1)
                d = np.flatnonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))
                raise AssertionError(f"{where} stream {i}: {d.size} values differ, first at {d[0]} ({a[d[0]]} vs {b[d[0]]}), last at {d[-1]}; "
                                     f"history {hist[i][-12:]}; rets so far {where}")
            assert gdevs[i].pendingSmiBytes() == sdevs[i].pendingSmiBytes(), (where, i)

    registered = False
    for step in range(36):
        for i in range(n):                                 # keep most FIFOs two to four batches deep
            while gdevs[i].pendingSmiBytes() < int(rng.integers(1, 4)) * NB:
                feed(i, rng.choice(["good"] * 12 + ["slip", "lost", "half"]))
        op = rng.choice(["call"] * 5 + ["half", "lone", "flush", "filter", "register", "maxread"] + ["filter_all"] * (2 * (filters == 1)))
        for x in gb + sb:
            x[...] = np.nan if not cs16 else SENT
        if op == "maxread":                                # one member's driver hands out shorter read()s from now on (or whole ones again)
            i = int(rng.integers(0, n)); m = int(rng.choice([0, NB // 2, 100000]))
            gdevs[i].setMaxRead(m); sdevs[i].setMaxRead(m); hist[i].append(f"maxread{m}")
        if op == "filter_all":                             # the same low-pass on everybody (lanes without stages: whole sub-batches through one filter launch), or off
            bw = float(rng.choice([100e3, 50e3, 1e6]))
            for i in range(n):
                gdevs[i].setBandwidth(S.SOAPY_SDR_RX, 0, bw); sdevs[i].setBandwidth(S.SOAPY_SDR_RX, 0, bw); hist[i].append(f"bw{bw:g}")
                filtered[i] = bw < 160e3
        if op == "lone" and not staged:
            i = int(rng.integers(0, n))
            assert gdevs[i].readStream(gsts[i], [gb[i]], MTU).ret == sdevs[i].readStream(ssts[i], [sb[i]], MTU).ret
            compare(("lone", step))
        elif op == "flush":
            i = int(rng.integers(0, n))
            assert gdevs[i].flushSmiFifo() == sdevs[i].flushSmiFifo() == 0
            hist[i].append("flush")
        elif op == "filter":
            i = int(rng.integers(0, n)); bw = float(rng.choice([100e3, 1e6]))
            gdevs[i].setBandwidth(S.SOAPY_SDR_RX, 0, bw); sdevs[i].setBandwidth(S.SOAPY_SDR_RX, 0, bw)
            hist[i].append(f"bw{bw:g}"); filtered[i] = bw < 160e3
        elif op == "register":
            if registered:
                grp.unregisterBuffers()
            else:
                grp.registerBuffers(gb)
            registered = not registered
        num = MTU // 2 if op == "half" else MTU
        if cs16 and rng.integers(0, 6) == 0:
            num = int(rng.choice([MTU + 4096, 2 * MTU]))    # a chunk loop of several read()s, member by member
        for x in gb + sb:
            x[...] = np.nan if not cs16 else SENT
        from cariboulite_amd import hip as _hip
        assert _hip.lib().clhip_debug_sticky_error() == 0, ("a handled HIP error was left pending before the group call", step, str(op))
        _, rets = grp.readStream(gb, num)
        assert _hip.lib().clhip_debug_sticky_error() == 0, ("the group call left a handled HIP error pending", step, str(op), num)
        srets = []
        for i in range(n):
            srets.append(sdevs[i].readStream(ssts[i], [sb[i]], num).ret)
            e = _hip.lib().clhip_debug_sticky_error()
            if e:
                print("DIAG lone read", step, str(op), num, i, srets, "sticky", e, "hip:", _hip.last_error(), "dev:", sdevs[i].lastError(), "hist", hist[i], flush=True)
            assert e == 0, ("a lone read left a handled HIP error pending", step, i)
        for i in range(n):
            hist[i].append(f"<{op}:{num}:{rets[i]}>")
        if rets != srets:
            from cariboulite_amd import hip
            bad = [i for i in range(n) if rets[i] != srets[i]]
            raise AssertionError((step, str(op), num, rets, srets, [(i, sdevs[i].lastError(), gdevs[i].lastError()) for i in bad], hip.last_error(), grp.lastError()))
        compare((str(op), step))
    st = grp.stats()
    assert st["errors"] == 0 and st["single_reads"] > 5 and (filters or st["ahead_reads"] > 40), st
    grp.close()
    for d in gdevs + sdevs:
        d.close()


@pytest.mark.parametrize("fmt,dtype,staged", [("CS16", np.int16, False), ("CF32", np.float32, False), ("CF32", np.float32, True)])
def test_the_low_pass_over_whole_sub_batches(S, orc, fmt, dtype, staged):
    """setBandwidth below 160 kHz on EVERY member (Cariboulite.cpp:395-417: the Butterworth-6 of CaribouliteStream.cpp:282-301): a
    sub-batch whose members all have the same filter selected goes through ONE multi-stream filter launch fed from the raw words --
    and every stream still equals its lone device bit for bit (the filter's fp64 state carried from call to call), through: another
    filter selected on everybody, one member switched off (its sub-batch falls apart: those members go through their own devices and
    take their filters' state with them), a member read through its own device between two calls, a damaged batch, the filter on
    again (the state comes back), the group closed (the state goes home: the lone devices read on in step)."""
    n = 12
    chan = lambda i: "S1G"
    args = {"FIR": "64:1000000", "RESAMP": "3/2"} if staged else None      # (staged: the filtered samples go on through the lane's pipe)
    full = MTU * 3 // 2 if staged else MTU
    gdevs, gsts = make_devices(S, n, getattr(S, "SOAPY_SDR_" + fmt), args, chan)
    sdevs, ssts = make_devices(S, n, getattr(S, "SOAPY_SDR_" + fmt), args, chan)
    def bw_all(bw, who=range(n)):
        for i in who:
            gdevs[i].setBandwidth(S.SOAPY_SDR_RX, 0, bw); sdevs[i].setBandwidth(S.SOAPY_SDR_RX, 0, bw)
    bw_all(100e3)
    grp = S.Group(gdevs, {"SUBBATCH": "4"})
    gb, sb = sentinel_buffers(n, (full + 8, 2), dtype), sentinel_buffers(n, (full + 8, 2), dtype)
    fed = [0]
    def step(script=None, expect_single=None):
        c = fed[0]; fed[0] += 1
        for i in range(n):
            b = batch_bytes(i, c, 0)
            if script and script.get(i) == "slip":
                b = slipped(b, 4)
            gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)
        s0 = grp.stats()["single_reads"]
        _, rets = grp.readStream(gb, MTU)
        srets = [sdevs[i].readStream(ssts[i], [sb[i]], MTU).ret for i in range(n)]
        assert rets == srets == [full] * n, (c, rets, srets)
        for i in range(n):
            assert same_behind_a_filter(gb[i], sb[i], 1200 if staged else 4), (c, i)
        if expect_single is not None:
            assert grp.stats()["single_reads"] - s0 == expect_single, (c, grp.stats())
    step(expect_single=0); step(expect_single=0)            # three filter launches per call, nobody through its own device
    assert grp.stats()["launches"] == 2 * 3 * (2 if staged else 1)       # (staged: a filter launch and a pipe launch per sub-batch)
    bw_all(20e3); step(expect_single=0); bw_all(100e3); step(expect_single=0)   # the 100 kHz filter continues from where it was (state is per filter)
    bw_all(1e6, [5]); step(expect_single=3)                 # member 5 without a filter: rows 4, 6, 7 go through their own devices, 5 batched alone
    step(expect_single=3)
    if not staged:                                          # (a staged member's FIR history lives in the group's pipe: not read on its own in between)
        for x in (gb[6], sb[6]):
            x[...] = 0
        assert gdevs[6].readStream(gsts[6], [gb[6]], MTU).ret == sdevs[6].readStream(ssts[6], [sb[6]], MTU).ret == 0   # (nothing pending: a read through the device itself)
        b = batch_bytes(6, 900, 0); gdevs[6].feedSmiBytes(b); sdevs[6].feedSmiBytes(b)
        assert gdevs[6].readStream(gsts[6], [gb[6]], MTU).ret == sdevs[6].readStream(ssts[6], [sb[6]], MTU).ret == MTU
        assert same_behind_a_filter(gb[6], sb[6])
    bw_all(100e3, [5]); step(expect_single=0)               # whole again: the states move back into the group's object
    step(script={9: "slip"}, expect_single=4)               # a slipped batch in member 9: its sub-batch goes home for this call
    step(expect_single=0)                                   # (the slipped batch was a whole read(): everybody is whole and in step again)
    grp.close()
    for c in range(2):
        for i in range(n):
            b = batch_bytes(i, 50 + c, 0)
            gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)
        for i in range(n):
            assert gdevs[i].readStream(gsts[i], [gb[i]], MTU).ret == sdevs[i].readStream(ssts[i], [sb[i]], MTU).ret
            if not staged:                                  # (the devices' own pipes start from rest; their FILTERS continue: the unstaged variants check that)
                assert same_behind_a_filter(gb[i], sb[i]), ("after the group", c, i)
    for d in gdevs + sdevs:
        d.close()


def test_a_group_filter_launch_that_gives_up_is_made_again(S):
    """the single-pass filter kernel's bounded polls (clhip_iir_set_poll_bound(-1): every poll gives up): the sub-batch's launch reports
    an overrun, its state is back where it was, the group filters the sub-batch again on the scan path inside the same call -- same
    samples as the lone devices, whose own objects go through the same repair; the members' iir_overruns counters move"""
    n = 4
    gdevs, gsts = make_devices(S, n, S.SOAPY_SDR_CS16, None, lambda i: "HiF")
    sdevs, ssts = make_devices(S, n, S.SOAPY_SDR_CS16, None, lambda i: "HiF")
    for d in gdevs + sdevs:
        d.setBandwidth(S.SOAPY_SDR_RX, 0, 50e3)
    grp = S.Group(gdevs)
    gb, sb = sentinel_buffers(n, (MTU + 2, 2), np.int16), sentinel_buffers(n, (MTU + 2, 2), np.int16)
    for c in range(3):
        if c == 1:
            grp.setIirPollBound(-1)
        for i in range(n):
            b = batch_bytes(i, c, 1)
            gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)
        _, rets = grp.readStream(gb, MTU)
        srets = [sdevs[i].readStream(ssts[i], [sb[i]], MTU).ret for i in range(n)]
        assert rets == srets == [MTU] * n
        for i in range(n):
            assert same_behind_a_filter(gb[i], sb[i]), (c, i)
    assert [gdevs[i].streamStats(gsts[i])["iir_overruns"] for i in range(n)] == [1] * n
    assert grp.stats()["single_reads"] == 0 and grp.stats()["errors"] == 0
    grp.close()
    for d in gdevs + sdevs:
        d.close()


def test_a_member_with_a_reader_thread_is_read_through_its_ring(S):
    """ASYNC=1 on one member (the reference's USE_ASYNC reader thread + ring, CaribouliteStream.cpp:16-49): inside the group call that
    member pops its ring like its lone twin, the others stay batched"""
    import time
    n = 4
    args_of = lambda i: {"ASYNC": "1"} if i == 2 else None
    def make():
        devs, sts = [], []
        for i in range(n):
            d = S.Device(dict(driver="Cariboulite", channel="S1G"))
            sts.append(d.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CS16, args=args_of(i)))
            d.activateStream(sts[-1])
            devs.append(d)
        return devs, sts
    gdevs, gsts = make()
    sdevs, ssts = make()
    grp = S.Group(gdevs)
    gb, sb = sentinel_buffers(n, (MTU + 2, 2), np.int16), sentinel_buffers(n, (MTU + 2, 2), np.int16)
    for c in range(4):
        for i in range(n):
            b = batch_bytes(i, c, 0)
            gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)
        t0 = time.time()
        while gdevs[2].streamQueueSize(gsts[2]) < MTU or sdevs[2].streamQueueSize(ssts[2]) < MTU:
            assert time.time() - t0 < 10
            time.sleep(0.001)
        _, rets = grp.readStream(gb, MTU, timeoutUs=1_000_000)
        srets = [sdevs[i].readStream(ssts[i], [sb[i]], MTU, timeoutUs=1_000_000).ret for i in range(n)]
        assert rets == srets == [MTU] * n, (c, rets, srets)
        for i in range(n):
            assert same(gb[i], sb[i]), (c, i)
    st = grp.stats()
    assert st["single_reads"] == 4 and st["batched_reads"] == 12 and st["errors"] == 0
    grp.close()
    for d in gdevs + sdevs:
        d.close()


@pytest.mark.parametrize("fmt,dtype,args,shape", [("CF32", np.float32, {"FIR": "64:1000000", "RESAMP": "3/2"}, (MTU * 3 // 2 + 8, 2)),
                                                   ("CS16", np.int16, None, (MTU + 2, 2))])
def test_results_made_ahead_for_registered_buffers_leave_by_the_route_the_call_names(S, fmt, dtype, args, shape):
    """With registered client buffers the NEXT call's results are computed ahead into device rows (no mirror to store to, no client
    pointer yet) and the call only scatters: into the registered buffers -- or, for a member the client hands ANOTHER buffer in that
    call, into the mirror and on by the last hop; a member whose slipped batch is found after its row was read ahead goes through its own
    device; a call of another length gives the work back.  Eleven members (sub-batches of 4 / 8 and a rest), six calls with every
    batch pending beforehand, against lone twins bit for bit; the untouched tails of the buffers included."""
    n, calls = 11, 6
    gdevs, _ = make_devices(S, n, fmt, args, lambda i: "S1G" if i % 3 else "HiF")
    sdevs, ssts = make_devices(S, n, fmt, args, lambda i: "S1G" if i % 3 else "HiF")
    grp = S.Group(gdevs)
    reg = sentinel_buffers(n, shape, dtype)
    grp.registerBuffers(reg)
    for c in range(calls):
        for i in range(n):
            b = batch_bytes(i, c, 0 if i % 3 else 1)
            if (c, i) == (3, 5):
                b = slipped(b, 6)
            gdevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)
    foreign = {2: (1, 7), 4: (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10)}      # call -> members that get an unregistered buffer in it
    for c in range(calls):
        num = MTU // 2 if c == 5 else MTU                      # (the last call asks for half a batch: what was made ahead for a whole one is given back)
        other = sentinel_buffers(n, shape, dtype)
        gb = [other[i] if i in foreign.get(c, ()) else reg[i] for i in range(n)]
        for b in gb:
            b[...] = np.nan if np.issubdtype(dtype, np.floating) else SENT
        sb = sentinel_buffers(n, shape, dtype)
        nd, rets = grp.readStream(gb, num)
        srets = [sdevs[i].readStream(ssts[i], [sb[i]], num).ret for i in range(n)]
        assert rets == srets, (c, rets, srets)
        for i in range(n):
            assert same(gb[i], sb[i]), (c, i)
    st = grp.stats()
    assert st["errors"] == 0 and st["direct_reads"] >= (calls - 1) * n - 2 - 11 - 2 and st["ahead_reads"] >= 3 * n
    grp.close()
    for d in gdevs + sdevs:
        d.close()
