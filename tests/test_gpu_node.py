"""-m gpu: cl_node -- ONE call over the stream groups of several GPUs (cariboulite_amd/csrc/host/cl_node.c).

The reference's unit is one SoapySDR device per channel (soapy_api/SoapyCariboulite.cpp:46-69); independent streams shard over the GPUs
of a node with no data-path collective.  A node sorts its devices into one cl_group per GPU and runs the groups'
calls at once, a thread each.  On this one-GPU box the several-groups-at-once shape is rehearsed with SHARDS=<k> (k groups on GPU 0, k - 1
worker threads); the contract is the group's: every member delivers, call after call, bit for bit what its own device delivers when
it is read (written) alone with the same bytes (samples), return values included, in the order the devices were given."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NB = 524288
MTU = 131072


@pytest.fixture(scope="module")
def S():
    import torch
    from cariboulite_amd import hip, soapy
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return soapy


def rx_devices(S, cfg):
    devs, sts = [], []
    for i, (fmt, args) in enumerate(cfg):
        d = S.Device(dict(driver="Cariboulite", channel="S1G" if i % 2 else "HiF", gpu="0"))
        sts.append(d.setupStream(S.SOAPY_SDR_RX, fmt, args=args))
        d.activateStream(sts[-1])
        devs.append(d)
    return devs, sts


def test_node_read_equals_lone_devices(S):
    """14 members in three shards (5 + 5 + 4): FIR64 + 3/2 lanes and plain CS16 lanes in every shard, five calls; member 3's second batch
    arrives six bytes late (re-sync inside its shard's call), member 9's third batch carries no sync word (its call returns 0 and the
    batch is gone), member 12 has half a batch pending in the last call -- the others, in the same and in the other shards, see nothing
    of it."""
    from cariboulite_amd import synth
    pipe = {"FIR": "64:1000000", "RESAMP": "3/2"}
    cfg = [("CF32", pipe) if i % 5 < 3 else ("CS16", None) for i in range(14)]
    n = len(cfg)
    ndevs, _ = rx_devices(S, cfg)
    sdevs, ssts = rx_devices(S, cfg)
    node = S.Node(ndevs, {"SHARDS": "3"})
    assert node.shards() == 3 and [node.shardOf(i) for i in range(n)] == [0] * 5 + [1] * 5 + [2] * 4
    shape = lambda i: (MTU * 3 // 2 + 8, 2)
    held = [np.full(shape(i), np.nan, np.float32) if cfg[i][0] == "CF32" else np.full((MTU * 3 // 2 + 8, 2), -21846, np.int16) for i in range(n)]
    for call in range(5):
        if call == 2:
            node.registerBuffers(held)                     # (cl_node_register_buffers: every shard's members write their buffers directly from now on)
        for i in range(n):
            b = synth.smi_stream_bytes(MTU, 0 if i % 2 else 1, stream=300 + i, n0=call * MTU)[0].copy()
            if (call, i) == (1, 3):
                b = np.concatenate([np.full(6, 0x11, np.uint8), b[:-6]])
            if (call, i) == (2, 9):
                b[:] = 0
            if (call, i) == (4, 12):
                b = b[: NB // 2]
            ndevs[i].feedSmiBytes(b); sdevs[i].feedSmiBytes(b)
        nb = [np.full(shape(i), np.nan, np.float32) if cfg[i][0] == "CF32" else np.full((MTU + 8, 2), -21846, np.int16) for i in range(n)]
        sb = [x.copy() for x in nb]
        if call >= 2:                                          # the registered buffers (the int16 ones are longer than a call needs: only its slots compare)
            for i in range(n):
                held[i][...] = np.nan if cfg[i][0] == "CF32" else -21846
            nb = [held[i] if cfg[i][0] == "CF32" else held[i][: MTU + 8] for i in range(n)]
        nd, rets = node.readStream(nb, MTU)
        srets = [sdevs[i].readStream(ssts[i], [sb[i]], MTU, timeoutUs=1000).ret for i in range(n)]
        assert rets == srets, (call, rets, srets)
        assert nd == sum(r > 0 for r in srets)
        for i in range(n):
            assert nb[i].tobytes() == sb[i].tobytes(), (call, i)
    st = node.stats()
    assert st["errors"] == 0 and st["batched_reads"] >= 5 * n - 6 and st["calls"] == 15 and st["direct_reads"] >= 3 * n - 6
    node.unregisterBuffers()
    node.close()
    for d in ndevs + sdevs:
        d.close()


def test_node_write_equals_lone_devices_modulators_included(S):
    """A node of TX devices in two shards: six plain CF32 members and eight with MOD=FM + RESAMP=2/3 (a modulator lane in each shard),
    four calls of whole and ragged lengths, the words piling up in the FIFOs: byte for byte the lone twins'."""
    mod = {"MOD": "FM:75000", "RESAMP": "2/3"}
    cfg = [mod if i % 7 < 4 else None for i in range(14)]
    n = len(cfg)

    def tx_devices():
        devs, sts = [], []
        for i in range(n):
            d = S.Device(dict(driver="Cariboulite", channel="S1G" if i % 2 else "HiF", gpu="0"))
            sts.append(d.setupStream(S.SOAPY_SDR_TX, "CF32", args=cfg[i]))
            d.activateStream(sts[-1])
            devs.append(d)
        return devs, sts

    ndevs, _ = tx_devices()
    sdevs, ssts = tx_devices()
    node = S.Node(ndevs, {"SHARDS": "2"})
    assert node.shards() == 2
    rng = np.random.default_rng(41)
    for num in (MTU, 4097, MTU - 1, MTU):
        bufs = [((rng.random((num, 2)) - 0.5) * 0.9).astype(np.float32) for _ in range(n)]
        nd, rets = node.writeStream(bufs, num)
        srets = [sdevs[i].writeStream(ssts[i], [bufs[i]], num).ret for i in range(n)]
        assert rets == srets == [num] * n and nd == n
    assert node.flush() == 0
    for i in range(n):
        g, s = ndevs[i].drainSmiBytes(), sdevs[i].drainSmiBytes()
        assert g.size == s.size and g.tobytes() == s.tobytes(), i
    st = node.stats()
    assert st["errors"] == 0 and st["single_reads"] == 0 and st["batched_reads"] == 4 * n
    node.close()
    for d in ndevs + sdevs:
        d.close()


def test_a_node_reads_or_writes_and_reports_its_shards_errors(S):
    rx = S.Device(dict(driver="Cariboulite", channel="S1G")); rx.activateStream(rx.setupStream(S.SOAPY_SDR_RX, "CS16"))
    rx2 = S.Device(dict(driver="Cariboulite", channel="HiF")); rx2.activateStream(rx2.setupStream(S.SOAPY_SDR_RX, "CS16"))
    tx = S.Device(dict(driver="Cariboulite", channel="HiF")); tx.activateStream(tx.setupStream(S.SOAPY_SDR_TX, "CS16"))
    with pytest.raises(RuntimeError, match="other direction"):
        S.Node([rx, tx])
    with pytest.raises(RuntimeError, match="no devices"):
        S.Node([])
    node = S.Node([rx, rx2], {"SHARDS": "8"})                # (more shards than members: one each)
    assert node.shards() == 2
    buf = [np.zeros((MTU, 2), np.int16), np.zeros((MTU, 2), np.int16)]
    assert node.readStream(buf, MTU, timeoutUs=1000) == (0, [0, 0])       # nothing pending: N empty reads
    assert node.writeStream(buf, MTU)[0] == -1 and "RX" in node.lastError() and "shard 0" in node.lastError()
    node.close(); rx.close(); rx2.close(); tx.close()


def test_config4_256_streams_fir128_resample_5_4_in_eight_shards(S, orc):
    """BASELINE config 4 at the drop-in boundary: 256 independent 4 MS/s streams, 128-tap FIR + 5/4 polyphase, sharded eight ways with
    nothing exchanged between the shards -- here the eight shards are eight groups of 32 on GPU 0 (SHARDS=8; on a node `gpus="0,..,7"`
    puts them on eight GPUs, the calls are the same).  Three calls of one MTU per stream; member 40's second batch arrives five bytes
    late, member 200's second batch carries no sync word, member 255 has half a batch pending in the last call.  A member of every shard
    and every damaged one equal their own lone devices bit for bit, call after call (the twins are made one at a time afterwards and
    fed the same bytes); three of them also equal the oracle chain within 1e-5."""
    from cariboulite_amd import synth
    from conftest import load_golden
    n, calls, shards = 256, 3, 8
    args = {"FIR": "128:1200000", "RESAMP": "5/4"}
    n_full = MTU * 5 // 4
    ch_of = lambda i: 0 if i % 2 else 1

    def batch(i, c):
        b = synth.smi_stream_bytes(MTU, ch_of(i), stream=1000 + i, n0=c * MTU)[0].copy()
        if (c, i) == (1, 40):
            b = np.concatenate([np.full(5, 0x11, np.uint8), b[:-5]])
        if (c, i) == (1, 200):
            b[:] = 0
        if (c, i) == (2, 255):
            b = b[: NB // 2]
        return b

    ndevs, _ = rx_devices(S, [("CF32", args)] * n)
    node = S.Node(ndevs, {"SHARDS": str(shards)})
    assert node.shards() == shards and [node.shardOf(i) for i in (0, 31, 32, 255)] == [0, 0, 1, 7]
    watched = sorted(set(range(3, n, 32)) | {40, 200, 255, 0})
    bufs = [np.empty((n_full + 8, 2), np.float32) for _ in range(n)]
    log = []
    for c in range(calls):
        for i in range(n):
            ndevs[i].feedSmiBytes(batch(i, c))
        for x in bufs:
            x[...] = np.nan
        nd, rets = node.readStream(bufs, MTU)
        assert nd == sum(r > 0 for r in rets)
        assert all(r == n_full for i, r in enumerate(rets) if (c, i) not in ((1, 40), (1, 200), (2, 255))), (c, rets)
        log.append((rets, {i: bufs[i].copy() for i in watched}))
    st = node.stats()
    assert st["errors"] == 0 and st["calls"] == calls * shards and st["batched_reads"] >= n * calls - 6
    node.close()
    for d in ndevs:
        d.close()

    for i in watched:                                          # the lone twins, one at a time, member i's channel
        d = S.Device(dict(driver="Cariboulite", channel="S1G" if i % 2 else "HiF", gpu="0"))
        s = d.setupStream(S.SOAPY_SDR_RX, "CF32", args=args)
        d.activateStream(s)
        for c in range(calls):
            d.feedSmiBytes(batch(i, c))
            sb = np.full((n_full + 8, 2), np.nan, np.float32)
            r = d.readStream(s, [sb], MTU, timeoutUs=1000).ret
            assert r == log[c][0][i], (c, i, r, log[c][0][i])
            assert sb.tobytes() == log[c][1][i].tobytes(), (c, i)
        d.close()

    t = load_golden("taps.npz")
    for i in (0, 40, 200):                                     # the oracle chain: smi_read -> persistent buffer -> /4096 -> FIR128 -> 5/4
        fir, rs = orc.FIR(t["fir128_c4"]), orc.Resampler(t["rs_5_4"], 5, 4)
        interm = np.zeros((MTU + 2, 2), np.int16)
        for c in range(calls):
            ret, iq, _ = orc.smi_read(ch_of(i), batch(i, c), MTU, NB, fill=-21846)
            rets, kept = log[c]
            if ret < 0:
                assert rets[i] == 0 and np.isnan(kept[i]).all()
                continue
            touched = (iq != -21846).any(axis=1)
            interm[touched] = iq[touched]
            want = rs.f64(fir.f64(orc.cs16_to_cf32(interm[:MTU])))
            assert rets[i] == want.shape[0]
            assert np.max(np.abs(kept[i][: rets[i]] - want)) <= 1e-5 * np.max(np.abs(want)), (i, c)
            assert np.isnan(kept[i][rets[i]:]).all()
