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
