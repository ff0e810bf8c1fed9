"""-m gpu: a stream group of TX devices (cl_group_writeStream, cariboulite_amd/csrc/host/cl_group.c).

The contract: one group call IS N single cl_writeStream calls (Stream::WriteSamplesGen, soapy_api/CaribouliteStream.cpp:199-258, over
caribou_smi_write, caribou_smi/caribou_smi.c:720-762; one Soapy device per channel, soapy_api/SoapyCariboulite.cpp:46-69): every
member's TX FIFO receives, call after call, byte for byte what its own device produces when it is written alone with the same
samples -- and what the oracle's conversion + caribou_smi_generate_data produce -- return values and counters included."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MTU = 131072


@pytest.fixture(scope="module")
def S():
    import torch
    from cariboulite_amd import hip, soapy
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return soapy


def make_tx(S, n, fmt, args_of=lambda i: None):
    devs, sts = [], []
    for i in range(n):
        d = S.Device(dict(driver="Cariboulite", channel="S1G" if i % 2 else "HiF"))
        sts.append(d.setupStream(S.SOAPY_SDR_TX, fmt, args=args_of(i)))
        d.activateStream(sts[-1])
        devs.append(d)
    return devs, sts


def samples(rng, fmt, n):
    if fmt == "CS16":
        return rng.integers(-4096, 4096, (n, 2)).astype(np.int16)
    if fmt == "CS8":
        return rng.integers(-128, 128, (n, 2)).astype(np.int8)
    x = (rng.random((n, 2)) - 0.5) * 1.9
    return x.astype(np.float32) if fmt == "CF32" else x


@pytest.mark.parametrize("fmt", ["CS16", "CF32", "CS8", "CF64"])
def test_group_write_equals_lone_devices_and_the_oracle(S, orc, fmt):
    """13 streams (two launches: 8 + 5), five calls -- a whole MTU, ragged lengths, a call above the MTU (clamped for every format but
    CS16, which the reference does not clamp: those members go one by one) -- without draining in between: each FIFO holds the
    calls' words one behind the other."""
    from cariboulite_amd import hip
    n = 13
    gdevs, gsts = make_tx(S, n, fmt)
    sdevs, ssts = make_tx(S, n, fmt)
    grp = S.Group(gdevs)
    rng = np.random.default_rng(11)
    to16 = {"CS16": lambda a: a, "CF32": orc.cf32_to_cs16, "CF64": orc.cf64_to_cs16, "CS8": orc.cs8_to_cs16}[fmt]
    want = [[] for _ in range(n)]
    for num in (MTU, 1000, MTU - 1, MTU + 4096, 4):
        bufs = [samples(rng, fmt, num) for _ in range(n)]
        nd, rets = grp.writeStream(bufs, num)
        srets = [sdevs[i].writeStream(ssts[i], [bufs[i]], num).ret for i in range(n)]
        assert rets == srets and nd == n, (num, rets, srets)
        took = num if fmt == "CS16" else min(num, MTU)
        assert rets == [took] * n
        for i in range(n):
            want[i].append(orc.generate_data(to16(bufs[i][:took]), hip.TX_DOCUMENTED))
    st = grp.stats()
    assert st["errors"] == 0 and st["single_reads"] == (n if fmt == "CS16" else 0) and st["batched_reads"] == n * (4 if fmt == "CS16" else 5)
    assert st["launches"] == 2 * (4 if fmt == "CS16" else 5)
    for i in range(n):
        g, s = gdevs[i].drainSmiBytes(), sdevs[i].drainSmiBytes()
        assert g.tobytes() == s.tobytes(), i
        assert g.tobytes() == np.concatenate(want[i]).tobytes(), i
        gs, ss = gdevs[i].streamStats(gsts[i]), sdevs[i].streamStats(ssts[i])
        for key in ("write_calls", "elements_written", "writes_empty"):
            assert gs[key] == ss[key], (i, key, gs, ss)
        assert gdevs[i].smiStats()["samples_written"] == sdevs[i].smiStats()["samples_written"]
    grp.close()
    for d in gdevs + sdevs:
        d.close()


def test_members_with_a_modulator_or_another_pack_mode_take_their_own_device(S, orc):
    """CF32 x 6: member 2 has MOD=FM + RESAMP (its own pipe: phase and resampler history persist across calls), member 4 packs "as
    written" (caribou_smi.c:700-701) while the others pack the documented layout -- both are written through their own devices inside
    the call, the rest in one launch; all equal their lone twins over three calls."""
    from cariboulite_amd import hip
    n = 6
    args_of = lambda i: {"MOD": "FM:75000", "RESAMP": "2/3"} if i == 2 else None
    gdevs, _ = make_tx(S, n, "CF32", args_of)
    sdevs, ssts = make_tx(S, n, "CF32", args_of)
    for d in (gdevs[4], sdevs[4]):
        d.setTxMode(hip.TX_AS_WRITTEN)
    grp = S.Group(gdevs)
    rng = np.random.default_rng(12)
    for call in range(3):
        bufs = [(samples(rng, "CF32", MTU) * 0.5).astype(np.float32) for _ in range(n)]
        nd, rets = grp.writeStream(bufs, MTU)
        srets = [sdevs[i].writeStream(ssts[i], [bufs[i]], MTU).ret for i in range(n)]
        assert rets == srets == [MTU] * n and nd == n
        for i in range(n):
            g, s = gdevs[i].drainSmiBytes(), sdevs[i].drainSmiBytes()
            assert g.size == s.size and g.tobytes() == s.tobytes(), (call, i)
            if i not in (2, 4):
                assert g.tobytes() == orc.generate_data(orc.cf32_to_cs16(bufs[i]), hip.TX_DOCUMENTED).tobytes(), (call, i)
        assert gdevs[4].drainSmiBytes().size == 0
    st = grp.stats()
    assert st["single_reads"] == 3 * 2 and st["batched_reads"] == 3 * 4 and st["errors"] == 0
    grp.close()
    for d in gdevs + sdevs:
        d.close()


def test_a_group_reads_or_writes(S):
    rx = S.Device(dict(driver="Cariboulite", channel="S1G")); rx.activateStream(rx.setupStream(S.SOAPY_SDR_RX, "CS16"))
    tx = S.Device(dict(driver="Cariboulite", channel="HiF")); tx.activateStream(tx.setupStream(S.SOAPY_SDR_TX, "CS16"))
    with pytest.raises(RuntimeError, match="other direction"):
        S.Group([rx, tx])
    g = S.Group([tx])
    buf = np.zeros((MTU, 2), np.int16)
    assert g.readStream([buf], MTU)[0] == -1 and "TX" in g.lastError()
    assert g.writeStream([buf], 0) == (0, [0])
    g.close()
    g = S.Group([rx])
    assert g.writeStream([buf], MTU)[0] == -1 and "RX" in g.lastError()
    g.close(); rx.close(); tx.close()


@pytest.mark.parametrize("seed,fmt", [(1, "CS16"), (2, "CF32"), (3, "CS8"), (4, "CF64"), (5, "CF32"), (6, "CS16")])
def test_random_walk_of_writes_against_lone_devices_and_the_oracle(S, orc, seed, fmt):
    """A seeded walk: nine TX members, every call another length (whole MTUs, ragged, tiny, above the MTU), now and then a member
    written through its own device between two group calls, a pack mode switched, the FIFOs drained at random moments or left to
    pile up -- after every drain every member's bytes equal its lone twin's and the oracle's conversion + pack of what was written."""
    from cariboulite_amd import hip
    rng = np.random.default_rng(700 + seed)
    n = 9
    gdevs, gsts = make_tx(S, n, fmt)
    sdevs, ssts = make_tx(S, n, fmt)
    grp = S.Group(gdevs)
    to16 = {"CS16": lambda a: a, "CF32": orc.cf32_to_cs16, "CF64": orc.cf64_to_cs16, "CS8": orc.cs8_to_cs16}[fmt]
    mode = [hip.TX_DOCUMENTED] * n
    want = [[] for _ in range(n)]
    for step in range(30):
        op = rng.choice(["call"] * 6 + ["lone", "mode", "drain"])
        if op == "lone":
            i = int(rng.integers(0, n)); num = int(rng.choice([MTU, 777]))
            b = samples(rng, fmt, num)
            assert gdevs[i].writeStream(gsts[i], [b], num).ret == sdevs[i].writeStream(ssts[i], [b], num).ret == num
            want[i].append(orc.generate_data(to16(b), mode[i]))
        elif op == "mode":
            i = int(rng.integers(0, n)); mode[i] = hip.TX_AS_WRITTEN if mode[i] == hip.TX_DOCUMENTED else hip.TX_DOCUMENTED
            gdevs[i].setTxMode(mode[i]); sdevs[i].setTxMode(mode[i])
        elif op == "drain":
            for i in range(n):
                g, s = gdevs[i].drainSmiBytes(), sdevs[i].drainSmiBytes()
                w = np.concatenate(want[i]) if want[i] else np.zeros(0, np.uint8)
                assert g.tobytes() == s.tobytes() == w.tobytes(), (step, i, g.size, s.size, w.size)
                want[i] = []
        num = int(rng.choice([MTU] * 4 + [MTU - 1, 1000, 3, MTU + 5000]))
        bufs = [samples(rng, fmt, num) for _ in range(n)]
        nd, rets = grp.writeStream(bufs, num)
        srets = [sdevs[i].writeStream(ssts[i], [bufs[i]], num).ret for i in range(n)]
        took = num if fmt == "CS16" else min(num, MTU)
        assert rets == srets == [took] * n and nd == n, (step, num, rets, srets)
        for i in range(n):
            want[i].append(orc.generate_data(to16(bufs[i][:took]), mode[i]))
    for i in range(n):
        g, s = gdevs[i].drainSmiBytes(), sdevs[i].drainSmiBytes()
        assert g.tobytes() == s.tobytes() == np.concatenate(want[i]).tobytes(), i
    st = grp.stats()
    assert st["errors"] == 0 and st["batched_reads"] > 50
    grp.close()
    for d in gdevs + sdevs:
        d.close()
