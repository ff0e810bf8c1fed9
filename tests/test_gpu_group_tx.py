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


MOD_CASES = {"fm_2_3": {"MOD": "FM:75000", "RESAMP": "2/3"}, "rs_2_3": {"RESAMP": "2/3"}, "fm": {"MOD": "FM:25000"}, "rs_3_2": {"RESAMP": "3/2"}}


def drain_equal(gdevs, sdevs, tag):
    for i, (gd, sd) in enumerate(zip(gdevs, sdevs)):
        g, s = gd.drainSmiBytes(), sd.drainSmiBytes()
        assert g.size == s.size and g.tobytes() == s.tobytes(), (tag, i, g.size, s.size)


@pytest.mark.parametrize("case", list(MOD_CASES))
def test_modulator_lanes_equal_lone_devices(S, case):
    """11 members with one modulator configuration (two launches: 8 + 3 streams through the group's own multi-stream TX pipes), six
    calls -- whole MTUs, ragged lengths, a call above the MTU (clamped), a tiny one: the modulator's phase, the resampler's history
    and the polyphase position carry from call to call inside the group's pipes, and every member's FIFO holds, call after call,
    byte for byte what its lone twin produces (which tests/test_gpu_soapy.py and test_gpu_walk_oracle.py hold against the oracle)."""
    n = 11
    args = MOD_CASES[case]
    gdevs, gsts = make_tx(S, n, "CF32", lambda i: args)
    sdevs, ssts = make_tx(S, n, "CF32", lambda i: args)
    grp = S.Group(gdevs)
    rng = np.random.default_rng(21)
    calls = (MTU, 1000, MTU - 1, MTU + 4096, 4, MTU)
    for k, num in enumerate(calls):
        bufs = [(samples(rng, "CF32", num) * 0.5).astype(np.float32) for _ in range(n)]
        nd, rets = grp.writeStream(bufs, num)
        srets = [sdevs[i].writeStream(ssts[i], [bufs[i]], num).ret for i in range(n)]
        assert rets == srets == [min(num, MTU)] * n and nd == n, (num, rets, srets)
        if k % 2:
            drain_equal(gdevs, sdevs, (case, k))                      # (the other calls' words pile up behind one another)
    drain_equal(gdevs, sdevs, (case, "end"))
    st = grp.stats()
    assert st["errors"] == 0 and st["single_reads"] == 0 and st["batched_reads"] == n * len(calls) and st["launches"] == 2 * len(calls), st
    for i in range(n):
        gs, ss = gdevs[i].streamStats(gsts[i]), sdevs[i].streamStats(ssts[i])
        for key in ("write_calls", "elements_written", "writes_empty", "tx_overruns"):
            assert gs[key] == ss[key], (i, key, gs, ss)
    # the group goes: the state is the streams' own again, and they carry on as their twins do
    grp.close()
    bufs = [(samples(rng, "CF32", 5000) * 0.5).astype(np.float32) for _ in range(n)]
    for i in range(n):
        assert gdevs[i].writeStream(gsts[i], [bufs[i]], 5000).ret == sdevs[i].writeStream(ssts[i], [bufs[i]], 5000).ret == 5000
    drain_equal(gdevs, sdevs, (case, "after the group"))
    for d in gdevs + sdevs:
        d.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_walk_of_modulated_writes_against_lone_devices(S, seed):
    """A seeded walk over two modulator lanes in one group (FM + 2/3 x 10, 2/3 alone x 3) and two plain CF32 members: every call another
    length; now and then a member written through its own device between two group calls -- with the length its mates were written
    (it rejoins its sub-batch: the state moves home and back) or with another one (its polyphase position differs from its mates':
    the sub-batch goes through the members' own devices until the positions meet again); a member left out of a call (no buffer);
    drains at random moments.  After every drain every member's bytes equal its lone twin's."""
    rng = np.random.default_rng(900 + seed)
    cfg = [MOD_CASES["fm_2_3"]] * 10 + [MOD_CASES["rs_2_3"]] * 3 + [None] * 2
    n = len(cfg)
    gdevs, gsts = make_tx(S, n, "CF32", lambda i: cfg[i])
    sdevs, ssts = make_tx(S, n, "CF32", lambda i: cfg[i])
    grp = S.Group(gdevs)
    last = MTU
    for step in range(24):
        op = rng.choice(["call"] * 5 + ["lone_same", "lone_other", "skip", "drain"])
        if op in ("lone_same", "lone_other"):
            i = int(rng.integers(0, n)); num = last if op == "lone_same" else int(rng.choice([777, 1000, MTU - 2]))
            b = (samples(rng, "CF32", num) * 0.5).astype(np.float32)
            assert gdevs[i].writeStream(gsts[i], [b], num).ret == sdevs[i].writeStream(ssts[i], [b], num).ret == min(num, MTU)
        elif op == "drain":
            drain_equal(gdevs, sdevs, (seed, step))
        num = int(rng.choice([MTU] * 3 + [MTU - 1, 1000, 3, MTU + 5000]))
        last = num
        bufs = [(samples(rng, "CF32", num) * 0.5).astype(np.float32) for _ in range(n)]
        left_out = int(rng.integers(0, n)) if op == "skip" else -1
        gb = [None if i == left_out else b for i, b in enumerate(bufs)]
        nd, rets = grp.writeStream(gb, num)
        srets = [0 if i == left_out else sdevs[i].writeStream(ssts[i], [bufs[i]], num).ret for i in range(n)]
        assert rets == srets and nd == n - (left_out >= 0), (step, num, rets, srets)
    drain_equal(gdevs, sdevs, (seed, "end"))
    st = grp.stats()
    assert st["errors"] == 0 and st["batched_reads"] > 40, st
    grp.close()
    for d in gdevs + sdevs:
        d.close()


def test_a_modulator_launch_that_gives_up_loses_nothing_but_its_own_words(S):
    """The group asks its modulator launches for their verdict when their words are committed (clhip_tx_pipe_status; a look-back that
    gave up): the sub-batch is launched again in ticket order, and a second failure -- forced here with a poll bound of 0, which holds
    in both orders -- commits nothing, is reported by the flush (the call itself had returned already: write-behind) and leaves the
    pipes where they were before the call: the members carry on exactly like twins that never saw the lost call."""
    n = 9
    args = MOD_CASES["fm_2_3"]
    gdevs, gsts = make_tx(S, n, "CF32", lambda i: args)
    sdevs, ssts = make_tx(S, n, "CF32", lambda i: args)
    grp = S.Group(gdevs)
    rng = np.random.default_rng(31)

    def both(num, twins=True):
        bufs = [(samples(rng, "CF32", num) * 0.5).astype(np.float32) for _ in range(n)]
        nd, rets = grp.writeStream(bufs, num)
        if twins:
            assert rets == [sdevs[i].writeStream(ssts[i], [bufs[i]], num).ret for i in range(n)]
        return nd, rets

    both(MTU); both(5000)
    assert grp.flush() == 0
    grp.setTxPollBound(0)
    nd, rets = both(60000, twins=False)
    assert nd == n and rets == [60000] * n                                # told before the launches ran
    assert grp.flush() == -1 and "lost" in grp.lastError()
    assert all(gdevs[i].streamStats(gsts[i])["tx_overruns"] == 2 for i in range(n))      # (both attempts, as cl_writeStream counts them)
    grp.setTxPollBound(-1)
    both(MTU - 3); both(MTU)
    assert grp.flush() == 0
    drain_equal(gdevs, sdevs, "after the lost call")
    grp.close()
    for d in gdevs + sdevs:
        d.close()
