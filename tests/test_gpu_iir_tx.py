"""-m gpu: IIR (fp64 blocked scan), FM mod/demod, CW tone and the TX pipe through the C-ABI."""
import os
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def G():
    import torch
    from cariboulite_amd import hip
    import gpu_util
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return gpu_util


def _sos5(orc_iir):
    s = orc_iir.sos()
    return np.concatenate([s[:, :3], s[:, 4:]], 1)


@pytest.mark.parametrize("bw_khz", [20, 50, 100])
def test_iir_vs_oracle_and_scipy(G, orc, bw_khz):
    """CaribouliteStream.cpp:291-298 semantics; int16 outputs equal the sequential fp64 oracle
    (+-1 LSB allowed only at truncation boundaries: 'Truncation cliffs')."""
    import torch
    from cariboulite_amd import hip, synth
    n = 131072 + 777                       # one native batch + ragged tail, several scan tiles
    _, i, q = synth.smi_stream_bytes(n, 0, stream=4)
    iq = np.stack([i, q], 1).astype(np.int16)
    ref = orc.IIR(6, 4e6, bw_khz * 1e3 / 2)
    want = ref.apply_cs16(iq)
    f = hip.IIR(_sos5(orc.IIR(6, 4e6, bw_khz * 1e3 / 2)))
    d = torch.from_numpy(iq.copy()).to(G.DEV)
    f.run(d, n)
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and np.mean(diff != 0) < 1e-5, (diff.max(), np.mean(diff != 0))
    # the carried state equals the oracle's DF-II state
    st = f.state[0]
    for s in range(3):
        assert abs(st[2 * s] - ref.fi.v1[s]) <= 1e-9 * max(1.0, abs(ref.fi.v1[s]))
        assert abs(st[8 + 2 * s + 1] - ref.fq.v2[s]) <= 1e-9 * max(1.0, abs(ref.fq.v2[s]))
    # scipy fixture (first 8192 samples, different input): +-1 LSB
    g = load_golden("dsp_float.npz")
    f2 = hip.IIR(_sos5(orc.IIR(6, 4e6, bw_khz * 1e3 / 2)))
    d2 = torch.from_numpy(g["iq_int16"].copy()).to(G.DEV)
    f2.run(d2, 8192)
    got2 = d2.cpu().numpy()
    assert np.max(np.abs(got2[:, 0] - np.trunc(g[f"iir_{bw_khz}k__y_i"]))) <= 1
    assert np.max(np.abs(got2[:, 1] - np.trunc(g[f"iir_{bw_khz}k__y_q"]))) <= 1


def test_iir_streaming_and_batch(G, orc):
    """State persists across calls (never reset: SURVEY.md section 8 a6); streams are independent."""
    import torch
    from cariboulite_amd import hip, synth
    n, ns = 40000, 3
    iqs = [np.stack(synth.smi_stream_bytes(n, 0, stream=20 + s)[1:], 1).astype(np.int16) for s in range(ns)]
    sos = _sos5(orc.IIR(6, 4e6, 25e3))
    want = [orc.IIR(6, 4e6, 25e3).apply_cs16(x) for x in iqs]
    # batch, one shot
    f = hip.IIR(sos, ns)
    d = torch.from_numpy(np.stack(iqs)).to(G.DEV)
    f.run(d, n, stride=n)
    got = d.cpu().numpy()
    for s in range(ns):
        assert np.max(np.abs(got[s].astype(int) - want[s].astype(int))) <= 1
        assert np.mean(got[s] != want[s]) < 1e-4
    # chunked on stream 0: identical to one shot except at rare truncation boundaries
    f1 = hip.IIR(sos, 1)
    outs, pos = [], 0
    for cn in (1, 63, 64, 65, 16384, 7, 23416):
        dd = torch.from_numpy(iqs[0][pos:pos + cn].copy()).to(G.DEV)
        f1.run(dd, cn)
        outs.append(dd.cpu().numpy()); pos += cn
    assert pos == n
    chunked = np.concatenate(outs)
    assert np.max(np.abs(chunked.astype(int) - want[0].astype(int))) <= 1
    assert np.mean(chunked != want[0]) < 1e-4
    # zeros in -> zeros out, impulse decays (sanity of the scan tables)
    z = torch.zeros((70000, 2), dtype=torch.int16, device=G.DEV)
    z[5, 0] = 4000
    fz = hip.IIR(sos, 1); fz.run(z, 70000)
    zz = z.cpu().numpy()
    assert np.all(zz[:, 1] == 0) and np.all(zz[60000:, 0] == 0) and zz[:2000, 0].max() > 0


def test_fm_demod_and_cw(G, orc):
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(3)
    n = 100003
    x = (rng.standard_normal((n, 2)) * 0.3).astype(np.float32)
    want, _ = orc.fm_demod_f64(x.astype(np.float64))
    d = torch.from_numpy(x).to(G.DEV)
    prev = torch.zeros(2, dtype=torch.float32, device=G.DEV)
    out = torch.empty(n, dtype=torch.float32, device=G.DEV)
    # two calls: the last sample is carried
    hip.fm_demod(d, 50000, prev, out)
    hip.fm_demod(d[50000:], n - 50000, prev, out[50000:])
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    dd = np.abs(got - want); dd = np.minimum(dd, 2 * np.pi - dd)
    assert got[0] == 0.0 and np.max(dd) <= TOL * np.pi
    tone = torch.empty((4_000_000, 2), dtype=torch.float32, device=G.DEV)
    hip.cw_tone(100e3, 4e6, 0.25, 4_000_000, tone)
    t = tone.cpu().numpy()
    k = np.arange(4_000_000)
    ph = 2 * np.pi * 100e3 * k / 4e6 + 0.25
    assert np.max(np.abs(t[:, 0] - np.cos(ph))) <= TOL and np.max(np.abs(t[:, 1] - np.sin(ph))) <= TOL
    o, _ = orc.cw_tone(100e3, 4e6, 4096, 0.25)
    assert np.max(np.abs(t[:4096] - o)) <= 1e-6


def test_fm_mod_vs_oracle(G, orc):
    import torch
    from cariboulite_amd import hip
    g = load_golden("dsp_float.npz")
    m = g["fm_msg"]; kf = float(g["fm_mod_kf"])
    d = torch.from_numpy(m).to(G.DEV)
    ws = torch.empty(hip.lib().clhip_fm_mod_workspace_bytes(m.size) // 8 + 1, dtype=torch.float64, device=G.DEV)
    ph = torch.zeros(1, dtype=torch.float64, device=G.DEV)
    out = torch.empty((m.size, 2), dtype=torch.float32, device=G.DEV)
    hip.fm_mod(d, m.size, kf, 4e6, ph, out, ws)
    torch.cuda.synchronize()
    assert np.max(np.abs(out.cpu().numpy() - g["fm_mod_iq"])) <= TOL
    _, want_ph = orc.fm_mod_f64(m, kf, 4e6)
    dph = abs(float(ph[0]) - want_ph); dph = min(dph, 2 * np.pi - dph)
    assert dph < 1e-9
    # long message, chunked with carried phase == oracle's sequential fp64 accumulation
    rng = np.random.default_rng(12)
    n = 1_000_003
    msg = (0.5 * np.sin(2 * np.pi * 3e3 * np.arange(n) / 4e6) + 0.2 * rng.standard_normal(n)).astype(np.float32)
    want, _ = orc.fm_mod_f64(msg, 75e3, 4e6)
    dm = torch.from_numpy(msg).to(G.DEV)
    ph.zero_()
    o = torch.empty((n, 2), dtype=torch.float32, device=G.DEV)
    ws = torch.empty(hip.lib().clhip_fm_mod_workspace_bytes(n) // 8 + 1, dtype=torch.float64, device=G.DEV)
    pos = 0
    for cn in (1, 1023, 1024, 1025, 500000, n - 503073):
        hip.fm_mod(dm[pos:], cn, 75e3, 4e6, ph, o[pos:], ws); pos += cn
    assert pos == n
    torch.cuda.synchronize()
    assert np.max(np.abs(o.cpu().numpy() - want)) <= TOL


def _tx_words_to_iq(words):
    """(i, q) as signed 13-bit integers from TX words in the documented layout (caribou_smi.c:693-696): in memory order
    byte0 = [SOF, TXC, CTX, I12..I8], byte1 = [0, I7..I1], byte2 = [0, I0, Q12..Q7], byte3 = [0, Q6..Q0]"""
    w = words.astype(np.uint32)
    b0, b1, b2, b3 = w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFF, (w >> 24) & 0xFF
    i13 = ((b0 & 0x1F) << 8) | ((b1 & 0x7F) << 1) | ((b2 >> 6) & 1)
    q13 = ((b2 & 0x3F) << 7) | (b3 & 0x7F)
    sx = lambda v: ((v.astype(np.int32) + 4096) & 0x1FFF) - 4096
    return np.stack([sx(i13), sx(q13)], 1)


def test_tx_pipe_config5(G, orc):
    """float message -> FM mod -> 2/3 resample -> x4096 truncate -> int13 pack (BASELINE config 5).
    Float stages to 1e-5 (tap output); the integer tail is bit-exact on the GPU's own floats."""
    import torch
    from cariboulite_amd import hip
    g, t = load_golden("dsp_float.npz"), load_golden("taps.npz")
    m = g["fm_msg"]; n = m.size
    pipe = hip.TxPipe(1, float(g["fm_mod_kf"]), 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    no = pipe.out_count(n)
    assert no == -(-n * 2 // 3)
    d = torch.from_numpy(m).to(G.DEV)
    by = torch.zeros(4 * no, dtype=torch.uint8, device=G.DEV)
    tap = torch.zeros((no, 2), dtype=torch.float32, device=G.DEV)
    assert pipe.run(hip.TXPIPE_IN_FM_MESSAGE, d, 0, n, by, 4 * no, tap, no) == no
    torch.cuda.synchronize()
    tp = tap.cpu().numpy()
    assert np.max(np.abs(tp - g["fm_mod_rs_2_3"])) <= TOL * np.max(np.abs(g["fm_mod_rs_2_3"]))
    # integer tail in isolation: oracle quantise + pack of the SAME floats -> identical bytes
    want_bytes = orc.generate_data(orc.cf32_to_cs16(tp), orc.TX_DOCUMENTED)
    assert np.array_equal(by.cpu().numpy(), want_bytes)
    # FPGA parser round trip recovers the quantised samples
    w = orc.fpga_tx_parse(by.cpu().numpy())
    _, iq, _ = orc.rx_data_analyze(0, (w & ~np.uint32(1 << 16)).view(np.uint8))
    q13 = ((orc.cf32_to_cs16(tp).astype(np.int32) + 4096) & 0x1FFF) - 4096     # the packer keeps 13 bits (& 0x1FFF)
    assert np.array_equal(iq[:no], q13)
    # chunked == one shot (phase + resampler history carried)
    pipe2 = hip.TxPipe(1, float(g["fm_mod_kf"]), 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    outs, pos = [], 0
    for cn in (3, 1, 2, 4093, 4093):
        k = pipe2.out_count(cn)
        b2 = torch.zeros(4 * max(k, 1), dtype=torch.uint8, device=G.DEV)
        assert pipe2.run(hip.TXPIPE_IN_FM_MESSAGE, d[pos:], 0, cn, b2, 4 * max(k, 1)) == k
        outs.append(b2.cpu().numpy()[:4 * k]); pos += cn
    assert pos == n
    chunked = np.concatenate(outs).view(np.uint32)
    one = by.cpu().numpy().view(np.uint32)
    # identical up to rare 1-LSB truncation flips: the fp64 phase sums are re-associated, and a superblock's first
    # sub-block is worked relative to its own start and rotated into place afterwards (one more fp32 rounding)
    assert chunked.size == one.size and np.mean(chunked != one) < 2e-3
    assert np.abs(_tx_words_to_iq(chunked) - _tx_words_to_iq(one)).max() <= 1          # and a flip is one LSB
    # as-written mode reproduces the reference's constant output
    pipe3 = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_AS_WRITTEN)
    assert pipe3.run(hip.TXPIPE_IN_FM_MESSAGE, d, 0, 300, by, 4 * no) == 200
    assert by.cpu().numpy()[:800].reshape(-1, 4).tolist() == [[0xFF, 0x7F, 0x40, 0x00]] * 200
    # CF32 input, no resampler: pure Soapy writeStream(CF32) tail
    x = (np.random.default_rng(0).standard_normal((5000, 2)) * 0.4).astype(np.float32)
    pipe4 = hip.TxPipe(1, 0.0, 4e6, None, 1, 1, hip.TX_DOCUMENTED)
    b4 = torch.zeros(4 * 5000, dtype=torch.uint8, device=G.DEV)
    assert pipe4.run(hip.TXPIPE_IN_CF32, torch.from_numpy(x).to(G.DEV), 0, 5000, b4, 4 * 5000) == 5000
    assert np.array_equal(b4.cpu().numpy(), orc.generate_data(orc.cf32_to_cs16(x), orc.TX_DOCUMENTED))


def test_tx_pipe_streaming_any_start_phase(G, orc):
    """Config 5 in chunks of every residue mod 3 and across the fast kernel's sub-block (3072) and superblock
    (18432; 12288 in round 1) boundaries: float tap vs the fp64 oracle (FM mod -> upfirdn 2/3), words == quantise+pack of the tap."""
    import torch
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    rng = np.random.default_rng(77)
    n = 200_003
    msg = (0.4 * np.sin(2 * np.pi * 2e3 * np.arange(n) / 4e6) + 0.3 * rng.standard_normal(n)).astype(np.float32)
    iq, _ = orc.fm_mod_f64(msg, 75e3, 4e6)
    want = orc.Resampler(t["rs_2_3"], 2, 3).f64(iq)
    d = torch.from_numpy(msg).to(G.DEV)
    pipe = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    sizes = [1, 1, 2, 5, 3071, 3072, 3073, 1, 12287, 12288, 12289, 4, 7, 40000, 2, 18431, 18432, 18433]
    sizes.append(n - sum(sizes))
    taps_out, words_out, pos = [], [], 0
    for cn in sizes:
        k = pipe.out_count(cn)
        by = torch.zeros(4 * max(k, 1) + 16, dtype=torch.uint8, device=G.DEV)
        tp = torch.zeros((max(k, 1) + 2, 2), dtype=torch.float32, device=G.DEV)
        # odd element offsets: the kernel must not need 16-byte aligned buffers
        assert pipe.run(hip.TXPIPE_IN_FM_MESSAGE, d[pos:], 0, cn, by[4:], 4 * max(k, 1), tp[1:], max(k, 1)) == k
        taps_out.append(tp[1:1 + k].cpu().numpy()); words_out.append(by[4:4 + 4 * k].cpu().numpy()); pos += cn
    assert pos == n
    got = np.concatenate(taps_out)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))
    assert np.array_equal(np.concatenate(words_out), orc.generate_data(orc.cf32_to_cs16(got), orc.TX_DOCUMENTED))
    # one shot == chunked to rounding
    pipe1 = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    k = pipe1.out_count(n)
    tp1 = torch.zeros((k, 2), dtype=torch.float32, device=G.DEV)
    by1 = torch.zeros(4 * k, dtype=torch.uint8, device=G.DEV)
    assert pipe1.run(hip.TXPIPE_IN_FM_MESSAGE, d, 0, n, by1, 4 * k, tp1, k) == k
    assert np.max(np.abs(tp1.cpu().numpy() - got)) <= 2e-6
    # two streams in one call (strided buffers)
    pipe2 = hip.TxPipe(2, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    n2 = 30001
    d2 = torch.stack([d[:n2], d[50000:50000 + n2]]).contiguous()
    k2 = pipe2.out_count(n2)
    tp2 = torch.zeros((2, k2, 2), dtype=torch.float32, device=G.DEV)
    by2 = torch.zeros((2, 4 * k2), dtype=torch.uint8, device=G.DEV)
    assert pipe2.run(hip.TXPIPE_IN_FM_MESSAGE, d2, n2, n2, by2, 4 * k2, tp2, k2) == k2
    for si, off in ((0, 0), (1, 50000)):
        iq2, _ = orc.fm_mod_f64(msg[off:off + n2], 75e3, 4e6)
        w2 = orc.Resampler(t["rs_2_3"], 2, 3).f64(iq2)
        assert np.max(np.abs(tp2[si].cpu().numpy() - w2)) <= TOL * np.max(np.abs(w2))


@pytest.mark.parametrize("order", [2, 4, 6, 8])
def test_iir_orders_and_long_streams(G, orc, order):
    """1..4 biquads; a stream long enough for several tile groups (256 tiles of 8192 samples per group), a ragged
    tail, then a second call that continues from the carried state; two streams with an odd stride."""
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(100 + order)
    n1, n2 = 2 * 256 * 8192 + 8192 * 3 + 77, 300_001
    x = rng.integers(-4096, 4096, size=(n1 + n2, 2), dtype=np.int16)
    ref = orc.IIR(order, 4e6, 60e3)
    want = ref.apply_cs16(x.copy())
    f = hip.IIR(_sos5(orc.IIR(order, 4e6, 60e3)))
    d = torch.from_numpy(x.copy()).to(G.DEV)
    f.run(d, n1)
    f.run(d[n1:], n2)
    got = d.cpu().numpy()
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and np.mean(diff != 0) < 1e-4, (diff.max(), np.mean(diff != 0))
    # two streams, stride larger than the length, unaligned start (falls back to scalar staging)
    ns, n, stride = 2, 100_003, 100_100
    buf = rng.integers(-4096, 4096, size=(ns * stride + 1, 2), dtype=np.int16)
    d2 = torch.from_numpy(buf.copy()).to(G.DEV)
    f2 = hip.IIR(_sos5(orc.IIR(order, 4e6, 60e3)), ns)
    f2.run(d2[1:], n, stride=stride)
    got2 = d2.cpu().numpy()
    for s in range(ns):
        lo = 1 + s * stride
        w = orc.IIR(order, 4e6, 60e3).apply_cs16(buf[lo:lo + n].copy())
        dd = np.abs(got2[lo:lo + n].astype(np.int32) - w.astype(np.int32))
        assert dd.max() <= 1 and np.mean(dd != 0) < 1e-4
        assert np.array_equal(got2[lo + n:lo + stride], buf[lo + n:lo + stride])      # the gap is untouched
    assert np.array_equal(got2[0], buf[0])


def test_tx_pipe_long_message_lookback(G, orc):
    """3.2 M messages = 174 superblocks per stream: the single-launch look-back runs many workgroups deep;
    two consecutive calls (epochs) on the same pipe, both against the fp64 oracle."""
    import torch
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    rng = np.random.default_rng(5)
    n = 3_200_017
    msg = (0.5 * np.sin(2 * np.pi * 1.5e3 * np.arange(2 * n) / 4e6) + 0.25 * rng.standard_normal(2 * n)).astype(np.float32)
    iq, _ = orc.fm_mod_f64(msg, 75e3, 4e6)
    want = orc.Resampler(t["rs_2_3"], 2, 3).f64(iq)
    d = torch.from_numpy(msg).to(G.DEV)
    pipe = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    got, pos = [], 0
    for cn in (n, n):
        k = pipe.out_count(cn)
        by = torch.zeros(4 * k, dtype=torch.uint8, device=G.DEV)
        tp = torch.zeros((k, 2), dtype=torch.float32, device=G.DEV)
        assert pipe.run(hip.TXPIPE_IN_FM_MESSAGE, d[pos:], 0, cn, by, 4 * k, tp, k) == k
        got.append(tp.cpu().numpy()); pos += cn
    got = np.concatenate(got)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))


def test_tx_lookback_overrun_is_reported_by_the_same_call(G, orc):
    """The look-back's bounded poll: forced to give up (poll bound 0), the call's verdict is -1 right after its own
    stream sync, the pipe is back in its pre-call state, and repeating the call gives a fresh pipe's bytes."""
    import torch
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    rng = np.random.default_rng(9)
    n0, n = 40_000, 80_000                                      # 3 and 5 superblocks of 18432 messages
    msg = (0.4 * rng.standard_normal(n0 + n)).astype(np.float32)
    d = torch.from_numpy(msg).to(G.DEV)

    def run(pipe, lo, cn):
        k = pipe.out_count(cn)
        by = torch.zeros(4 * k, dtype=torch.uint8, device=G.DEV)
        assert pipe.run(hip.TXPIPE_IN_FM_MESSAGE, d[lo:], 0, cn, by, 4 * k) == k
        torch.cuda.synchronize()
        return by.cpu().numpy()

    good = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    w0 = run(good, 0, n0); assert good.status() == 0
    w1 = run(good, n0, n); assert good.status() == 0
    p = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    assert np.array_equal(run(p, 0, n0), w0) and p.status() == 0
    before = p.out_count(n)
    p.set_poll_bound(0)
    run(p, n0, n)
    assert p.status() == -1 and "look-back" in hip.last_error()
    assert p.status() == 0                                      # reported once
    assert p.out_count(n) == before                             # polyphase phase restored
    p.set_poll_bound(-1)
    assert np.array_equal(run(p, n0, n), w1) and p.status() == 0   # FM phase and resampler history restored


def test_iir_second_round_of_tile_groups(G, orc):
    """71 M samples = 68 tile groups: K2b's group scan takes a second 64-group round.  Bit-for-bit check of the int16
    outputs against the sequential fp64 oracle (about 10 s of host time)."""
    import torch
    from cariboulite_amd import hip
    n = (1 << 26) + (1 << 22) + 12345
    x = np.random.default_rng(3).integers(-4096, 4096, size=(n, 2), dtype=np.int16)
    ref = orc.IIR(6, 4e6, 25e3)
    want = ref.apply_cs16(x.copy())
    f = hip.IIR(_sos5(orc.IIR(6, 4e6, 25e3)))
    d = torch.from_numpy(x).to(G.DEV)
    f.run(d, n)
    got = d.cpu().numpy()
    del d
    diff = np.abs(got[-(1 << 23):].astype(np.int32) - want[-(1 << 23):].astype(np.int32))      # the part behind group 64
    assert diff.max() <= 1 and np.mean(diff != 0) < 1e-4
    assert np.mean(got[: 1 << 23] != want[: 1 << 23]) < 1e-4


def test_iir_four_kernel_scan_still_agrees(G, orc):
    """CLHIP_IIR_ONEPASS=0: the four-kernel blocked scan (the path of filters whose memory is too long for the
    single-pass kernel's horizon) against the same oracle.  Child process: the switch is latched at first use."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        from cariboulite_amd import hip, soapy
        from oracle import oracle as orc
        rng = np.random.default_rng(5)
        n = 3 * 256 * 4096 + 4096 * 7 + 123
        x = rng.integers(-4096, 4096, size=(n, 2), dtype=np.int16)
        ref = orc.IIR(6, 4e6, 10e3)
        want = ref.apply_cs16(x.copy())
        s = ref.sos()
        sos5 = np.concatenate([s[:, :3], s[:, 4:]], 1)
        f = hip.IIR(sos5)
        d = torch.from_numpy(x.copy()).to("cuda:0")
        f.run(d, 2 * 256 * 4096 + 11)
        f.run(d[2 * 256 * 4096 + 11:], n - (2 * 256 * 4096 + 11))
        diff = np.abs(d.cpu().numpy().astype(np.int32) - want.astype(np.int32))
        assert diff.max() <= 1 and np.mean(diff != 0) < 1e-4, (diff.max(), np.mean(diff != 0))
        print("legacy ok")
    """) % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CLHIP_IIR_ONEPASS="0"), capture_output=True, text=True)
    assert r.returncode == 0 and "legacy ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_iir_narrowest_reference_filter_many_tiles(G, orc):
    """fc = 10 kHz (the reference's 20 kHz-bandwidth filter: the longest memory of the three, horizon of several
    tiles in the single-pass kernel), 32 streams of 40 tiles + a ragged tail, two calls: carried state + look-back
    near the stream start (the carried state enters through the first `horizon` tiles)."""
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(11)
    ns, n1, n2 = 6, 40 * 4096 + 1000, 9 * 4096 + 17
    x = rng.integers(-4096, 4096, size=(ns, n1 + n2, 2), dtype=np.int16)
    f = hip.IIR(_sos5(orc.IIR(6, 4e6, 10e3)), ns)
    d = torch.from_numpy(x.copy()).to(G.DEV)
    f.run(d, n1, stride=n1 + n2)
    f.run(d[:, n1:], n2, stride=n1 + n2)
    got = d.cpu().numpy()
    for s in range(ns):
        want = orc.IIR(6, 4e6, 10e3).apply_cs16(x[s].copy())
        diff = np.abs(got[s].astype(np.int32) - want.astype(np.int32))
        assert diff.max() <= 1 and np.mean(diff != 0) < 1e-4, (s, diff.max(), np.mean(diff != 0))


def test_iir_overrun_is_reported_rolled_back_and_repaired(G, orc):
    """The single-pass kernel's polls are bounded.  With the bound forced below 0 every tile that looks at a predecessor's
    aggregate gives up: clhip_iir_status() says so for that very call (the object's own pinned word), the carried state is
    back where it was (the launch wrote the other half of the ping-pong), no NaN has travelled, and the object takes the
    four-kernel scan from then on.  Out of place, clhip_iir_finish() repeats the call by itself; in place the caller
    re-produces the input.  A second object on the same GPU never sees the first one's verdict."""
    if os.environ.get("CLHIP_IIR_ONEPASS") == "0":
        pytest.skip("the A/B switch in force replaces the single-pass kernel whose polls this test forces to give up")
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(5)
    n1, n2 = 37 * 4096 + 5, 600 * 4096 + 123
    x = rng.integers(-4096, 4096, size=(n1 + n2, 2), dtype=np.int16)
    sos = _sos5(orc.IIR(6, 4e6, 50e3))
    want = orc.IIR(6, 4e6, 50e3).apply_cs16(x.copy())
    f, other = hip.IIR(sos, 1), hip.IIR(sos, 1)
    d = torch.from_numpy(x.copy()).to(G.DEV)
    out = torch.zeros_like(d)
    f.run(d, n1, out=out)                                 # a good first call: the state is no longer at rest
    other.run(d, n1, out=torch.empty_like(d))
    torch.cuda.synchronize()
    assert f.status() == 0 and other.status() == 0 and not f.on_scan_path()
    st0 = f.state.copy()
    assert np.abs(st0).max() > 0
    # forced overrun, out of place
    f.set_poll_bound(-1)
    f.run(d[n1:], n2, out=out[n1:])
    torch.cuda.synchronize()
    assert other.status() == 0                            # not the other object's business
    assert f.finish() == 1                                # overran -> rolled back -> repeated on the scan path
    assert f.on_scan_path() and f.status() == 0
    got = out.cpu().numpy()
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and np.mean(diff != 0) < 1e-4, (diff.max(), np.mean(diff != 0))
    assert np.all(np.isfinite(f.state))
    # forced overrun, in place: the verdict, the restored state, and the repeat is the caller's
    g = hip.IIR(sos, 1)
    g.run(d, n1, out=out)
    torch.cuda.synchronize()
    assert g.status() == 0 and np.array_equal(g.state, st0)
    g.set_poll_bound(-1)
    e = d[n1:].clone()
    g.run(e, n2)
    torch.cuda.synchronize()
    assert g.finish() == -1 and np.array_equal(g.state, st0) and g.on_scan_path()
    e.copy_(d[n1:])                                       # re-produce the input
    g.run(e, n2)
    assert g.finish() == 0
    diff = np.abs(e.cpu().numpy().astype(np.int32) - want[n1:].astype(np.int32))
    assert diff.max() <= 1 and np.mean(diff != 0) < 1e-4
    # an unconsulted overrun is refused by the next run instead of silently chaining garbage
    h = hip.IIR(sos, 1)
    h.set_poll_bound(-1)
    h.run(d, n2, out=out)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError):
        h.run(d, n2, out=out)


@pytest.mark.parametrize("seg", [16, 32, 64])
@pytest.mark.parametrize("dynamic", [1, 0])
def test_iir_every_segment_length_and_tile_order(G, orc, seg, dynamic):
    """The single-pass kernel's three segment lengths (16 / 32 / 64 samples per lane, picked from the call's size) and
    both tile orders (a ticket per tile / a rank per wave), forced here: three streams, a ragged tail, a second call
    from the carried state, the narrowest reference filter where its horizon fits (fc = 10 kHz at 16-sample segments
    would need 40 predecessor tiles: it takes 32) -- all against the sequential fp64 oracle."""
    if os.environ.get("CLHIP_IIR_ONEPASS") == "0":
        pytest.skip("the A/B switch in force replaces the single-pass kernel")
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(seg + dynamic)
    ns, n1, n2 = 3, 70 * 32 * seg + 3 * seg + 5, 9 * 32 * seg + 17
    x = rng.integers(-4096, 4096, size=(ns, n1 + n2, 2), dtype=np.int16)
    for fc in (50e3, 25e3, 10e3):
        f = hip.IIR(_sos5(orc.IIR(6, 4e6, fc)), ns)
        hip.lib().clhip_iir_set_shape(f.h, seg, dynamic)
        d = torch.from_numpy(x.copy()).to(G.DEV)
        f.run(d, n1, stride=n1 + n2)
        f.run(d[:, n1:], n2, stride=n1 + n2)
        assert f.finish() == 0
        got = d.cpu().numpy()
        assert not f.on_scan_path()                        # (fc = 10 kHz at 16-sample segments: that CALL took the scan, the object did not switch)
        for s in range(ns):
            want = orc.IIR(6, 4e6, fc).apply_cs16(x[s].copy())
            diff = np.abs(got[s].astype(np.int32) - want.astype(np.int32))
            assert diff.max() <= 1 and np.mean(diff != 0) < 1e-4, (fc, s, diff.max(), np.mean(diff != 0))


@pytest.mark.parametrize("order", [4, 6, 8])
def test_iir_sections_with_general_numerators(G, orc, order):
    """The C ABI takes any biquad cascade.  A Chebyshev-II low-pass has its zeros off z = -1, so no section is
    b = (1, 2, 1): the single-pass kernel takes its five-operation stage form (the Butterworth designs of the other tests
    all take the four-operation one).  Oracle: the same DF-II cascade (orc_iir_step) with these coefficients."""
    import ctypes as C
    import torch
    from scipy import signal
    from cariboulite_amd import hip
    sos = signal.cheby2(order, 40, 120e3, "low", fs=4e6, output="sos")
    assert not any(np.array_equal(s[:3], [1.0, 2.0, 1.0]) for s in sos[1:])
    sos5 = np.concatenate([sos[:, :3] / sos[:, 3:4], sos[:, 4:] / sos[:, 3:4]], 1)

    def oracle_filter():
        f = orc.IIR(order, 4e6, 50e3)                    # any design of the same order: only the struct is reused
        for rail in (f.fi, f.fq):
            for s in range(order // 2):
                rail.b0[s], rail.b1[s], rail.b2[s], rail.a1[s], rail.a2[s] = sos5[s]
                rail.v1[s] = rail.v2[s] = 0.0
        return f

    rng = np.random.default_rng(order)
    n1, n2 = 70 * 4096 + 333, 5 * 4096 + 1
    x = rng.integers(-4096, 4096, size=(n1 + n2, 2), dtype=np.int16)
    f = hip.IIR(sos5, 1)
    d = torch.from_numpy(x.copy()).to(G.DEV)
    f.run(d, n1)
    f.run(d[n1:], n2)                                    # carried state, ragged tails
    assert f.finish() == 0
    want = oracle_filter().apply_cs16(x.copy())
    diff = np.abs(d.cpu().numpy().astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and np.mean(diff != 0) < 1e-4, (diff.max(), np.mean(diff != 0))


@pytest.mark.parametrize("bw_khz", [100, 50, 20])
def test_iir_time_slices_with_a_halo_equal_one_filter(G, orc, bw_khz):
    """SURVEY.md section 8e names a carried-state hand-off for slicing one long stream through the IIR over several GPUs.  The
    filter's memory is finite (clhip_iir_memory_samples: nothing of an older state is left above 1e-12), so a slice owner
    starts from rest that many samples before its slice and discards the halo's outputs: the concatenated slices equal ONE
    filter's output over the whole stream, sample for sample -- no state message, no collective (shard.run_iir_time_slice)."""
    import torch
    from cariboulite_amd import hip, shard, soapy as S
    n = 6_000_000 + 12345
    g = torch.Generator(device=G.DEV); g.manual_seed(1000 + bw_khz)
    iq = torch.randint(-4096, 4096, (n, 2), dtype=torch.int16, device=G.DEV, generator=g)
    sos = S.design_butter_lowpass(6, 4e6, bw_khz * 1e3 / 2)
    one = hip.IIR(sos)
    ref = torch.empty_like(iq)
    one.run(iq, n, out=ref)
    torch.cuda.synchronize()
    assert one.status() == 0
    mem = one.memory_samples()
    assert 0 < mem <= 65536, mem
    for world in (2, 5):
        got = torch.zeros_like(iq)
        scratch = torch.empty((mem + 8, 2), dtype=torch.int16, device=G.DEV)
        for (a, b) in shard.time_slices(n, world, 1):
            f = hip.IIR(sos)
            assert shard.run_iir_time_slice(f, iq, a, b, got[a:], scratch) == b - a
            torch.cuda.synchronize()
            assert f.status() == 0
        # Equal sample for sample -- up to the fp64 rounding of two different histories: behind the halo the slice's state agrees
        # with the one filter's to 1e-12 but not bit for bit, the two trajectories stay an ulp of the state (1e-11 ... 1e-10)
        # apart, and an output crosses an integer with that probability per sample: one (int16) in 10^10 may differ by one LSB.
        # (Seen once in ~150 suite runs on unseeded input: 1 sample of 12 M; the input is seeded now.)
        d = (got.to(torch.int32) - ref.to(torch.int32)).abs()
        assert int(d.max()) <= 1 and int((d != 0).sum()) <= 2, (bw_khz, world, int(d.max()), int((d != 0).sum()))


def test_tx_fm_time_slices_with_a_phase_hand_off_equal_one_pipe(G, orc):
    """the FM modulator's phase is a prefix sum over the whole message stream -- the one state of the
    TX pipe a halo cannot rebuild.  Time slices of one long stream (one per GPU in production) therefore get the phase at
    their start as an 8-byte hand-off (an exclusive prefix over the slices' fp64 phase sums: shard.fm_slice_phases, one
    all_gather of a double under torch.distributed) and rebuild the resampler history from a short halo
    (clhip_tx_pipe_seek + shard.run_fm_time_slice): the concatenated words equal ONE pipe's up to the rare one-LSB
    truncation flips that any re-association of the fp64 phase sums produces (the bar of the chunked-vs-one-shot test)."""
    import torch
    from cariboulite_amd import hip, shard
    t = load_golden("taps.npz")
    dev = G.DEV
    n = 3 * 1_000_001
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    msg = 0.3 * torch.randn(n, device=dev, generator=gen)
    kf, fs = 75e3, 4e6
    one = hip.TxPipe(1, kf, fs, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    no = one.out_count(n)
    ref = torch.zeros(4 * no, dtype=torch.uint8, device=dev)
    assert one.run(hip.TXPIPE_IN_FM_MESSAGE, msg, 0, n, ref, 4 * no) == no
    torch.cuda.synchronize()
    assert one.status() == 0
    for world in (2, 5):
        slices = shard.time_slices(n, world, 3)                    # slice starts on polyphase phase 0 (M = 3)
        phases = shard.fm_slice_phases(msg, slices, kf, fs)
        got = torch.zeros_like(ref)
        scratch = torch.zeros(4 * 64, dtype=torch.uint8, device=dev)
        pos = 0
        for (a, b), ph in zip(slices, phases):
            if b <= a:
                continue
            p = hip.TxPipe(1, kf, fs, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
            assert pos == a * 2 // 3
            k = shard.run_fm_time_slice(p, msg, a, b, ph, kf, fs, got[4 * pos:], scratch, halo=63)
            torch.cuda.synchronize()
            assert p.status() == 0
            pos += k
        assert pos == no
        wa, wb = got.cpu().numpy().view(np.uint32), ref.cpu().numpy().view(np.uint32)
        assert np.mean(wa != wb) < 2e-3, (world, float(np.mean(wa != wb)))
        d13 = np.abs(_tx_words_to_iq(wa) - _tx_words_to_iq(wb)); d13 = np.minimum(d13, 8192 - d13)
        assert d13.max() <= 1, world


@pytest.mark.parametrize("ch", [0, 1])
def test_iir_fed_from_raw_smi_words_equals_unpack_then_filter(G, orc, ch):
    """clhip_iir_run_smi: the 13-bit field extraction of caribou_smi_rx_data_analyze as the filter's own input conversion.
    Same outputs and same carried state, bit for bit, as clhip_smi_unpack -> clhip_iir_run on the same words, for both channel
    types, across calls (one native batch, a ragged one, a long one) -- and against the sequential oracle."""
    import torch
    from cariboulite_amd import hip, synth, soapy as S
    sos = S.design_butter_lowpass(6, 4e6, 50e3)
    fa, fb = hip.IIR(sos), hip.IIR(sos)
    if fa.on_scan_path():                  # CLHIP_IIR_ONEPASS=0: the scan takes int16 samples, run_smi says -2 and the caller unpacks first
        assert fa.run_smi(ch, torch.zeros(64, dtype=torch.int32, device=G.DEV), 64, torch.zeros((64, 2), dtype=torch.int16, device=G.DEV)) == -2
        pytest.skip("the single-pass kernel is switched off")
    calls = [131072, 4097, 3_000_000, 64]
    n = sum(calls)
    b, i, q = synth.smi_stream_bytes(n, ch, stream=31 + ch)
    words = torch.from_numpy(b.view(np.int32).copy()).to(G.DEV)
    iq = torch.from_numpy(np.stack([i, q], 1).astype(np.int16)).to(G.DEV)
    oa, ob_ = torch.zeros((n, 2), dtype=torch.int16, device=G.DEV), torch.zeros((n, 2), dtype=torch.int16, device=G.DEV)
    pos = 0
    for cn in calls:
        assert fa.run_smi(ch, words[pos:], cn, oa[pos:]) == 0
        fb.run(iq[pos:], cn, out=ob_[pos:])
        torch.cuda.synchronize()
        assert fa.status() == 0 and fb.status() == 0
        pos += cn
    assert torch.equal(oa, ob_)
    assert np.array_equal(fa.state, fb.state)
    want = orc.IIR(6, 4e6, 50e3).apply_cs16(np.stack([i, q], 1).astype(np.int16))
    d = np.abs(oa.cpu().numpy().astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 1e-5


def test_two_large_iir_launches_share_the_gpu(G, orc):
    """Two filters, each on its own HIP stream, each launch large enough to want every resident wave of the chip, in
    flight together (round-2 review: 'no test runs two IIR streams at once').  Chunks are taken by ticket, so a wave only
    ever waits for chunks whose owners are already running: neither launch can starve the other, whatever share of the
    chip each gets.  Each object's verdict is its own; a launch that did give up (static assignment: clhip_iir_set_shape(f, 0, 0))
    is repaired by finish() and still has to be right."""
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(77)
    n = (1 << 24) + 1234
    xs = [rng.integers(-4096, 4096, size=(n, 2), dtype=np.int16) for _ in range(2)]
    bws = (50e3, 500e3)
    wants = [orc.IIR(6, 4e6, bw).apply_cs16(x.copy()) for x, bw in zip(xs, bws)]
    fs = [hip.IIR(_sos5(orc.IIR(6, 4e6, bw)), 1) for bw in bws]
    ds = [torch.from_numpy(x).to(G.DEV) for x in xs]
    outs = [torch.zeros_like(d) for d in ds]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    half = n // 2 + 77
    for lo, cnt in ((0, half), (half, n - half)):                      # two rounds: the second starts from a carried state
        for f, d, o, s in zip(fs, ds, outs, streams):
            f.run(d[lo:], cnt, out=o[lo:], stream=s.cuda_stream)       # no synchronisation between the two launches
        verdicts = [f.finish() for f in fs]
        assert all(v in (0, 1) for v in verdicts), verdicts
    torch.cuda.synchronize()
    for o, want in zip(outs, wants):
        diff = np.abs(o.cpu().numpy().astype(np.int32) - want.astype(np.int32))
        assert diff.max() <= 1 and np.mean(diff != 0) < 1e-4, (diff.max(), np.mean(diff != 0))


def test_tx_pipe_calls_on_either_side_of_the_kernel_choice_equal_one_shot(G, orc):
    """clhip_tx_pipe_run picks its kernel by the call's size (up to 2^22 messages: one sub-block per workgroup; above: superblocks
    of eight).  One pipe fed calls on either side of that line, in turn, gives the SMI words of one call over the whole message
    -- phase,
    resampler history and polyphase phase are one state whatever kernel wrote it."""
    import torch
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    sizes = [5_000_000, 131_072, 4_500_000, 1_000, (1 << 22) + 1, 1 << 22]
    n = sum(sizes)
    g = torch.Generator(device=G.DEV); g.manual_seed(77)
    msg = torch.randn(n, device=G.DEV, generator=g) * 0.35
    one = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    k = one.out_count(n)
    want = torch.zeros(4 * k, dtype=torch.uint8, device=G.DEV)
    assert one.run(hip.TXPIPE_IN_FM_MESSAGE, msg, 0, n, want, 4 * k) == k
    torch.cuda.synchronize()
    assert one.status() == 0
    p = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    got, pos = [], 0
    for cn in sizes:
        kc = p.out_count(cn)
        by = torch.zeros(4 * max(kc, 1), dtype=torch.uint8, device=G.DEV)
        assert p.run(hip.TXPIPE_IN_FM_MESSAGE, msg[pos:], 0, cn, by, 4 * max(kc, 1)) == kc
        torch.cuda.synchronize()
        assert p.status() == 0
        got.append(by[:4 * kc]); pos += cn
    got = torch.cat(got)
    assert got.numel() == want.numel()

    def iq(b):          # documented TX layout (caribou_smi.c:693-696) -> signed 13-bit (i, q)
        b = b.view(-1, 4).to(torch.int32)
        i13 = ((b[:, 0] & 0x1F) << 8) | ((b[:, 1] & 0x7F) << 1) | ((b[:, 2] >> 6) & 1)
        q13 = ((b[:, 2] & 0x3F) << 7) | (b[:, 3] & 0x7F)
        sx = lambda v: ((v + 4096) & 0x1FFF) - 4096
        return torch.stack([sx(i13), sx(q13)], 1)

    # the frame bits are the same words; the samples agree to the quantiser's last bit: two calls that cut the message at
    # different places round differently (the fp64 phase scan, the fp32 rotation of a sub-block into place), and an
    # (int16)(f * 4096) now and then falls on the other side of an integer -- the bar of the float stages (1e-5 relative) is five orders of magnitude above that
    a, b = iq(got), iq(want)
    d = (a - b).abs()
    d = torch.minimum(d, 8192 - d)       # 13-bit words: a unit phasor's 4096 and 4095 sit on either side of the wrap
    assert int(d.max()) <= 1, int(d.max())
    # (how often: a sub-block worked relative to its own start is rotated into place in fp32, so where the calls cut the message
    # moves 1e-7-sized roundings around: 3e-4 of the samples with the one-sub-block kernel on both sides)
    assert float((d != 0).float().mean()) < 1e-3, float((d != 0).float().mean())
    assert torch.equal(got.view(-1, 4)[:, 0] & 0xE0, want.view(-1, 4)[:, 0] & 0xE0)


def test_iir_last_call_can_be_taken_back(G, orc):
    """clhip_iir_unrun (a stream group that filtered ahead of a client who then went another way): run A, run B, take B back, run C --
    the outputs of C equal those of an object that ran A and C only, bit for bit, on both launch paths; with nothing to take back it
    refuses."""
    import torch
    from cariboulite_amd import hip, synth
    n, ns = 131072, 3
    sos = _sos5(orc.IIR(6, 4e6, 50e3))
    x = [torch.from_numpy(np.stack([np.stack(synth.smi_stream_bytes(n, 0, stream=60 + 3 * k + s)[1:], 1).astype(np.int16) for s in range(ns)])).to(G.DEV) for k in range(3)]
    for onepass in (True, False):
        f, ref = hip.IIR(sos, ns), hip.IIR(sos, ns)
        if not onepass:
            for o in (f, ref):
                o.set_poll_bound(-1); y = x[0].clone(); o.run(y, n, stride=n); torch.cuda.synchronize(); assert o.status() == -1     # (now on the scan path)
        a1, a2 = x[0].clone(), x[0].clone()
        f.run(a1, n, stride=n); ref.run(a2, n, stride=n)
        b = x[1].clone(); f.run(b, n, stride=n)
        torch.cuda.synchronize()
        assert f.status() == 0 and f.unrun() == 0
        c1, c2 = x[2].clone(), x[2].clone()
        f.run(c1, n, stride=n); ref.run(c2, n, stride=n)
        torch.cuda.synchronize()
        assert f.status() == 0 and ref.status() == 0
        assert torch.equal(a1, a2) and torch.equal(c1, c2), onepass
        assert f.unrun() == 0 and f.unrun() != 0                    # one level


def test_tx_pipe_stream_state_moves_between_pipes(G, orc):
    """clhip_tx_pipe_move_stream / _position / _set_position (what a stream group's modulator lanes rest on): a stream run for a while
    in a pipe of its own, moved into stream 2 of a four-stream pipe, run on there beside three others, moved home and run again yields,
    piece by piece and bit for bit, the words ONE pipe yields over the same calls -- phase, resampler history and polyphase position carried; the
    pieces' lengths leave every residue mod 3 at the moves.  Pipes of another configuration refuse."""
    import torch
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    rng = np.random.default_rng(78)
    n = 90_000
    msg = (0.4 * np.sin(2 * np.pi * 3e3 * np.arange(n) / 4e6) + 0.3 * rng.standard_normal(n)).astype(np.float32)
    others = (0.5 * rng.standard_normal((4, n))).astype(np.float32)
    d = torch.from_numpy(msg).to(G.DEV)
    plan = (("own", 10_001), ("grp", 20_000), ("grp", 7), ("own", 30_001), ("grp", 12_345), ("own", n - 72_354))
    one = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)      # (the same calls: another split re-associates the fp64 phase sums)
    want, pos = [], 0
    for _, cn in plan:
        k = one.out_count(cn)
        by = torch.zeros(4 * max(k, 1), dtype=torch.uint8, device=G.DEV)
        assert one.run(hip.TXPIPE_IN_FM_MESSAGE, d[pos:], 0, cn, by, 4 * max(k, 1)) == k
        want.append(by[:4 * k].cpu().numpy()); pos += cn
    want = np.concatenate(want)
    own = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    grp = hip.TxPipe(4, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    got, pos = [], 0
    for where, cn in plan:
        if where == "grp" and pos and got[-1][0] == "own":
            grp.set_position(own.position()); grp.take_stream_from(2, own, 0)
        if where == "own" and got and got[-1][0] == "grp":
            own.set_position(grp.position()); own.take_stream_from(0, grp, 2)
        if where == "own":
            kk = own.out_count(cn)
            by = torch.zeros(4 * max(kk, 1), dtype=torch.uint8, device=G.DEV)
            assert own.run(hip.TXPIPE_IN_FM_MESSAGE, d[pos:], 0, cn, by, 4 * max(kk, 1)) == kk
            torch.cuda.synchronize(); assert own.status() == 0
            got.append(("own", by[:4 * kk].cpu().numpy()))
        else:
            kk = grp.out_count(cn)
            rows = torch.from_numpy(others[:, pos:pos + cn].copy()).to(G.DEV)
            rows[2] = d[pos:pos + cn]
            by = torch.zeros((4, 4 * max(kk, 1)), dtype=torch.uint8, device=G.DEV)
            assert grp.run(hip.TXPIPE_IN_FM_MESSAGE, rows, cn, cn, by, 4 * max(kk, 1)) == kk
            torch.cuda.synchronize(); assert grp.status() == 0
            got.append(("grp", by[2, :4 * kk].cpu().numpy()))
        pos += cn
    assert pos == n
    assert np.array_equal(np.concatenate([g for _, g in got]), want)
    other_cfg = hip.TxPipe(1, 25e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    with pytest.raises(RuntimeError, match="configured differently"):
        other_cfg.take_stream_from(0, own, 0)


def test_words_to_rows_and_rows_to_rows(G):
    """The row launches of the stream group: rows of packed words stored into n_rows destinations of their own (the members' pinned TX
    FIFOs: 16-byte aligned and not), and rows of results of any length and alignment into destinations of their own (registered client
    buffers)."""
    import ctypes as C
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(79)
    rows, n, stride = 5, 10_007, 10_040
    w = torch.from_numpy(rng.integers(0, 2**32, (rows, stride), dtype=np.uint64).astype(np.uint32).view(np.int32)).to(G.DEV)
    dst = torch.zeros((rows, n + 16), dtype=torch.int32, device=G.DEV)
    ptrs = (C.c_void_p * rows)(*[dst[r].data_ptr() + 4 * (r % 3) for r in range(rows)])      # offsets of 0 / 4 / 8 bytes
    assert hip.lib().clhip_words_to_rows(w.data_ptr(), 4 * stride, n, rows, ptrs, None) == 0
    torch.cuda.synchronize()
    for r in range(rows):
        o = r % 3
        assert torch.equal(dst[r, o:o + n], w[r, :n]) and int(dst[r, :o].abs().sum()) == 0 and int(dst[r, o + n:].abs().sum()) == 0
    # byte rows: lengths and alignments of their own (16-byte, 4-byte and odd addresses)
    src = torch.from_numpy(rng.integers(0, 256, (rows, 70_000), dtype=np.uint8)).to(G.DEV)
    out = torch.zeros((rows, 70_064), dtype=torch.uint8, device=G.DEV)
    lens = [65_536, 70_000, 3, 33_333, 0]
    so, do = [0, 4, 1, 16, 0], [0, 8, 3, 5, 0]
    sp = (C.c_void_p * rows)(*[src[r].data_ptr() + so[r] for r in range(rows)])
    dp = (C.c_void_p * rows)(*[out[r].data_ptr() + do[r] for r in range(rows)])
    ln = (C.c_size_t * rows)(*[min(l, 70_000 - so[r]) for r, l in enumerate(lens)])
    assert hip.lib().clhip_rows_to_rows(sp, dp, ln, rows, None) == 0
    torch.cuda.synchronize()
    for r in range(rows):
        k = int(ln[r])
        assert torch.equal(out[r, do[r]:do[r] + k], src[r, so[r]:so[r] + k]), r
        assert int(out[r, :do[r]].sum()) == 0 and int(out[r, do[r] + k:].sum()) == 0, r


def test_tx_lookback_epoch_wrap_in_a_long_lived_pipe(G):
    """The single-launch FM path tags its look-back words with a 14-bit launch epoch (no reset between launches) and forgets every old
    entry when the epoch wraps, after 16 383 runs -- nine minutes of MTU calls at 4 MS/s.  A pipe is run up to its 16 000th call; its
    stream's state then moves into a fresh pipe (epoch 1) and both run on in lock step across the first one's wrap: bit for bit the same
    words, call after call."""
    import torch
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    rng = np.random.default_rng(80)
    n = 6_150                                                # (three look-back units per call; not a multiple of 3: every polyphase start)
    msgs = torch.from_numpy((0.4 * rng.standard_normal((7, n))).astype(np.float32)).to(G.DEV)
    old = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    by = torch.zeros(4 * (n * 2 // 3 + 2), dtype=torch.uint8, device=G.DEV)
    for k in range(16_000):
        old.run(hip.TXPIPE_IN_FM_MESSAGE, msgs[k % 7], 0, n, by, by.numel())
    torch.cuda.synchronize()
    assert old.status() == 0
    new = hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    new.set_position(old.position()); new.take_stream_from(0, old, 0)
    by2 = torch.zeros_like(by)
    for k in range(16_000, 16_800):                          # the wrap is at the old pipe's 16 384th run
        ka = old.run(hip.TXPIPE_IN_FM_MESSAGE, msgs[k % 7], 0, n, by, by.numel())
        kb = new.run(hip.TXPIPE_IN_FM_MESSAGE, msgs[k % 7], 0, n, by2, by2.numel())
        assert ka == kb
        if k % 16 == 0 or 16_376 <= k <= 16_392:
            torch.cuda.synchronize()
            assert old.status() == 0 and new.status() == 0
            assert torch.equal(by[:4 * ka], by2[:4 * kb]), k
