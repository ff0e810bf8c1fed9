"""-m gpu: the five BASELINE.json configurations at the sizes bench.py runs them, checked through size-independent
properties through the C ABI.  These tests allocate many gigabytes of device memory; they live in a
file of their own that sorts LAST so that the rest of the suite runs before the allocator has seen them, and each gives
its memory back when it is done."""
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


@pytest.fixture(autouse=True)
def _give_the_memory_back():
    yield
    import gc
    import torch
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def _tx_words_to_iq(words):
    """(i, q) as signed 13-bit integers from TX words in the documented layout (caribou_smi.c:693-696)"""
    w = words.astype(np.uint32)
    b0, b1, b2, b3 = w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFF, (w >> 24) & 0xFF
    i13 = ((b0 & 0x1F) << 8) | ((b1 & 0x7F) << 1) | ((b2 >> 6) & 1)
    q13 = ((b2 & 0x3F) << 7) | (b3 & 0x7F)
    sx = lambda v: ((v.astype(np.int32) + 4096) & 0x1FFF) - 4096
    return np.stack([sx(i13), sx(q13)], 1)



def test_config2_at_baseline_size_by_properties(G, orc):
    """BASELINE.json config 2 at the size bench.py runs it -- ONE 2^28-sample stream, 1 GiB in, 3 GiB out, output byte offsets
    past 2^31 -- checked through properties that do not need an oracle pass over 2^28 samples:
      * impulses scattered over the whole stream (tile seams, chunk seams, the last tile) come out as the oracle's response
        to the same impulse, and every other output is exactly zero;
      * four calls of 2^26 samples equal one call of 2^28, bit for bit (streaming state across calls);
      * the pipe is linear in its input: pipe(a) + pipe(b) == pipe(a + b) to rounding, on the synthetic stream."""
    import torch
    from cariboulite_amd import hip, synth
    t = load_golden("taps.npz")
    n = 1 << 28
    dev = G.DEV
    mk = lambda: hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    zero_word = int(synth.iq_to_words([0], [0], 0)[0])
    words = torch.full((n,), zero_word - (1 << 32), dtype=torch.int32, device=dev)       # (the sync bits set: 0x80004000)
    spots = [5, 4063, 4064, 4095, 4096, 131071, 131072, (1 << 27) + 12345, (1 << 28) - 4100, (1 << 28) - 200]
    amps = [(4095, -4096), (-4096, 4095), (1000, 0), (0, -1000), (77, 78), (-1, 1), (4095, 4095), (-4096, -4096), (2048, -2048), (1, 0)]
    for p, (ai, aq) in zip(spots, amps):
        w = int(synth.iq_to_words([ai], [aq], 0)[0])
        words[p] = w - (1 << 32) if w >= (1 << 31) else w
    pipe = mk()
    no = pipe.out_count(n)
    assert no == n * 3 // 2 and no * 8 > (1 << 31)
    out = torch.full((no + 8, 2), float("nan"), dtype=torch.float32, device=dev)
    assert pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0) == no
    torch.cuda.synchronize()
    assert torch.isnan(out[no:]).all() and not torch.isnan(out[:no]).any()
    # the oracle's response to one impulse, placed at each spot: FIR64 then 3/2 with 24 taps -> outputs m with 2m/3 in [p, p + 64 + 8)
    span = 128
    nz = torch.zeros(no, dtype=torch.bool, device=dev)
    clusters, cur = [], [0]
    for k in range(1, len(spots)):
        if spots[k] - spots[k - 1] < 2 * span: cur.append(k)
        else: clusters.append(cur); cur = [k]
    clusters.append(cur)
    for cl in clusters:                                            # impulses closer than a response length are checked together
        lo = (spots[cl[0]] // 2) * 2                               # an even input index: the resampler's phase there is 0
        ln = (spots[cl[-1]] - lo) // 2 * 2 + 2 * span
        x = np.zeros((ln, 2))
        for k in cl: x[spots[k] - lo] = (amps[k][0] / 4096.0, amps[k][1] / 4096.0)
        want = orc.Resampler(t["rs_3_2"], 3, 2).f64(orc.FIR(t["fir64_c2"]).f64(x))
        m0 = lo * 3 // 2
        m1 = min(m0 + want.shape[0], no)
        got = out[m0:m1].cpu().numpy()
        scale = max(np.max(np.abs(want)), 1e-30)
        assert np.max(np.abs(got - want[:m1 - m0])) <= TOL * scale, (cl, np.max(np.abs(got - want[:m1 - m0])), scale)
        nz[m0:m1] = True
        assert np.all(want[(spots[cl[-1]] - lo + 80) * 3 // 2:] == 0)   # the window covers the whole response (64 + 8 input samples)
    assert int(torch.count_nonzero(out[:no][~nz])) == 0           # everything else: exactly zero
    del nz
    # streaming: four calls of 2^26 == the one call, bit for bit
    pipe2 = mk()
    out2 = torch.empty((no, 2), dtype=torch.float32, device=dev)
    q = n // 4
    for k in range(4):
        assert pipe2.run(hip.PIPE_IN_SMI_WORDS, words.data_ptr() + 4 * k * q, 0, q, out2.data_ptr() + 8 * (k * q * 3 // 2), 0) == q * 3 // 2
    torch.cuda.synchronize()
    assert torch.equal(out[:no], out2)
    # linearity on the synthetic stream: a = the stream, b = a constant offset stream; a + b stays inside 13 bits
    del out2
    a = synth.torch_smi_words(n, dev, 0, 7)
    ai = ((a >> 17) & 0x1FFF).to(torch.int64); ai = torch.where(ai >= 4096, ai - 8192, ai) // 4
    aq = ((a >> 1) & 0x1FFF).to(torch.int64); aq = torch.where(aq >= 4096, aq - 8192, aq) // 4
    bi, bq = 1500, -900

    def pack(i, q_):
        w = 0x80004000 | ((i & 0x1FFF) << 17) | ((q_ & 0x1FFF) << 1)
        return ((w + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32)
    wa, wab = pack(ai, aq), pack(ai + bi, aq + bq)
    del a, ai, aq
    ya = torch.empty((no, 2), dtype=torch.float32, device=dev)
    assert mk().run(hip.PIPE_IN_SMI_WORDS, wa, 0, n, ya, 0) == no
    assert mk().run(hip.PIPE_IN_SMI_WORDS, wab, 0, n, out, 0) == no
    torch.cuda.synchronize()
    # pipe(b) for a constant b: the DC gains of FIR and resampler legs -- take it from the run itself, far from the start
    d = (out[:no] - ya)
    dc = d[3000:3003].double().mean(0)                             # three consecutive outputs = the three polyphase legs' mean
    legs = d[3000:3000 + 3 * 1000].double().reshape(1000, 3, 2).mean(0)
    want_i, want_q = bi / 4096.0, bq / 4096.0
    assert abs(float(dc[0]) - want_i) < 2e-3 and abs(float(dc[1]) - want_q) < 2e-3       # unit DC gain of both filters, to their ripple
    dev_ = (d[3000:].double().reshape(-1, 3, 2) - legs).abs().max()
    assert float(dev_) <= 4e-6, float(dev_)                        # the difference is the same constant everywhere: linear, shift-invariant


def test_configs_3_and_4_at_baseline_size_by_properties(G, orc):
    """BASELINE.json config 3 (S1G + HiF, 2^27 samples each, FIR64 + FM demod) and one GPU's share of config 4 (32 streams x
    2^24 samples, FIR128 + 5/4) at the sizes bench.py runs them, through properties: a clean tone demodulates to its
    constant phase step everywhere (a frequency the FIR passes), on both channel types; in the 32-stream pipe every stream
    answers an impulse at its own position with the oracle's response scaled by its own amplitude and is zero elsewhere --
    no stream sees another's samples."""
    import torch
    from cariboulite_amd import hip, synth
    t = load_golden("taps.npz")
    dev = G.DEV
    # ---- config 3
    n = 1 << 27
    f0 = 137e3
    step = 2 * np.pi * f0 / 4e6
    for ch in (0, 1):
        idx = torch.arange(n, device=dev, dtype=torch.float64)
        i = torch.round(3000.0 * torch.cos(step * idx)).to(torch.int64) & 0x1FFF
        q = torch.round(3000.0 * torch.sin(step * idx)).to(torch.int64) & 0x1FFF
        a, b = (i, q) if ch == 0 else (q, i)
        w = 0x80004000 | (a << 17) | (b << 1)
        words = ((w + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32)
        del idx, i, q, a, b, w
        pipe = hip.RxPipe(1, ch, t["fir64_c3"], None, 1, 1, hip.PIPE_OUT_FM_DEMOD)
        out = torch.full((n + 8,), float("nan"), dtype=torch.float32, device=dev)
        assert pipe.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, out, 0) == n
        torch.cuda.synchronize()
        assert torch.isnan(out[n:]).all()
        err = (out[64:n].double() - step).abs().max()              # behind the FIR's transient
        assert float(err) < 2e-3, float(err)                       # 13-bit quantisation of a 3000-LSB tone: ~1/3000 rad
        assert abs(float(out[64:n].double().mean()) - step) < 1e-6
        del words, out
    # ---- config 4, one GPU's share
    ns, n4 = 32, 1 << 24
    zero = int(synth.iq_to_words([0], [0], 0)[0]) - (1 << 32)
    words = torch.full((ns, n4), zero, dtype=torch.int32, device=dev)
    pos = [1000 + 524287 * s for s in range(ns)]                   # every stream somewhere else, the last near the end
    for s in range(ns):
        wv = int(synth.iq_to_words([100 * (s + 1)], [-50 * (s + 1)], 0)[0])
        words[s, pos[s]] = wv - (1 << 32) if wv >= (1 << 31) else wv
    pipe = hip.RxPipe(ns, 0, t["fir128_c4"], t["rs_5_4"], 5, 4, hip.PIPE_OUT_IQ)
    no = pipe.out_count(n4)
    out = torch.full((ns, no + 4, 2), float("nan"), dtype=torch.float32, device=dev)
    assert pipe.run(hip.PIPE_IN_SMI_WORDS, words, n4, n4, out, no + 4) == no
    torch.cuda.synchronize()
    assert torch.isnan(out[:, no:]).all()
    for s in range(ns):
        lo = pos[s] // 4 * 4                                       # a multiple of M: the resampler's phase there is 0
        x = np.zeros((512, 2)); x[pos[s] - lo] = (100 * (s + 1) / 4096.0, -50 * (s + 1) / 4096.0)
        want = orc.Resampler(t["rs_5_4"], 5, 4).f64(orc.FIR(t["fir128_c4"]).f64(x))
        m0 = lo * 5 // 4
        got = out[s, m0:m0 + want.shape[0]].cpu().numpy()
        assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want)), s
        rest = out[s, :no].clone(); rest[m0:m0 + want.shape[0]] = 0
        assert int(torch.count_nonzero(rest)) == 0, s



def test_config1_and_pack_round_trip_at_baseline_size(G):
    """BASELINE.json config 1 at full size (2^28 samples, 2048 native chunks, one launch of each kernel), checked exactly
    against the same integer arithmetic written with torch on the device -- (w >> 17 | w >> 1) & 0x1FFF, sign-extended, / 4096 --
    and the encode -> decode round trip of the TX pack (documented layout) over 2^27 samples: unpacking the 13-bit fields
    of the packed bytes (firmware/smi_ctrl.v:194-243 byte order) gives back every sample's low 13 bits."""
    import torch
    from cariboulite_amd import hip, synth
    dev = G.DEV
    n = 1 << 28
    nch = n * 4 // 524288
    for ch in (0, 1):
        words = synth.torch_smi_words(n, dev, ch, 11 + ch)
        offs = torch.full((nch,), -9, dtype=torch.int32, device=dev)
        out = torch.empty((n, 2), dtype=torch.float32, device=dev)
        hip.smi_find_offsets(words, 4 * n, 524288, 524288, nch, offs)
        hip.smi_unpack(ch, words, 4 * n, 524288, 524288, nch, offs, hip.FORMAT_CF32, out, None)
        torch.cuda.synchronize()
        assert int(torch.count_nonzero(offs)) == 0
        for col, sh in ((0, 17), (1, 1)):
            f = ((words >> sh) & 0x1FFF).to(torch.int32)
            f = torch.where(f >= 4096, f - 8192, f).to(torch.float32) / 4096.0
            assert torch.equal(out[:, col if ch == 0 else 1 - col], f), (ch, col)     # HiF: the two fields swap roles (caribou_smi.c:342-378)
            del f
        del words, out
    m = 1 << 27
    iq = torch.randint(-32768, 32768, (m, 2), dtype=torch.int16, device=dev)
    by = torch.zeros(4 * m, dtype=torch.uint8, device=dev)
    hip.smi_pack(hip.TX_DOCUMENTED, iq, m, by)
    torch.cuda.synchronize()
    b = by.view(m, 4).to(torch.int32)
    assert bool(((b[:, 0] & 0xE0) == 0xE0).all()) and not bool((b[:, 1:] & 0x80).any())      # frame bits: 111 | 0 | 0 | 0
    i13 = ((b[:, 0] & 0x1F) << 8) | (b[:, 1] << 1) | ((b[:, 2] >> 6) & 1)
    q13 = ((b[:, 2] & 0x3F) << 7) | b[:, 3]
    assert torch.equal(i13, iq[:, 0].to(torch.int32) & 0x1FFF) and torch.equal(q13, iq[:, 1].to(torch.int32) & 0x1FFF)



def test_tx_config5_at_baseline_size_by_properties(G, orc):
    """BASELINE.json config 5 at the size it is benchmarked (2^27 messages -> 2^27 * 2/3 SMI words) through properties:
    every word carries the frame bits 111|0|0|0; decoded with torch on the device, the samples are a unit phasor through a
    unit-DC-gain resampler (|z| = 4096 to the filter's ripple and the quantiser's step wherever the phase moves slowly);
    and four calls of 2^25 messages give the one call's words up to rare one-LSB truncation flips (the fp64 phase sums are
    re-associated across calls), both pipes continuing from the same phase afterwards."""
    import torch
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    dev = G.DEV
    n = 1 << 27
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    msg = 0.05 * torch.randn(n, device=dev, generator=gen)        # slow phase: the 2/3 resampler passes the phasor almost untouched
    mk = lambda: hip.TxPipe(1, 75e3, 4e6, t["rs_2_3"], 2, 3, hip.TX_DOCUMENTED)
    p1 = mk()
    no = p1.out_count(n)
    by = torch.zeros(4 * no, dtype=torch.uint8, device=dev)
    assert p1.run(hip.TXPIPE_IN_FM_MESSAGE, msg, 0, n, by, 4 * no) == no and p1.status() == 0
    torch.cuda.synchronize()
    b = by.view(no, 4).to(torch.int32)
    assert bool(((b[:, 0] & 0xE0) == 0xE0).all()) and not bool((b[:, 1:] & 0x80).any())
    sx = lambda v: torch.where(v >= 4096, v - 8192, v)
    i13 = sx(((b[:, 0] & 0x1F) << 8) | (b[:, 1] << 1) | ((b[:, 2] >> 6) & 1)).double()
    q13 = sx(((b[:, 2] & 0x3F) << 7) | b[:, 3]).double()
    mag = torch.sqrt(i13 * i13 + q13 * q13)[64:]
    # 13 bits hold [-4096, 4095]: +4096 wraps, so a component at full scale is excluded from the magnitude property
    ok = (i13[64:].abs() < 4090) & (q13[64:].abs() < 4090)
    assert float(ok.double().mean()) > 0.9
    assert float((mag[ok] - 4096.0).abs().max()) < 24.0, float((mag[ok] - 4096.0).abs().max())
    del b, i13, q13, mag, ok
    p4 = mk()
    by4 = torch.zeros(4 * no, dtype=torch.uint8, device=dev)
    q, pos_out = n // 4, 0
    for k in range(4):
        ko = p4.out_count(q)
        assert p4.run(hip.TXPIPE_IN_FM_MESSAGE, msg.data_ptr() + 4 * k * q, 0, q, by4.data_ptr() + 4 * pos_out, 4 * ko) == ko and p4.status() == 0
        pos_out += ko
    torch.cuda.synchronize()
    assert pos_out == no
    diff = by.view(torch.int32) != by4.view(torch.int32)
    assert float(diff.double().mean()) < 1e-3                      # truncation flips only
    # ... and both pipes continue from the same phase and history: one more call each
    tail = 0.05 * torch.randn(3072 * 7, device=dev, generator=gen)
    ka = p1.out_count(tail.numel())
    ba, bb = torch.zeros(4 * ka, dtype=torch.uint8, device=dev), torch.zeros(4 * ka, dtype=torch.uint8, device=dev)
    assert p1.run(hip.TXPIPE_IN_FM_MESSAGE, tail, 0, tail.numel(), ba, 4 * ka) == ka
    assert p4.run(hip.TXPIPE_IN_FM_MESSAGE, tail, 0, tail.numel(), bb, 4 * ka) == ka
    torch.cuda.synchronize()
    ia, ib = _tx_words_to_iq(ba.cpu().numpy().view(np.uint32)), _tx_words_to_iq(bb.cpu().numpy().view(np.uint32))
    d13 = np.abs(ia - ib); d13 = np.minimum(d13, 8192 - d13)
    assert d13.max() <= 1


def test_pps_tags_at_baseline_size_by_properties(G, orc):
    """The meta plane of ONE 2^28-sample stream (config 1 / 2's size): unpack writes it, clhip_sync_tags compacts it.
      * the synthetic stream marks every 4 000 000th sample: the tags are exactly 0, 4e6, 8e6, ...;
      * a plane with ~3 % markers: the count equals torch's, the positions are strictly increasing, every position holds a 1,
        and their checksum equals the checksum of torch.nonzero; a capacity smaller than the count keeps the first `cap`."""
    import torch
    from cariboulite_amd import hip, synth
    dev = G.DEV
    n = 1 << 28
    words = synth.torch_smi_words(n, torch.device(dev), channel=0, stream=3)
    nch = n // 131072
    offs = torch.zeros(nch, dtype=torch.int32, device=dev)
    iq = torch.empty((n, 2), dtype=torch.int16, device=dev)
    meta = torch.full((n,), 0xAA, dtype=torch.uint8, device=dev)
    hip.smi_find_offsets(words, 4 * n, 524288, 524288, nch, offs)
    hip.smi_unpack(0, words, 4 * n, 524288, 524288, nch, offs, hip.FORMAT_CS16, iq, meta)
    del words, iq
    ws = torch.empty(hip.lib().clhip_sync_tags_ws_bytes(n), dtype=torch.uint8, device=dev)
    idx = torch.full((1024,), -1, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    hip.sync_tags(meta, n, idx, 1024, cnt, ws)
    torch.cuda.synchronize()
    want = list(range(0, n, 4_000_000))
    assert int(cnt.item()) == len(want) and idx.cpu().numpy()[:len(want)].tolist() == want

    g = torch.Generator(device=dev); g.manual_seed(9)
    meta = (torch.rand(n, device=dev, generator=g) < 0.03).to(torch.uint8) * 1 + \
           (torch.rand(n, device=dev, generator=g) < 0.01).to(torch.uint8) * 2          # values 0, 1, 2, 3: only 1 is a marker
    k = int((meta == 1).sum().item())
    idx = torch.full((k + 16,), -1, dtype=torch.int32, device=dev)
    hip.sync_tags(meta, n, idx, k + 16, cnt, ws)
    torch.cuda.synchronize()
    assert int(cnt.item()) == k
    got = idx[:k].to(torch.int64)
    assert bool((idx[k:] == -1).all())
    assert bool((got[1:] > got[:-1]).all()) and int(got[0]) >= 0 and int(got[-1]) < n
    assert bool((meta[got] == 1).all())
    ref = torch.nonzero(meta == 1).flatten()
    assert bool((ref == got).all())
    del ref
    cap = 100_000
    idx2 = torch.full((cap + 16,), -1, dtype=torch.int32, device=dev)
    hip.sync_tags(meta, n, idx2, cap, cnt, ws)
    torch.cuda.synchronize()
    assert int(cnt.item()) == k and bool((idx2[:cap] == idx[:cap]).all()) and bool((idx2[cap:] == -1).all())
