"""-m gpu: the fused RX pipe (unpack -> FIR -> resample | FM demod) through the C-ABI vs the
fp64 oracle (tolerance 1e-5 of the stage peak, BASELINE.json north_star), vs the committed
scipy fixtures, and vs the generic kernels (bit-identical by construction)."""
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


CONFIGS = {
    "c2": dict(fir="fir64_c2", rs="rs_3_2", L=3, M=2, mode=0),
    "c3": dict(fir="fir64_c3", rs=None, L=1, M=1, mode=1),
    "c4": dict(fir="fir128_c4", rs="rs_5_4", L=5, M=4, mode=0),
    "f64": dict(fir="fir64_c2", rs=None, L=1, M=1, mode=0),
    "f128": dict(fir="fir128_c4", rs=None, L=1, M=1, mode=0),
    "d2": dict(fir="fir64_c2", rs="rs_1_2", L=1, M=2, mode=0),       # decimating shapes
    "d4": dict(fir="fir64_c2", rs="rs_1_4", L=1, M=4, mode=0),
    "d34": dict(fir="fir64_c2", rs="rs_3_4", L=3, M=4, mode=0),
}


def make_pipe(cfg, n_streams=1, channel=0):
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    c = CONFIGS[cfg]
    return hip.RxPipe(n_streams, channel, t[c["fir"]], t[c["rs"]] if c["rs"] else None, c["L"], c["M"], c["mode"])


def oracle_chain(orc, cfg, x_cf32):
    """fp64 oracle of the float stages on CF32 input (streaming objects returned for chunk tests)."""
    t = load_golden("taps.npz")
    c = CONFIGS[cfg]
    y = orc.FIR(t[c["fir"]]).f64(x_cf32)
    if c["mode"] == 1:
        return orc.fm_demod_f64(y)[0]
    if c["rs"]:
        return orc.Resampler(t[c["rs"]], c["L"], c["M"]).f64(y)
    return y


def check_fm(G, orc, got, words_u8, n, channel, y_oracle):
    """FM demod parity.  atan2 is ill-conditioned where |y| -> 0 (a 1e-7 error of the fp32 FIR
    output becomes an arbitrary angle error), so the 1e-5 bound is applied (a) to the demod
    stage in isolation -- oracle demod of the SAME fp32 FIR outputs the fused kernel produced
    (the FIR-only config runs the identical FIR code) -- and (b) end to end on the
    magnitude-weighted phasor, which is well conditioned."""
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    fir_only = hip.RxPipe(1, channel, t["fir64_c3"], None, 1, 1, hip.PIPE_OUT_IQ)
    y_gpu = run_pipe(G, fir_only, words_u8, n)
    want_iso = orc.fm_demod_f64(y_gpu.astype(np.float64))[0]
    d = np.abs(got - want_iso)
    d = np.minimum(d, 2 * np.pi - d)                      # +-pi wrap is the same angle
    assert np.max(d) <= TOL * np.pi, np.max(d)
    want = orc.fm_demod_f64(y_oracle)[0]
    mag = np.hypot(y_oracle[:, 0], y_oracle[:, 1])
    mag_prev = np.concatenate([[0.0], mag[:-1]])
    w = np.minimum(mag, mag_prev)
    err = np.abs(np.exp(1j * got) - np.exp(1j * want)) * w
    assert np.max(err) <= TOL * np.max(mag), (np.max(err), np.max(mag))


def run_pipe(G, pipe, words_u8, n, in_kind=None, chunks=None):
    import torch
    from cariboulite_amd import hip
    in_kind = hip.PIPE_IN_SMI_WORDS if in_kind is None else in_kind
    d_in = torch.from_numpy(np.ascontiguousarray(words_u8)).to(G.DEV)
    cols = 1 if pipe.out_mode == hip.PIPE_OUT_FM_DEMOD else 2
    outs = []
    pos = 0
    esz = 8 if in_kind == hip.PIPE_IN_CF32 else 4
    for cn in (chunks or [n]):
        no = pipe.out_count(cn)
        out = torch.full((no + 8, cols), float("nan"), dtype=torch.float32, device=G.DEV)
        got = pipe.run(in_kind, d_in.data_ptr() + pos * esz, 0, cn, out, 0)
        assert got == no
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert np.all(np.isnan(o[no:])), "wrote past the end"
        outs.append(o[:no])
        pos += cn
    return np.concatenate(outs)


@pytest.mark.parametrize("cfg", list(CONFIGS))
@pytest.mark.parametrize("channel", [0, 1])
def test_fused_vs_oracle(G, orc, cfg, channel):
    from cariboulite_amd import hip, synth
    n = 3 * 4088 + 1234 if cfg != "c4" else 2 * 4088 + 600   # several tiles + ragged tail
    n -= n % 4
    b, i, q = synth.smi_stream_bytes(n, channel, stream=11)
    pipe = make_pipe(cfg, 1, channel)
    assert pipe.uses_fused(n)
    got = run_pipe(G, pipe, b, n)
    offs, iq, _ = orc.rx_data_analyze(channel, b)
    if cfg == "c3":
        y = orc.FIR(load_golden("taps.npz")["fir64_c3"]).f64(orc.cs16_to_cf32(iq[:n]))
        check_fm(G, orc, got[:, 0], b, n, channel, y)
        return
    want = oracle_chain(orc, cfg, orc.cs16_to_cf32(iq[:n]))
    want = want.reshape(got.shape)
    peak = np.max(np.abs(want))
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= TOL * peak, (np.max(np.abs(got - want)), peak)


def test_c2_vs_scipy_fixture(G, orc):
    from cariboulite_amd import hip
    g = load_golden("dsp_float.npz")
    pipe = make_pipe("c2")
    got = run_pipe(G, pipe, g["bytes"], 8192)
    want = g["fir64_c2__rs_3_2"]
    assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))
    pipe = make_pipe("c3")
    got = run_pipe(G, pipe, g["bytes"], 8192)[:, 0]
    check_fm(G, orc, got, g["bytes"], 8192, 0, g["fir64_c3__y"])
    big = np.hypot(g["fir64_c3__y"][:, 0], g["fir64_c3__y"][:, 1]) > 0.05      # well-conditioned samples
    big[1:] &= big[:-1]
    assert np.max(np.abs(got - g["fir64_c3__fm_demod"])[big]) <= TOL * np.pi
    pipe = make_pipe("c4")
    got = run_pipe(G, pipe, g["bytes"], 8192)
    want = g["fir128_c4__rs_5_4"]
    assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))


@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_fused_vs_generic_second_implementation(G, orc, cfg, monkeypatch):
    """Two implementations of one spec.  The direct-form fused kernel uses the generic kernels'
    summation order: identical bits.  The default 2-parallel fast FIR re-associates the sums:
    equal to rounding (a few 1e-7 of the peak)."""
    from cariboulite_amd import hip, synth
    n = 2 * 4088 + 36
    b, _, _ = synth.smi_stream_bytes(n, 0, stream=2)
    p1, p2 = make_pipe(cfg), make_pipe(cfg)
    p2.force_generic(True)
    assert p1.uses_fused(n) and not p2.uses_fused(n)
    a, g = run_pipe(G, p1, b, n), run_pipe(G, p2, b, n)
    if CONFIGS[cfg]["mode"] == 1:
        d = np.abs(a - g); d = np.minimum(d, 2 * np.pi - d)
        big = np.abs(orc.FIR(load_golden("taps.npz")["fir64_c3"]).f64(
            orc.cs16_to_cf32(orc.rx_data_analyze(0, b)[1][:n]))).max(axis=1) > 0.02
        big[1:] &= big[:-1]
        assert np.max(d[big]) <= 2e-5
    else:
        assert np.max(np.abs(a - g)) <= 2e-6 * np.max(np.abs(g))


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4", "d2", "d4", "d34"])
def test_streaming_chunks_equal_one_shot(G, orc, cfg):
    """History carried across calls: chunked == one big call (SURVEY.md section 5 checkpoint row)."""
    from cariboulite_amd import hip, synth
    n = 4 * 4096
    b, _, _ = synth.smi_stream_bytes(n, 0, stream=5)
    one = run_pipe(G, make_pipe(cfg), b, n)
    chunks = [4096, 8, 4088, 4, 2048, 6140]          # all multiples of M=4/2: fused path, same lane parity
    assert sum(chunks) == n
    p = make_pipe(cfg)
    many = run_pipe(G, p, b, n, chunks=chunks)
    assert np.array_equal(one, many)                 # bit-identical: state is exact, arithmetic identical
    # ragged chunk lengths (odd phases) go through the generic / direct-form kernels: same spec, equal to rounding
    chunks2 = [1, 2, 3, 4091, 5, 4090, 8184, 8]
    assert sum(chunks2) == n
    many2 = run_pipe(G, make_pipe(cfg), b, n, chunks=chunks2)
    if CONFIGS[cfg]["mode"] == 1:
        d = np.abs(one - many2); d = np.minimum(d, 2 * np.pi - d)
        assert np.mean(d > 2e-5) < 0.05               # ill-conditioned only where |y| ~ 0
    else:
        assert np.max(np.abs(one - many2)) <= 2e-6 * np.max(np.abs(one))


def test_direct_form_is_bit_identical_to_generic(G, orc):
    """CLHIP_FFA=0 (direct-form FIR) keeps the generic kernels' summation order: identical bits.
    Run in a child process because the variant is latched from the environment at first use."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from cariboulite_amd import hip, synth
        t = np.load(%r)
        n = 2 * 4088 + 36
        b = synth.smi_stream_bytes(n, 0, stream=2)[0]
        d = torch.from_numpy(b.copy()).to("cuda:0")
        outs = []
        for gen in (False, True):
            p = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, 0)
            p.force_generic(gen)
            o = torch.zeros((p.out_count(n), 2), dtype=torch.float32, device="cuda:0")
            p.run(hip.PIPE_IN_SMI_WORDS, d, 0, n, o, 0); torch.cuda.synchronize()
            outs.append(o.cpu().numpy())
        assert np.array_equal(outs[0], outs[1]), float(np.max(np.abs(outs[0] - outs[1])))
        print("bit-identical")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
            os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "taps.npz"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CLHIP_FFA="0"), capture_output=True, text=True)
    assert r.returncode == 0 and "bit-identical" in r.stdout, r.stderr[-2000:]


def test_multi_stream_and_input_kinds(G, orc):
    import torch
    from cariboulite_amd import hip, synth
    n, ns = 4088 + 512, 5
    t = load_golden("taps.npz")
    words = np.stack([synth.smi_stream_bytes(n, 0, stream=s)[0].view(np.uint32) for s in range(ns)])
    d_in = torch.from_numpy(words.view(np.int32)).to(G.DEV)
    pipe = hip.RxPipe(ns, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, 0)
    no = pipe.out_count(n)
    out = torch.zeros((ns, no + 4, 2), dtype=torch.float32, device=G.DEV)
    assert pipe.run(hip.PIPE_IN_SMI_WORDS, d_in, n, n, out, no + 4) == no
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    single = [run_pipe(G, make_pipe("c2"), words[s].view(np.uint8), n) for s in range(ns)]
    for s in range(ns):
        assert np.array_equal(got[s, :no], single[s])
    # CS16 and CF32 inputs give the same bits as raw words (exact conversions)
    _, iq, _ = orc.rx_data_analyze(0, words[0].view(np.uint8))
    cs16 = iq[:n].copy()
    a = run_pipe(G, make_pipe("c2"), cs16.view(np.uint8), n, in_kind=hip.PIPE_IN_CS16)
    assert np.array_equal(a, single[0])
    cf32 = orc.cs16_to_cf32(cs16)
    b = run_pipe(G, make_pipe("c2"), cf32.view(np.uint8), n, in_kind=hip.PIPE_IN_CF32)
    assert np.max(np.abs(b - single[0])) <= 1e-6 * np.max(np.abs(single[0]))


@pytest.mark.parametrize("channel", [0, 1])
def test_c4_share_of_one_gpu_32_streams(G, orc, channel):
    """Config 4 as a workload: the 32 streams one GPU owns (256 streams / 8 GPUs), FIR128 + 5/4, ONE pipe and one
    launch per call, ragged tail, two calls (carried history) -- every stream against the fp64 oracle."""
    import torch
    from cariboulite_amd import hip, synth
    t = load_golden("taps.npz")
    ns = 32
    calls = [3 * 4088 + 1000, 2 * 4088 + 604]                  # multiples of M = 4: both calls start on phase 0
    n = sum(calls)
    words = np.stack([synth.smi_stream_bytes(n, channel, stream=200 + s)[0].view(np.uint32) for s in range(ns)])
    d_in = torch.from_numpy(words.view(np.int32)).to(G.DEV)
    pipe = hip.RxPipe(ns, channel, t["fir128_c4"], t["rs_5_4"], 5, 4, hip.PIPE_OUT_IQ)
    got, pos = [], 0
    for cn in calls:
        assert pipe.uses_fused(cn)
        no = pipe.out_count(cn)
        out = torch.full((ns, no + 4, 2), float("nan"), dtype=torch.float32, device=G.DEV)
        assert pipe.run(hip.PIPE_IN_SMI_WORDS, d_in.data_ptr() + 4 * pos, n, cn, out, no + 4) == no
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert np.all(np.isnan(o[:, no:]))
        got.append(o[:, :no])
        pos += cn
    got = np.concatenate(got, axis=1)
    for s in range(ns):
        _, iq, _ = orc.rx_data_analyze(channel, words[s].view(np.uint8))
        want = oracle_chain(orc, "c4", orc.cs16_to_cf32(iq[:n]))
        assert got[s].shape == want.shape
        assert np.max(np.abs(got[s] - want)) <= TOL * np.max(np.abs(want)), s


@pytest.mark.parametrize("ntaps,rs,L,M,mode", [(33, "rs_3_2", 3, 2, 0), (48, None, 1, 1, 1), (100, "rs_3_2", 3, 2, 0),
                                                (97, None, 1, 1, 1), (20, "rs_5_4", 5, 4, 0), (1, "rs_3_4", 3, 4, 0),
                                                (127, "rs_5_4", 5, 4, 0), (65, None, 1, 1, 0)])
def test_arbitrary_tap_counts_stay_fused(G, orc, ntaps, rs, L, M, mode):
    """Any FIR=<ntaps> stream kwarg up to 128 taps (odd lengths included) runs the fused kernel: the taps are
    zero-padded to the next instantiated length.  Against the fp64 oracle of the UNPADDED filter, chunked."""
    from cariboulite_amd import hip, soapy, synth
    t = load_golden("taps.npz")
    h = soapy.design_lowpass(ntaps, 100e3 if mode == 1 else 900e3, 4e6) if ntaps > 1 else np.ones(1, np.float32)
    n = 3 * 4088 + 1236
    b, _, _ = synth.smi_stream_bytes(n, 0, stream=21)
    pipe = hip.RxPipe(1, 0, h, t[rs] if rs else None, L, M, mode)
    assert pipe.uses_fused(n)
    got = run_pipe(G, pipe, b, n, chunks=[2 * 4088, n - 2 * 4088])          # second call: carried (longer) history
    _, iq, _ = orc.rx_data_analyze(0, b)
    y = orc.FIR(h).f64(orc.cs16_to_cf32(iq[:n]))
    if mode == 1:
        # demod stage in isolation on the GPU's own FIR outputs + the magnitude-weighted end-to-end bound (see check_fm)
        fir_only = hip.RxPipe(1, 0, h, None, 1, 1, hip.PIPE_OUT_IQ)
        y_gpu = run_pipe(G, fir_only, b, n)
        assert np.max(np.abs(y_gpu - y)) <= TOL * np.max(np.abs(y))
        d = np.abs(got[:, 0] - orc.fm_demod_f64(y_gpu.astype(np.float64))[0]); d = np.minimum(d, 2 * np.pi - d)
        assert np.max(d) <= TOL * np.pi
        return
    want = orc.Resampler(t[rs], L, M).f64(y) if rs else y
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))


def test_linearity_and_impulse_full_size(G, orc):
    """Size-independent properties at a bench-like size: an impulse returns the taps;
    response to a constant settles at the DC gain."""
    import torch
    from cariboulite_amd import hip, synth
    t = load_golden("taps.npz")
    n = 1 << 22
    i = np.zeros(n, np.int64); q = np.zeros(n, np.int64)
    i[1000] = 4095; q[1000] = -4096
    i[2_000_000] = 2048
    words = synth.iq_to_words(i, q, 0)
    pipe = hip.RxPipe(1, 0, t["fir64_c2"], None, 1, 1, 0)
    got = run_pipe(G, pipe, words.view(np.uint8), n)
    h = t["fir64_c2"].astype(np.float64)
    assert np.max(np.abs(got[1000:1064, 0] - h * (4095 / 4096))) < 1e-6
    assert np.max(np.abs(got[1000:1064, 1] - h * (-1.0))) < 1e-6
    assert np.max(np.abs(got[2_000_000:2_000_064, 0] - h * 0.5)) < 1e-6
    nz = np.zeros(n, bool); nz[1000:1064] = True; nz[2_000_000:2_000_064] = True
    assert np.all(got[~nz] == 0)


def test_time_sliced_single_stream_equals_one_pipe(G, orc):
    """one long stream cut into contiguous time slices, each processed by its own pipe
    (its own GPU in production) after priming with the halo before the slice: bit-identical to one pipe."""
    import torch
    from cariboulite_amd import hip, shard, synth
    t = load_golden("taps.npz")
    n = 1_000_000
    words = synth.torch_smi_words(n, G.DEV, 0, 9)
    one = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    n_out = one.out_count(n)
    ref = torch.empty((n_out, 2), dtype=torch.float32, device=G.DEV)
    assert one.run(hip.PIPE_IN_SMI_WORDS, words, 0, n, ref, 0) == n_out
    for world in (2, 3, 8):
        got = torch.zeros_like(ref)
        scratch = torch.empty((1024, 2), dtype=torch.float32, device=G.DEV)
        pos = 0
        for (a, b) in shard.time_slices(n, world, 2):        # lcm(M = 2, 2)
            p = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
            assert a * 3 % 2 == 0 and pos == a * 3 // 2
            k = shard.run_time_slice(p, hip.PIPE_IN_SMI_WORDS, words, a, b, got[pos:], None, scratch)
            pos += k
        assert pos == n_out
        torch.cuda.synchronize()
        assert torch.equal(got, ref), world


def test_streams_of_one_pipe_advance_independently(G, orc):
    """clhip_rx_pipe_epoch_begin / _run_range / _epoch_end (what cl_group_readStream launches through): the streams of ONE 5-stream
    pipe driven by range runs -- all together, in two runs around a skipped stream, one alone from int16 samples, with different
    input counts -- equal five single-stream pipes driven with the same inputs bit for bit, call after call; a skipped stream keeps
    its state; a range over streams on different polyphase phases is refused, and so is the whole-pipe call once the streams have
    drifted apart."""
    import torch
    from cariboulite_amd import hip, synth
    t = load_golden("taps.npz")
    ns, n = 5, 3 * 4096 + 40
    multi = hip.RxPipe(ns, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    lone = [hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ) for _ in range(ns)]
    stride_in, stride_out = n + 16, n * 3 // 2 + 32
    def words(call):
        return torch.stack([torch.from_numpy(np.concatenate([synth.smi_stream_bytes(n, 0, stream=50 + s, n0=call * n)[0], np.zeros(64, np.uint8)]).view(np.int32).copy()) for s in range(ns)]).to(G.DEV)
    def cs16_of(w_row, count):
        o = torch.zeros((count + 2, 2), dtype=torch.int16, device=G.DEV)
        offs = torch.zeros(1, dtype=torch.int32, device=G.DEV)
        hip.smi_unpack(0, w_row, 4 * count, 4 * count, 4 * count, 1, offs, hip.FORMAT_CS16, o)
        return o
    # call plans: list of (first, count, kind, n_in); streams not named are skipped in that call
    plans = [[(0, 5, "words", n)],
             [(0, 2, "words", n), (3, 2, "words", n)],                      # stream 2 skipped
             [(0, 1, "words", n), (1, 1, "cs16", n), (2, 3, "words", n)],
             [(0, 5, "words", n)],
             [(0, 4, "words", n), (4, 1, "words", 2 * 1024)],                # stream 4 takes fewer inputs: still phase 0 (a multiple of 4)
             [(0, 5, "words", n)]]
    for call, plan in enumerate(plans):
        w = words(call)
        assert w.shape[1] >= stride_in
        out = torch.full((ns, stride_out, 2), float("nan"), dtype=torch.float32, device=G.DEV)
        ref = torch.full((ns, stride_out, 2), float("nan"), dtype=torch.float32, device=G.DEV)
        multi.epoch_begin()
        for first, count, kind, n_in in plan:
            if kind == "words":
                got = multi.run_range(first, count, hip.PIPE_IN_SMI_WORDS, w[first], w.shape[1], n_in, out[first], stride_out)
            else:
                got = multi.run_range(first, count, hip.PIPE_IN_CS16, cs16_of(w[first], n_in), 0, n_in, out[first], stride_out)
            for s in range(first, first + count):
                assert got == lone[s].out_count(n_in)
                if kind == "words":
                    assert lone[s].run(hip.PIPE_IN_SMI_WORDS, w[s], 0, n_in, ref[s], 0) == got
                else:
                    assert lone[s].run(hip.PIPE_IN_CS16, cs16_of(w[s], n_in), 0, n_in, ref[s], 0) == got
        multi.epoch_end()
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == ref.cpu().numpy().tobytes(), call      # NaN where nothing was written, on both sides
    assert [multi.stream_total(s) for s in range(ns)] == [6 * n, 6 * n, 5 * n, 6 * n, 5 * n + 2048]
    # drifted apart: the whole-pipe call refuses; a range over two phase classes refuses; an epoch cannot be opened twice
    w = words(9)
    out = torch.zeros((ns, stride_out, 2), dtype=torch.float32, device=G.DEV)
    with pytest.raises(RuntimeError, match="no longer move as one"):
        multi.run(hip.PIPE_IN_SMI_WORDS, w, w.shape[1], n, out, stride_out)
    multi.epoch_begin()
    with pytest.raises(RuntimeError):
        multi.epoch_begin()
    assert multi.run_range(0, 1, hip.PIPE_IN_SMI_WORDS, w[0], 0, 6, out[0], 0) == 9        # 6 inputs: stream 0 is now off the others' phase class (6 mod 4)
    with pytest.raises(RuntimeError, match="already ran"):
        multi.run_range(0, 2, hip.PIPE_IN_SMI_WORDS, w[0], w.shape[1], n, out[0], stride_out)
    multi.epoch_end()
    multi.epoch_begin()
    with pytest.raises(RuntimeError, match="different polyphase phases"):
        multi.run_range(0, 2, hip.PIPE_IN_SMI_WORDS, w[0], w.shape[1], n, out[0], stride_out)
    multi.epoch_end()
    torch.cuda.synchronize()


def test_a_range_run_of_the_open_epoch_can_be_taken_back(G, orc):
    """clhip_rx_pipe_unrun_stream (what a stream group does with results it computed ahead of a client who then went another way): in
    an epoch all four streams run; the run of stream 1 is taken back and made again with OTHER input, that of stream 2 is taken back
    and not made again -- afterwards every stream continues exactly like a lone pipe that was given what finally counted."""
    import torch
    from cariboulite_amd import hip, synth
    t = load_golden("taps.npz")
    ns, n = 4, 2 * 4096
    multi = hip.RxPipe(ns, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    lone = [hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ) for _ in range(ns)]
    stride_out = n * 3 // 2 + 32
    def words(call):
        return torch.stack([torch.from_numpy(np.concatenate([synth.smi_stream_bytes(n, 0, stream=70 + s, n0=call * n)[0], np.zeros(64, np.uint8)]).view(np.int32).copy()) for s in range(ns)]).to(G.DEV)
    def epoch(w_multi, counted, unrun=(), rerun=None):
        """counted[s] = the words row stream s finally consumed in this epoch (None: nothing)"""
        out = torch.full((ns, stride_out, 2), float("nan"), dtype=torch.float32, device=G.DEV)
        ref = torch.full((ns, stride_out, 2), float("nan"), dtype=torch.float32, device=G.DEV)
        multi.epoch_begin()
        multi.run_range(0, ns, hip.PIPE_IN_SMI_WORDS, w_multi[0], w_multi.shape[1], n, out[0], stride_out)
        for s in unrun:
            multi.unrun_stream(s, n)
            out[s] = float("nan")
        if rerun is not None:
            s, row = rerun
            multi.run_range(s, 1, hip.PIPE_IN_SMI_WORDS, row, 0, n, out[s], 0)
        multi.epoch_end()
        for s in range(ns):
            if counted[s] is not None:
                lone[s].run(hip.PIPE_IN_SMI_WORDS, counted[s], 0, n, ref[s], 0)
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == ref.cpu().numpy().tobytes()
    w0, w1, w2, other = words(0), words(1), words(2), words(7)
    epoch(w0, [w0[s] for s in range(ns)])
    epoch(w1, [w1[0], other[1], None, w1[3]], unrun=(1, 2), rerun=(1, other[1]))
    assert [multi.stream_total(s) for s in range(ns)] == [2 * n, 2 * n, n, 2 * n]
    epoch(w2, [w2[s] for s in range(ns)])                  # ... and all of them go on from the state that counted
    multi.epoch_begin()
    with pytest.raises(RuntimeError, match="no run of the open epoch"):
        multi.unrun_stream(0, n)
    multi.epoch_end()
    torch.cuda.synchronize()


def test_outputs_do_not_depend_on_input_kind_chunking_or_what_lies_behind_the_input(G):
    """Three properties the host side relies on when it decides HOW a call's samples reach the pipe (one read() or several, raw words
    or int16 samples, a buffer that ends with the call or a FIFO slot with the next batch behind it): the fused kernels' outputs are
    bit-identical (a) from raw words and from the int16 samples unpacked from them, (b) in one run and in two runs that split the
    input anywhere (a multiple of 2 M), (c) whatever the words behind the run's last input are."""
    import torch
    from cariboulite_amd import hip, synth
    t = load_golden("taps.npz")
    n = 131072
    b = synth.smi_stream_bytes(n + 64, 0, stream=5)[0]
    w = torch.from_numpy(b.view(np.int32).copy()).to(G.DEV)
    def pipe():
        return hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    def cs16_of(wr, count):
        o = torch.zeros((count + 2, 2), dtype=torch.int16, device=G.DEV)
        offs = torch.zeros(1, dtype=torch.int32, device=G.DEV)
        hip.smi_unpack(0, wr, 4 * count, 4 * count, 4 * count, 1, offs, hip.FORMAT_CS16, o)
        return o
    so = n * 3 // 2 + 64
    ref = torch.zeros((so, 2), device=G.DEV)
    g = pipe().run(hip.PIPE_IN_SMI_WORDS, w, 0, n, ref, 0)
    out = torch.zeros((so, 2), device=G.DEV)
    assert pipe().run(hip.PIPE_IN_CS16, cs16_of(w, n), 0, n, out, 0) == g                       # (a)
    torch.cuda.synchronize()
    assert torch.equal(out[:g], ref[:g])
    for first in (n // 2, n // 4, n - 4096, 4):                                                 # (b)
        p = pipe(); out.zero_()
        g1 = p.run(hip.PIPE_IN_SMI_WORDS, w, 0, first, out, 0)
        g2 = p.run(hip.PIPE_IN_SMI_WORDS, w[first:], 0, n - first, out[g1:], 0)
        torch.cuda.synchronize()
        assert g1 + g2 == g and torch.equal(out[:g], ref[:g]), first
    for m in (n, 65536, 1001, n - 3):                                                           # (c)
        wz = w.clone(); wz[m:] = 0
        oa, ob = torch.zeros((so, 2), device=G.DEV), torch.zeros((so, 2), device=G.DEV)
        ga = pipe().run(hip.PIPE_IN_SMI_WORDS, w, 0, m, oa, 0)
        gb = pipe().run(hip.PIPE_IN_SMI_WORDS, wz, 0, m, ob, 0)
        torch.cuda.synchronize()
        assert ga == gb and torch.equal(oa[:ga], ob[:gb]), m
