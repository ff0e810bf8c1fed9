"""-m gpu: integer stages through the C-ABI vs the oracle and the reference fixtures (bit-exact)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    from cariboulite_amd import hip
    import gpu_util
    assert torch.cuda.is_available()
    arch = hip.require_gpu()
    assert arch.startswith("gfx950"), arch
    return gpu_util


def _rx_names():
    return [str(n) for n in load_golden("smi_rx_cases.npz")["names"]]


@pytest.mark.parametrize("name", _rx_names())
def test_unpack_vs_reference_fixture(G, orc, name):
    from cariboulite_amd import hip
    g = load_golden("smi_rx_cases.npz")
    buf = g[f"{name}__bytes"]
    for ch, cn in ((0, "s1g"), (1, "hif")):
        offs, iq, meta = G.gpu_rx_data_analyze(ch, buf)
        assert int(offs[0]) == int(g[f"{name}__offs_{cn}"]) == int(g[f"{name}__find"])
        if int(offs[0]) >= 0:
            assert np.array_equal(iq, g[f"{name}__iq_{cn}"])      # incl. untouched sentinel slots
            assert np.array_equal(meta, g[f"{name}__meta_{cn}"])
        else:
            assert np.all(iq == -21846) and np.all(meta == 0xAA)


def test_unpack_formats_vs_oracle(G, orc):
    from cariboulite_amd import hip, synth
    b, i, q = synth.smi_stream_bytes(10007, 0, stream=3)
    want_offs, want_iq, want_meta = orc.rx_data_analyze(0, b)
    n = 10007
    for fmt, conv in ((hip.FORMAT_CF32, orc.cs16_to_cf32), (hip.FORMAT_CF64, orc.cs16_to_cf64),
                      (hip.FORMAT_CS8, orc.cs16_to_cs8), (hip.FORMAT_CS16, lambda a: a)):
        offs, out, meta = G.gpu_rx_data_analyze(0, b, fmt)
        assert offs[0] == 0
        assert np.array_equal(out[:n], conv(want_iq[:n]))
        assert np.array_equal(meta[:n], want_meta[:n])
    # misaligned stream (extrapolated sample must be converted too)
    bm = np.concatenate([np.zeros(5, np.uint8), b])[:4 * 2000]
    wo, wi, wm = orc.rx_data_analyze(1, bm)
    assert wo == 5
    offs, out, meta = G.gpu_rx_data_analyze(1, bm, hip.FORMAT_CF32)
    nvalid = (bm.size - 4 * 2) // 4 + 1
    assert offs[0] == 5 and np.array_equal(out[:nvalid], orc.cs16_to_cf32(wi[:nvalid]))
    assert np.all(np.isnan(out[nvalid:]))


def test_unpack_chunked_like_smi_read(G, orc):
    """The chunk loop of caribou_smi_read (caribou_smi.c:643-679) as ONE batched launch."""
    from cariboulite_amd import hip
    g = load_golden("smi_read_cases.npz")
    for name in ("five_chunks_aligned", "five_chunks_hif", "chunk2_misaligned_by_2", "stream_offset_6"):
        ch, n, batch = [int(v) for v in g[f"{name}__args"]]
        stream = g[f"{name}__bytes"][: 4 * n]
        n_chunks = -(-stream.size // batch)
        offs, iq, meta = G.gpu_rx_data_analyze(ch, stream, chunk_len=batch, n_chunks=n_chunks)
        ret, wiq, wmeta = orc.smi_read(ch, stream, n, batch)
        assert ret == int(g[f"{name}__ret"]) and np.all(offs >= 0)
        assert np.array_equal(iq[: n + 2], g[f"{name}__iq"])
        assert np.array_equal(meta[: n + 2], g[f"{name}__meta"])


def test_find_offsets_unaligned_base_and_failure(G, orc):
    import torch
    from cariboulite_amd import hip, synth
    b, _, _ = synth.smi_stream_bytes(3000, 0)
    for lead in (1, 2, 3):
        for off in (0, 3, 9, 1001):
            buf = np.concatenate([np.full(off, 0xFF, np.uint8), b])[:8000]
            d = G.dev_bytes(np.concatenate([np.zeros(lead, np.uint8), buf]))
            offs = torch.full((1,), -7, dtype=torch.int32, device=G.DEV)
            hip.smi_find_offsets(d.data_ptr() + lead, buf.size, buf.size, buf.size, 1, offs)
            torch.cuda.synchronize()
            assert int(offs[0]) == orc.find_buffer_offset(buf) == off
    bad = np.zeros(524288, np.uint8)
    offs, iq, meta = G.gpu_rx_data_analyze(0, bad)
    assert offs[0] == -1 == orc.find_buffer_offset(bad)


def test_native_batch_many_chunks_random_offsets(G, orc):
    """64 native-size chunks (caribou_smi.c:78), random per-chunk misalignment, one launch."""
    from cariboulite_amd import hip, synth
    rng = np.random.default_rng(5)
    nb = 65536
    chunks = []
    want_offs = []
    for c in range(24):
        w = synth.iq_to_words(rng.integers(-4096, 4096, nb // 4), rng.integers(-4096, 4096, nb // 4), 0,
                              rng.integers(0, 2, nb // 4)).view(np.uint8)
        o = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 6, 11, 250]))
        c_bytes = np.concatenate([np.zeros(o, np.uint8), w])[:nb]
        chunks.append(c_bytes); want_offs.append(o)
    stream = np.concatenate(chunks)
    offs, iq, meta = G.gpu_rx_data_analyze(0, stream, chunk_len=nb, n_chunks=24)
    ret, wiq, wmeta = orc.smi_read(0, stream, stream.size // 4, nb)
    assert offs.tolist() == want_offs and ret == stream.size // 4
    assert np.array_equal(iq, wiq) and np.array_equal(meta, wmeta)


def test_pack_and_conversions(G, orc):
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(8)
    n = 70001
    iq = rng.integers(-32768, 32768, (n, 2)).astype(np.int16)
    iq[:4] = [[4095, -4096], [-4096, 4095], [0, 0], [-1, 1]]
    d_iq = torch.from_numpy(iq).to(G.DEV)
    for mode in (hip.TX_DOCUMENTED, hip.TX_AS_WRITTEN):
        out = torch.zeros(4 * n, dtype=torch.uint8, device=G.DEV)
        hip.smi_pack(mode, d_iq, n, out)
        assert np.array_equal(out.cpu().numpy(), orc.generate_data(iq, mode))
    g = load_golden("smi_tx_as_written.npz")
    out = torch.zeros(4 * 256, dtype=torch.uint8, device=G.DEV)
    hip.smi_pack(hip.TX_AS_WRITTEN, torch.from_numpy(g["iq"]).to(G.DEV), 256, out)
    assert np.array_equal(out.cpu().numpy(), g["bytes"])       # the reference's own output
    # CS16 -> fmt -> CS16
    f32 = torch.empty((n, 2), dtype=torch.float32, device=G.DEV)
    hip.convert_from_cs16(d_iq, n, hip.FORMAT_CF32, f32)
    assert np.array_equal(f32.cpu().numpy(), orc.cs16_to_cf32(iq))
    f64 = torch.empty((n, 2), dtype=torch.float64, device=G.DEV)
    hip.convert_from_cs16(d_iq, n, hip.FORMAT_CF64, f64)
    assert np.array_equal(f64.cpu().numpy(), orc.cs16_to_cf64(iq))
    s8 = torch.empty((n, 2), dtype=torch.int8, device=G.DEV)
    hip.convert_from_cs16(d_iq, n, hip.FORMAT_CS8, s8)
    assert np.array_equal(s8.cpu().numpy(), orc.cs16_to_cs8(iq))
    # TX direction incl. truncation, wrap and out-of-range behaviour of the x86 reference build
    x = (rng.standard_normal((n, 2)) * 3).astype(np.float32)
    x[:6] = [[0.99999 / 4096, -0.99999 / 4096], [7.99999, -8.0], [8.0, 1e9], [-1e9, np.nan], [np.inf, -np.inf], [1e-30, -1e-30]]
    back = torch.empty((n, 2), dtype=torch.int16, device=G.DEV)
    hip.convert_to_cs16(torch.from_numpy(x).to(G.DEV), hip.FORMAT_CF32, n, back)
    assert np.array_equal(back.cpu().numpy(), orc.cf32_to_cs16(x))
    xd = x.astype(np.float64) * 1.0000001
    hip.convert_to_cs16(torch.from_numpy(xd).to(G.DEV), hip.FORMAT_CF64, n, back)
    assert np.array_equal(back.cpu().numpy(), orc.cf64_to_cs16(xd))
    i8 = rng.integers(-128, 128, (n, 2)).astype(np.int8)
    hip.convert_to_cs16(torch.from_numpy(i8).to(G.DEV), hip.FORMAT_CS8, n, back)
    assert np.array_equal(back.cpu().numpy(), orc.cs8_to_cs16(i8))


@pytest.mark.parametrize("fmt_name", ["CS16", "CF32", "CF64", "CS8"])
def test_convert_pack_in_one_launch_equals_the_two_steps_and_the_oracle(orc, fmt_name):
    """writeStream's conversion loop (CaribouliteStream.cpp:199-244) + caribou_smi_generate_data (caribou_smi.c:684-717) in one
    launch: the same bytes as clhip_convert_to_cs16 -> clhip_smi_pack and as the oracle, ragged lengths, both pack modes,
    an output that is only 4-byte aligned; float inputs include the values the x86 truncating conversion wraps."""
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(21)
    fmt = getattr(hip, "FORMAT_" + fmt_name)
    for n in (1, 3, 4, 1023, 131072, 131075):
        if fmt_name == "CS16":
            host = rng.integers(-32768, 32768, (n, 2)).astype(np.int16); to16 = lambda a: a
        elif fmt_name == "CF32":
            host = ((rng.random((n, 2)) - 0.5) * 20).astype(np.float32); host[:: 97] = [[7.99, -8.0]]; to16 = orc.cf32_to_cs16
        elif fmt_name == "CF64":
            host = (rng.random((n, 2)) - 0.5) * 20; host[:: 89] = [[-8.0, 7.999999]]; to16 = orc.cf64_to_cs16
        else:
            host = rng.integers(-128, 128, (n, 2)).astype(np.int8); to16 = orc.cs8_to_cs16
        d_in = torch.from_numpy(host.copy()).to("cuda:0")
        for mode, mis in ((hip.TX_DOCUMENTED, 0), (hip.TX_AS_WRITTEN, 0), (hip.TX_DOCUMENTED, 4)):
            raw = torch.full((4 * n + 64,), 0xEE, dtype=torch.uint8, device="cuda:0")
            assert hip.lib().clhip_convert_pack(d_in.data_ptr(), fmt, n, mode, raw.data_ptr() + mis, hip.current_stream()) == 0
            iq = torch.empty((n, 2), dtype=torch.int16, device="cuda:0")
            two = torch.empty(4 * n, dtype=torch.uint8, device="cuda:0")
            hip.convert_to_cs16(d_in, fmt, n, iq)
            hip.smi_pack(mode, iq, n, two)
            torch.cuda.synchronize()
            got = raw.cpu().numpy()
            assert (got[:mis] == 0xEE).all() and (got[mis + 4 * n:] == 0xEE).all()
            assert np.array_equal(got[mis:mis + 4 * n], two.cpu().numpy())
            assert np.array_equal(got[mis:mis + 4 * n], orc.generate_data(to16(host), mode))


@pytest.mark.parametrize("fmt_name", ["CS16", "CF32", "CF64", "CS8"])
def test_convert_pack_rows_equals_the_single_row_launch_row_by_row(orc, fmt_name):
    """clhip_convert_pack_rows (a stream group's writeStream: up to 8 streams per launch, rows wherever they are): every row's bytes
    equal the oracle's (CaribouliteStream.cpp:199-244 + caribou_smi.c:684-717), rows scattered and one of them only 4-byte aligned,
    nothing written between or behind the rows; 0 rows and more than 8 are refused / empty."""
    import ctypes as C
    import torch
    from cariboulite_amd import hip
    rng = np.random.default_rng(5)
    fmt = getattr(hip, "FORMAT_" + fmt_name)
    for n, rows in ((131072, 8), (1023, 3), (4, 1)):
        if fmt_name == "CS16":
            host = rng.integers(-32768, 32768, (rows, n, 2)).astype(np.int16); to16 = lambda a: a
        elif fmt_name == "CF32":
            host = ((rng.random((rows, n, 2)) - 0.5) * 20).astype(np.float32); to16 = orc.cf32_to_cs16
        elif fmt_name == "CF64":
            host = (rng.random((rows, n, 2)) - 0.5) * 20; to16 = orc.cf64_to_cs16
        else:
            host = rng.integers(-128, 128, (rows, n, 2)).astype(np.int8); to16 = orc.cs8_to_cs16
        d_in = torch.from_numpy(host.copy()).to("cuda:0")
        pitch = 4 * n + 64
        raw = torch.full((rows * pitch + 64,), 0xEE, dtype=torch.uint8, device="cuda:0")
        order = list(rng.permutation(rows))                          # row r's words land in slot order[r]
        offs = [int(order[r]) * pitch + (4 if r == 1 else 0) for r in range(rows)]
        ins = (C.c_void_p * rows)(*[d_in[r].data_ptr() for r in range(rows)])
        outs = (C.c_void_p * rows)(*[raw.data_ptr() + offs[r] for r in range(rows)])
        assert hip.lib().clhip_convert_pack_rows(ins, fmt, n, rows, hip.TX_DOCUMENTED, outs, hip.current_stream()) == 0
        torch.cuda.synchronize()
        got = raw.cpu().numpy()
        mask = np.ones(got.size, bool)
        for r in range(rows):
            assert np.array_equal(got[offs[r]: offs[r] + 4 * n], orc.generate_data(to16(host[r]), hip.TX_DOCUMENTED)), (n, r)
            mask[offs[r]: offs[r] + 4 * n] = False
        assert (got[mask] == 0xEE).all()
    assert hip.lib().clhip_convert_pack_rows(ins, fmt, 4, 0, hip.TX_DOCUMENTED, outs, hip.current_stream()) == 0
    assert hip.lib().clhip_convert_pack_rows(ins, fmt, 4, 9, hip.TX_DOCUMENTED, outs, hip.current_stream()) != 0


def test_take_i_rail():
    import torch
    from cariboulite_amd import hip
    for n in (1, 255, 131072, 1 << 20):
        x = torch.randn((n, 2), device="cuda:0")
        m = torch.full((n + 4,), 7.0, device="cuda:0")
        assert hip.lib().clhip_take_i_rail(x.data_ptr(), n, m.data_ptr(), hip.current_stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(m[:n], x[:, 0]) and bool((m[n:] == 7.0).all())
