"""-m gpu: the failure branch of the benchmarked configuration.

The fused RX kernel checks the per-chunk sync-search results on the device
(clhip_rx_pipe_set_sync_check).  When a chunk is out of sync the call must end up exactly where the
reference's caribou_smi_read -> FIR -> L/M chain ends up (caribou_smi.c:235-292 search, :319-325 skip,
:382-389 extrapolated sample, :665-668 "-3"): the raw-word run is rolled back and redone from re-synchronised
int16 samples, a chunk without sync leaves the pipe where it was.  Outputs, the carried history (through the
following call) and the return codes are compared with orc.smi_read -> orc.FIR -> orc.Resampler on the same bytes.
"""
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-5
SENT = -21846            # 0xAAAA: what "untouched" slots hold on both sides


@pytest.fixture(scope="module")
def G():
    import torch
    from cariboulite_amd import hip
    import gpu_util
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return gpu_util


def slipped_chunk(words_u8, k, rng):
    """A chunk that lost sync by k bytes: k junk bytes in front, the tail cut so the length stays."""
    junk = rng.integers(0, 256, k, dtype=np.uint8)
    junk &= 0x3F                                 # never looks like a sync word
    return np.concatenate([junk, words_u8[: words_u8.size - k]])


def build_calls(channel, chunk_len, n_chunks, slips, seed):
    """A list of calls (bytes, kind).  Call 0 aligned; one call per slip with chunk 1 (and for k = 6 also the
    last chunk) offset by k bytes; one call whose chunk 2 has no sync at all; a final aligned call."""
    from cariboulite_amd import synth
    rng = np.random.default_rng(seed)
    per = chunk_len // 4
    calls, pos = [], 0

    def fresh():
        nonlocal pos
        b = synth.smi_stream_bytes(per * n_chunks, channel, stream=seed, n0=pos)[0].copy()
        pos += per * n_chunks
        return b

    calls.append((fresh(), "aligned"))
    for k in slips:
        b = fresh()
        b[chunk_len:2 * chunk_len] = slipped_chunk(b[chunk_len:2 * chunk_len], k, rng)
        if k == 6:
            b[-chunk_len:] = slipped_chunk(b[-chunk_len:], 5, rng)
        calls.append((b, f"slip{k}"))
    b = fresh()
    b[2 * chunk_len:3 * chunk_len] = 0           # (0 & 0xC001C000) != 0x80004000 everywhere: no sync
    calls.append((b, "lost"))
    calls.append((fresh(), "aligned"))
    return calls


def oracle_calls(orc, channel, calls, chunk_len, fir, rs, L, M):
    """caribou_smi_read -> /4096 -> FIR -> L/M with streaming state; a -3 call delivers nothing (Stream::Read
    squashes it to 0 samples, CaribouliteStream.cpp:266-276) and leaves the float stages where they were."""
    f, r = orc.FIR(fir), orc.Resampler(rs, L, M)
    res = []
    for b, kind in calls:
        n = b.size // 4
        ret, iq, _ = orc.smi_read(channel, b, n, chunk_len, fill=SENT)
        if ret < 0:
            res.append((ret, None))
            continue
        assert ret == n
        res.append((ret, r.f64(f.f64(orc.cs16_to_cf32(iq[:n])))))
    return res


def gpu_calls(G, pipe, calls, chunk_len, n_streams=1, stream_of_interest=0, other=None):
    import torch
    outs = []
    for ci, (b, kind) in enumerate(calls):
        n = b.size // 4
        if n_streams == 1:
            d = G.dev_bytes(b)
            stride = 0
        else:
            rows = [other[ci] if s != stream_of_interest else b for s in range(n_streams)]
            d = torch.from_numpy(np.stack(rows)).to(G.DEV)
            stride = b.size
        nch = -(-b.size // chunk_len)
        offs = torch.full((n_streams * nch,), 77, dtype=torch.int32, device=G.DEV)
        cs16 = torch.full((n_streams, n + 2, 2), SENT, dtype=torch.int16, device=G.DEV)
        no = pipe.out_count(n)
        out = torch.full((n_streams, no + 8, 2), float("nan"), dtype=torch.float32, device=G.DEV)
        h_offs = np.full(n_streams * nch, 99, dtype=np.int32)
        if os.environ.get("CL_TEST_TRACE_PTRS") == "1":           # (robustness record: a GPU page fault's address can be placed)
            os.write(2, ("[ptrs] call %d %s n=%d d=%#x+%d offs=%#x cs16=%#x+%d out=%#x+%d\n" % (
                ci, kind, n, d.data_ptr(), d.numel(), offs.data_ptr(), cs16.data_ptr(), cs16.numel() * 2,
                out.data_ptr(), out.numel() * 4)).encode())
        rc = pipe.run_smi(d, stride, b.size, chunk_len, offs, cs16, out, no + 8, h_offs=h_offs)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        if rc >= 0:
            assert rc == no
            assert np.all(np.isnan(o[:, no:])), "wrote past the end"
        outs.append((rc, o[:, :max(rc, 0)], h_offs.reshape(n_streams, nch).copy()))
    return outs


@pytest.mark.parametrize("channel", [0, 1])
@pytest.mark.parametrize("chunk_len,n_chunks", [(4 * 8192, 5), (524288, 3)])
def test_c2_misaligned_and_lost_chunks_equal_smi_read_chain(G, orc, channel, chunk_len, n_chunks):
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    calls = build_calls(channel, chunk_len, n_chunks, (1, 3, 6), seed=40 + channel)
    want = oracle_calls(orc, channel, calls, chunk_len, t["fir64_c2"], t["rs_3_2"], 3, 2)
    pipe = hip.RxPipe(1, channel, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    assert pipe.uses_fused(calls[0][0].size // 4)
    got = gpu_calls(G, pipe, calls, chunk_len)
    kinds = [k for _, k in calls]
    for (b, kind), (wret, wout), (rc, o, h_offs) in zip(calls, want, got):
        if kind == "lost":
            assert wret == -3 and rc == -3                          # caribou_smi.c:665-668
            assert h_offs[0, 2] == -1
            continue
        assert rc == wout.shape[0]
        if kind.startswith("slip"):
            k = int(kind[4:])
            assert h_offs[0, 1] == k and h_offs[0, 0] == 0
        else:
            assert not h_offs.any()
        peak = np.max(np.abs(wout))
        assert np.max(np.abs(o[0] - wout)) <= TOL * peak, (kind, np.max(np.abs(o[0] - wout)), peak)
    assert kinds[-1] == "aligned"        # the call after the failures: carried history and phase are the oracle's


@pytest.mark.parametrize("source", ["search_results", "in_kernel"])
def test_rollback_restores_the_pre_call_state(G, orc, source):
    """The armed check on its own: a bad launch writes nothing for the affected tiles, raises the flag, and
    rollback() puts the pipe back so that re-running the same (repaired) call gives a clean pipe's outputs."""
    import torch
    from cariboulite_amd import hip, synth
    t = load_golden("taps.npz")
    per, nch = 8192, 4
    n = per * nch
    b0 = synth.smi_stream_bytes(n, 0, stream=7)[0]
    b1 = synth.smi_stream_bytes(n, 0, stream=7, n0=n)[0]
    bad = b1.copy()
    bad[4 * per:8 * per] = slipped_chunk(bad[4 * per:8 * per], 2, np.random.default_rng(1))

    def run(pipe, b, armed):
        d = G.dev_bytes(b)
        offs = torch.zeros(nch, dtype=torch.int32, device=G.DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=G.DEV)
        hip.smi_find_offsets(d, b.size, 4 * per, 4 * per, nch, offs)
        if armed:
            pipe.set_sync_check(offs if source == "search_results" else None, per, flag)   # None: tiles test the chunk heads themselves
        no = pipe.out_count(n)
        out = torch.full((no, 2), float("nan"), dtype=torch.float32, device=G.DEV)
        assert pipe.run(hip.PIPE_IN_SMI_WORDS, d, 0, n, out, 0) == no
        torch.cuda.synchronize()
        pipe.set_sync_check(None, per, None)
        return out.cpu().numpy(), int(flag.item())

    clean = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, 0)
    a0, _ = run(clean, b0, False)
    a1, _ = run(clean, b1, False)
    p = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, 0)
    assert p.rollback() == -1                                   # nothing to undo yet
    g0, f0 = run(p, b0, True)
    assert f0 == 0 and np.array_equal(g0, a0)
    gb, fb = run(p, bad, True)
    assert fb == 1
    lo, hi = per * 3 // 2, 2 * per * 3 // 2                     # outputs that depend only on the bad chunk
    assert np.all(np.isnan(gb[lo + 200:hi - 200])), "tiles of an out-of-sync chunk must not be stored"
    assert p.rollback() == 0 and p.rollback() == -1             # one level
    g1, f1 = run(p, b1, True)
    assert f1 == 0 and np.array_equal(g1, a1)                   # history and phase were those after call 0


def test_multi_stream_c4_one_stream_slips(G, orc):
    """FIR128 + 5/4, four streams in one pipe, stream 2 slips in one call: every stream equals its own
    smi_read chain (the whole call is redone from int16 samples; streams in sync are unaffected)."""
    from cariboulite_amd import hip, synth
    t = load_golden("taps.npz")
    chunk_len, n_chunks, ns = 4 * 8192, 3, 4
    calls = build_calls(0, chunk_len, n_chunks, (3,), seed=60)
    others = {s: [synth.smi_stream_bytes(chunk_len // 4 * n_chunks, 0, stream=70 + s, n0=ci * chunk_len // 4 * n_chunks)[0]
                  for ci in range(len(calls))] for s in range(ns) if s != 2}
    pipe = hip.RxPipe(ns, 0, t["fir128_c4"], t["rs_5_4"], 5, 4, hip.PIPE_OUT_IQ)
    import torch
    outs = []
    for ci, (b, kind) in enumerate(calls):
        rows = [b if s == 2 else others[s][ci] for s in range(ns)]
        d = torch.from_numpy(np.stack(rows)).to(G.DEV)
        n = b.size // 4
        offs = torch.zeros(ns * n_chunks, dtype=torch.int32, device=G.DEV)
        cs16 = torch.full((ns, n + 2, 2), SENT, dtype=torch.int16, device=G.DEV)
        no = pipe.out_count(n)
        out = torch.full((ns, no, 2), float("nan"), dtype=torch.float32, device=G.DEV)
        rc = pipe.run_smi(d, b.size, b.size, chunk_len, offs, cs16, out, no)
        outs.append((rc, out.cpu().numpy()))
    for s in range(ns):
        mine = [(b if s == 2 else others[s][ci], k) for ci, (b, k) in enumerate(calls)]
        # a "-3" anywhere in the call fails the call for the whole pipe (one polyphase counter): model that
        f, r = orc.FIR(t["fir128_c4"]), orc.Resampler(t["rs_5_4"], 5, 4)
        for ci, (b, kind) in enumerate(mine):
            rc, o = outs[ci]
            if calls[ci][1] == "lost":
                assert rc == -3
                continue
            n = b.size // 4
            ret, iq, _ = orc.smi_read(0, b, n, chunk_len, fill=SENT)
            assert ret == n
            w = r.f64(f.f64(orc.cs16_to_cf32(iq[:n])))
            assert rc == w.shape[0]
            assert np.max(np.abs(o[s] - w)) <= TOL * np.max(np.abs(w)), (s, ci)


@pytest.mark.parametrize("variant", ["generic", "odd_chunk"])
def test_run_smi_without_the_device_check(G, orc, variant):
    """Pipes / chunkings the device-side check does not cover (generic kernels; chunk length not a power of
    two) take the search-first route and give the same results."""
    from cariboulite_amd import hip
    t = load_golden("taps.npz")
    chunk_len = 4 * 8192 if variant == "generic" else 4 * 6000
    calls = build_calls(0, chunk_len, 4, (2,), seed=80)
    want = oracle_calls(orc, 0, calls, chunk_len, t["fir64_c2"], t["rs_3_2"], 3, 2)
    pipe = hip.RxPipe(1, 0, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
    if variant == "generic":
        pipe.force_generic(True)
    got = gpu_calls(G, pipe, calls, chunk_len)
    for (b, kind), (wret, wout), (rc, o, _) in zip(calls, want, got):
        if kind == "lost":
            assert rc == -3 == wret
            continue
        assert rc == wout.shape[0]
        assert np.max(np.abs(o[0] - wout)) <= TOL * np.max(np.abs(wout)), kind
