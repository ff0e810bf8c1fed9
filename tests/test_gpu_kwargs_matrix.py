"""-m gpu: readStream over the PRODUCT of its configuration axes -- format x low-pass (setBandwidth) x extension stages (FIR +
RESAMP | FIR + DEMOD) x ASYNC x ZEROCOPY -- each case a byte stream with a re-synchronising and a lost batch in it, checked
against the oracle's chain  caribou_smi_read -> [IIR] -> [/4096 | >> 5 | FIR -> L/M | FIR -> FM]  (caribou_smi.c:632-682,
CaribouliteStream.cpp:282-382) with the Stream's persistent intermediate buffer modelled (slots a re-sync leaves untouched keep
the previous batch's samples).  Every pair of axis values occurs together at least once; the cases replace round 3's sweep of
the whole suite under 23 one-at-a-time environment switches (the switches are gone: one route per configuration)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
NB, MTU, SENT = 524288, 131072, -21846

#        format  iir     stages      async  zerocopy
CASES = [("CS16", None,  None,       0, 0), ("CS16", 100e3, None,       0, 1), ("CS16", None,  None,       1, 0), ("CS16", 50e3,  None,       1, 1),
         ("CS8",  None,  None,       0, 1), ("CS8",  20e3,  None,       0, 0), ("CS8",  None,  None,       1, 1),
         ("CF32", None,  None,       0, 0), ("CF32", 100e3, None,       0, 1), ("CF32", None,  "rs32",     0, 0), ("CF32", None,  "rs32",     0, 1),
         ("CF32", 100e3, "rs32",     0, 0), ("CF32", None,  "rs32",     1, 1), ("CF32", None,  "fm",       0, 0), ("CF32", 50e3,  "fm",       1, 0),
         ("CF32", None,  "fm",       0, 1), ("CF64", None,  None,       0, 1), ("CF64", 100e3, None,       1, 0), ("CF64", None,  None,       0, 0)]


def test_every_pair_of_axis_values_is_covered():
    val = lambda c, k: str(bool(c[k])) if k == 1 else str(c[k])         # (the low-pass axis: off / on; all three filters occur)
    axes = [sorted({val(c, k) for c in CASES}) for k in range(5)]
    assert {c[1] for c in CASES} == {None, 20e3, 50e3, 100e3}
    for a in range(5):
        for b in range(a + 1, 5):
            if {a, b} == {0, 2}:
                continue                                   # extension stages need CF32 (setupStream refuses the others)
            for va in axes[a]:
                for vb in axes[b]:
                    assert any(val(c, a) == va and val(c, b) == vb for c in CASES), (a, va, b, vb)


@pytest.fixture(scope="module")
def S():
    import torch
    from cariboulite_amd import hip, soapy
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return soapy


@pytest.mark.parametrize("fmt,bw,stages,use_async,zc", CASES)
def test_read_stream_configuration(S, orc, fmt, bw, stages, use_async, zc):
    from cariboulite_amd import synth
    t = load_golden("taps.npz")
    ch = (len(fmt) + (1 if bw else 0) + use_async) % 2
    args = {}
    if stages == "rs32":
        args.update(FIR="64:1000000", RESAMP="3/2")
    elif stages == "fm":
        args.update(FIR="64:100000", DEMOD="FM")
    if use_async:
        args["ASYNC"] = "1"
    if zc:
        args["ZEROCOPY"] = "1"
    sdr = S.Device(dict(driver="Cariboulite", channel="S1G" if ch == 0 else "HiF"))
    rx = sdr.setupStream(S.SOAPY_SDR_RX, fmt, args=args)
    if bw:
        sdr.setBandwidth(S.SOAPY_SDR_RX, 0, bw)
    sdr.activateStream(rx)
    # five batches: good, 6 bytes late (re-sync: one extrapolated sample, one untouched slot), good, no sync at all, good
    b = synth.smi_stream_bytes(5 * MTU, ch, stream=77 + len(fmt))[0].copy()
    b[NB:2 * NB] = np.concatenate([np.full(6, 0x11, np.uint8), b[NB:2 * NB - 6]])
    b[3 * NB:4 * NB] = 0
    dt = {"CS16": np.int16, "CS8": np.int8, "CF32": np.float32, "CF64": np.float64}[fmt]
    n_out = MTU * 3 // 2 + 8 if stages == "rs32" else MTU + 8
    out = np.zeros((n_out, 2) if stages != "fm" else (n_out,), dt)
    if zc:
        sdr.registerStreamBuffer(rx, out)
    # the oracle's chain
    iir = orc.IIR(6, 4e6, bw / 2) if bw else None
    fir = orc.FIR(t["fir64_c2"] if stages == "rs32" else t["fir64_c3"]) if stages else None
    rs = orc.Resampler(t["rs_3_2"], 3, 2) if stages == "rs32" else None
    fm_prev = None
    interm = np.zeros((MTU + 2, 2), np.int16)
    sdr.feedSmiBytes(b)
    delivered = 0
    for k in range(5):
        ret, iq, _ = orc.smi_read(ch, b[k * NB:(k + 1) * NB], MTU, NB, fill=SENT)
        if use_async and ret < 0:
            continue                                       # the reader thread drops a failed read; the consumer never sees it
        if not (fmt == "CS16" and bw):
            out[...] = 0                                   # (CS16 behind the low-pass: the client's buffer IS the buffer the reference filters in place -- left as delivered)
        sr = sdr.readStream(rx, [out], MTU, timeoutUs=3_000_000 if use_async else 100000)
        if ret < 0:
            assert sr.ret == 0 and (not out.any() or (fmt == "CS16" and bw)), k       # Stream::Read squashes -3 to 0 (CaribouliteStream.cpp:266-276)
            continue
        assert ret == MTU
        touched = (iq != SENT).any(axis=1)
        interm[touched] = iq[touched]
        stale = ~touched[:MTU]                             # slot MTU-1 of the re-synchronised batch
        x = interm[:MTU].copy()
        if fmt == "CS16" and not bw and not stages and not use_async:
            # plain CS16: only the slots the reference writes reach the client's buffer (caribou_smi.c:344-389)
            assert sr.ret == MTU
            assert np.array_equal(out[:MTU][~stale], x[~stale]) and not out[:MTU][stale].any(), k
            delivered += 1
            continue
        if iir:
            y = iir.apply_cs16(x)
            if not use_async:                              # the reference filters IN PLACE (CaribouliteStream.cpp:291-298): the buffer keeps what was delivered,
                interm[:MTU] = y                           # and the slot a later re-sync leaves untouched goes through the filter again.  (With the reader
        else:                                              # thread the re-sync happens in ITS buffer, interm_native_buffer1, which nobody filters: :16-49.)
            y = x
        if not stages:
            want = {"CS16": lambda v: v, "CS8": orc.cs16_to_cs8, "CF32": orc.cs16_to_cf32, "CF64": orc.cs16_to_cf64}[fmt](y)
            assert sr.ret == MTU
            got = out[:MTU]
            if iir:
                lsb = {"CS16": 1, "CS8": 1, "CF32": 1.0 / 4096, "CF64": 1.0 / 4096}[fmt]
                d = np.abs(got.astype(np.float64) - want.astype(np.float64))
                assert d.max() <= lsb * (1 + 1e-9) and np.mean(d != 0) < 1e-3, (k, d.max())
            else:
                assert np.array_equal(got, want), k       # integer / power-of-two arithmetic: exact, stale slot included
        else:
            z = fir.f64(orc.cs16_to_cf32(y))
            tol = 1e-5 if not iir else 2e-3               # one int16 LSB behind the IIR is 2.4e-4 of full scale in front of the FIR
            if stages == "rs32":
                want = rs.f64(z)
                assert sr.ret == want.shape[0]
                assert np.max(np.abs(out[:sr.ret] - want)) <= tol * np.max(np.abs(want)), k
            else:
                want, fm_prev = orc.fm_demod_f64(z, fm_prev)
                assert sr.ret == MTU
                # atan2 is ill-conditioned where the phasor is small: judged where it is not (behind the 25 kHz low-pass the 250 kHz
                # tone is gone and only filtered noise is left: a lower bar, and the rare one-LSB crossings of the IIR's int16
                # output -- one sample in ~10^4 -- are left to the 99.9th percentile)
                mag = np.hypot(z[:, 0], z[:, 1]); ok = mag > (0.02 if not iir else 0.008); ok[1:] &= ok[:-1]; ok[0] = False
                d = np.abs(out[:MTU] - want); d = np.minimum(d, 2 * np.pi - d)
                if not iir:
                    assert ok.mean() > 0.5 and np.max(d[ok]) <= 1e-4, (k, np.max(d[ok]))
                else:
                    assert ok.mean() > 0.1 and np.quantile(d[ok], 0.999) <= 1e-3 and np.max(d[ok]) <= 0.1, (k, ok.mean(), np.max(d[ok]))
            assert not out[sr.ret:].any()
        delivered += 1
    assert delivered == 4
    st = sdr.streamStats(rx)
    if zc:
        assert st["zero_copy_registrations"] == 1 and st["zero_copy_reads"] >= 3, st
    else:
        assert st["zero_copy_reads"] == 0
    sdr.close()
