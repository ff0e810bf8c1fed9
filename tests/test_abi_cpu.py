"""not-gpu: the C-ABI libraries load and export every symbol include/cariboulite_hip.h declares;
host-side design helpers agree with scipy / the oracle; no GPU => loud failure, no CPU fallback."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _declared():
    src = open(os.path.join(ROOT, "include", "cariboulite_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:clhip|cl)_[A-Za-z0-9_]+)\s*\(", src)))


def _exports(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if " T " in l}


def test_every_declared_symbol_is_exported_and_bound():
    from cariboulite_amd import hip, soapy
    hip.lib(); soapy.lib()                       # both load without a GPU
    decl = _declared()
    exp = _exports(hip.LIB_PATH) | _exports(soapy.LIB_PATH)
    missing = [s for s in decl if s not in exp]
    assert not missing, missing
    assert len(decl) > 60
    bound = set(hip.exported_symbols()) | set(soapy.exported_symbols())
    assert not [s for s in decl if s not in bound], "python binding lags the header"
    # layer split: HIP shim exports clhip_*, host layer exports cl_*
    assert all(s in _exports(hip.LIB_PATH) for s in decl if s.startswith("clhip_"))
    assert all(s in _exports(soapy.LIB_PATH) for s in decl if not s.startswith("clhip_"))


def test_fanout_library_exports_what_its_header_declares():
    """include/cariboulite_fanout.h (the RCCL point-to-point fan-out): every declared clfan_* symbol is exported by
    libcariboulite_fanout.so, bound by the ctypes face, and the stream -> rank rule matches shard.assign_streams."""
    from cariboulite_amd import fanout, shard, _build
    _build.build_fanout()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cariboulite_fanout.h")).read(), flags=re.S)
    decl = sorted(set(re.findall(r"\b(clfan_[A-Za-z0-9_]+)\s*\(", src)))
    assert len(decl) >= 9
    exp = _exports(fanout.LIB_PATH)
    assert not [s for s in decl if s not in exp]
    assert sorted(decl) == fanout.exported_symbols()
    dyn = subprocess.run(["ldd", fanout.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in dyn                                           # the transfers are RCCL's
    for n, w in ((256, 8), (5, 2), (3, 4), (0, 2)):
        for r in range(w):
            assert fanout.lib().clfan_local_count(n, w, r) == len(shard.assign_streams(n, w, r))


def test_product_does_not_link_or_import_the_oracle():
    from cariboulite_amd import hip, soapy
    for lib in (hip.LIB_PATH, soapy.LIB_PATH):
        dyn = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
        assert "liboracle" not in dyn and "libref_smi" not in dyn
        assert not [s for s in _exports(lib) if s.startswith("orc_") or s.startswith("ref_")]
    for dp, _, fs in os.walk(os.path.join(ROOT, "cariboulite_amd")):
        for f in fs:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "cl_oracle" not in txt, f


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cariboulite_amd import hip, soapy
    assert hip.lib().clhip_device_count() == 0
    with pytest.raises(RuntimeError):
        hip.require_gpu()
    with pytest.raises(RuntimeError):
        soapy.Device(dict(driver="Cariboulite", channel="S1G"))
    assert not soapy.lib().cl_smi_init(0)


def test_design_lowpass_matches_scipy_firwin():
    from cariboulite_amd import soapy
    t = load_golden("taps.npz")
    for name, n, fc in (("fir64_c2", 64, 1.0e6), ("fir64_c3", 64, 100e3), ("fir128_c4", 128, 1.2e6)):
        h = soapy.design_lowpass(n, fc, 4e6)
        assert np.max(np.abs(h.astype(np.float64) - t[name + "__f64"])) < 4e-8       # fp32 rounding of 0.5-ish taps
        assert np.max(np.abs(h - t[name])) <= 6e-8 and np.mean(h != t[name]) < 0.1    # vs scipy's own fp32 rounding
    for L, M in ((3, 2), (5, 4), (2, 3)):
        h = soapy.design_lowpass(8 * L, 1.0 / max(L, M), 2.0, gain=L)
        assert np.max(np.abs(h.astype(np.float64) - t[f"rs_{L}_{M}__f64"])) < 1e-7
    with pytest.raises(ValueError):
        soapy.design_lowpass(64, 3e6, 4e6)


def test_design_butter_matches_oracle_and_scipy(orc):
    from cariboulite_amd import soapy
    g = load_golden("dsp_float.npz")
    import numpy.polynomial.polynomial as P
    for bw in (20, 50, 100):
        sos = soapy.design_butter_lowpass(6, 4e6, bw * 1e3 / 2)
        o = orc.IIR(6, 4e6, bw * 1e3 / 2).sos()
        # same algorithm, different compiler flags (FMA contraction): agree to rounding
        assert np.allclose(sos[:, :3], o[:, :3], rtol=1e-9, atol=0) and np.allclose(sos[:, 3:], o[:, 4:], rtol=1e-13, atol=0)
        den = np.array([1.0]); den_s = np.array([1.0])
        for s in sos:
            den = P.polymul(den, [1.0, s[3], s[4]])
        for s in g[f"iir_{bw}k__sos"]:
            den_s = P.polymul(den_s, s[3:])
        assert np.allclose(den, den_s, rtol=1e-9, atol=0)


def test_cpp_api_library_exports_the_reference_method_names():
    from cariboulite_amd import _build
    _build.build_all()
    out = subprocess.run(["nm", "-DC", "--defined-only", _build.CPP_LIB], capture_output=True, text=True, check=True).stdout
    for m in ("CaribouLiteRadio::ReadSamples(std::complex<float>*", "CaribouLiteRadio::ReadSamples(std::complex<short>*",
              "CaribouLiteRadio::WriteSamples(std::complex<float>*", "CaribouLiteRadio::WriteSamples(std::complex<short>*",
              "CaribouLiteRadio::StartReceiving(", "CaribouLiteRadio::StopReceiving()", "CaribouLiteRadio::GetNativeMtuSample()"):
        assert m in out, m


def test_host_memory_operation_ring_and_its_signal_safe_dump(tmp_path):
    """clhip_debug_ops / _ops_dump (what the SIGABRT tracer of the test sessions prints): registrations are noted whether or not
    they succeed -- without a GPU they fail, which is what makes this testable here -- oldest first, at most 256, and the text dump
    uses write(2) only."""
    import ctypes as C
    from cariboulite_amd import hip

    class Rec(C.Structure):
        _fields_ = [("seq", C.c_uint64), ("op", C.c_uint32), ("pad", C.c_uint32), ("base", C.c_uint64), ("len", C.c_uint64)]
    L = hip.lib()
    recs = (Rec * 300)()
    n0 = L.clhip_debug_ops(recs, 300)
    buf = np.zeros(1 << 16, np.uint8)
    L.clhip_host_register.restype = C.c_void_p
    for k in range(300):
        assert not L.clhip_host_register(C.c_void_p(buf.ctypes.data + 4096 * (k % 8)), C.c_size_t(4096 + k))
    n = L.clhip_debug_ops(recs, 300)
    assert n == 256 and n0 <= 256
    seqs = [recs[k].seq for k in range(n)]
    assert seqs == sorted(seqs) and seqs[-1] - seqs[0] == 255
    assert all(recs[k].op == 2 for k in range(n))                                           # CLHIP_OP_REGISTER_FAILED
    assert recs[n - 1].len == 4096 + 299 and recs[n - 1].base == buf.ctypes.data + 4096 * (299 % 8)
    out = tmp_path / "dump.txt"
    fd = os.open(str(out), os.O_WRONLY | os.O_CREAT)
    L.clhip_debug_ops_dump(fd)
    os.close(fd)
    lines = out.read_text().splitlines()
    assert lines[0].startswith("[clhip] host-memory operations") and len(lines) == 257
    assert lines[-1].split()[1] == "register-failed" and lines[-1].split()[3] == "+0x%x" % (4096 + 299)
