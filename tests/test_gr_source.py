"""The GNU Radio source block over the GPU path (cariboulite_amd/csrc/gr_source; the caller SURVEY.md section 8(f) rank 1
names: gr-caribouLite/lib/caribouLiteSource_impl.cc:104-121).  GNU Radio's headers are not in the build image, so the
adaptor is compiled against a compile-check stub of the API slice it uses (tests/cpp/gr_api_stub -- NOT GNU Radio): on CPU
as a shared module (it must compile and link against the C++ API), on the GPU box as a program that drives work() and
compares samples, meta bytes and pps tags with the oracle."""
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "cariboulite_amd")
STUB = os.path.join(ROOT, "tests", "cpp", "gr_api_stub")
ADAPTOR = os.path.join(PKG, "csrc", "gr_source", "caribouLiteSourceHip.cc")


def _common():
    from cariboulite_amd import _build
    _build.build_all()
    return ["-std=c++17", "-Wall", "-Wextra", "-I", STUB, "-I", os.path.join(PKG, "csrc", "gr_source"),
            "-I", os.path.join(PKG, "csrc", "cpp_api"), "-I", os.path.join(ROOT, "include"),
            "-L", PKG, "-lcariboulite_cpp", "-lcariboulite_host", "-lcariboulite_hip", "-lpthread", f"-Wl,-rpath,{PKG}"]


def test_adaptor_builds_as_a_module(tmp_path):
    so = str(tmp_path / "libgnuradio-caribouLiteHip.so")
    r = subprocess.run(["g++", "-fPIC", "-shared", ADAPTOR, "-o", so] + _common(), capture_output=True, text=True)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr[-3000:]
    syms = subprocess.run(["nm", "-D", "-C", "--undefined-only", so], capture_output=True, text=True).stdout
    for f in ("CaribouLiteRadio::ReadSamples(std::complex<float>*", "CaribouLiteRadio::GetSyncTags", "CaribouLiteRadio::EnableSyncTags"):
        assert f in syms, f


@pytest.mark.gpu
def test_work_against_the_oracle(tmp_path):
    from oracle import oracle as orc
    orc.lib()
    exe = str(tmp_path / "test_gr_source")
    r = subprocess.run(["g++", "-O1", "-g", os.path.join(ROOT, "tests", "cpp", "test_gr_source.cpp"), ADAPTOR, "-o", exe,
                        "-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "liboracle.so"),
                        f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}"] + _common(), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gr source ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
