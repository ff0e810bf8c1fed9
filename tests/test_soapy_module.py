"""The real SoapySDR::Device adaptor (cariboulite_amd/csrc/soapy_module/SoapyCaribouliteHip.cpp).  SoapySDR's
headers are not in the build image, so the adaptor is compiled against a compile-check stub of the API slice it
overrides (tests/cpp/soapy_api_stub -- NOT SoapySDR): on CPU as a shared module (it must compile and link against
the host layer), on the GPU box as a program that drives it through the Device virtuals."""
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "cariboulite_amd")
STUB = os.path.join(ROOT, "tests", "cpp", "soapy_api_stub")
ADAPTOR = os.path.join(PKG, "csrc", "soapy_module", "SoapyCaribouliteHip.cpp")


def _common():
    from cariboulite_amd import _build
    _build.build_all()
    return ["-std=c++11", "-Wall", "-Wextra", "-I", STUB, "-I", os.path.join(ROOT, "include"),
            "-L", PKG, "-lcariboulite_host", "-lcariboulite_hip", f"-Wl,-rpath,{PKG}"]


def test_adaptor_builds_as_a_module(tmp_path):
    so = str(tmp_path / "libSoapyCaribouliteHip.so")
    r = subprocess.run(["g++", "-fPIC", "-shared", ADAPTOR, "-o", so] + _common(), capture_output=True, text=True)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr[-3000:]
    syms = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    for f in ("cl_device_make", "cl_setupStream", "cl_readStream", "cl_writeStream", "cl_smi_feed_bytes"):
        assert f in syms


@pytest.mark.gpu
def test_adaptor_through_device_virtuals(tmp_path):
    exe = str(tmp_path / "test_soapy_module")
    r = subprocess.run(["g++", os.path.join(ROOT, "tests", "cpp", "test_soapy_module.cpp"), "-o", exe] + _common(),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK soapy module" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
