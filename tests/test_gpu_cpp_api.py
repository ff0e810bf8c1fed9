"""-m gpu: the C++ API seam -- builds and runs tests/cpp/test_cpp_api.cpp,
which drives CaribouLiteRadio (sync + async callback thread) and checks it against the oracle."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_api_program(tmp_path):
    from cariboulite_amd import _build
    from oracle import oracle as orc
    orc.lib()
    _build.build_all()
    exe = str(tmp_path / "test_cpp_api")
    pkg = os.path.join(ROOT, "cariboulite_amd")
    cmd = ["g++", "-std=c++14", "-O1", "-g", os.path.join(ROOT, "tests", "cpp", "test_cpp_api.cpp"),
           "-I", os.path.join(pkg, "csrc", "cpp_api"), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
           "-L", pkg, "-lcariboulite_cpp", "-lcariboulite_host", "-lcariboulite_hip",
           os.path.join(ROOT, "oracle", "liboracle.so"), "-lpthread",
           f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-o", exe]
    subprocess.run(cmd, check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cpp api ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
