#!/usr/bin/env python3
"""Builds and runs tools/cpp/bench_cpp_api.cpp: microseconds per CaribouLiteRadio::ReadSamples call of one MTU, PCIe-inclusive."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cariboulite_amd import _build
_build.build_all()
pkg = os.path.join(ROOT, "cariboulite_amd")
exe = os.path.join(tempfile.mkdtemp(), "bench_cpp_api")
subprocess.run(["g++", "-std=c++14", "-O2", os.path.join(ROOT, "tools", "cpp", "bench_cpp_api.cpp"), "-I", os.path.join(pkg, "csrc", "cpp_api"),
                "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lcariboulite_cpp", "-lcariboulite_host", "-lcariboulite_hip", "-lpthread",
                f"-Wl,-rpath,{pkg}", "-o", exe], check=True)
sys.exit(subprocess.run([exe]).returncode)
