"""not-gpu: the world > 1 branch of the RCCL fan-out (cariboulite_amd/csrc/fanout/clfan.cpp: grouped ncclSend / ncclRecv,
peer and row arithmetic) compiled against an in-process model of the RCCL calls it uses (tests/cpp/rccl_mock: threads as
ranks, host memory; TEST ONLY) and run at world 1, 2, 3 and 8 with ragged strides: every row checked on both sides, the
schedule checked to be one group per call, and a deliberately unmatched send shown to FAIL (so a deadlocking schedule
cannot pass by hanging).  The product library links the real librccl; 8-GPU runs are the driver's."""
import os
import subprocess

from conftest import ROOT

MOCK = os.path.join(ROOT, "tests", "cpp", "rccl_mock")


def test_fanout_schedule_on_a_mock_rccl(tmp_path):
    exe = str(tmp_path / "test_clfan_mock")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-pthread", "-I", MOCK, "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "cariboulite_amd", "csrc", "fanout", "clfan.cpp"), os.path.join(MOCK, "rccl_mock.cpp"),
           os.path.join(ROOT, "tests", "cpp", "test_clfan_mock.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK clfan mock" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_fanout_schedule_under_thread_sanitizer(tmp_path):
    """the same program under TSan: the model's mailboxes and clfan's per-thread error buffer are race-free, so a hang or a
    torn row on the real node is not the harness's own doing"""
    exe = str(tmp_path / "test_clfan_mock_tsan")
    cmd = ["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-pthread", "-I", MOCK, "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "cariboulite_amd", "csrc", "fanout", "clfan.cpp"), os.path.join(MOCK, "rccl_mock.cpp"),
           os.path.join(ROOT, "tests", "cpp", "test_clfan_mock.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "OK clfan mock" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout[-2000:] + r.stderr[-3000:]
