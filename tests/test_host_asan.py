"""not-gpu: the host-side bookkeeping of the ingest path (three-cursor byte FIFO, sample ring span calls) under
AddressSanitizer + UBSan on the CPU build, driven by random op sequences against a plain model
(tests/cpp/test_fifo_ring_asan.c).  Sanitizers are a CPU-build affair on this pool."""
import os
import subprocess

import pytest

from conftest import ROOT


def test_fifo_and_ring_bookkeeping_under_asan(tmp_path):
    from cariboulite_amd import _build
    _build.build_all()
    pkg = os.path.join(ROOT, "cariboulite_amd")
    host = os.path.join(pkg, "csrc", "host")
    exe = str(tmp_path / "fifo_ring_asan")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "cpp", "test_fifo_ring_asan.c"), os.path.join(host, "cl_ring.c"), os.path.join(host, "cl_smi.c"),
           "-I", host, "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lcariboulite_hip", f"-Wl,-rpath,{pkg}", "-lpthread", "-lm", "-o", exe]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "harness ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_ring_span_calls_two_threads_under_tsan(tmp_path):
    """The ring is not locked between a span's _begin and _end (round 3): one producer and one consumer thread hammer it under
    ThreadSanitizer -- a storage access of one inside the other's open span would be a reported race -- and the consumer
    checks that what it reads is strictly increasing (tests/cpp/test_ring_tsan.c)."""
    from cariboulite_amd import _build
    _build.build_all()
    pkg = os.path.join(ROOT, "cariboulite_amd")
    host = os.path.join(pkg, "csrc", "host")
    exe = str(tmp_path / "ring_tsan")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "cpp", "test_ring_tsan.c"), os.path.join(host, "cl_ring.c"),
           "-I", host, "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lcariboulite_hip", f"-Wl,-rpath,{pkg}", "-lpthread", "-o", exe]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ring tsan harness ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_group_copy_pool_under_tsan(tmp_path):
    """the stream group's last hop (cl_group.c: copy threads + the calling thread draining one queue, non-temporal stores) under
    ThreadSanitizer: ragged rows, an overflowing queue, 0 / 1 / 3 workers, every byte checked after every drain"""
    from cariboulite_amd import _build
    _build.build_all()
    pkg = os.path.join(ROOT, "cariboulite_amd")
    host = os.path.join(pkg, "csrc", "host")
    exe = str(tmp_path / "group_pool_tsan")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "cpp", "test_group_pool_tsan.c"), os.path.join(host, "cl_smi.c"), os.path.join(host, "cl_soapy.c"), os.path.join(host, "cl_ring.c"),
           "-I", host, "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lcariboulite_hip", f"-Wl,-rpath,{pkg}", "-lpthread", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "group pool tsan harness ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_stream_group_host_side_over_a_threaded_hip_model(tmp_path, sanitizer):
    """cl_group.c + cl_smi.c + cl_soapy.c over tests/cpp/hip_mock/clhip_mock.c, a CPU model of the clhip_* layer in which every HIP
    stream is a thread: what the group queues (copies out of the members' FIFOs, launches that read offset tables and write the
    mirrors, the next call's batches read and computed AHEAD) runs concurrently with the caller, with a feeder thread per member and
    with the copy threads, while the client reads members through their own devices, asks for half batches and registers / releases
    its buffers in between.  ThreadSanitizer: a host write under queued work is a race; AddressSanitizer: a buffer freed or
    overrun under queued work.  Every delivered sample is checked against what was fed (tests/cpp/test_group_mock.c)."""
    host = os.path.join(ROOT, "cariboulite_amd", "csrc", "host")
    exe = str(tmp_path / "group_mock")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "cpp", "test_group_mock.c"), os.path.join(ROOT, "tests", "cpp", "hip_mock", "clhip_mock.c"),
           os.path.join(ROOT, "oracle", "cl_oracle.c")] + [os.path.join(host, f) for f in ("cl_group.c", "cl_node.c", "cl_smi.c", "cl_soapy.c", "cl_ring.c")] + \
          ["-I", host, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), "-lpthread", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, "30" if sanitizer == "thread" else "40"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "group mock harness ok" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_single_stream_host_paths_over_a_threaded_hip_model(tmp_path, sanitizer):
    """readStream with ASYNC=1 (reader thread -> device ring -> client), readStream on the calling thread with the seam's read-ahead
    (plain and into a registered ZEROCOPY buffer, with a flush in between) and writeStream CS16 / CF32 into the pinned TX FIFO, each
    racing a producer / drainer thread, over the threaded model of the HIP layer (tests/cpp/test_stream_mock.c); every delivered
    block names its own place in the stream: contiguous inside, strictly ascending across calls."""
    host = os.path.join(ROOT, "cariboulite_amd", "csrc", "host")
    exe = str(tmp_path / "stream_mock")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "cpp", "test_stream_mock.c"), os.path.join(ROOT, "tests", "cpp", "hip_mock", "clhip_mock.c"),
           os.path.join(ROOT, "oracle", "cl_oracle.c")] + [os.path.join(host, f) for f in ("cl_group.c", "cl_node.c", "cl_smi.c", "cl_soapy.c", "cl_ring.c")] + \
          ["-I", host, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), "-lpthread", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "stream mock harness ok" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
