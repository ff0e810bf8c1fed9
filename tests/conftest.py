import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The HIP runtime copies pageable host memory of 1 MiB and more by pinning the caller's pages in place (and keeps such
# pinnings cached by address).  The test sessions upload thousands of numpy temporaries that die right after the copy; now
# and then the GPU then faults on a HOST heap address inside such a copy ("Memory access fault by GPU ... on address
# 0x621f06932000", main thread in tests/gpu_util.dev_bytes; DESIGN.md section 7, robustness record) and the HSA runtime
# aborts the process.  With this threshold (MiB) out of reach the runtime stages pageable copies through its own pinned
# buffers instead, and the device never touches numpy's heap.  Has to be in the environment before the runtime initialises.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The native libraries are build artefacts (git-ignored): build them in-tree when a fresh
    checkout runs the tests before __graft_entry__.build() (hipcc cross-compiles without a GPU)."""
    import torch  # noqa: F401  -- torch's bundled HIP runtime has to be the first one in the process: a test that loads a
    #                      HIP library of ours first (tests/test_binding_b.py) would otherwise leave torch.cuda blind
    from cariboulite_amd import _build
    if not (os.path.exists(_build.HIP_LIB) and os.path.exists(_build.HOST_LIB)):
        _build.build_all()
    from oracle import oracle as o
    if not os.path.exists(o.LIB_PATH):
        o.build(ref=os.path.isdir("/root/reference"))
    _install_abort_trace()


def _install_abort_trace():
    """A SIGABRT anywhere in the process prints the native stack of the thread that raised it (tests/cpp/abrt_trace.c)
    before python's faulthandler / the default action takes over: the robustness record's open item is an abort with no
    message, and the python frames alone do not say who called abort().  Best effort; CL_ABRT_TRACE=0 turns it off."""
    if os.environ.get("CL_ABRT_TRACE", "1") == "0":
        return
    import ctypes
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "abrt_trace.c")
    lib = os.path.join(ROOT, "tests", "cpp", "build", "libabrt_trace.so")
    try:
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(lib), exist_ok=True)
            subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", lib, src], check=True)
        import faulthandler
        if not faulthandler.is_enabled():
            faulthandler.enable()                 # (first: ours chains to the handler it finds)
        out = os.environ.get("CL_ABRT_TRACE_FILE", os.path.join(ROOT, "gpurun_out", "abrt_trace.txt"))
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tr = ctypes.CDLL(lib)
        tr.abrt_trace_install(out.encode())                 # (under pytest fd 2 is a capture file that dies with the process)
        # ... followed by the product library's own record of what it did with host memory it does not own (registrations,
        # releases, copies: clhip_debug_ops_dump), so that a GPU fault address on the host heap can be set against it
        from cariboulite_amd import hip
        tr.abrt_trace_set_dump.argtypes = [ctypes.c_void_p]
        tr.abrt_trace_set_dump(ctypes.cast(hip.lib().clhip_debug_ops_dump, ctypes.c_void_p))
    except Exception as e:                        # a diagnostic must never fail a session
        print("abort trace not installed:", e, file=sys.stderr)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/cl_oracle.c via ctypes) -- the checker, never the product."""
    from oracle import oracle as o
    o.lib()
    return o


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def pytest_sessionfinish(session, exitstatus):
    """Leave nothing of ours for the interpreter's shutdown to destroy in arbitrary order: objects that own GPU memory, HIP
    streams or a reader thread (pipes, filters, Soapy devices a failed test left open) are collected here, while the HIP
    runtime is certainly alive, and the GPU is idle when the process starts to exit."""
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
