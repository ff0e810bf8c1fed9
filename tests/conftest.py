import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


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
