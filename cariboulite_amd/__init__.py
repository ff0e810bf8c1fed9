"""cariboulite_amd -- MI355X (gfx950) implementation of the CaribouLite host
sample-stream hot path behind the reference's own call surface.

The product is the pair of in-tree C-ABI libraries declared in
include/cariboulite_hip.h:

    libcariboulite_hip.so   hand-written HIP kernels + thin launch shim
    libcariboulite_host.so  host C: SMI seam, radio trio, SoapySDR stream calls

This package is only the Python face used by tests and bench.py (ctypes; torch
supplies device memory, streams and torch.distributed).  There is no CPU
fallback: loading fails loudly when the HIP library is missing.
"""
# (Nothing here touches the process's HIP runtime configuration: the C libraries copy caller-owned pageable memory in pieces
# below the runtime's in-place pinning threshold themselves, clhip_memcpy_h2d / _d2h.  tests/conftest.py and bench.py set
# GPU_PINNED_MIN_XFER_SIZE for THEIR OWN torch uploads of numpy temporaries)

from . import hip  # noqa: F401,E402

__all__ = ["hip"]
