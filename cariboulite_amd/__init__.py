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
import os as _os

# The tests, bench.py and the tools move numpy arrays to and from the device with torch.  The HIP runtime copies pageable
# host memory of 1 MiB and more by pinning the caller's pages in place, and that path has ended long test sessions with a
# GPU page fault on a host heap address.  With the threshold (MiB) out of reach the runtime stages
# such copies through its own pinned buffers.  Only a default, and only effective when this package is imported before the
# process's first HIP call (importing torch is not one); the C libraries themselves never rely on it -- they copy
# caller-owned pageable memory in pieces below the threshold (clhip_memcpy_h2d / _d2h).
_os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

from . import hip  # noqa: F401,E402

__all__ = ["hip"]
