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
from . import hip  # noqa: F401

__all__ = ["hip"]
