"""ctypes face of libcariboulite_fanout.so (include/cariboulite_fanout.h): the fan-out / fan-in of raw stream buffers
between the GPUs of one node over RCCL point-to-point.  One process per GPU; the 128-byte RCCL id is made on rank 0
and carried to the others by the caller (torch.distributed here)."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libcariboulite_fanout.so")
ID_BYTES = 128

_SIGS = {
    "clfan_unique_id": (C.c_int, [C.c_void_p]),
    "clfan_create": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "clfan_destroy": (None, [C.c_void_p]),
    "clfan_world": (C.c_int, [C.c_void_p]),
    "clfan_rank": (C.c_int, [C.c_void_p]),
    "clfan_last_error": (C.c_char_p, []),
    "clfan_local_count": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "clfan_scatter_streams": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clfan_gather_streams": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
}
_lib = None


def exported_symbols():
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


class Comm:
    """clfan_comm of this process (its GPU must be the current device)."""

    def __init__(self, world, rank, unique_id=None):
        idbuf = (C.c_uint8 * ID_BYTES)(*(unique_id or [0] * ID_BYTES))
        self.h = lib().clfan_create(idbuf, world, rank)
        if not self.h:
            raise RuntimeError("clfan_create failed: " + lib().clfan_last_error().decode())
        self.world, self.rank = world, rank

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * ID_BYTES)()
        if lib().clfan_unique_id(buf) != 0:
            raise RuntimeError("clfan_unique_id failed: " + lib().clfan_last_error().decode())
        return list(buf)

    def scatter(self, root, d_root, root_stride, stream_bytes, n_streams, d_mine, mine_stride, stream):
        if lib().clfan_scatter_streams(self.h, root, d_root, root_stride, stream_bytes, n_streams, d_mine, mine_stride, stream) != 0:
            raise RuntimeError("clfan_scatter_streams failed: " + lib().clfan_last_error().decode())

    def gather(self, root, d_mine, mine_stride, stream_bytes, n_streams, d_root, root_stride, stream):
        if lib().clfan_gather_streams(self.h, root, d_mine, mine_stride, stream_bytes, n_streams, d_root, root_stride, stream) != 0:
            raise RuntimeError("clfan_gather_streams failed: " + lib().clfan_last_error().decode())

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.clfan_destroy(self.h)
        self.h = None

    __del__ = close
