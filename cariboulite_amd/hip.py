"""ctypes binding of libcariboulite_hip.so (layer 1 of include/cariboulite_hip.h).

Accepts torch CUDA(HIP) tensors or raw integer device pointers.  Fails loudly
(ImportError / RuntimeError) when the library is absent -- there is no CPU path.
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libcariboulite_hip.so")

CHANNEL_S1G, CHANNEL_HIF = 0, 1
FORMAT_CF32, FORMAT_CS16, FORMAT_CS8, FORMAT_CF64 = 0, 1, 2, 3
PIPE_IN_SMI_WORDS, PIPE_IN_CS16, PIPE_IN_CF32 = 0, 1, 2
PIPE_OUT_IQ, PIPE_OUT_FM_DEMOD = 0, 1
TX_DOCUMENTED, TX_AS_WRITTEN = 0, 1
TXPIPE_IN_FM_MESSAGE, TXPIPE_IN_CF32 = 0, 1
NATIVE_BATCH_LEN = 524288

_lib = None

_SIGS = {
    "clhip_device_count": (C.c_int, []),
    "clhip_set_device": (C.c_int, [C.c_int]),
    "clhip_last_error": (C.c_char_p, []),
    "clhip_arch_name": (C.c_char_p, []),
    "clhip_malloc": (C.c_void_p, [C.c_size_t]),
    "clhip_free": (None, [C.c_void_p]),
    "clhip_host_alloc": (C.c_void_p, [C.c_size_t]),
    "clhip_host_free": (None, [C.c_void_p]),
    "clhip_host_device_ptr": (C.c_void_p, [C.c_void_p]),
    "clhip_host_register": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "clhip_host_unregister": (None, [C.c_void_p]),
    "clhip_smi_unpack_aligned": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clhip_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clhip_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clhip_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clhip_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "clhip_stream_create": (C.c_void_p, []),
    "clhip_stream_destroy": (None, [C.c_void_p]),
    "clhip_stream_sync": (C.c_int, [C.c_void_p]),
    "clhip_event_create": (C.c_void_p, []),
    "clhip_event_destroy": (None, [C.c_void_p]),
    "clhip_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clhip_event_elapsed_ms": (C.c_float, [C.c_void_p, C.c_void_p]),
    "clhip_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clhip_memcpy2d_h2d": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "clhip_event_sync": (C.c_int, [C.c_void_p]),
    "clhip_debug_ops": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "clhip_debug_ops_dump": (None, [C.c_int]),
    "clhip_debug_copy_counters": (None, [C.c_void_p]),
    "clhip_debug_sticky_error": (C.c_int, []),
    "clhip_smi_find_offsets": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "clhip_smi_unpack": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clhip_convert_from_cs16": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "clhip_convert_to_cs16": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    "clhip_smi_debug_analyze": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]),
    "clhip_smi_pack": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "clhip_convert_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "clhip_convert_pack_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "clhip_take_i_rail": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "clhip_rows_to_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "clhip_words_to_rows": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "clhip_sync_tags_ws_bytes": (C.c_size_t, [C.c_size_t]),
    "clhip_sync_tags": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clhip_iir_create": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "clhip_iir_destroy": (None, [C.c_void_p]),
    "clhip_iir_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "clhip_iir_status": (C.c_int, [C.c_void_p]),
    "clhip_iir_finish": (C.c_int, [C.c_void_p]),
    "clhip_iir_unrun": (C.c_int, [C.c_void_p]),
    "clhip_iir_set_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clhip_iir_get_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clhip_iir_set_poll_bound": (None, [C.c_void_p, C.c_int]),
    "clhip_iir_set_shape": (None, [C.c_void_p, C.c_int, C.c_int]),
    "clhip_iir_on_scan_path": (C.c_int, [C.c_void_p]),
    "clhip_iir_debug_stamps": (C.c_size_t, [C.c_void_p, C.c_void_p]),
    "clhip_iir_memory_samples": (C.c_size_t, [C.c_void_p]),
    "clhip_iir_run_smi": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "clhip_rx_pipe_create": (C.c_void_p, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "clhip_rx_pipe_destroy": (None, [C.c_void_p]),
    "clhip_rx_pipe_reset": (None, [C.c_void_p]),
    "clhip_rx_pipe_seek": (None, [C.c_void_p, C.c_ulonglong]),
    "clhip_rx_pipe_halo": (C.c_size_t, [C.c_void_p]),
    "clhip_rx_pipe_out_count": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "clhip_rx_pipe_uses_fused": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int]),
    "clhip_rx_pipe_run": (C.c_long, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clhip_rx_pipe_force_generic": (None, [C.c_void_p, C.c_int]),
    "clhip_rx_pipe_set_diag": (None, [C.c_void_p, C.c_void_p]),
    "clhip_rx_pipe_set_sync_check": (None, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clhip_rx_pipe_rollback": (C.c_int, [C.c_void_p]),
    "clhip_rx_pipe_epoch_begin": (C.c_int, [C.c_void_p]),
    "clhip_rx_pipe_run_range": (C.c_long, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clhip_rx_pipe_epoch_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clhip_rx_pipe_unrun_stream": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "clhip_rx_pipe_out_count_stream": (C.c_size_t, [C.c_void_p, C.c_int, C.c_size_t]),
    "clhip_rx_pipe_stream_total": (C.c_ulonglong, [C.c_void_p, C.c_int]),
    "clhip_rx_pipe_set_host_sink": (None, [C.c_void_p, C.c_void_p]),
    "clhip_rx_pipe_set_offs_writeback": (None, [C.c_void_p, C.c_int]),
    "clhip_rx_pipe_out_elem_bytes": (C.c_size_t, [C.c_void_p]),
    "clhip_rx_pipe_run_smi": (C.c_long, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clhip_tx_pipe_create": (C.c_void_p, [C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "clhip_tx_pipe_destroy": (None, [C.c_void_p]),
    "clhip_tx_pipe_reset": (None, [C.c_void_p]),
    "clhip_tx_pipe_out_count": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "clhip_tx_pipe_seek": (C.c_int, [C.c_void_p, C.c_ulonglong, C.c_void_p]),
    "clhip_tx_pipe_run": (C.c_long, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "clhip_tx_pipe_status": (C.c_int, [C.c_void_p]),
    "clhip_tx_pipe_set_poll_bound": (None, [C.c_void_p, C.c_int]),
    "clhip_tx_pipe_position": (C.c_ulonglong, [C.c_void_p]),
    "clhip_tx_pipe_pack_mode": (C.c_int, [C.c_void_p]),
    "clhip_tx_pipe_set_position": (C.c_int, [C.c_void_p, C.c_ulonglong]),
    "clhip_tx_pipe_move_stream": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "clhip_fm_demod": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clhip_fm_mod": (C.c_int, [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clhip_fm_mod_workspace_bytes": (C.c_size_t, [C.c_size_t]),
    "clhip_cw_tone": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_size_t, C.c_void_p, C.c_void_p]),
}


def exported_symbols():
    """Every layer-1 symbol include/cariboulite_hip.h declares."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("CLHIP_LIB") or LIB_PATH      # an alternative build of the shim (diagnostic / ablation builds)
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP extension is required; there is no CPU fallback)")
        _lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
    return _lib


def last_error():
    return lib().clhip_last_error().decode()


def _check(rc, what):
    if rc is None or rc < 0:
        raise RuntimeError(f"{what} failed: {last_error()}")
    return rc


def ptr(t):
    """Device pointer of a torch tensor (or pass through an int / None)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    l = lib()
    if l.clhip_device_count() <= 0:
        raise RuntimeError("no HIP device visible: the cariboulite_amd hot path needs an MI355X")
    return l.clhip_arch_name().decode()


# ------------------------------------------------------------------ stages
def smi_find_offsets(d_bytes, total_bytes, chunk_stride, chunk_len, n_chunks, d_offs, stream=None):
    _check(lib().clhip_smi_find_offsets(ptr(d_bytes), total_bytes, chunk_stride, chunk_len, n_chunks,
                                        ptr(d_offs), stream if stream is not None else current_stream()),
           "clhip_smi_find_offsets")


def smi_unpack(channel, d_bytes, total_bytes, chunk_stride, chunk_len, n_chunks, d_offs, fmt, d_out,
               d_meta=None, stream=None):
    _check(lib().clhip_smi_unpack(channel, ptr(d_bytes), total_bytes, chunk_stride, chunk_len, n_chunks,
                                  ptr(d_offs), fmt, ptr(d_out), ptr(d_meta),
                                  stream if stream is not None else current_stream()), "clhip_smi_unpack")


def sync_tags(d_meta, n, d_idx, cap, d_count, d_ws=None, stream=None):
    """pps tags of a meta plane (caribouLiteSource_impl.cc:113-119): d_idx[:min(count, cap)] = ascending positions of
    meta == 1, d_count[0] = how many there are.  d_meta may be a torch tensor or a raw device address (any alignment)."""
    _check(lib().clhip_sync_tags(ptr(d_meta), n, ptr(d_idx), cap, ptr(d_count),
                                 ptr(d_ws), stream if stream is not None else current_stream()), "clhip_sync_tags")


def convert_from_cs16(d_iq, n, fmt, d_out, stream=None):
    _check(lib().clhip_convert_from_cs16(ptr(d_iq), n, fmt, ptr(d_out),
                                         stream if stream is not None else current_stream()), "clhip_convert_from_cs16")


def convert_to_cs16(d_in, fmt, n, d_iq, stream=None):
    _check(lib().clhip_convert_to_cs16(ptr(d_in), fmt, n, ptr(d_iq),
                                       stream if stream is not None else current_stream()), "clhip_convert_to_cs16")


def smi_pack(mode, d_iq, n, d_bytes, stream=None):
    _check(lib().clhip_smi_pack(mode, ptr(d_iq), n, ptr(d_bytes),
                                stream if stream is not None else current_stream()), "clhip_smi_pack")


class RxPipe:
    """clhip_rx_pipe: unpack -> FIR -> [L/M resample | FM demod] for n_streams streams."""

    def __init__(self, n_streams, channel, fir_taps, rs_taps=None, up=1, down=1, out_mode=PIPE_OUT_IQ):
        fir = np.ascontiguousarray(fir_taps, dtype=np.float32)
        rs = np.ascontiguousarray(rs_taps, dtype=np.float32) if rs_taps is not None else None
        self.n_streams, self.up, self.down, self.out_mode = n_streams, up, down, out_mode
        self.h = lib().clhip_rx_pipe_create(n_streams, channel, fir.ctypes.data, fir.size,
                                            rs.ctypes.data if rs is not None else None,
                                            rs.size if rs is not None else 0, up, down, out_mode)
        if not self.h:
            raise RuntimeError("clhip_rx_pipe_create failed: " + last_error())

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.clhip_rx_pipe_destroy(self.h)
        self.h = None

    __del__ = close

    def reset(self):
        lib().clhip_rx_pipe_reset(self.h)

    def seek(self, n_total):
        lib().clhip_rx_pipe_seek(self.h, n_total)

    def halo(self):
        return lib().clhip_rx_pipe_halo(self.h)

    def out_count(self, n_in):
        return lib().clhip_rx_pipe_out_count(self.h, n_in)

    def uses_fused(self, n_in, in_kind=PIPE_IN_SMI_WORDS):
        return bool(lib().clhip_rx_pipe_uses_fused(self.h, n_in, in_kind))

    def force_generic(self, on=True):
        lib().clhip_rx_pipe_force_generic(self.h, int(on))

    def set_diag(self, d_buf):
        self._diag = d_buf
        lib().clhip_rx_pipe_set_diag(self.h, ptr(d_buf))

    def set_sync_check(self, d_offs, chunk_samples, d_bad_flag):
        self._chk = (d_offs, d_bad_flag)          # keep the tensors alive
        lib().clhip_rx_pipe_set_sync_check(self.h, ptr(d_offs), chunk_samples, ptr(d_bad_flag))

    def run(self, in_kind, d_in, in_stride, n_in, d_out, out_stride, stream=None):
        return _check(lib().clhip_rx_pipe_run(self.h, in_kind, ptr(d_in), in_stride, n_in, ptr(d_out), out_stride,
                                              stream if stream is not None else current_stream()),
                      "clhip_rx_pipe_run")

    def rollback(self):
        return lib().clhip_rx_pipe_rollback(self.h)

    # streams advancing independently (stream groups): an epoch of range runs
    def epoch_begin(self):
        _check(lib().clhip_rx_pipe_epoch_begin(self.h), "clhip_rx_pipe_epoch_begin")

    def run_range(self, first, count, in_kind, d_in, in_stride, n_in, d_out, out_stride, stream=None):
        """d_in / d_out: tensors (or views) that start at stream `first`'s row"""
        return _check(lib().clhip_rx_pipe_run_range(self.h, first, count, in_kind, ptr(d_in), in_stride, n_in, ptr(d_out), out_stride,
                                                    stream if stream is not None else current_stream()), "clhip_rx_pipe_run_range")

    def epoch_end(self, stream=None):
        _check(lib().clhip_rx_pipe_epoch_end(self.h, stream if stream is not None else current_stream()), "clhip_rx_pipe_epoch_end")

    def unrun_stream(self, s, n_in):
        _check(lib().clhip_rx_pipe_unrun_stream(self.h, s, n_in), "clhip_rx_pipe_unrun_stream")

    def out_count_stream(self, s, n_in):
        return lib().clhip_rx_pipe_out_count_stream(self.h, s, n_in)

    def stream_total(self, s):
        return lib().clhip_rx_pipe_stream_total(self.h, s)

    def run_smi(self, d_bytes, stream_stride_bytes, n_bytes, chunk_len_bytes, d_offs, d_cs16, d_out, out_stride,
                h_offs=None, stream=None):
        """caribou_smi_read-shaped call; returns outputs per stream or CL_SMI_ERR_SYNC (-3) (other errors raise)."""
        rc = lib().clhip_rx_pipe_run_smi(self.h, ptr(d_bytes), stream_stride_bytes, n_bytes, chunk_len_bytes, ptr(d_offs),
                                         h_offs.ctypes.data if h_offs is not None else None, ptr(d_cs16), ptr(d_out),
                                         out_stride, stream if stream is not None else current_stream())
        if rc < 0 and rc not in (-3, -4):
            raise RuntimeError("clhip_rx_pipe_run_smi failed: " + last_error())
        return rc


class IIR:
    """clhip_iir: fp64 biquad cascade on CS16 streams; the object carries the state (one per Soapy stream and filter)."""

    def __init__(self, sos, n_streams=1, device="cuda:0"):
        sos = np.asarray(sos, dtype=np.float64)
        if sos.shape[1] == 6:                       # scipy layout b0 b1 b2 a0 a1 a2 -> {b0,b1,b2,a1,a2}
            sos = np.concatenate([sos[:, :3] / sos[:, 3:4], sos[:, 4:] / sos[:, 3:4]], 1)
        self.sos = np.ascontiguousarray(sos)
        self.n_streams = n_streams
        require_gpu()
        self.h = lib().clhip_iir_create(self.sos.ctypes.data, self.sos.shape[0], n_streams)
        if not self.h:
            raise RuntimeError("clhip_iir_create failed: " + last_error())

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:          # (module globals may be gone at interpreter exit)
            lib().clhip_iir_destroy(self.h)
            self.h = None

    def run(self, d_iq, n, stride=None, stream=None, out=None):
        """in place unless `out` is given; asynchronous"""
        _check(lib().clhip_iir_run(self.h, ptr(d_iq), ptr(out if out is not None else d_iq), n if stride is None else stride, n,
                                   stream if stream is not None else current_stream()), "clhip_iir_run")

    def run_smi(self, channel, d_words, n, out, stride=None, stream=None):
        """raw in-sync SMI words in, filtered CS16 out (out of place); returns -2 when the call would take the scan path"""
        rc = lib().clhip_iir_run_smi(self.h, channel, ptr(d_words), ptr(out), n if stride is None else stride, n,
                                     stream if stream is not None else current_stream())
        if rc == -1:
            raise RuntimeError("clhip_iir_run_smi failed: " + last_error())
        return rc

    def status(self):
        """after a synchronise: 0 = the last run is good, -1 = it overran (state restored, object now on the scan path)"""
        return lib().clhip_iir_status(self.h)

    def finish(self):
        """synchronise + verdict + repeat on the scan path when the call was out of place: 0 good, 1 repaired, -1 failed"""
        return lib().clhip_iir_finish(self.h)

    def unrun(self):
        """take the last run back (after a synchronise): the carried state is what it was before it"""
        return lib().clhip_iir_unrun(self.h)

    def set_poll_bound(self, polls):
        lib().clhip_iir_set_poll_bound(self.h, int(polls))

    def memory_samples(self):
        """samples after which nothing of an earlier state is left above 1e-12 (0: longer than the kernel's horizon)"""
        return lib().clhip_iir_memory_samples(self.h)

    def on_scan_path(self):
        return bool(lib().clhip_iir_on_scan_path(self.h))

    def debug_stamps(self):
        """([64 waves][16 tiles][12 phases] stamps, [8192 waves][start, end, tiles | HW_ID << 16 | XCC_ID << 48, ticks spent polling,
        first chunk | class << 32]) of the last launch (CLHIP_IIR_STAMPS=1), or None"""
        n = lib().clhip_iir_debug_stamps(self.h, None)
        if not n:
            return None
        out = np.zeros(n, dtype=np.uint64)
        lib().clhip_iir_debug_stamps(self.h, out.ctypes.data)
        return out[:64 * 16 * 12].reshape(64, 16, 12), out[64 * 16 * 12:].reshape(-1, 5)

    @property
    def state(self):
        st = np.zeros((self.n_streams, 16), dtype=np.float64)
        _check(lib().clhip_iir_get_state(self.h, st.ctypes.data), "clhip_iir_get_state")
        return st

    @state.setter
    def state(self, st):
        st = None if st is None else np.ascontiguousarray(st, dtype=np.float64).reshape(self.n_streams, 16)
        _check(lib().clhip_iir_set_state(self.h, st.ctypes.data if st is not None else None), "clhip_iir_set_state")


class TxPipe:
    """clhip_tx_pipe: [FM modulate] -> L/M resample -> quantise -> int13 pack."""

    def __init__(self, n_streams, kf_hz, fs_hz, rs_taps=None, up=1, down=1, pack_mode=TX_DOCUMENTED):
        rs = np.ascontiguousarray(rs_taps, dtype=np.float32) if rs_taps is not None else None
        self.h = lib().clhip_tx_pipe_create(n_streams, kf_hz, fs_hz, rs.ctypes.data if rs is not None else None,
                                            rs.size if rs is not None else 0, up, down, pack_mode)
        if not self.h:
            raise RuntimeError("clhip_tx_pipe_create failed: " + last_error())

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.clhip_tx_pipe_destroy(self.h)
        self.h = None

    __del__ = close

    def reset(self):
        lib().clhip_tx_pipe_reset(self.h)

    def out_count(self, n_in):
        return lib().clhip_tx_pipe_out_count(self.h, n_in)

    def seek(self, n_total, phase_rad):
        """reset + place the pipe at message n_total of a longer stream with the modulator's phase (radians, per stream) there"""
        ph = np.ascontiguousarray(np.atleast_1d(phase_rad), dtype=np.float64)
        _check(lib().clhip_tx_pipe_seek(self.h, int(n_total), ph.ctypes.data), "clhip_tx_pipe_seek")

    def status(self):
        """0 = the last run's bytes are valid (ask after synchronising its stream), -1 = look-back overrun."""
        return lib().clhip_tx_pipe_status(self.h)

    def set_poll_bound(self, polls):
        lib().clhip_tx_pipe_set_poll_bound(self.h, polls)

    def position(self):
        return int(lib().clhip_tx_pipe_position(self.h))

    def set_position(self, n_total):
        _check(lib().clhip_tx_pipe_set_position(self.h, int(n_total)), "clhip_tx_pipe_set_position")

    def take_stream_from(self, dst_stream, src, src_stream, stream=None):
        """the carried state of src's stream `src_stream` moves into this pipe's stream `dst_stream`"""
        _check(lib().clhip_tx_pipe_move_stream(self.h, dst_stream, src.h, src_stream, stream), "clhip_tx_pipe_move_stream")

    def run(self, in_kind, d_in, in_stride, n_in, d_bytes, out_stride_bytes, d_tap=None, tap_stride=0, stream=None):
        return _check(lib().clhip_tx_pipe_run(self.h, in_kind, ptr(d_in), in_stride, n_in, ptr(d_bytes),
                                              out_stride_bytes, ptr(d_tap), tap_stride,
                                              stream if stream is not None else current_stream()),
                      "clhip_tx_pipe_run")


def fm_demod(d_iq, n, d_prev, d_out, stream=None):
    _check(lib().clhip_fm_demod(ptr(d_iq), n, ptr(d_prev), ptr(d_out),
                                stream if stream is not None else current_stream()), "clhip_fm_demod")


def fm_mod(d_msg, n, kf_hz, fs_hz, d_phase, d_iq_out, d_ws, stream=None):
    _check(lib().clhip_fm_mod(ptr(d_msg), n, kf_hz, fs_hz, ptr(d_phase), ptr(d_iq_out), ptr(d_ws),
                              d_ws.numel() * d_ws.element_size(),
                              stream if stream is not None else current_stream()), "clhip_fm_mod")


def cw_tone(f_hz, fs_hz, phase0, n, d_iq_out, stream=None):
    _check(lib().clhip_cw_tone(f_hz, fs_hz, phase0, n, ptr(d_iq_out),
                               stream if stream is not None else current_stream()), "clhip_cw_tone")
