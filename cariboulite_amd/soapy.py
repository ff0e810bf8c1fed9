"""SoapySDR-shaped Python face of libcariboulite_host.so (layer 2 of
include/cariboulite_hip.h) so that tests and examples read like the reference's
own SoapySDR clients (examples/python/read_test.py):

    sdr = Device(dict(driver="Cariboulite", channel="S1G"))
    rx = sdr.setupStream(SOAPY_SDR_RX, SOAPY_SDR_CS16)
    sdr.activateStream(rx)
    sr = sdr.readStream(rx, [buf], len(buf))      # sr.ret

The only addition is the byte injection that replaces /dev/smi:
sdr.feedSmiBytes(b) / sdr.drainSmiBytes().
"""
import ctypes as C
import os

import numpy as np

from . import hip as _hip

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libcariboulite_host.so")

SOAPY_SDR_TX, SOAPY_SDR_RX = 0, 1
SOAPY_SDR_CS16, SOAPY_SDR_CS8, SOAPY_SDR_CF32, SOAPY_SDR_CF64 = "CS16", "CS8", "CF32", "CF64"
SOAPY_SDR_NOT_SUPPORTED = -5
SMI_ERR_IO, SMI_ERR_DEBUGMODE, SMI_ERR_SYNC = -1, -2, -3

_SIGS = {
    "cl_smi_init": (C.c_void_p, [C.c_int]),
    "cl_smi_close": (C.c_int, [C.c_void_p]),
    "cl_smi_feed_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_smi_feed_reserve": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "cl_smi_feed_commit": (C.c_int, [C.c_void_p, C.c_size_t]),
    "cl_smi_pending_bytes": (C.c_size_t, [C.c_void_p]),
    "cl_smi_feed_fd": (C.c_long, [C.c_void_p, C.c_int, C.c_size_t]),
    "cl_smi_feed_file": (C.c_long, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t]),
    "cl_smi_drain_to_fd": (C.c_long, [C.c_void_p, C.c_int, C.c_size_t]),
    "cl_smi_set_max_read": (None, [C.c_void_p, C.c_size_t]),
    "cl_smi_drain_bytes": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_smi_set_tx_mode": (None, [C.c_void_p, C.c_int]),
    "cl_smi_read": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_smi_write": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "cl_smi_get_native_batch_samples": (C.c_size_t, [C.c_void_p]),
    "cl_smi_set_debug_mode": (None, [C.c_void_p, C.c_int]),
    "cl_smi_get_debug_data": (C.c_void_p, [C.c_void_p]),
    "cl_smi_set_debug_clock": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cl_smi_read_to_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_smi_write_from_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "cl_smi_flush_fifo": (C.c_int, [C.c_void_p]),
    "cl_smi_stream": (C.c_void_p, [C.c_void_p]),
    "cl_smi_device": (C.c_int, [C.c_void_p]),
    "cl_radio_read_samples_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_radio_write_samples_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_radio_smi": (C.c_void_p, [C.c_void_p]),
    "cl_ring_create": (C.c_void_p, [C.c_size_t, C.c_size_t, C.c_int, C.c_int]),
    "cl_ring_create_device": (C.c_void_p, [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int]),
    "cl_ring_storage": (C.c_void_p, [C.c_void_p]),
    "cl_ring_on_device": (C.c_int, [C.c_void_p]),
    "cl_ring_put_begin": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "cl_ring_put_end": (None, [C.c_void_p, C.c_size_t]),
    "cl_ring_put_cancel": (None, [C.c_void_p]),
    "cl_ring_put_abandon": (None, [C.c_void_p]),
    "cl_ring_get_begin": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "cl_ring_get_end": (None, [C.c_void_p, C.c_size_t]),
    "cl_ring_destroy": (None, [C.c_void_p]),
    "cl_ring_put": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_ring_get": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "cl_ring_reset": (None, [C.c_void_p]),
    "cl_ring_size": (C.c_size_t, [C.c_void_p]),
    "cl_ring_capacity": (C.c_size_t, [C.c_void_p]),
    "cl_radio_create": (C.c_void_p, [C.c_void_p, C.c_int]),
    "cl_radio_destroy": (None, [C.c_void_p]),
    "cl_radio_read_samples": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_radio_write_samples": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_radio_get_native_mtu_size_samples": (C.c_size_t, [C.c_void_p]),
    "cl_device_make": (C.c_void_p, [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_size_t]),
    "cl_device_unmake": (None, [C.c_void_p]),
    "cl_device_smi": (C.c_void_p, [C.c_void_p]),
    "cl_device_last_error": (C.c_char_p, [C.c_void_p]),
    "cl_getStreamFormats": (C.c_size_t, [C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t]),
    "cl_getNativeStreamFormat": (C.c_char_p, [C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_double)]),
    "cl_setupStream": (C.c_void_p, [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_size_t]),
    "cl_stream_register_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cl_stream_unregister_buffers": (None, [C.c_void_p, C.c_void_p]),
    "cl_closeStream": (None, [C.c_void_p, C.c_void_p]),
    "cl_getStreamMTU": (C.c_size_t, [C.c_void_p, C.c_void_p]),
    "cl_activateStream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_size_t]),
    "cl_deactivateStream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong]),
    "cl_readStream": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_int),
                                C.POINTER(C.c_longlong), C.c_long]),
    "cl_writeStream": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_int),
                                 C.c_longlong, C.c_long]),
    "cl_setBandwidth": (None, [C.c_void_p, C.c_int, C.c_size_t, C.c_double]),
    "cl_getDigitalFilter": (C.c_int, [C.c_void_p]),
    "cl_stream_queue_size": (C.c_size_t, [C.c_void_p, C.c_void_p]),
    "cl_stream_iir_overruns": (C.c_ulong, [C.c_void_p]),
    "cl_getStreamStats": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cl_smi_get_stats": (None, [C.c_void_p, C.c_void_p]),
    "cl_stream_set_iir_poll_bound": (None, [C.c_void_p, C.c_int]),
    "cl_group_make": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_size_t]),
    "cl_group_unmake": (None, [C.c_void_p]),
    "cl_group_size": (C.c_size_t, [C.c_void_p]),
    "cl_group_readStream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_int), C.c_long]),
    "cl_group_writeStream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_int), C.c_long]),
    "cl_group_set_iir_poll_bound": (None, [C.c_void_p, C.c_int]),
    "cl_group_set_tx_poll_bound": (None, [C.c_void_p, C.c_int]),
    "cl_group_flush": (C.c_int, [C.c_void_p]),
    "cl_node_make": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_size_t]),
    "cl_node_unmake": (None, [C.c_void_p]),
    "cl_node_size": (C.c_size_t, [C.c_void_p]),
    "cl_node_shards": (C.c_size_t, [C.c_void_p]),
    "cl_node_group": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "cl_node_shard_of": (C.c_int, [C.c_void_p, C.c_size_t]),
    "cl_node_readStream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_int), C.c_long]),
    "cl_node_writeStream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_int), C.c_long]),
    "cl_node_flush": (C.c_int, [C.c_void_p]),
    "cl_node_register_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "cl_node_unregister_buffers": (None, [C.c_void_p]),
    "cl_node_last_error": (C.c_char_p, [C.c_void_p]),
    "cl_group_last_error": (C.c_char_p, [C.c_void_p]),
    "cl_group_getStats": (None, [C.c_void_p, C.c_void_p]),
    "cl_group_register_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "cl_group_unregister_buffers": (None, [C.c_void_p]),
    "cl_design_lowpass": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "cl_design_butter_lowpass": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_void_p]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        _hip.lib()                       # RTLD_GLOBAL: libcariboulite_hip.so first
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def _kwargs(d):
    d = {str(k): str(v) for k, v in (d or {}).items()}
    n = len(d)
    K = (C.c_char_p * max(n, 1))(*[k.encode() for k in d])
    V = (C.c_char_p * max(n, 1))(*[v.encode() for v in d.values()])
    return K, V, n


def design_lowpass(n_taps, cutoff_hz, fs_hz, gain=1.0):
    out = np.empty(n_taps, dtype=np.float32)
    if lib().cl_design_lowpass(n_taps, cutoff_hz, fs_hz, gain, out.ctypes.data) != 0:
        raise ValueError("cl_design_lowpass: bad arguments")
    return out


def design_butter_lowpass(order, fs_hz, fc_hz):
    out = np.empty((order // 2, 5), dtype=np.float64)
    if lib().cl_design_butter_lowpass(order, fs_hz, fc_hz, out.ctypes.data) != 0:
        raise ValueError("cl_design_butter_lowpass: bad arguments")
    return out


class SmiDebugData(C.Structure):
    _fields_ = [("error_accum_counter", C.c_uint32), ("cur_err_cnt", C.c_uint32),
                ("last_correct_byte", C.c_uint8), ("error_rate", C.c_double), ("bitrate", C.c_double),
                ("last_time_sec", C.c_long), ("last_time_usec", C.c_long)]


SMI_CLOCK_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long))


class Ring:
    """cl_ring over uint32 elements (the reference's circular_buffer<T>)."""

    def __init__(self, size, override_write=True, block_read=True, device=None):
        if device is None:
            self.h = lib().cl_ring_create(size, 4, int(override_write), int(block_read))
        else:                                        # storage in device memory
            self.h = lib().cl_ring_create_device(device, size, 4, int(override_write), int(block_read))
        if not self.h:
            raise RuntimeError("cl_ring_create failed")

    def put(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint32)
        return lib().cl_ring_put(self.h, d.ctypes.data, d.size)

    def get(self, n, timeout_us=1000):
        out = np.zeros(max(n, 1), dtype=np.uint32)
        k = lib().cl_ring_get(self.h, out.ctypes.data, n, int(timeout_us))
        return k, out[:k].copy()

    def size(self):
        return lib().cl_ring_size(self.h)

    def capacity(self):
        return lib().cl_ring_capacity(self.h)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.cl_ring_destroy(self.h); self.h = None


class StreamResult:
    def __init__(self, ret, flags=0, timeNs=0):
        self.ret, self.flags, self.timeNs = ret, flags, timeNs


class Device:
    """Stands where SoapySDR.Device(dict(driver="Cariboulite", ...)) stands."""

    def __init__(self, args):
        K, V, n = _kwargs(args)
        self.h = lib().cl_device_make(K, V, n)
        if not self.h:
            # Cariboulite.cpp:25-34 throws; no GPU also lands here (there is no CPU fallback)
            raise RuntimeError("Cariboulite device make failed: channel must be S1G or HiF and an MI355X must be visible")
        self.smi = lib().cl_device_smi(self.h)

    def close(self):
        grp = getattr(self, "_group", None)
        grp = grp() if grp is not None else None
        if grp is not None:
            grp.close()                             # (a group holds its members' seams: it goes first, they are ordinary devices again)
        if getattr(self, "h", None) and _lib is not None:  # (at interpreter shutdown the module's globals may be gone already)
            _lib.cl_device_unmake(self.h)
        self.h = None

    __del__ = close

    # ---- injection replacing /dev/smi
    def feedSmiBytes(self, b):
        b = np.ascontiguousarray(b, dtype=np.uint8)
        return lib().cl_smi_feed_bytes(self.smi, b.ctypes.data, b.size)

    def feedSmiFile(self, path, offset=0, max_bytes=1 << 62):
        return lib().cl_smi_feed_file(self.smi, str(path).encode(), offset, max_bytes)

    def feedSmiFd(self, fd, max_bytes=1 << 62):
        return lib().cl_smi_feed_fd(self.smi, fd, max_bytes)

    def drainSmiToFd(self, fd, max_bytes=1 << 62):
        return lib().cl_smi_drain_to_fd(self.smi, fd, max_bytes)

    def pendingSmiBytes(self):
        return lib().cl_smi_pending_bytes(self.smi)

    def flushSmiFifo(self):
        return lib().cl_smi_flush_fifo(self.smi)

    def setMaxRead(self, n):
        lib().cl_smi_set_max_read(self.smi, n)

    def setTxMode(self, mode):
        lib().cl_smi_set_tx_mode(self.smi, mode)

    def drainSmiBytes(self, max_bytes=1 << 26):
        out = np.empty(max_bytes, dtype=np.uint8)
        n = lib().cl_smi_drain_bytes(self.smi, out.ctypes.data, max_bytes)
        return out[:n].copy()

    def setSmiDebugMode(self, mode):
        lib().cl_smi_set_debug_mode(self.smi, mode)

    def smiDebugData(self):
        d = C.cast(lib().cl_smi_get_debug_data(self.smi), C.POINTER(SmiDebugData)).contents
        return (d.error_accum_counter, d.cur_err_cnt, d.last_correct_byte, d.error_rate)

    def smiDebugBitrate(self):
        d = C.cast(lib().cl_smi_get_debug_data(self.smi), C.POINTER(SmiDebugData)).contents
        return d.bitrate, (d.last_time_sec, d.last_time_usec)

    def setSmiDebugClock(self, readings):
        """Replay (sec, usec) clock readings, one per analysed chunk (None: back to gettimeofday)."""
        if readings is None:
            self._clock_cb = None
            lib().cl_smi_set_debug_clock(self.smi, None, None)
            return
        it = iter(readings)

        def now(_user, sec, usec):
            s_, u_ = next(it)
            sec[0], usec[0] = s_, u_
        self._clock_cb = SMI_CLOCK_FN(now)          # kept alive with the device
        lib().cl_smi_set_debug_clock(self.smi, self._clock_cb, None)

    # ---- lower seam (caribou_smi_read / caribou_smi_write)
    def smiRead(self, channel, n, want_meta=True, fill=-21846):
        iq = np.full((n + 2, 2), fill, dtype=np.int16)
        meta = np.full(n + 2, 0xAA, dtype=np.uint8) if want_meta else None
        ret = lib().cl_smi_read(self.smi, channel, iq.ctypes.data, meta.ctypes.data if want_meta else None, n)
        return ret, iq, meta

    def smiWrite(self, channel, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
        return lib().cl_smi_write(self.smi, channel, iq.ctypes.data, iq.shape[0])

    # ---- SoapySDR::Device surface
    def getStreamFormats(self, direction, channel):
        arr = (C.c_char_p * 8)()
        n = lib().cl_getStreamFormats(self.h, direction, channel, arr, 8)
        return [arr[i].decode() for i in range(n)]

    def getNativeStreamFormat(self, direction, channel):
        fs = C.c_double(0)
        f = lib().cl_getNativeStreamFormat(self.h, direction, channel, C.byref(fs))
        return f.decode(), fs.value

    def setupStream(self, direction, fmt, channels=(0,), args=None):
        K, V, n = _kwargs(args)
        ch = (C.c_size_t * max(len(channels), 1))(*channels)
        st = lib().cl_setupStream(self.h, direction, fmt.encode(), ch, len(channels), K, V, n)
        if not st:
            raise RuntimeError(lib().cl_device_last_error(self.h).decode())     # std::runtime_error in the reference
        return st

    def lastError(self):
        return lib().cl_device_last_error(self.h).decode()

    def registerStreamBuffer(self, st, buf):
        """ZEROCOPY=1 streams: `buf` (a numpy array the caller keeps alive) may be written by the last kernel of a read"""
        if lib().cl_stream_register_buffer(self.h, st, buf.ctypes.data, buf.nbytes) != 0:
            raise RuntimeError(self.lastError())
        self._zc_bufs = getattr(self, "_zc_bufs", []) + [buf]

    def unregisterStreamBuffers(self, st):
        lib().cl_stream_unregister_buffers(self.h, st)
        self._zc_bufs = []

    def closeStream(self, st):
        lib().cl_closeStream(self.h, st)

    def getStreamMTU(self, st):
        return lib().cl_getStreamMTU(self.h, st)

    def activateStream(self, st, flags=0, timeNs=0, numElems=0):
        return lib().cl_activateStream(self.h, st, flags, timeNs, numElems)

    def deactivateStream(self, st, flags=0, timeNs=0):
        return lib().cl_deactivateStream(self.h, st, flags, timeNs)

    def readStream(self, st, buffs, numElems, flags=0, timeoutUs=100000):
        p = (C.c_void_p * 1)(buffs[0].ctypes.data)
        fl, tn = C.c_int(flags), C.c_longlong(0)
        ret = lib().cl_readStream(self.h, st, p, numElems, C.byref(fl), C.byref(tn), timeoutUs)
        return StreamResult(ret, fl.value, tn.value)

    def writeStream(self, st, buffs, numElems, flags=0, timeNs=0, timeoutUs=100000):
        p = (C.c_void_p * 1)(buffs[0].ctypes.data)
        fl = C.c_int(flags)
        ret = lib().cl_writeStream(self.h, st, p, numElems, C.byref(fl), timeNs, timeoutUs)
        return StreamResult(ret, fl.value, timeNs)

    def setBandwidth(self, direction, channel, bw):
        lib().cl_setBandwidth(self.h, direction, channel, bw)

    def getDigitalFilter(self):
        return lib().cl_getDigitalFilter(self.h)

    def streamQueueSize(self, st):
        return lib().cl_stream_queue_size(self.h, st)

    def streamStats(self, st):
        out = (C.c_uint64 * 10)()
        lib().cl_getStreamStats(self.h, st, out)
        return dict(zip(("read_calls", "elements_read", "reads_empty", "iir_overruns", "write_calls", "elements_written",
                         "writes_empty", "tx_overruns", "zero_copy_registrations", "zero_copy_reads"), [int(v) for v in out]))

    def smiStats(self):
        out = (C.c_uint64 * 6)()
        lib().cl_smi_get_stats(self.smi, out)
        return dict(zip(("samples_read", "resyncs", "sync_losses", "timeouts", "io_errors", "samples_written"), [int(v) for v in out]))

    def streamIirOverruns(self, st):
        return lib().cl_stream_iir_overruns(st)

    def setStreamIirPollBound(self, st, polls):
        lib().cl_stream_set_iir_poll_bound(st, int(polls))


class Group:
    """cl_group: N devices of one GPU read in one call (include/cariboulite_hip.h, "stream group").  Make it after
    setupStream of every member; `readStream(buffs, numElems)` returns (streams delivered, [ret per member])."""

    def __init__(self, devices, args=None):
        self.devices = list(devices)                # kept alive with the group
        K, V, n = _kwargs(args)
        arr = (C.c_void_p * len(self.devices))(*[d.h for d in self.devices])
        self.h = lib().cl_group_make(arr, len(self.devices), K, V, n)
        if not self.h:
            raise RuntimeError(lib().cl_group_last_error(None).decode())
        self._rets = (C.c_int * len(self.devices))()
        self._ptrs = (C.c_void_p * len(self.devices))()
        import weakref
        for d in self.devices:
            d._group = weakref.ref(self)

    def readStream(self, buffs, numElems, timeoutUs=100000):
        last = getattr(self, "_last_buffs", ())
        if len(last) != len(buffs) or any(x is not y for x, y in zip(last, buffs)):      # (taking 32 addresses costs more than 50 us)
            for i, b in enumerate(buffs):
                self._ptrs[i] = b.ctypes.data
            self._last_buffs = tuple(buffs)         # (held: the identity test above stands for the addresses)
        n = lib().cl_group_readStream(self.h, self._ptrs, numElems, self._rets, timeoutUs)
        return n, list(self._rets)

    def writeStream(self, buffs, numElems, timeoutUs=100000):
        """a group of TX devices: (members that consumed elements, [ret per member])"""
        last = getattr(self, "_last_buffs", ())
        if len(last) != len(buffs) or any(x is not y for x, y in zip(last, buffs)):
            for i, b in enumerate(buffs):
                self._ptrs[i] = b.ctypes.data if b is not None else None      # (None: the member is left out of the call)
            self._last_buffs = tuple(buffs)
        n = lib().cl_group_writeStream(self.h, self._ptrs, numElems, self._rets, timeoutUs)
        return n, list(self._rets)

    def registerBuffers(self, buffs):
        arr = (C.c_void_p * len(buffs))(*[b.ctypes.data for b in buffs])
        if lib().cl_group_register_buffers(self.h, arr, buffs[0].nbytes) != 0:
            raise RuntimeError(self.lastError())
        self._registered = list(buffs)              # the client keeps them allocated while registered

    def unregisterBuffers(self):
        lib().cl_group_unregister_buffers(self.h)
        self._registered = None

    def lastError(self):
        return lib().cl_group_last_error(self.h).decode()

    def flush(self):
        return lib().cl_group_flush(self.h)

    def setIirPollBound(self, polls):
        lib().cl_group_set_iir_poll_bound(self.h, polls)

    def setTxPollBound(self, polls):
        lib().cl_group_set_tx_poll_bound(self.h, polls)

    def stats(self):
        out = (C.c_uint64 * 11)()
        lib().cl_group_getStats(self.h, out)
        return dict(zip(("calls", "batched_reads", "single_reads", "direct_reads", "launches", "errors", "copies_2d", "last_queue_us",
                         "last_arrive_us", "last_total_us", "ahead_reads"), [int(v) for v in out]))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.cl_group_unmake(self.h)
            for d in getattr(self, "devices", ()):
                d._group = None
        self.h = None

    __del__ = close


class Node:
    """cl_node: devices of SEVERAL GPUs (their `gpu` kwarg) read or written in one call -- one cl_group per GPU, every group's call at
    once on a thread of its own (include/cariboulite_hip.h, "ONE call over the stream groups of SEVERAL GPUs").  args: SHARDS=<k> groups
    per GPU (rehearsal on one GPU), the rest are the groups' kwargs."""

    def __init__(self, devices, args=None):
        self.devices = list(devices)
        K, V, n = _kwargs(args)
        arr = (C.c_void_p * len(self.devices))(*[d.h for d in self.devices])
        self.h = lib().cl_node_make(arr, len(self.devices), K, V, n)
        if not self.h:
            raise RuntimeError(lib().cl_node_last_error(None).decode())
        self._rets = (C.c_int * len(self.devices))()
        self._ptrs = (C.c_void_p * len(self.devices))()
        import weakref
        for d in self.devices:
            d._group = weakref.ref(self)

    def _set(self, buffs):
        last = getattr(self, "_last_buffs", ())
        if len(last) != len(buffs) or any(x is not y for x, y in zip(last, buffs)):
            for i, b in enumerate(buffs):
                self._ptrs[i] = b.ctypes.data if b is not None else None
            self._last_buffs = tuple(buffs)

    def readStream(self, buffs, numElems, timeoutUs=100000):
        self._set(buffs)
        n = lib().cl_node_readStream(self.h, self._ptrs, numElems, self._rets, timeoutUs)
        return n, list(self._rets)

    def writeStream(self, buffs, numElems, timeoutUs=100000):
        self._set(buffs)
        n = lib().cl_node_writeStream(self.h, self._ptrs, numElems, self._rets, timeoutUs)
        return n, list(self._rets)

    def flush(self):
        return lib().cl_node_flush(self.h)

    def registerBuffers(self, buffs):
        arr = (C.c_void_p * len(buffs))(*[b.ctypes.data for b in buffs])
        if lib().cl_node_register_buffers(self.h, arr, buffs[0].nbytes) != 0:
            raise RuntimeError(self.lastError())
        self._registered = list(buffs)              # the client keeps them allocated while registered

    def unregisterBuffers(self):
        lib().cl_node_unregister_buffers(self.h)
        self._registered = None

    def shards(self):
        return int(lib().cl_node_shards(self.h))

    def shardOf(self, member):
        return int(lib().cl_node_shard_of(self.h, member))

    def lastError(self):
        return lib().cl_node_last_error(self.h).decode()

    def stats(self):
        """the groups' statistics, summed (the three last_*_us fields: the slowest group's)"""
        names = ("calls", "batched_reads", "single_reads", "direct_reads", "launches", "errors", "copies_2d", "last_queue_us",
                 "last_arrive_us", "last_total_us", "ahead_reads")
        tot = dict.fromkeys(names, 0)
        for s in range(self.shards()):
            out = (C.c_uint64 * 11)()
            lib().cl_group_getStats(lib().cl_node_group(self.h, s), out)
            for k, v in zip(names, out):
                tot[k] = max(tot[k], int(v)) if k.startswith("last_") else tot[k] + int(v)
        return tot

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.cl_node_unmake(self.h)
            for d in getattr(self, "devices", ()):
                d._group = None
        self.h = None

    __del__ = close
