"""ctypes face of oracle/liboracle.so (cl_oracle.c) and, when present, of
oracle/_ref/libref_smi.so (the compiled reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the cariboulite_amd product package.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libref_smi.so")

CH_S1G, CH_HIF = 0, 1
TX_DOCUMENTED, TX_AS_WRITTEN = 0, 1
NATIVE_BATCH_LEN = 524288  # caribou_smi.c:78  (1024*1024/2 bytes)


def build(ref=True):
    """Compile the restatement (and oracle/_ref when /root/reference exists)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class _IIR(C.Structure):
    _fields_ = [("n_stages", C.c_int)] + [
        (k, C.c_double * 8) for k in ("b0", "b1", "b2", "a1", "a2", "v1", "v2")]


class _Src(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("pos", C.c_size_t),
                ("max_read", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_iir_step.restype = C.c_double
        _lib.orc_resamp_f32.restype = C.c_size_t
        _lib.orc_rx_pipe_f32.restype = C.c_size_t
        _lib.orc_fpga_tx_parse.restype = C.c_size_t
        _lib.orc_sync_tags.restype = C.c_size_t
    return _lib


def have_ref():
    return os.path.exists(REF_PATH)


_ref = None


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_PATH)
    return _ref


# ---------------------------------------------------------------- RX integer
def find_buffer_offset(buf):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    return lib().orc_find_buffer_offset(_p(buf, C.c_uint8), C.c_size_t(buf.size))


def rx_data_analyze(channel, buf, want_iq=True, want_meta=True, fill=-21846):
    """Returns (offs, iq[n,2] int16, meta[n] uint8); slots the reference leaves
    untouched keep the sentinel `fill` (0xAAAA) / 0xAA."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    n = buf.size // 4 + 2
    iq = np.full((n, 2), fill, dtype=np.int16) if want_iq else None
    meta = np.full(n, 0xAA, dtype=np.uint8) if want_meta else None
    offs = lib().orc_rx_data_analyze(int(channel), _p(buf, C.c_uint8), C.c_size_t(buf.size),
                                     _p(iq, C.c_int16), _p(meta, C.c_uint8))
    return offs, iq, meta


def smi_read(channel, stream_bytes, length_samples, native_batch_len=NATIVE_BATCH_LEN,
             max_read=0, fill=-21846, want_meta=True):
    b = np.ascontiguousarray(stream_bytes, dtype=np.uint8)
    src = _Src(b.ctypes.data, b.size, 0, max_read)
    iq = np.full((length_samples + 2, 2), fill, dtype=np.int16)
    meta = np.full(length_samples + 2, 0xAA, dtype=np.uint8) if want_meta else None
    ret = lib().orc_smi_read(C.byref(src), int(channel), _p(iq, C.c_int16), _p(meta, C.c_uint8),
                             C.c_size_t(length_samples), C.c_size_t(native_batch_len))
    return ret, iq, meta


def smi_read_pos(channel, stream_bytes, length_samples, native_batch_len=NATIVE_BATCH_LEN, max_read=0, fill=-21846):
    """smi_read, and how many bytes of the source the chunk loop consumed (caribou_smi.c:643-679: every read() it issued, the one whose
    analysis failed included) -- what a model of a sequence of calls over one byte FIFO needs"""
    b = np.ascontiguousarray(stream_bytes, dtype=np.uint8)
    src = _Src(b.ctypes.data if b.size else None, b.size, 0, max_read)
    iq = np.full((length_samples + 2, 2), fill, dtype=np.int16)
    ret = lib().orc_smi_read(C.byref(src), int(channel), _p(iq, C.c_int16), None, C.c_size_t(length_samples), C.c_size_t(native_batch_len))
    return ret, iq, int(src.pos)


# ---------------------------------------------------------------- TX integer
def generate_data(iq, mode=TX_DOCUMENTED):
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
    out = np.empty(iq.shape[0] * 4, dtype=np.uint8)
    lib().orc_generate_data(int(mode), _p(iq, C.c_int16), C.c_size_t(iq.shape[0]), _p(out, C.c_uint8))
    return out


def fpga_tx_parse(b):
    b = np.ascontiguousarray(b, dtype=np.uint8)
    w = np.empty(b.size, dtype=np.uint32)
    n = lib().orc_fpga_tx_parse(_p(b, C.c_uint8), C.c_size_t(b.size), _p(w, C.c_uint32))
    return w[:n].copy()


# ------------------------------------------------------------------ pps tags
def sync_tags(meta, cap=None):
    """(positions the GNU Radio source's work() tags, how many there are): caribouLiteSource_impl.cc:113-119."""
    meta = np.ascontiguousarray(meta, dtype=np.uint8)
    cap = meta.size if cap is None else cap
    idx = np.empty(max(cap, 1), dtype=np.uint32)
    k = lib().orc_sync_tags(_p(meta, C.c_uint8), C.c_size_t(meta.size), _p(idx, C.c_uint32), C.c_size_t(cap))
    return idx[:min(k, cap)].copy(), int(k)


# --------------------------------------------------------------- conversions
def _conv(name, a, in_t, out_dt, out_t):
    a = np.ascontiguousarray(a).reshape(-1, 2)
    out = np.empty(a.shape, dtype=out_dt)
    getattr(lib(), name)(_p(a, in_t), _p(out, out_t), C.c_size_t(a.shape[0]))
    return out


def cs16_to_cf32(a): return _conv("orc_cs16_to_cf32", np.asarray(a, np.int16), C.c_int16, np.float32, C.c_float)
def cs16_to_cf64(a): return _conv("orc_cs16_to_cf64", np.asarray(a, np.int16), C.c_int16, np.float64, C.c_double)
def cs16_to_cs8(a): return _conv("orc_cs16_to_cs8", np.asarray(a, np.int16), C.c_int16, np.int8, C.c_int8)
def cf32_to_cs16(a): return _conv("orc_cf32_to_cs16", np.asarray(a, np.float32), C.c_float, np.int16, C.c_int16)
def cf64_to_cs16(a): return _conv("orc_cf64_to_cs16", np.asarray(a, np.float64), C.c_double, np.int16, C.c_int16)
def cs8_to_cs16(a): return _conv("orc_cs8_to_cs16", np.asarray(a, np.int8), C.c_int8, np.int16, C.c_int16)


# ----------------------------------------------------------------------- IIR
class IIR:
    """iir1-style Butterworth LowPass<order> pair (I and Q rails)."""

    def __init__(self, order, fs, fc):
        self.fi, self.fq = _IIR(), _IIR()
        for f in (self.fi, self.fq):
            lib().orc_iir_butter_lowpass(C.byref(f), int(order), C.c_double(fs), C.c_double(fc))

    def sos(self):
        f = self.fi
        return np.array([[f.b0[s], f.b1[s], f.b2[s], 1.0, f.a1[s], f.a2[s]] for s in range(f.n_stages)])

    def apply_cs16(self, iq):
        iq = np.array(iq, dtype=np.int16).reshape(-1, 2)
        lib().orc_iir_apply_cs16(C.byref(self.fi), C.byref(self.fq), _p(iq, C.c_int16), C.c_size_t(iq.shape[0]))
        return iq

    def step_f64(self, x):
        """Unrounded fp64 output of the I rail for real input x (no int cast)."""
        return np.array([lib().orc_iir_step(C.byref(self.fi), C.c_double(v)) for v in x])


# -------------------------------------------------------------- float stages
class _Fir(C.Structure):
    _fields_ = [("n_taps", C.c_int), ("taps", C.POINTER(C.c_float)), ("hist", C.POINTER(C.c_double))]


class FIR:
    def __init__(self, taps):
        self.taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.hist = np.zeros(2 * max(self.taps.size - 1, 1), dtype=np.float64)
        self.hist32 = np.zeros(2 * max(self.taps.size - 1, 1), dtype=np.float32)
        self.s = _Fir(self.taps.size, _p(self.taps, C.c_float), _p(self.hist, C.c_double))

    def f64(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2)
        y = np.empty(x.shape, dtype=np.float64)
        lib().orc_fir_f64(C.byref(self.s), _p(x, C.c_float), C.c_size_t(x.shape[0]), _p(y, C.c_double))
        return y

    def f32(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2)
        y = np.empty(x.shape, dtype=np.float32)
        lib().orc_fir_f32(_p(self.taps, C.c_float), self.taps.size, _p(self.hist32, C.c_float),
                          _p(x, C.c_float), C.c_size_t(x.shape[0]), _p(y, C.c_float))
        return y


class _Rs(C.Structure):
    _fields_ = [("L", C.c_int), ("M", C.c_int), ("n_taps", C.c_int), ("taps", C.POINTER(C.c_float)),
                ("hist", C.POINTER(C.c_double)), ("hist_len", C.c_int), ("n_in", C.c_uint64)]


class Resampler:
    def __init__(self, taps, L, M):
        self.taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.L, self.M = int(L), int(M)
        K = (self.taps.size + L - 1) // L
        self.hist = np.zeros(2 * max(K - 1, 1), dtype=np.float64)
        self.hist32 = np.zeros(2 * max(K - 1, 1), dtype=np.float32)
        self.n_in32 = C.c_uint64(0)
        self.s = _Rs(L, M, self.taps.size, _p(self.taps, C.c_float), _p(self.hist, C.c_double), K - 1, 0)

    def f64(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        y = np.empty((x.shape[0] * self.L // self.M + 2, 2), dtype=np.float64)
        lib().orc_resamp_f64.restype = C.c_size_t
        n = lib().orc_resamp_f64(C.byref(self.s), _p(x, C.c_double), C.c_size_t(x.shape[0]), _p(y, C.c_double))
        return y[:n].copy()

    def f32(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2)
        y = np.empty((x.shape[0] * self.L // self.M + 2, 2), dtype=np.float32)
        n = lib().orc_resamp_f32(_p(self.taps, C.c_float), self.taps.size, self.L, self.M,
                                 _p(self.hist32, C.c_float), C.byref(self.n_in32),
                                 _p(x, C.c_float), C.c_size_t(x.shape[0]), _p(y, C.c_float))
        return y[:n].copy()


def fm_demod_f64(x, prev=None):
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
    prev = np.zeros(2) if prev is None else np.asarray(prev, dtype=np.float64)
    y = np.empty(x.shape[0], dtype=np.float64)
    lib().orc_fm_demod_f64(_p(prev, C.c_double), _p(x, C.c_double), C.c_size_t(x.shape[0]), _p(y, C.c_double))
    return y, prev


def fm_mod_f64(m, kf, fs, phase=0.0):
    m = np.ascontiguousarray(m, dtype=np.float32)
    ph = C.c_double(phase)
    out = np.empty((m.size, 2), dtype=np.float32)
    lib().orc_fm_mod_f64(C.byref(ph), C.c_double(kf), C.c_double(fs), _p(m, C.c_float), C.c_size_t(m.size), _p(out, C.c_float))
    return out, ph.value


def cw_tone(f, fs, n, phase=0.0):
    ph = C.c_double(phase)
    out = np.empty((n, 2), dtype=np.float32)
    lib().orc_cw_tone(C.byref(ph), C.c_double(f), C.c_double(fs), C.c_size_t(n), _p(out, C.c_float))
    return out, ph.value


class RxPipeF32:
    """The fp32 CPU pipe timed by bench.py as the "port" cpu_baseline."""

    def __init__(self, channel, fir_taps, rs_taps, L, M, max_samples):
        self.ch = int(channel)
        self.fir = np.ascontiguousarray(fir_taps, np.float32)
        self.rs = np.ascontiguousarray(rs_taps, np.float32)
        self.L, self.M = int(L), int(M)
        self.fir_hist = np.zeros(2 * self.fir.size, np.float32)
        self.rs_hist = np.zeros(2 * self.rs.size, np.float32)
        self.n_in = C.c_uint64(0)
        self.tmp_iq = np.empty((max_samples + 2, 2), np.int16)
        self.tmp_cf = np.empty((max_samples + 2, 2), np.float32)
        self.tmp_fir = np.empty((max_samples + 2, 2), np.float32)
        self.out = np.empty((max_samples * self.L // self.M + 2, 2), np.float32)

    def run(self, b):
        b = np.ascontiguousarray(b, np.uint8)
        n = lib().orc_rx_pipe_f32(self.ch, _p(b, C.c_uint8), C.c_size_t(b.size),
                                  _p(self.fir, C.c_float), self.fir.size, _p(self.fir_hist, C.c_float),
                                  _p(self.rs, C.c_float), self.rs.size, self.L, self.M,
                                  _p(self.rs_hist, C.c_float), C.byref(self.n_in),
                                  _p(self.tmp_iq, C.c_int16), _p(self.tmp_cf, C.c_float),
                                  _p(self.tmp_fir, C.c_float), _p(self.out, C.c_float))
        return self.out[:n]


def rx_pipe_f32_mt(channel, b, fir_taps, rs_taps, L, M, n_threads, native_batch_len=NATIVE_BATCH_LEN, bufs=None):
    """All-cores CPU pipe over one long stream from zero state.  Returns (out, bufs)."""
    b = np.ascontiguousarray(b, np.uint8)
    fir = np.ascontiguousarray(fir_taps, np.float32); rs = np.ascontiguousarray(rs_taps, np.float32)
    n = b.size // 4
    if bufs is None:
        bufs = (np.empty((n + 2, 2), np.int16), np.empty((n + fir.size + 8, 2), np.float32),
                np.empty((n + 16, 2), np.float32), np.empty((n * L // M + 2, 2), np.float32))
    iq, x, y, out = bufs
    lib().orc_rx_pipe_f32_mt.restype = C.c_size_t
    no = lib().orc_rx_pipe_f32_mt(int(channel), _p(b, C.c_uint8), C.c_size_t(b.size), C.c_size_t(native_batch_len),
                                  _p(fir, C.c_float), fir.size, _p(rs, C.c_float), rs.size, int(L), int(M),
                                  _p(iq, C.c_int16), _p(x, C.c_float), _p(y, C.c_float), _p(out, C.c_float),
                                  int(n_threads))
    return out[:no], bufs


# ------------------------------------------------------- link-integrity modes
DEBUG_NONE, DEBUG_LFSR, DEBUG_PUSH, DEBUG_PULL = 0, 1, 2, 3


class _Dbg(C.Structure):
    _fields_ = [("error_accum_counter", C.c_uint32), ("cur_err_cnt", C.c_uint32),
                ("last_correct_byte", C.c_uint8), ("error_rate", C.c_double)]


class DebugState:
    """caribou_smi_debug_data_st without the wall-clock fields."""

    def __init__(self):
        self.s = _Dbg(0, 0, 0, 0.0)

    def analyze(self, mode, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        return lib().orc_debug_analyze(C.byref(self.s), int(mode), _p(buf, C.c_uint8), C.c_size_t(buf.size))

    def tuple(self):
        return (self.s.error_accum_counter, self.s.cur_err_cnt, self.s.last_correct_byte, self.s.error_rate)


def lfsr_stream(n, seed=0x56):
    out = np.empty(n, dtype=np.uint8)
    v = seed
    lib().orc_lfsr.restype = C.c_uint8
    for i in range(n):
        v = lib().orc_lfsr(C.c_uint8(v))
        out[i] = v
    return out


def bitrate_ema(nbytes, old, cur, old_mbps):
    """smi_calculate_performance with the two clock readings (sec, usec) as arguments."""
    lib().orc_bitrate_ema.restype = C.c_double
    return lib().orc_bitrate_ema(C.c_size_t(nbytes), C.c_long(old[0]), C.c_long(old[1]), C.c_long(cur[0]), C.c_long(cur[1]), C.c_double(old_mbps))


def ref_bitrate_ema(nbytes, old, old_mbps):
    """The compiled reference's smi_calculate_performance: (result, the (sec, usec) wall-clock reading it used)."""
    ref().ref_bitrate_ema.restype = C.c_double
    cs, cu = C.c_long(0), C.c_long(0)
    r = ref().ref_bitrate_ema(C.c_size_t(nbytes), C.c_long(old[0]), C.c_long(old[1]), C.c_double(old_mbps), C.byref(cs), C.byref(cu))
    return r, (cs.value, cu.value)


def ref_debug_read(mode, stream_bytes, length_samples, state, native_batch_len=NATIVE_BATCH_LEN):
    """state = (accum, cur, last_byte, error_rate) -> (ret, new state) through the compiled reference."""
    b = np.ascontiguousarray(stream_bytes, dtype=np.uint8)
    a, c, l, r = C.c_uint32(state[0]), C.c_uint32(state[1]), C.c_uint8(state[2]), C.c_double(state[3])
    with tempfile.NamedTemporaryFile(suffix=".smi") as f:
        f.write(b.tobytes()); f.flush()
        ret = ref().ref_debug_read_file(f.name.encode(), int(mode), C.c_size_t(length_samples),
                                        C.c_size_t(native_batch_len), C.byref(a), C.byref(c), C.byref(l), C.byref(r))
    return ret, (a.value, c.value, l.value, r.value)


# ------------------------------------------------------------ circular buffer
class _Ring(C.Structure):
    _fields_ = [("buf", C.POINTER(C.c_uint32)), ("max_size", C.c_size_t), ("head", C.c_size_t),
                ("tail", C.c_size_t), ("override_write", C.c_int)]


class Ring:
    """circular_buffer<uint32_t> restated (no waiting: block_read only decides the short-read rule)."""

    def __init__(self, size, override_write=True, block_read=True):
        lib().orc_ring_capacity_for.restype = C.c_size_t
        cap = lib().orc_ring_capacity_for(C.c_size_t(size))
        self.store = np.zeros(cap, dtype=np.uint32)
        self.r = _Ring()
        self.block = int(block_read)
        lib().orc_ring_init(C.byref(self.r), C.c_size_t(size), int(override_write), _p(self.store, C.c_uint32))
        for f in ("orc_ring_put", "orc_ring_get", "orc_ring_size"):
            getattr(lib(), f).restype = C.c_size_t

    def put(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint32)
        return lib().orc_ring_put(C.byref(self.r), _p(d, C.c_uint32), C.c_size_t(d.size))

    def get(self, n):
        out = np.zeros(max(n, 1), dtype=np.uint32)
        k = lib().orc_ring_get(C.byref(self.r), _p(out, C.c_uint32), C.c_size_t(n), self.block)
        return k, out[:k].copy()

    def size(self):
        return lib().orc_ring_size(C.byref(self.r))

    def capacity(self):
        return self.r.max_size


REF_RING_PATH = os.path.join(HERE, "_ref", "libref_ring.so")
_ref_ring = None


def ref_ring_lib():
    global _ref_ring
    if _ref_ring is None:
        _ref_ring = C.CDLL(REF_RING_PATH)
        _ref_ring.ref_ring_new.restype = C.c_void_p
        for f in ("ref_ring_put", "ref_ring_get", "ref_ring_size", "ref_ring_capacity"):
            getattr(_ref_ring, f).restype = C.c_size_t
    return _ref_ring


class RefRing:
    """The reference's own circular_buffer<uint32_t>, compiled from /root/reference."""

    def __init__(self, size, override_write=True, block_read=True):
        self.h = C.c_void_p(ref_ring_lib().ref_ring_new(C.c_size_t(size), int(override_write), int(block_read)))

    def put(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint32)
        return ref_ring_lib().ref_ring_put(self.h, _p(d, C.c_uint32), C.c_size_t(d.size))

    def get(self, n, timeout_us=1000):
        out = np.zeros(max(n, 1), dtype=np.uint32)
        k = ref_ring_lib().ref_ring_get(self.h, _p(out, C.c_uint32), C.c_size_t(n), int(timeout_us))
        return k, out[:k].copy()

    def size(self):
        return ref_ring_lib().ref_ring_size(self.h)

    def capacity(self):
        return ref_ring_lib().ref_ring_capacity(self.h)

    def __del__(self):
        if self.h:
            ref_ring_lib().ref_ring_free(self.h); self.h = None


# ------------------------------------------------- compiled reference (_ref)
def ref_find_buffer_offset(buf):
    buf = np.ascontiguousarray(buf, dtype=np.uint8).copy()
    return ref().ref_find_buffer_offset(_p(buf, C.c_uint8), C.c_size_t(buf.size))


def ref_rx_data_analyze(channel, buf, want_meta=True, fill=-21846):
    buf = np.ascontiguousarray(buf, dtype=np.uint8).copy()
    n = buf.size // 4 + 2
    iq = np.full((n, 2), fill, dtype=np.int16)
    meta = np.full(n, 0xAA, dtype=np.uint8) if want_meta else None
    offs = ref().ref_rx_data_analyze(int(channel), _p(buf, C.c_uint8), C.c_size_t(buf.size),
                                     _p(iq, C.c_int16), _p(meta, C.c_uint8))
    return offs, iq, meta


def ref_smi_read(channel, stream_bytes, length_samples, native_batch_len=NATIVE_BATCH_LEN,
                 fill=-21846):
    b = np.ascontiguousarray(stream_bytes, dtype=np.uint8)
    iq = np.full((length_samples + 2, 2), fill, dtype=np.int16)
    meta = np.full(length_samples + 2, 0xAA, dtype=np.uint8)
    with tempfile.NamedTemporaryFile(suffix=".smi") as f:
        f.write(b.tobytes()); f.flush()
        # the reference prints a hexdump on sync failure: keep stdout clean
        ret = ref().ref_smi_read_file(f.name.encode(), int(channel), _p(iq, C.c_int16),
                                      _p(meta, C.c_uint8), C.c_size_t(length_samples),
                                      C.c_size_t(native_batch_len))
    return ret, iq, meta


def ref_generate_data(iq):
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
    out = np.empty(iq.shape[0] * 4, dtype=np.uint8)
    ref().ref_generate_data(_p(iq, C.c_int16), C.c_size_t(iq.shape[0]), _p(out, C.c_uint8))
    return out
