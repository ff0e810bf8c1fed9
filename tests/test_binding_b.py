"""Binding B is a committed C file.  Where the reference tree is present it is compiled
against the reference's OWN headers -- the three replaced signatures (cariboulite_radio.h:592-619), the struct members
it touches and the sample-type layouts are machine-checked -- into oracle/_ref/libbinding_b.so (oracle/Makefile `ref`),
which travels to the GPU box.  There it is DRIVEN: cariboulite_radio_read_samples on a non-blocking pipe that delivers a
short first read and then makes the call wait in poll(), against the compiled reference's caribou_smi_read on a pipe
fed the same way."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference/software/libcariboulite/src"
STUB = os.path.join(ROOT, "tests", "binding_b", "cariboulite_radio_hip.c")


def test_stub_is_committed_and_matches_integration_md():
    src = open(STUB).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "tests/binding_b/cariboulite_radio_hip.c" in doc
    for fn in ("cariboulite_radio_read_samples", "cariboulite_radio_write_samples", "cariboulite_radio_get_native_mtu_size_samples"):
        assert fn in src and fn in doc


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_stub_compiles_against_the_reference_header():
    cmd = ["gcc", "-fsyntax-only", "-std=gnu11", "-Wall", "-Werror", "-I", REF, "-I", os.path.join(REF, "caribou_smi"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "binding_b", "bb_harness.c"), STUB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # a wrong signature must be caught: flip one and expect a conflict with the reference's declaration
    bad = open(STUB).read().replace("size_t cariboulite_radio_get_native_mtu_size_samples(cariboulite_radio_state_st *radio)",
                                    "int cariboulite_radio_get_native_mtu_size_samples(cariboulite_radio_state_st *radio)")
    r = subprocess.run(cmd[:-1] + ["-x", "c", "-"], input=bad, capture_output=True, text=True)
    assert r.returncode != 0 and "conflicting types" in r.stderr


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_binding_is_built_for_real_against_the_reference_headers():
    """not -fsyntax-only: a shared object with the three reference symbols, linked against the host layer"""
    from oracle import oracle as orc
    orc.build(ref=True)
    lib = os.path.join(ROOT, "oracle", "_ref", "libbinding_b.so")
    assert os.path.exists(lib)
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    for fn in ("cariboulite_radio_read_samples", "cariboulite_radio_write_samples", "cariboulite_radio_get_native_mtu_size_samples"):
        assert f" T {fn}" in out
    und = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    for fn in ("cl_smi_feed_reserve", "cl_smi_feed_commit", "cl_smi_read", "cl_radio_write_samples", "poll", "ioctl", "pthread_once"):
        assert fn in und, fn
    src = open(STUB).read()
    assert "POLLIN" in src and "POLLOUT" in src and "SMI_STREAM_IOC_SET_STREAM_STATUS" in src and "pthread_once" in src


def _pipe_feeder(wfd, pieces, errs):
    """write piece k once the reader has emptied the pipe of piece k-1 (FIONREAD == 0): every read() of the consumer sees
    exactly one piece, and between pieces it finds the pipe empty and goes through poll()"""
    import array, fcntl, termios, time
    try:
        for pc in pieces:
            deadline = time.time() + 20
            while time.time() < deadline:
                buf = array.array("i", [0])
                fcntl.ioctl(wfd, termios.FIONREAD, buf)
                if buf[0] == 0:
                    break
                time.sleep(0.0005)
            time.sleep(0.0003)                   # the consumer has had time to run into the empty pipe (the tail's poll() waits 2 ms: caribou_smi.c:624-629)
            os.write(wfd, pc.tobytes())
    except Exception as e:                       # pragma: no cover
        errs.append(e)


@pytest.mark.gpu
def test_read_samples_on_a_pipe_with_a_short_first_read_equals_the_reference(orc):
    import ctypes as C
    import fcntl
    import threading
    import numpy as np
    from cariboulite_amd import synth
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libbinding_b.so")
    if not (os.path.exists(lib_path) and orc.have_ref()):
        pytest.skip("oracle/_ref was not built (no reference tree in the build container)")
    bb, ref = C.CDLL(lib_path), orc.ref()
    bb.bb_open.restype = C.c_void_p
    bb.bb_open.argtypes = [C.c_int, C.c_int, C.c_size_t]
    bb.bb_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    bb.bb_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    bb.bb_mtu.restype = C.c_size_t
    bb.bb_mtu.argtypes = [C.c_void_p]
    bb.bb_close.argtypes = [C.c_void_p]
    ref.ref_smi_read_fd.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    NB, MTU = 524288, 131072
    n = 2 * MTU + 1024 + 777                     # a short first read, two native batches, a ragged tail
    for ch in (0, 1):
        b, _, _ = synth.smi_stream_bytes(n, ch, stream=40 + ch)
        b = b.copy()
        b[4 * (1024 + MTU): 4 * (1024 + MTU) + 6] = 0            # the second full batch lost 6 bytes: re-sync + extrapolated sample
        pieces = [b[:4096], b[4096: 4096 + NB], b[4096 + NB: 4096 + 2 * NB], b[4096 + 2 * NB:]]
        res = {}
        for who in ("ref", "hip"):
            rfd, wfd = os.pipe()
            fcntl.fcntl(wfd, 1031, 1 << 20)                      # F_SETPIPE_SZ: a whole native batch fits
            fcntl.fcntl(rfd, fcntl.F_SETFL, fcntl.fcntl(rfd, fcntl.F_GETFL) | os.O_NONBLOCK)
            errs = []
            th = threading.Thread(target=_pipe_feeder, args=(wfd, pieces, errs))
            iq = np.full((n + 8, 2), -21846, np.int16)
            meta = np.full(n + 8, 0xAA, np.uint8)
            th.start()
            if who == "ref":
                ret = ref.ref_smi_read_fd(rfd, ch, iq.ctypes.data, meta.ctypes.data, n, NB)
            else:
                h = bb.bb_open(rfd, ch, NB)
                assert bb.bb_mtu(h) == MTU
                ret = bb.bb_read(h, iq.ctypes.data, meta.ctypes.data, n)
                # the TX side: a pipe is not the SMI driver, the stream-state ioctl fails and the call says so, like
                # caribou_smi_write does (caribou_smi.c:727-735)
                assert bb.bb_write(h, iq.ctypes.data, 100) == -1
                bb.bb_close(h)
            th.join(timeout=30)
            os.close(rfd); os.close(wfd)
            assert not errs
            res[who] = (ret, iq, meta)
        # (the call's last read() asks for 3108 bytes and polls for 2 ms: if the feeder is late BOTH time out there, with the
        # same count -- "Reading timed-out", caribou_smi.c:657-661)
        assert res["ref"][0] == res["hip"][0] and res["hip"][0] in (n, n - 777), (res["ref"][0], res["hip"][0])
        assert np.array_equal(res["ref"][1], res["hip"][1])      # every slot, the untouched ones included
        assert np.array_equal(res["ref"][2], res["hip"][2])
