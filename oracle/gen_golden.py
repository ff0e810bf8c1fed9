#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  Run in the build container only:

    python oracle/gen_golden.py

TEST INFRASTRUCTURE.  Two kinds of fixture:
  * smi_*.npz  -- outputs of the REFERENCE's own caribou_smi.c, compiled from
    /root/reference by `make -C oracle ref` (oracle/_ref/libref_smi.so), on the
    byte buffers stored beside them.  These pin the integer stages.
  * dsp_*.npz / taps.npz -- float64 scipy restatements of the stages the
    reference does not implement (FIR, upfirdn, FM, Butterworth): inputs, taps
    and expected outputs.  These pin the "parity unpinned" float stages.
Fixtures are data only (inputs + expected outputs).
"""
import os
import sys

import numpy as np
import scipy.signal as sg

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import oracle as orc          # noqa: E402
from cariboulite_amd import synth         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FILL16, FILL8 = -21846, 0xAA


def words_to_bytes(w):
    return np.asarray(w, dtype=np.uint32).view(np.uint8)


def gen_kat():
    """The eight known-answer words of SURVEY.md section 8c, re-captured from the reference."""
    words = np.array([0x80004000, 0x80027FFF, 0xBFFE4002, 0x9FFE6001,
                      0xA0005FFE, 0x89A45E3F, 0xB65C61C2, 0x80C87F39], dtype=np.uint32)
    res = {"words": words}
    for ch, name in ((0, "s1g"), (1, "hif")):
        offs, iq, meta = orc.ref_rx_data_analyze(ch, words_to_bytes(words))
        assert offs == 0
        res[f"iq_{name}"] = iq[:8]
        res[f"sync_{name}"] = meta[:8]
    np.savez_compressed(os.path.join(OUT, "smi_rx_kat.npz"), **res)


def rx_cases():
    rng = np.random.default_rng(0xC0FFEE)
    cases = []

    def rand_words(n, ch=0):
        i = rng.integers(-4096, 4096, n); q = rng.integers(-4096, 4096, n)
        s = rng.integers(0, 2, n)
        return synth.iq_to_words(i, q, ch, s)

    # aligned
    cases.append(("aligned_1024", words_to_bytes(rand_words(1024))))
    cases.append(("aligned_synth_4096", synth.smi_stream_bytes(4096)[0]))
    # byte offsets: garbage prefix of `o` bytes that cannot look like sync
    for o in (1, 2, 3, 4, 5, 7, 8, 13, 4097):
        n = 2048 if o < 100 else 3072
        body = words_to_bytes(rand_words(n))
        prefix = np.zeros(o, dtype=np.uint8)  # 0x00 words fail the sync mask
        cases.append((f"offset_{o}", np.concatenate([prefix, body])[: 4 * n]))
    # prefix made of random bytes (may contain accidental partial matches)
    body = words_to_bytes(rand_words(1024))
    cases.append(("offset_rand_prefix_37", np.concatenate([rng.integers(0, 256, 37).astype(np.uint8), body])))
    # a false 3-word match before the real alignment
    w = rand_words(600)
    b = words_to_bytes(w).copy()
    b2 = np.concatenate([b[:12], np.array([0xFF, 0xFF, 0xFF, 0xFF, 0x11], dtype=np.uint8), b[12:]])
    cases.append(("false_match_then_offset", b2))
    # short buffers: len <= 16 -> offset 0 without looking (caribou_smi.c:240-243)
    for ln in (0, 3, 4, 8, 12, 15, 16):
        cases.append((f"short_{ln}", rng.integers(0, 256, ln).astype(np.uint8)))
    # just above the threshold, aligned and not
    for ln in (17, 18, 19, 20, 21, 24, 32, 33):
        cases.append((f"edge_{ln}", words_to_bytes(rand_words(9))[:ln].copy()))
    cases.append(("edge_misaligned_24", np.concatenate([np.zeros(2, np.uint8), words_to_bytes(rand_words(8))])[:24]))
    # ragged length (not a multiple of 4), aligned and misaligned
    cases.append(("ragged_4001", words_to_bytes(rand_words(1001))[:4001].copy()))
    cases.append(("ragged_off3_4003", np.concatenate([np.zeros(3, np.uint8), words_to_bytes(rand_words(1001))])[:4003]))
    # no sync anywhere -> -1
    cases.append(("nosync_zeros", np.zeros(2048, np.uint8)))
    cases.append(("nosync_ff", np.full(2048, 0xFF, np.uint8)))
    bad = rand_words(512); bad ^= np.uint32(0x00008000)      # bit15 set -> mask fails everywhere
    cases.append(("nosync_bit15", words_to_bytes(bad)))
    # sync only in the last 16 bytes: the scan stops at len-16 (exclusive) -> not found
    tail = np.concatenate([np.zeros(2048 - 16, np.uint8), words_to_bytes(rand_words(4))])
    cases.append(("sync_only_in_last16", tail))
    tail2 = np.concatenate([np.zeros(2048 - 20, np.uint8), words_to_bytes(rand_words(5))])
    cases.append(("sync_at_len_minus_20", tail2))
    # corrupted single word in the middle of an aligned buffer: still offs 0
    c = rand_words(1024); c[500] = 0
    cases.append(("aligned_one_bad_word", words_to_bytes(c)))
    # first word corrupted: re-sync at +4
    c = rand_words(1024); c[0] = 0x12345678
    cases.append(("first_word_bad", words_to_bytes(c)))
    # extreme values
    ext = synth.iq_to_words([4095, -4096, 0, -1, 1, 4095, -4096, 2047, -2048, 1234],
                            [-4096, 4095, 0, 1, -1, 4095, -4096, -2048, 2047, -3871], 0,
                            [1, 0, 0, 1, 0, 1, 0, 1, 0, 1])
    cases.append(("extremes", words_to_bytes(ext)))
    return cases


def gen_rx():
    res = {}
    names = []
    for name, buf in rx_cases():
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        names.append(name)
        res[f"{name}__bytes"] = buf
        res[f"{name}__find"] = np.int32(orc.ref_find_buffer_offset(buf))
        for ch, cn in ((0, "s1g"), (1, "hif")):
            offs, iq, meta = orc.ref_rx_data_analyze(ch, buf)
            res[f"{name}__offs_{cn}"] = np.int32(offs)
            res[f"{name}__iq_{cn}"] = iq
            res[f"{name}__meta_{cn}"] = meta
    res["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "smi_rx_cases.npz"), **res)
    print("rx cases:", len(names))


def gen_read():
    """caribou_smi_read() through a file fd, with a small native batch so the
    chunk loop (caribou_smi.c:643-679) is exercised by small fixtures."""
    rng = np.random.default_rng(0xBEEF)
    nb = 4096  # bytes per native batch in these cases
    res = {}; names = []

    def words(n, ch=0):
        return synth.iq_to_words(rng.integers(-4096, 4096, n), rng.integers(-4096, 4096, n), ch,
                                 rng.integers(0, 2, n))

    def add(name, stream, length_samples, ch=0, batch=nb):
        stream = np.ascontiguousarray(stream, np.uint8)
        ret, iq, meta = orc.ref_smi_read(ch, stream, length_samples, batch)
        names.append(name)
        res[f"{name}__bytes"] = stream
        res[f"{name}__args"] = np.array([ch, length_samples, batch], dtype=np.int64)
        res[f"{name}__ret"] = np.int32(ret)
        res[f"{name}__iq"] = iq
        res[f"{name}__meta"] = meta

    s5 = words_to_bytes(words(5 * 1024))
    add("five_chunks_aligned", s5, 5 * 1024)
    add("five_chunks_hif", s5, 5 * 1024, ch=1)
    add("partial_last_chunk", s5, 4 * 1024 + 100)
    add("less_than_one_chunk", s5, 300)
    add("eof_timeout", s5[: 3 * 4096 + 2000], 5 * 1024)          # source drains: partial count
    # chunk 2 is misaligned by 2 bytes (drop two bytes at its start)
    mis = np.concatenate([s5[:4096], s5[4096 + 2:]])
    add("chunk2_misaligned_by_2", mis, 5 * 1024 - 1)
    # a chunk with no sync -> -3
    bad = s5.copy(); bad[2 * 4096:3 * 4096] = 0
    add("chunk3_no_sync", bad, 5 * 1024)
    # misaligned by 6 bytes: offs = 6 -> shortening 2, one slot left untouched
    mis6 = np.concatenate([np.zeros(6, np.uint8), s5])
    add("stream_offset_6", mis6, 3 * 1024)
    add("zero_length", s5, 0)
    add("native_batch_default", words_to_bytes(words(3000)), 3000, batch=524288)
    res["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "smi_read_cases.npz"), **res)
    print("read cases:", len(names))


def gen_tx():
    rng = np.random.default_rng(0x7A)
    iq = rng.integers(-4096, 4096, (256, 2)).astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "smi_tx_as_written.npz"), iq=iq,
                        bytes=orc.ref_generate_data(iq))


def gen_debug():
    """Link-integrity modes through the reference's own caribou_smi_read (returns -2) on files:
    chained calls carry last_correct_byte / error EMA exactly as caribou_smi_st does."""
    rng = np.random.default_rng(0xD1A6)
    nb = 8192
    res = {}; names = []

    def add(name, mode, stream, n_calls, length_samples):
        stream = np.ascontiguousarray(stream, np.uint8)
        state = (0, 0, 0, 0.0)
        rets, states = [], []
        pos = 0
        for _ in range(n_calls):
            ret, state = orc.ref_debug_read(mode, stream[pos:], length_samples, state, nb)
            rets.append(ret); states.append(state)
            pos += min(nb, 4 * length_samples)        # the reference consumes one chunk per call
        names.append(name)
        res[f"{name}__bytes"] = stream
        res[f"{name}__args"] = np.array([mode, n_calls, length_samples, nb], dtype=np.int64)
        res[f"{name}__rets"] = np.array(rets, dtype=np.int32)
        res[f"{name}__states"] = np.array([[s[0], s[1], s[2]] for s in states], dtype=np.int64)
        res[f"{name}__rates"] = np.array([s[3] for s in states], dtype=np.float64)

    lf = orc.lfsr_stream(4 * nb)
    add("lfsr_clean", orc.DEBUG_LFSR, lf, 4, 4096)
    bad = lf.copy(); bad[[5, 100, 9000, 9001, 20000]] ^= np.array([1, 0x80, 0xFF, 3, 0x10], np.uint8); bad[12345] = 0
    add("lfsr_errors", orc.DEBUG_LFSR, bad, 4, 4096)
    add("lfsr_short_calls", orc.DEBUG_LFSR, bad, 6, 700)
    add("lfsr_zeros", orc.DEBUG_LFSR, np.zeros(nb, np.uint8), 1, 2048)
    w = np.full(4 * nb // 4, 0xABCDEF01, np.uint32)
    add("push_clean", orc.DEBUG_PUSH, w.view(np.uint8), 3, 2048)
    w2 = w.copy(); w2[[3, 500, 2100, 4000]] ^= np.array([1, 0x80000000, 0xF0F0, 0xFFFFFFFF], np.uint32)
    add("push_errors", orc.DEBUG_PUSH, w2.view(np.uint8), 3, 2048)
    add("pull_misaligned_3", orc.DEBUG_PULL, np.concatenate([np.array([9, 8, 7], np.uint8), w2.view(np.uint8)]), 2, 2048)
    add("push_misaligned_6", orc.DEBUG_PUSH, np.concatenate([rng.integers(0, 256, 6).astype(np.uint8), w2.view(np.uint8)]), 2, 2048)
    add("push_nosync", orc.DEBUG_PUSH, np.zeros(nb, np.uint8), 1, 2048)
    add("push_short_16", orc.DEBUG_PUSH, w.view(np.uint8)[:16], 1, 4)
    res["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "smi_debug_cases.npz"), **res)
    print("debug cases:", len(names))


def gen_ring():
    """Op sequences through the reference's own circular_buffer<uint32_t> (compiled from
    datatypes/circular_buffer.h): per op the returned count, the data popped and the fill level."""
    rng = np.random.default_rng(0xB1F)
    res = {}; names = []
    for name, size, ov, blk in (("ov_block_100", 100, 1, 1), ("noov_block_64", 64, 0, 1), ("ov_noblock_33", 33, 1, 0),
                                ("noov_noblock_256", 256, 0, 0)):
        ring = orc.RefRing(size, ov, blk)
        ops, rets, sizes, popped = [], [], [], []
        ctr = 0
        for _ in range(400):
            if rng.random() < 0.55:
                n = int(rng.integers(0, ring.capacity()))
                d = np.arange(ctr, ctr + n, dtype=np.uint32); ctr += n
                r = ring.put(d); ops.append((0, n)); rets.append(r)
            else:
                n = int(rng.integers(0, ring.capacity()))
                k, d = ring.get(n, 100); ops.append((1, n)); rets.append(k); popped.append(d)
            sizes.append(ring.size())
        names.append(name)
        res[f"{name}__cfg"] = np.array([size, ov, blk, ring.capacity()], dtype=np.int64)
        res[f"{name}__ops"] = np.array(ops, dtype=np.int64)
        res[f"{name}__rets"] = np.array(rets, dtype=np.int64)
        res[f"{name}__sizes"] = np.array(sizes, dtype=np.int64)
        res[f"{name}__popped"] = np.concatenate(popped) if popped else np.zeros(0, np.uint32)
    res["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "ring_cases.npz"), **res)
    print("ring cases:", len(names))


def design_taps():
    t = {}
    t["fir64_c2"] = sg.firwin(64, 1.0e6, window="hamming", fs=4e6)
    t["fir64_c3"] = sg.firwin(64, 100e3, window="hamming", fs=4e6)
    t["fir128_c4"] = sg.firwin(128, 1.2e6, window="hamming", fs=4e6)
    for L, M in ((3, 2), (5, 4), (2, 3), (1, 2), (1, 4), (3, 4)):     # the last three: decimating shapes of the fused pipe
        t[f"rs_{L}_{M}"] = L * sg.firwin(8 * L, 1.0 / max(L, M), window="hamming")
    return {k: v.astype(np.float32) for k, v in t.items()}, t


def gen_dsp():
    taps32, taps64 = design_taps()
    np.savez_compressed(os.path.join(OUT, "taps.npz"), **taps32,
                        **{k + "__f64": v for k, v in taps64.items()})
    n = 8192
    b, i, q = synth.smi_stream_bytes(n, 0, stream=7)
    x = np.stack([i, q], 1).astype(np.float32) / np.float32(4096.0)
    xc = x[:, 0].astype(np.float64) + 1j * x[:, 1].astype(np.float64)
    res = {"bytes": b, "x_cf32": x}
    for fk in ("fir64_c2", "fir64_c3", "fir128_c4"):
        h = taps32[fk].astype(np.float64)
        y = sg.lfilter(h, [1.0], xc)
        res[f"{fk}__y"] = np.stack([y.real, y.imag], 1)
    for fk, L, M in (("fir64_c2", 3, 2), ("fir128_c4", 5, 4), ("fir64_c2", 2, 3)):
        y = res[f"{fk}__y"]; yc = y[:, 0] + 1j * y[:, 1]
        hr = taps32[f"rs_{L}_{M}"].astype(np.float64)
        z = sg.upfirdn(hr, yc, up=L, down=M)[: -(-n * L // M)]
        res[f"{fk}__rs_{L}_{M}"] = np.stack([z.real, z.imag], 1)
    # FM demod of the narrow-band FIR output
    y = res["fir64_c3__y"]; yc = y[:, 0] + 1j * y[:, 1]
    d = np.empty(n); d[0] = 0.0; d[1:] = np.angle(yc[1:] * np.conj(yc[:-1]))
    res["fir64_c3__fm_demod"] = d
    # FM mod: message = 1 kHz tone + noise, kf = 75 kHz deviation per unit
    rng = np.random.default_rng(0xF3)
    m = (0.8 * np.sin(2 * np.pi * 1e3 * np.arange(n) / 4e6 * 50) + 0.1 * rng.standard_normal(n)).astype(np.float32)
    ph = np.cumsum(2 * np.pi * 75e3 * m.astype(np.float64) / 4e6)
    res["fm_msg"] = m
    res["fm_mod_kf"] = np.float64(75e3)
    res["fm_mod_iq"] = np.stack([np.cos(ph), np.sin(ph)], 1)
    # TX chain C5: FM mod -> 2/3 resample (fp64)
    zc = sg.upfirdn(taps32["rs_2_3"].astype(np.float64), np.cos(ph) + 1j * np.sin(ph), up=2, down=3)[: -(-n * 2 // 3)]
    res["fm_mod_rs_2_3"] = np.stack([zc.real, zc.imag], 1)
    # Butterworth order 6 at the three reference cut-offs (fc = bw/2, fs = 4e6):
    xi = i.astype(np.float64)
    for bw in (20e3, 50e3, 100e3):
        sos = sg.butter(6, bw / 2, "low", fs=4e6, output="sos")
        res[f"iir_{int(bw/1e3)}k__sos"] = sos
        res[f"iir_{int(bw/1e3)}k__y_i"] = sg.sosfilt(sos, xi)
        res[f"iir_{int(bw/1e3)}k__y_q"] = sg.sosfilt(sos, q.astype(np.float64))
    res["iq_int16"] = np.stack([i, q], 1)
    np.savez_compressed(os.path.join(OUT, "dsp_float.npz"), **res)


def main():
    os.makedirs(OUT, exist_ok=True)
    orc.build(ref=True)
    assert orc.have_ref(), "compiled reference missing: run in the build container"
    gen_kat(); gen_rx(); gen_read(); gen_tx(); gen_debug(); gen_ring(); gen_dsp()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
