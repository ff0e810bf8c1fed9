#!/usr/bin/env python3
"""Diagnostic: run the sync-recovery call sequences (tests/test_gpu_sync_recovery.py) with every device buffer handed to
clhip_rx_pipe_run_smi embedded in a larger allocation whose surroundings hold a sentinel, and report any byte outside the
buffers that a kernel touched.  (An out-of-bounds store into memory that happens to be mapped goes unnoticed by the tests;
into unmapped memory it is an asynchronous GPU fault.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cariboulite_amd import hip
import test_gpu_sync_recovery as T

dev = torch.device("cuda:0")
GUARD = 1 << 20            # bytes on each side
t = np.load(os.path.join(ROOT, "tests", "golden", "taps.npz"))
bad = 0


def guarded(nbytes, fill_byte):
    """(whole uint8 tensor, offset of the payload)"""
    whole = torch.full((GUARD + nbytes + GUARD,), fill_byte, dtype=torch.uint8, device=dev)
    return whole, GUARD


def check(name, whole, off, nbytes, fill_byte, ci, kind):
    global bad
    lo, hi = whole[:off], whole[off + nbytes:]
    for side, g in (("before", lo), ("after", hi)):
        idx = torch.nonzero(g != fill_byte).flatten()
        if idx.numel():
            bad += 1
            first, last = int(idx[0]), int(idx[-1])
            base = 0 if side == "before" else off + nbytes
            print(f"  !! {name}: {idx.numel()} guard bytes {side} the buffer changed (call {ci} {kind}); offsets {first - (off if side == 'before' else 0)}..{last - (off if side == 'before' else 0)} relative to the buffer {'start' if side == 'before' else 'end'}")


for channel in (0, 1):
    for chunk_len, n_chunks in ((4 * 8192, 5), (524288, 3)):
        calls = T.build_calls(channel, chunk_len, n_chunks, (1, 3, 6), seed=40 + channel)
        pipe = hip.RxPipe(1, channel, t["fir64_c2"], t["rs_3_2"], 3, 2, hip.PIPE_OUT_IQ)
        for ci, (b, kind) in enumerate(calls):
            n = b.size // 4
            nch = -(-b.size // chunk_len)
            no = pipe.out_count(n)
            w_in, o_in = guarded(b.size, 0x5A)
            w_in[o_in:o_in + b.size] = torch.from_numpy(b).to(dev)
            w_off, o_off = guarded(4 * nch, 0x5B)
            w_cs, o_cs = guarded(4 * (n + 2), 0x5C)
            w_out, o_out = guarded(8 * (no + 8), 0x5D)
            h_offs = np.full(nch, 99, dtype=np.int32)
            rc = hip.lib().clhip_rx_pipe_run_smi(pipe.h, w_in.data_ptr() + o_in, 0, b.size, chunk_len, w_off.data_ptr() + o_off,
                                                 h_offs.ctypes.data, w_cs.data_ptr() + o_cs, w_out.data_ptr() + o_out, no + 8,
                                                 torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            tmp = w_in.clone(); tmp[o_in:o_in + b.size] = 0x5A          # the input must be untouched altogether
            check("input", w_in, o_in, b.size, 0x5A, ci, kind)
            check("offs", w_off, o_off, 4 * nch, 0x5B, ci, kind)
            check("cs16", w_cs, o_cs, 4 * (n + 2), 0x5C, ci, kind)
            check("out", w_out, o_out, 8 * (no + 8), 0x5D, ci, kind)
            print(f"ch {channel} chunk {chunk_len} call {ci} {kind}: rc {rc}")
print("guard violations:", bad)
