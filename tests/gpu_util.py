"""Helpers shared by the -m gpu tests: torch supplies device memory only."""
import numpy as np
import torch

from cariboulite_amd import hip

DEV = "cuda:0"


def dev_bytes(buf, pad=64):
    """uint8 numpy -> device tensor with a little slack after the data."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    t = torch.zeros(buf.size + pad, dtype=torch.uint8, device=DEV)
    if buf.size:
        t[:buf.size] = torch.from_numpy(buf.copy()).to(DEV)
    return t


def gpu_rx_data_analyze(channel, buf, fmt=hip.FORMAT_CS16, want_meta=True, chunk_len=None, n_chunks=1,
                        stride=None):
    """find_offsets + unpack through the C-ABI on sentinel-filled outputs, like orc.rx_data_analyze."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    total = buf.size
    chunk_len = total if chunk_len is None else chunk_len
    stride = chunk_len if stride is None else stride
    if n_chunks == 1:
        stride = (stride + 3) // 4 * 4 + 4
    d = dev_bytes(buf)
    offs = torch.full((max(n_chunks, 1),), -7, dtype=torch.int32, device=DEV)
    hip.smi_find_offsets(d, total, max(stride, 4), chunk_len, n_chunks, offs)
    n_slots = total // 4 + 2
    if fmt == hip.FORMAT_CS16:
        out = torch.full((n_slots, 2), -21846, dtype=torch.int16, device=DEV)
    elif fmt == hip.FORMAT_CF32:
        out = torch.full((n_slots, 2), float("nan"), dtype=torch.float32, device=DEV)
    elif fmt == hip.FORMAT_CF64:
        out = torch.full((n_slots, 2), float("nan"), dtype=torch.float64, device=DEV)
    else:
        out = torch.full((n_slots, 2), -86, dtype=torch.int8, device=DEV)
    meta = torch.full((n_slots,), 0xAA, dtype=torch.uint8, device=DEV) if want_meta else None
    if total:
        hip.smi_unpack(channel, d, total, max(stride, 4), chunk_len, n_chunks, offs, fmt, out, meta)
    torch.cuda.synchronize()
    return offs.cpu().numpy(), out.cpu().numpy(), (meta.cpu().numpy() if want_meta else None)
