"""Synthetic SMI byte streams for tests and bench.

RX word layout (caribou_smi.c:338-340, firmware/smi_ctrl.v:122-140 -> LE u32):
    [31:30]=10 | [29:17]=A(13) | [16]=0 | [15:14]=01 | [13:1]=B(13) | [0]=sync
    S1G: A=I, B=Q        HiF: A=Q, B=I
"""
import numpy as np

SEED = 0xCA41B0
FS = 4_000_000
CH_S1G, CH_HIF = 0, 1


def iq_to_words(i13, q13, channel=CH_S1G, sync=None):
    """int arrays in [-4096, 4095] -> uint32 RX words."""
    i13 = np.asarray(i13).astype(np.int64) & 0x1FFF
    q13 = np.asarray(q13).astype(np.int64) & 0x1FFF
    a, b = (i13, q13) if channel == CH_S1G else (q13, i13)
    w = 0x80004000 | (a << 17) | (b << 1)
    if sync is not None:
        w = w | (np.asarray(sync).astype(np.int64) & 1)
    return w.astype(np.uint32)


def tone_noise_iq(n, stream=0, f_tone=250e3, amp=1800.0, noise=0.3, n0=0):
    """round(amp * (tone + noise * N(0,1))) clipped to 13 bits; tone at 250 kHz."""
    rng = np.random.default_rng(SEED + stream)
    t = (np.arange(n0, n0 + n, dtype=np.float64)) / FS
    ph = 2 * np.pi * f_tone * t
    i = amp * (np.cos(ph) + noise * rng.standard_normal(n))
    q = amp * (np.sin(ph) + noise * rng.standard_normal(n))
    i = np.clip(np.rint(i), -4096, 4095).astype(np.int16)
    q = np.clip(np.rint(q), -4096, 4095).astype(np.int16)
    return i, q


def smi_stream_bytes(n, channel=CH_S1G, stream=0, n0=0):
    """n samples of the standard synthetic stream as the byte buffer /dev/smi would deliver."""
    i, q = tone_noise_iq(n, stream=stream, n0=n0)
    sync = ((np.arange(n0, n0 + n) % FS) == 0)
    return iq_to_words(i, q, channel, sync).view(np.uint8), i, q


def torch_smi_words(n, device, channel=CH_S1G, stream=0, chunk=1 << 24):
    """Device-side generator for bench-sized buffers (int32 tensor of RX words).
    Same distribution as tone_noise_iq (torch's RNG, so not the same bytes)."""
    import torch
    out = torch.empty(n, dtype=torch.int32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(SEED + stream)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        idx = torch.arange(s, s + m, device=device, dtype=torch.float64)
        ph = (2 * np.pi * 250e3 / FS) * idx
        i = 1800.0 * (torch.cos(ph) + 0.3 * torch.randn(m, device=device, generator=g, dtype=torch.float64))
        q = 1800.0 * (torch.sin(ph) + 0.3 * torch.randn(m, device=device, generator=g, dtype=torch.float64))
        i = torch.clamp(torch.round(i), -4096, 4095).to(torch.int64) & 0x1FFF
        q = torch.clamp(torch.round(q), -4096, 4095).to(torch.int64) & 0x1FFF
        a, b = (i, q) if channel == CH_S1G else (q, i)
        w = 0x80004000 | (a << 17) | (b << 1)
        w = w | ((torch.arange(s, s + m, device=device) % FS) == 0).to(torch.int64)
        # reinterpret the low 32 bits as int32
        out[s:s + m] = ((w + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32)
        del idx, ph, i, q, a, b, w
    return out
