#!/usr/bin/env python3
"""Memory-system floor for config 2's traffic shape (1 GiB read + 3 GiB written per step)."""
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 28
a = torch.empty(n, dtype=torch.int32, device=dev).random_()
out = torch.empty((n * 3 // 2, 2), dtype=torch.float32, device=dev)
def t(fn, reps=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
w = t(lambda: out.fill_(1.0))
print("fill 3 GiB: %.3f ms  %.2f TB/s write" % (w * 1e3, out.numel() * 4 / w / 1e12))
src = torch.empty_like(out)
c = t(lambda: out.copy_(src))
print("copy 3 GiB->3 GiB: %.3f ms  %.2f TB/s total" % (c * 1e3, 2 * out.numel() * 4 / c / 1e12))
o3 = out.view(-1)[: 3 * n].view(3, n)
af = a.view(torch.float32)
def expand():
    o3[0].copy_(af); 
r = t(lambda: torch.add(af, 1.0, out=o3[0]))
print("read 1 GiB + write 1 GiB: %.3f ms  %.2f TB/s" % (r * 1e3, 2 * n * 4 / r / 1e12))
# 1 read : 3 write shape
def shape():
    torch.add(af, 1.0, out=o3[0]); o3[1].fill_(2.0); o3[2].fill_(3.0)
s = t(shape)
print("1 GiB read + 3 GiB write (3 kernels): %.3f ms  -> %.2f TB/s" % (s * 1e3, 4 * n * 4 / s / 1e12))
