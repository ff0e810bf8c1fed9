import os, sys, time
import numpy as np, torch
DEV = torch.device("cuda", 0)
torch.zeros(1, device=DEV)
for size in (512 * 1024, 1536 * 1024, 4 * 1024 * 1024):
    h = np.random.default_rng(0).integers(0, 256, size=size, dtype=np.uint8)
    d = torch.zeros(size, dtype=torch.uint8, device=DEV)
    ht = torch.from_numpy(h)
    for _ in range(5): d.copy_(ht); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): d.copy_(ht)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    o = torch.empty(size, dtype=torch.uint8)
    for _ in range(5): o.copy_(d)
    t2 = time.perf_counter()
    for _ in range(50): o.copy_(d)
    t3 = time.perf_counter()
    print(f"{size/1024:7.0f} KiB  H2D {1e6*(t1-t0)/50:7.1f} us   D2H {1e6*(t3-t2)/50:7.1f} us", flush=True)
