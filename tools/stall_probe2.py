"""developer probe: does the HIP runtime stall once after N un-synchronised launches? (plain torch kernels)"""
import time, sys
import numpy as np
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1 << 25, device=dev)      # ~60 us kernel
y = torch.zeros(64, device=dev)
for name, t, n in (("long kernels", x, 5000), ("tiny kernels", y, 20000)):
    torch.cuda.synchronize()
    ts = np.empty(n)
    t00 = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter()
        t.add_(1.0)
        ts[i] = time.perf_counter() - t0
    tq = time.perf_counter() - t00
    torch.cuda.synchronize()
    tt = time.perf_counter() - t00
    big = np.nonzero(ts > 2e-3)[0]
    print(name, "enqueue %.1f ms wall %.1f ms median %.2f us; launches > 2 ms:" % (tq * 1e3, tt * 1e3, np.median(ts) * 1e6),
          [(int(i), round(float(ts[i]) * 1e3, 1)) for i in big[:12]])
