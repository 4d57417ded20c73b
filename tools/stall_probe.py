import sys, time, os
sys.path.insert(0, "/root/repo")
import torch
from pointwise_amd import stack, synth, _lib
dev = torch.device("cuda:0")
B, N = 32, 2048
Ps = [torch.from_numpy(synth.modelnet_like(B, N, seed=10 + i)).to(dev) for i in range(4)]
ups = [torch.from_numpy(synth.upstream_grad(B, N, 9, 77 + i)).to(dev) for i in range(4)]
mode = sys.argv[1] if len(sys.argv) > 1 else "prefetch"
st = stack.Conv3pStack(3, None, device=dev, seed=1, overlap_search=(mode != "serial"))
ts = []
import gc
if len(sys.argv) > 2 and sys.argv[2] == 'freeze':
    gc.collect(); gc.freeze()
torch.cuda.synchronize()
t00 = time.perf_counter()
for i in range(300):
    t0 = time.perf_counter()
    st.forward(Ps[i % 4], Ps[i % 4])
    if mode == "prefetch":
        st.prefetch(Ps[(i + 1) % 4])
    st.backward(ups)
    ts.append(time.perf_counter() - t0)
tq = time.perf_counter() - t00
torch.cuda.synchronize()
tt = time.perf_counter() - t00
import numpy as np
ts = np.array(ts) * 1e3
print(mode, "enqueue total %.1f ms, wall %.1f ms, median step enqueue %.3f ms" % (tq * 1e3, tt * 1e3, np.median(ts)))
idx = np.argsort(-ts)[:8]
print("slowest enqueues:", [(int(i), round(float(ts[i]), 2)) for i in sorted(idx)])
