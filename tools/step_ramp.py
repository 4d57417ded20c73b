"""developer (ON THE GPU BOX): duration of every one of the first steps of the headline loop (HIP events around each
step), to see what a short timed region (the driver's --steps 20 --warmup 5) measures."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pointwise_amd import synth, stack
dev = torch.device("cuda:0")
B, N = 32, 2048
tPs = [torch.from_numpy(synth.modelnet_like(B, N, seed=1236 + i)).to(dev) for i in range(4)]
gcat = torch.cat([torch.from_numpy(synth.upstream_grad(B, N, 9, 77 + li)).to(dev) for li in range(4)], dim=2).contiguous()
st = stack.Conv3pStack(3, None, device=dev, seed=1234)
st.tune(tPs[0]); st.prepare(B, N)
torch.cuda.synchronize()
n = 60
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
ev[0].record()
for i in range(n):
    t0 = time.perf_counter()
    st.forward(tPs[i % 4], tPs[i % 4]); st.prefetch(tPs[(i + 1) % 4]); st.backward(gcat)
    host.append((time.perf_counter() - t0) * 1e3)
    ev[i + 1].record()
torch.cuda.synchronize()
d = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("gpu ms per step:", " ".join("%.3f" % x for x in d))
print("host enqueue ms:", " ".join("%.3f" % x for x in host))
