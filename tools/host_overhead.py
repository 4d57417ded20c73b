"""How long does the host need to ENQUEUE one step (no synchronisation) vs the GPU to execute it?"""
import sys, time
import torch
sys.path.insert(0, ".")
from pointwise_amd import stack, synth
dev = torch.device("cuda:0")
Ps = [torch.from_numpy(synth.modelnet_like(32, 2048, seed=i)).to(dev) for i in range(4)]
ups = [torch.from_numpy(synth.upstream_grad(32, 2048, 9, 7 + i)).to(dev) for i in range(4)]
st = stack.Conv3pStack(3, None, device=dev)
def step(i):
    st.forward(Ps[i % 4], Ps[i % 4]); st.backward(ups)
for i in range(10): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50): step(i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("enqueue %.3f ms/step, total %.3f ms/step" % (t_enq / 50 * 1e3, t_all / 50 * 1e3))
