"""developer (ON THE GPU BOX): the cfg2 step as 8 independent cached op calls (+ SELU ops), no hints -- what the TF shim
does.  Prints ms/step; under rocprofv3 --kernel-trace gives the timeline (tools/timeline.py)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from pointwise_amd import conv3p_op as op, synth, stack

dev = torch.device("cuda:0")
B, N = 32, 2048
tPs = [torch.from_numpy(synth.modelnet_like(B, N, seed=1236 + i)).to(dev) for i in range(4)]
tXs = [t.clone() for t in tPs]
gcat = torch.cat([torch.from_numpy(synth.upstream_grad(B, N, 9, 77 + li)).to(dev) for li in range(4)], dim=2).contiguous()
st = stack.Conv3pStack(3, None, device=dev, seed=1234)
cache = op.NeighborCache(B, N, torch.float32, dev, slots=4, max_taps=27, max_cin=9, max_cout=9)
ctr = [0]
def step():
    i = ctr[0] % 4
    ctr[0] += 1
    return bench.op_boundary_cached_step(st, cache, tPs[i], tXs[i], gcat)
print("op boundary, cached, no hints: %.4f ms/step" % (bench.timed(dev, step, 20, 5) * 1e3))
