"""developer (ON THE GPU BOX): the geometry of a cfg2 / cfg4 step alone on an idle GPU -- sort + ONE multi-stride search
launch (what the headline's prefetch enqueues), and the four strides as four single launches.
usage: python tools/search_time.py [cfg2|cfg4|cfg5]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pointwise_amd import conv3p_op as op, synth, stack

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device("cuda:0")
if cfg == "cfg2":
    B, N, mk = 32, 2048, lambda i: synth.modelnet_like(32, 2048, seed=100 + i)
elif cfg == "cfg4":
    B, N, mk = 16, 4096, lambda i: synth.room_like(16, 4096, 40 + i)
else:
    B, N, mk = 16, 8192, lambda i: synth.room_like(16, 8192, 7 + i, extent=(2.4, 2.4, 3.0))
Ps = [torch.from_numpy(mk(i)).to(dev) for i in range(3)]
strides = [(1, 1, 1)] if cfg == "cfg5" else [(s, s, s) for s in (1, 2, 3, 4)]
cache = op.NeighborCache(B, N, torch.float32, dev, slots=4, max_taps=27, max_cin=9, max_cout=9)


def timed(fn, n=30, warm=6):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


multi = timed(lambda i: op.cache_prepare_multi(Ps[i % 3], (3, 3, 3), strides, stack.VOXEL, cache))
def singles(i):
    for k, s in enumerate(strides):
        op.cache_prepare(Ps[i % 3], (3, 3, 3), s, stack.VOXEL, cache, points_unchanged=k > 0)
single = timed(singles)
print("%s geometry alone: multi %.1f us, %d single launches %.1f us" % (cfg, multi * 1e3, len(strides), single * 1e3))
