"""developer (ON THE GPU BOX): backward of a 9 -> 9 layer with a 4 x 4 x 4 (64-tap) filter at the cfg2 size, dilated, with the
populated-rows kernel (64-bit tap sets) and with the dense-G kernel.  Under rocprofv3 --kernel-trace --stats: the kernel names."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pointwise_amd import conv3p_op as op, synth
dev = torch.device("cuda:0")
B, N, ci, co = 32, 2048, 9, 9
f = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 4, 4)
S = (2, 2, 2)
P = synth.modelnet_like(B, N, 40)
t = lambda a: torch.from_numpy(a).to(dev)
tp, tx, tw, tdy = t(P), t(synth.features(B, N, ci, 1, points=P)), t(synth.filter_weights(f[0], f[1], f[2], ci, co, 2)), t(synth.upstream_grad(B, N, co, 3))
for hint in (True, False):
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=f[0] * f[1] * f[2], max_cin=ci, max_cout=co, sparse_neighbourhoods=hint)
    g = lambda: op.conv3p_grad(tdy, tp, tx, tw, S, 0.1, cache=cache)
    g(); g(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g()
    torch.cuda.synchronize()
    print("filter %s stride 2, 9 -> 9, B=32 N=2048: backward with %s: %.3f ms" % (f, "populated rows" if hint else "dense G", (time.perf_counter() - t0) / 20 * 1e3))
