"""developer: per-kernel times of the deep path, forward and backward separately (cfg5 shard)"""
import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from pointwise_amd import _lib, conv3p_op as op, synth
lib = _lib.load(); dev = torch.device("cuda:0")
def kinds():
    out = {}
    for k in range(lib.conv3p_profile_kinds()):
        n, ms = ctypes.c_uint64(0), ctypes.c_double(0.0)
        lib.conv3p_profile_read(k, ctypes.byref(n), ctypes.byref(ms))
        if n.value: out[lib.conv3p_profile_name(k).decode()] = (n.value, round(ms.value / n.value * 1e3, 1))
    return out
ci, co = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 256)
B, N = 16, 8192
t = lambda a: torch.from_numpy(a).to(dev)
P = synth.room_like(B, N, 7, extent=(2.4, 2.4, 3.0)); tp = t(P); tx = t(synth.features(B, N, ci, 8, points=P)); tdy = t(synth.upstream_grad(B, N, co, 9))
tw = t(synth.filter_weights(3, 3, 3, ci, co, 5))
cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
for _ in range(2):
    y = op.conv3p(tp, tx, tw, (1, 1, 1), 0.1, cache=cache); dx, dw = op.conv3p_grad(tdy, tp, tx, tw, (1, 1, 1), 0.1, cache=cache)
torch.cuda.synchronize()
for name, fn in (("forward", lambda: op.conv3p(tp, tx, tw, (1, 1, 1), 0.1, cache=cache)),
                 ("backward", lambda: op.conv3p_grad(tdy, tp, tx, tw, (1, 1, 1), 0.1, cache=cache))):
    lib.conv3p_profile_reset(); lib.conv3p_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    lib.conv3p_profile_enable(0)
    k = kinds()
    print("%s %d->%d: %.2f ms;" % (name, ci, co, dt * 1e3), {n: v for n, v in k.items() if n.startswith("deep") or n.startswith("reduce")})
