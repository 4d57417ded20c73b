"""developer: fp32 vs fp64 on the register path (cfg2 size, 9->9, stride 2)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pointwise_amd import conv3p_op as op, synth
dev = torch.device("cuda:0")
B, N = 32, 2048
for dt, tdt in ((np.float32, torch.float32), (np.float64, torch.float64)):
    P = torch.from_numpy(synth.modelnet_like(B, N, 5).astype(dt)).to(dev)
    X = torch.from_numpy(synth.features(B, N, 9, 6, dtype=dt)).to(dev)
    W = torch.from_numpy(synth.filter_weights(3, 3, 3, 9, 9, 7, dtype=dt)).to(dev)
    dY = torch.from_numpy(synth.upstream_grad(B, N, 9, 8, dtype=dt)).to(dev)
    cache = op.NeighborCache(B, N, tdt, dev, slots=1, max_taps=27, max_cin=9, max_cout=9)
    for _ in range(3):
        y = op.conv3p(P, X, W, (2, 2, 2), 0.1, cache=cache); dx, dw = op.conv3p_grad(dY, P, X, W, (2, 2, 2), 0.1, cache=cache)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        y = op.conv3p(P, X, W, (2, 2, 2), 0.1, cache=cache, points_unchanged=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20):
        dx, dw = op.conv3p_grad(dY, P, X, W, (2, 2, 2), 0.1, cache=cache, points_unchanged=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(dt.__name__, "9->9 s2 cached geometry: forward %.3f ms, backward %.3f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t1) / 20 * 1e3))
