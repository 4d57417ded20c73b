"""developer: time the shapes outside the models' own (any-shape paths: padded matrix-core classes, fp64 blocks, > 256 channels),
at the cfg2 size"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pointwise_amd import conv3p_op as op, synth
dev = torch.device("cuda:0")
B, N = 32, 2048
for (ci, co, dt, tdt) in ((5, 7, np.float32, torch.float32), (36, 13, np.float64, torch.float64), (16, 16, np.float32, torch.float32),
                          (5, 7, np.float64, torch.float64), (32, 64, np.float64, torch.float64), (320, 320, np.float32, torch.float32)):
    P = torch.from_numpy(synth.modelnet_like(B, N, 5).astype(dt)).to(dev)
    X = torch.from_numpy(synth.features(B, N, ci, 6, dtype=dt)).to(dev)
    W = torch.from_numpy(synth.filter_weights(3, 3, 3, ci, co, 7, dtype=dt)).to(dev)
    dY = torch.from_numpy(synth.upstream_grad(B, N, co, 8, dtype=dt)).to(dev)
    cache = op.NeighborCache(B, N, tdt, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
    for _ in range(2):
        y = op.conv3p(P, X, W, (2, 2, 2), 0.1, cache=cache); dx, dw = op.conv3p_grad(dY, P, X, W, (2, 2, 2), 0.1, cache=cache)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = int(os.environ.get('REPS', '10'))
    for _ in range(reps):
        y = op.conv3p(P, X, W, (2, 2, 2), 0.1, cache=cache, points_unchanged=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(reps):
        dx, dw = op.conv3p_grad(dY, P, X, W, (2, 2, 2), 0.1, cache=cache, points_unchanged=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s %d->%d s2 cached geometry: forward %.3f ms, backward %.3f ms" % (dt.__name__, ci, co, (t1 - t0) / reps * 1e3, (t2 - t1) / reps * 1e3))
