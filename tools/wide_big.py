import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import oracle
from pointwise_amd import conv3p_op as op
from tests.parity_util import make_case, rel_err
dev = torch.device("cuda:0")
for kind, B, N, s in (("room", 1, 16454, 1), ("room", 2, 9000, 1), ("cube", 1, 20000, 2), ("room", 1, 16454, 2)):
    P, X, W, dY = make_case(kind, B, N, 36, 13, seed=77)
    if kind == "room": P = (P * 0.7).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for cached in (False, True):
        cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=36, max_cout=13) if cached else None
        dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), (s,) * 3, 0.1, cache=cache)
        rdx, rdw = oracle.backward(dY, P, X, W, (s,) * 3, 0.1, nthreads=16)
        print(kind, B, N, s, "cached" if cached else "stateless", "dX %.2e dW %.2e" % (rel_err(dx.cpu().numpy(), rdx), rel_err(dw.cpu().numpy(), rdw)))
