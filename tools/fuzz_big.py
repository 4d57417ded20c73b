"""developer (ON THE GPU BOX): the register-path shapes on clouds searched in SEVERAL groups (N > 8192: four consecutive
lanes per centre instead of lanes by list length) and beyond the fused search (N > 16384), against the oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import oracle
from pointwise_amd import conv3p_op as op
from tests.parity_util import make_case, rel_err
dev = torch.device("cuda:0"); bad = 0
for N in (8193, 9000, 12000, 16384, 17000, 20000):
    for (ci, co), s, kind in (((3, 9), (1, 1, 1), "modelnet"), ((9, 9), (2, 2, 2), "room"), ((36, 13), (1, 1, 1), "cube"), ((9, 9), (3, 3, 3), "modelnet")):
        P, X, W, dY = make_case(kind, 2, N, ci, co, (3, 3, 3), seed=N + ci)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        cnt = op.neighbor_count(t(P), (3, 3, 3), s, 0.1).cpu().numpy()
        y = op.conv3p(t(P), t(X), t(W), s, 0.1).cpu().numpy()
        dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, 0.1)
        nthr = min(32, os.cpu_count() or 1)
        ry = oracle.forward(P, X, W, s, 0.1, nthreads=nthr); rdx, rdw = oracle.backward(dY, P, X, W, s, 0.1, nthreads=nthr)
        e = (np.array_equal(cnt, oracle.neighbor_count(P, (3, 3, 3), s, 0.1)), rel_err(y, ry), rel_err(dx.cpu().numpy(), rdx), rel_err(dw.cpu().numpy(), rdw))
        ok = e[0] and e[1] <= 1e-5 and e[2] <= 1e-5 and e[3] <= 5e-5
        bad += 0 if ok else 1
        print(N, ci, co, s, kind, "counts", e[0], "y %.1e dx %.1e dw %.1e" % e[1:], "" if ok else "  <-- BAD")
print("bad:", bad)
