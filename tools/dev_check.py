"""Developer smoke: HIP path vs CPU oracle on a few shapes, with timings (run on the GPU box)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import oracle
from pointwise_amd import conv3p_op as op
from tests.parity_util import make_case, rel_err

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))
cases = [("modelnet", 2, 256, 3, 9, (3, 3, 3), (1, 1, 1), np.float32),
         ("modelnet", 4, 2048, 9, 9, (3, 3, 3), (2, 2, 2), np.float32),
         ("lattice", 2, 512, 9, 9, (3, 3, 3), (3, 3, 3), np.float32),
         ("room", 2, 1000, 36, 13, (3, 3, 3), (1, 1, 1), np.float32),
         ("modelnet", 2, 300, 5, 7, (3, 3, 3), (4, 4, 4), np.float32),
         ("modelnet", 2, 300, 5, 7, (2, 1, 3), (1, 2, 3), np.float64),
         ("lattice", 1, 256, 3, 9, (3, 3, 3), (2, 2, 2), np.float64)]
for kind, B, N, ci, co, fzyx, s, dt in cases:
    P, X, W, dY = make_case(kind, B, N, ci, co, fzyx, seed=5, dtype=dt)
    t = lambda a: torch.from_numpy(a).to(dev)
    cnt = op.neighbor_count(t(P), fzyx, s, 0.1).cpu().numpy()
    cnt_ref = oracle.neighbor_count(P, fzyx, s, 0.1)
    y = op.conv3p(t(P), t(X), t(W), s, 0.1).cpu().numpy()
    dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, 0.1)
    y_ref = oracle.forward(P, X, W, s, 0.1)
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, 0.1)
    print(kind, B, N, ci, co, fzyx, s, dt.__name__, "count_equal", np.array_equal(cnt, cnt_ref),
          "y %.2e dx %.2e dw %.2e" % (rel_err(y, y_ref), rel_err(dx.cpu().numpy(), dx_ref),
                                      rel_err(dw.cpu().numpy(), dw_ref)))

# timing: cfg2 layer shapes
for ci, co, s in [(3, 9, 1), (9, 9, 2), (9, 9, 3), (9, 9, 4)]:
    P, X, W, dY = make_case("modelnet", 32, 2048, ci, co, seed=9)
    tp, tx, tw, tdy = [torch.from_numpy(a).to(dev) for a in (P, X, W, dY)]
    for fn, name in [(lambda: op.conv3p(tp, tx, tw, (s, s, s), 0.1), "fwd"),
                     (lambda: op.conv3p_grad(tdy, tp, tx, tw, (s, s, s), 0.1), "bwd")]:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print("B32 N2048 %d->%d s%d %s: %.1f us" % (ci, co, s, name, (time.perf_counter() - t0) / 20 * 1e6))
