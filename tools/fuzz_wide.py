"""developer (ON THE GPU BOX): the wide populated-rows backward (36 -> 13) against the oracle on dense clouds -- tiles over the
capacity of G (split by centres), several search groups, sizes that are not multiples of a tile, non-finite rows.
usage: python tools/fuzz_wide.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import oracle
from pointwise_amd import conv3p_op as op
from tests.parity_util import make_case, rel_err, TOL
dev = torch.device("cuda:0")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for it in range(ncase):
    kind = ["room", "cube", "identical", "modelnet", "lattice"][rng.integers(5)]
    B = int(rng.integers(1, 4))
    N = int(rng.choice([65, 127, 300, 640, 1000, 2048, 3000, 4096, 5000]))
    if kind == "identical" and N > 300: N = 300
    s = (int(rng.integers(1, 3)),) * 3
    f = (3, 3, 3) if rng.random() < 0.8 else (3, 3, 1)
    P, X, W, dY = make_case(kind, B, N, 36, 13, f, seed=3000 + it)
    if kind == "room" and rng.random() < 0.5:
        P = (P * 0.6).astype(np.float32)      # denser: more neighbours per tap, more populated rows per tile
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=36, max_cout=13) if rng.random() < 0.5 else None
    dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, 0.1, cache=cache)
    dx2, dw2 = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, 0.1, cache=cache)
    rdx, rdw = oracle.backward(dY, P, X, W, s, 0.1, nthreads=8)
    r64 = oracle.backward(dY.astype(np.float64), P.astype(np.float64), X.astype(np.float64), W.astype(np.float64), s, 0.1, nthreads=8)
    floor = rel_err(rdw, r64[1])
    ex, ew = rel_err(dx.cpu().numpy(), rdx), rel_err(dw.cpu().numpy(), rdw)
    rep = bool(torch.equal(dx, dx2) and torch.equal(dw, dw2))
    ok = ex <= 1e-5 and ew <= max(2e-5, 4 * floor) and rep
    if not ok: bad += 1
    print("%s case %d %s B=%d N=%d f=%s s=%d %s: dX %.2e dW %.2e (ref's own %.1e) reproducible %s" % ("ok " if ok else "BAD", it, kind, B, N, f, s[0], "cached" if cache is not None else "stateless", ex, ew, floor, rep))
print("bad:", bad)
