"""developer (ON THE GPU BOX): random op configurations against the oracle -- sizes that are not multiples of a tile,
one-cloud batches, anisotropic strides, 1 / 2 / 3 / 5-tap axes, lattice / identical / isolated clouds, the models' and
other channel shapes, stateless and cached.  usage: python tools/fuzz_gpu.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import oracle
from pointwise_amd import conv3p_op as op
from tests.parity_util import make_case, rel_err, TOL
dev = torch.device("cuda:0")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
shapes = [(3, 9), (9, 9), (6, 9), (12, 9), (36, 13), (3, 3), (9, 3), (5, 7), (16, 16), (33, 20)]
kinds = ["modelnet", "room", "cube", "lattice", "vlattice", "identical", "isolated"]
bad = 0
for it in range(ncase):
    kind = kinds[rng.integers(len(kinds))]
    ci, co = shapes[rng.integers(len(shapes))]
    B = int(rng.integers(1, 5))
    N = int(rng.choice([1, 2, 63, 64, 65, 100, 127, 129, 200, 500, 777, 1024, 1500, 2048, 2500]))
    if kind in ("identical",) and N > 300: N = 300
    if ci * co > 300 and N > 600: N = 600
    f = tuple(int(v) for v in rng.choice([1, 2, 3, 3, 3, 5], size=3))
    if f[0] * f[1] * f[2] > 64: f = (3, 3, 3)
    s = tuple(int(v) for v in rng.integers(1, 5, size=3)) if rng.random() < 0.4 else (int(rng.integers(1, 5)),) * 3
    dt = np.float64 if rng.random() < 0.15 else np.float32
    P, X, W, dY = make_case(kind, B, N, ci, co, f, seed=1000 + it, dtype=dt)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cache = op.NeighborCache(B, N, torch.float32 if dt == np.float32 else torch.float64, dev, slots=2, max_taps=f[0] * f[1] * f[2], max_cin=ci, max_cout=co) if rng.random() < 0.5 else None
    try:
        cnt = op.neighbor_count(t(P), f, s, 0.1).cpu().numpy()
        y = op.conv3p(t(P), t(X), t(W), s, 0.1, cache=cache).cpu().numpy()
        dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, 0.1, cache=cache)
        dx, dw = dx.cpu().numpy(), dw.cpu().numpy()
    except Exception as e:
        print("CASE", it, kind, B, N, ci, co, f, s, dt.__name__, "raised", repr(e)[:200]); bad += 1; continue
    rc = oracle.neighbor_count(P, f, s, 0.1)
    ry = oracle.forward(P, X, W, s, 0.1); rdx, rdw = oracle.backward(dY, P, X, W, s, 0.1)
    ty, tw = TOL[np.dtype(dt)]
    floor = 0.0
    if kind == "identical":
        d = np.float64
        floor = rel_err(rdw, oracle.backward(dY.astype(d), P.astype(d), X.astype(d), W.astype(d), s, 0.1)[1])
    ok = np.array_equal(cnt, rc) and rel_err(y, ry) <= ty and rel_err(dx, rdx) <= ty and rel_err(dw, rdw) <= max(tw, 4 * floor)
    if not ok and np.array_equal(cnt, rc) and dt == np.float32:
        # many coincident points (lattice clouds): sums of thousands of equal-sign terms, where the reference's own fp32
        # loops are the looser side -- judge both against the exact sums over the oracle's pair lists
        from tests.parity_util import exact_from_oracle_lists
        worst = 0.0
        for b in range(B):
            ye, dxe, dwe = exact_from_oracle_lists(P[b], X[b], W, dY[b], s, 0.1)
            worst = max(worst, rel_err(y[b], ye) / ty, rel_err(dx[b], dxe) / ty)
        dwe = sum(exact_from_oracle_lists(P[b], X[b], W, dY[b], s, 0.1)[2] for b in range(B))
        e_hip, e_ref = rel_err(dw, dwe), rel_err(rdw, dwe)
        print("  case", it, "vs exact sums: y/dx %.2f of tol, dw hip %.2e, dw reference-fp32 %.2e" % (worst, e_hip, e_ref))
        ok = worst <= 1.0 and (e_hip <= tw or e_hip <= e_ref)
    if not ok:
        bad += 1
        print("CASE", it, kind, B, N, ci, co, f, s, dt.__name__, "cache" if cache else "stateless", "counts", np.array_equal(cnt, rc),
              "y %.2e dx %.2e dw %.2e" % (rel_err(y, ry), rel_err(dx, rdx), rel_err(dw, rdw)))
print("%d cases, %d bad" % (ncase, bad))
