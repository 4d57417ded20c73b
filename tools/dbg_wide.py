"""developer (ON THE GPU BOX): 36 -> 13 backward on rooms against the oracle, stateless and cached; where the errors are"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import oracle
from pointwise_amd import conv3p_op as op, synth
from tests.parity_util import make_case
dev = torch.device("cuda:0")
B, N = int(sys.argv[1]), int(sys.argv[2])
S = (int(sys.argv[3]),) * 3 if len(sys.argv) > 3 else (1, 1, 1)
P, X, W, dY = make_case("room", B, N, 36, 13, seed=5)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
tp, tx, tw, tdy = t(P), t(X), t(W), t(dY)
ref = oracle.backward(dY, P, X, W, S, 0.1)
for mode in ("stateless", "cached"):
    cache = None if mode == "stateless" else op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=36, max_cout=13)
    dx, dw = op.conv3p_grad(tdy, tp, tx, tw, S, 0.1, cache=cache) if cache is not None else op.conv3p_grad(tdy, tp, tx, tw, S, 0.1)
    torch.cuda.synchronize()
    dx, dw = dx.cpu().numpy(), dw.cpu().numpy()
    ex = np.abs(dx - ref[0]).max(axis=2)
    bad = ex > 1e-4 * max(1, np.abs(ref[0]).max())
    print(mode, "dX max err %.3g, bad points %d of %d; dW max err %.3g (max|dW| %.3g)" % (ex.max(), bad.sum(), bad.size, np.abs(dw - ref[1]).max(), np.abs(ref[1]).max()))
    if bad.any():
        ew = np.abs(dw - ref[1]).max(axis=(1, 2))
        print("  dW err per tap:", np.array2string(ew, precision=2))
        b0 = np.argwhere(bad)[:, 0]
        print("  bad per cloud:", np.bincount(b0, minlength=B))
        # ratio of wrong value to right value at the worst point
        w = np.unravel_index(np.argmax(ex), ex.shape)
        print("  worst point", w, "got", dx[w][:6], "ref", ref[0][w][:6])
