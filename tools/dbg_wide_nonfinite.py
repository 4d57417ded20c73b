"""developer: the wide path with non-finite inputs, per-output error report"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from parity_util import make_case
from pointwise_amd import conv3p_op as op
from oracle import oracle
dev = torch.device("cuda:0")
ci, co = 300, 70
B, N = 2, 300
P, X, W, dY = make_case("room", B, N, ci, co, seed=1400)
X = X.copy(); dY = dY.copy()
X[0, 17, 3] = np.inf; X[1, 200, ci - 1] = np.nan; dY[0, 99, co - 1] = np.inf
s = (1, 1, 1)
with np.errstate(all="ignore"):
    y_ref = oracle.forward(P, X, W, s, 0.1); dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, 0.1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
y = op.conv3p(t(P), t(X), t(W), s, 0.1).cpu().numpy()
dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, 0.1)
for name, got, ref in (("y", y, y_ref), ("dx", dx.cpu().numpy(), dx_ref), ("dw", dw.cpu().numpy(), dw_ref)):
    bad = ~np.isfinite(ref); ok = ~bad
    d = np.abs(got[ok] - ref[ok])
    print(name, "masks equal", np.array_equal(~np.isfinite(got), bad), "max abs err", d.max(), "max ref", np.abs(ref[ok]).max(), "n bad", bad.sum())
    if name == "dw":
        e = np.abs(np.where(bad, 0, got - ref)); idx = np.unravel_index(np.argmax(e), e.shape); print("  worst at", idx, got[idx], ref[idx])
