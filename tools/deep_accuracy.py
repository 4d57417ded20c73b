"""developer (ON THE GPU BOX): cfg5-sized layer, one cloud -- max error of the HIP forward / backward and of the fp32
oracle (the reference's own arithmetic) against the fp64 oracle, on channel slices."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import oracle
from pointwise_amd import conv3p_op as op, synth
dev = torch.device("cuda:0")
N, ci, co, s = 8192, 128, 256, (1, 1, 1)
P = synth.room_like(16, N, 7, extent=(2.4, 2.4, 3.0))[3:4]
X = synth.features(16, N, ci, 8, points=synth.room_like(16, N, 7, extent=(2.4, 2.4, 3.0)))[3:4]
W = synth.filter_weights(3, 3, 3, ci, co, 5)
dY = synth.upstream_grad(16, N, co, 9)[3:4]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
y = op.conv3p(t(P), t(X), t(W), s, 0.1).cpu().numpy()
dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, 0.1)
dx, dw = dx.cpu().numpy(), dw.cpu().numpy()
for c0 in (0, 96, 240):
    cs = slice(c0, c0 + 16)
    Ws = np.ascontiguousarray(W[..., cs])
    y32 = oracle.forward(P, X, Ws, s, 0.1)
    y64 = oracle.forward(P.astype(np.float64), X.astype(np.float64), Ws.astype(np.float64), s, 0.1)
    m = np.abs(y64).max()
    print("y[%3d:%3d]  max|y| %.3f   hip vs f64 %.3e   oracle32 vs f64 %.3e   hip vs oracle32 %.3e  (all / max|y|)" % (
        c0, c0 + 16, m, np.abs(y[..., cs] - y64).max() / m, np.abs(y32 - y64).max() / m, np.abs(y[..., cs] - y32).max() / m))
for k0 in (0, 64, 120):
    ks = slice(k0, k0 + 8)
    Xs, Ws = np.ascontiguousarray(X[..., ks]), np.ascontiguousarray(W[:, :, :, ks, :])
    dx32, dw32 = oracle.backward(dY, P, Xs, Ws, s, 0.1)
    dx64, dw64 = oracle.backward(dY.astype(np.float64), P.astype(np.float64), Xs.astype(np.float64), Ws.astype(np.float64), s, 0.1)
    m, mw = np.abs(dx64).max(), np.abs(dw64).max()
    print("dX[%3d:%3d] max %.3f   hip vs f64 %.3e   oracle32 vs f64 %.3e   hip vs oracle32 %.3e" % (
        k0, k0 + 8, m, np.abs(dx[..., ks] - dx64).max() / m, np.abs(dx32 - dx64).max() / m, np.abs(dx[..., ks] - dx32).max() / m))
    print("dW[%3d:%3d] max %.3f   hip vs f64 %.3e   oracle32 vs f64 %.3e   hip vs oracle32 %.3e" % (
        k0, k0 + 8, mw, np.abs(dw[:, :, :, ks, :] - dw64).max() / mw, np.abs(dw32 - dw64).max() / mw, np.abs(dw[:, :, :, ks, :] - dw32).max() / mw))
