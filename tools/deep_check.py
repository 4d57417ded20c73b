"""Developer check of the deep-channel (matrix-core) path: which kernels ran, parity vs the oracle, timing."""
import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from oracle import oracle
from pointwise_amd import _lib, conv3p_op as op, synth
from tests.parity_util import rel_err
lib = _lib.load(); dev = torch.device("cuda:0")
def kinds():
    out = {}
    for k in range(lib.conv3p_profile_kinds()):
        n, ms = ctypes.c_uint64(0), ctypes.c_double(0.0)
        lib.conv3p_profile_read(k, ctypes.byref(n), ctypes.byref(ms))
        if n.value: out[lib.conv3p_profile_name(k).decode()] = (n.value, round(ms.value / n.value * 1e3, 1))
    return out
B, N, ci, co = 1, 2048, 128, 256
P = synth.room_like(B, N, 3); X = synth.features(B, N, ci, 4, points=P); W = synth.filter_weights(3, 3, 3, ci, co, 5)
dY = synth.upstream_grad(B, N, co, 6)
t = lambda a: torch.from_numpy(a).to(dev)
tp, tx, tw, tdy = t(P), t(X), t(W), t(dY)
lib.conv3p_profile_reset(); lib.conv3p_profile_enable(1)
y = op.conv3p(tp, tx, tw, (1, 1, 1), 0.1); dx, dw = op.conv3p_grad(tdy, tp, tx, tw, (1, 1, 1), 0.1)
torch.cuda.synchronize(); lib.conv3p_profile_enable(0)
print("kernels (launches, avg us):", kinds())
t0 = time.time(); yr = oracle.forward(P, X, W, (1, 1, 1), 0.1, nthreads=1); dxr, dwr = oracle.backward(dY, P, X, W, (1, 1, 1), 0.1)
print("oracle %.1f s; y %.2e dx %.2e dw %.2e" % (time.time() - t0, rel_err(y.cpu().numpy(), yr), rel_err(dx.cpu().numpy(), dxr), rel_err(dw.cpu().numpy(), dwr)))
# cfg5 per-GPU shard size: B=16, N=8192
B, N = 16, 8192
P = synth.room_like(B, N, 7, extent=(2.4, 2.4, 3.0)); tp = t(P); tx = t(synth.features(B, N, ci, 8, points=P)); tdy = t(synth.upstream_grad(B, N, co, 9))
for _ in range(2):
    y = op.conv3p(tp, tx, tw, (1, 1, 1), 0.1); dx, dw = op.conv3p_grad(tdy, tp, tx, tw, (1, 1, 1), 0.1)
torch.cuda.synchronize()
lib.conv3p_profile_reset(); lib.conv3p_profile_enable(1)
t0 = time.perf_counter()
for _ in range(3):
    y = op.conv3p(tp, tx, tw, (1, 1, 1), 0.1); dx, dw = op.conv3p_grad(tdy, tp, tx, tw, (1, 1, 1), 0.1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
lib.conv3p_profile_enable(0)
print("cfg5 shard B=16 N=8192 128->256: %.2f ms fwd+bwd -> %.2f Mpoints/s" % (dt * 1e3, B * N / dt / 1e6))
print("kernels (launches, avg us):", kinds())
cnt = op.neighbor_count(tp, (3, 3, 3), (1, 1, 1), 0.1); print("mean neighbours/point %.1f" % (cnt.sum().item() / (B * N)))
