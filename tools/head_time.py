"""developer: time the classification head's FC kernels at model size (B=32, K=73728, N=512) -> GB/s of W1 traffic"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pointwise_amd import _lib, head
lib = _lib.load(); dev = torch.device("cuda:0")
M, K, N = 32, 73728, 512
x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) / K ** 0.5; b = torch.zeros(N, device=dev)
dy = torch.randn(M, N, device=dev)
y = head.fully_connected(x, W, b); dW = torch.empty_like(W)
for _ in range(3):
    head.fully_connected(x, W, b); head.fully_connected_grad(x, W, y, dy, dW_out=dW)
torch.cuda.synchronize()
def t(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
tf = t(lambda: head.fully_connected(x, W, b)); tb = t(lambda: head.fully_connected_grad(x, W, y, dy, dW_out=dW))
wb = K * N * 4
print("fc1 forward  %.1f us  -> %.0f GB/s of W" % (tf * 1e6, wb / tf / 1e9))
print("fc1 backward %.1f us  -> %.0f GB/s (W read + dW write)" % (tb * 1e6, 2 * wb / tb / 1e9))
lib.conv3p_profile_reset(); lib.conv3p_profile_enable(1)
for _ in range(5): head.fully_connected(x, W, b); head.fully_connected_grad(x, W, y, dy, dW_out=dW)
torch.cuda.synchronize(); lib.conv3p_profile_enable(0)
for k in range(lib.conv3p_profile_kinds()):
    n, ms = ctypes.c_uint64(0), ctypes.c_double(0.0)
    lib.conv3p_profile_read(k, ctypes.byref(n), ctypes.byref(ms))
    if n.value: print("  %-22s %.1f us" % (lib.conv3p_profile_name(k).decode(), ms.value / n.value * 1e3))
tm = t(lambda: torch.matmul(x, W))
print("torch.matmul (rocBLAS) forward for comparison: %.1f us" % (tm * 1e6))
