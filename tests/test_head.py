"""The classification model's dense head on the FC kernels, against the float64 restatement.
Tolerance (stated): |delta| <= 2e-4 * max(1, max|ref|) -- the kernels accumulate up to 73 728 fp32 products per
output in a fixed split-K order."""
import numpy as np
import pytest

from oracle import head_numpy as ref

TOL = 2e-4


def rel(got, want):
    want = np.asarray(want)
    return float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / max(1.0, np.abs(want).max()))


def test_oracle_head_gradients_are_finite_difference_consistent():
    rng = np.random.default_rng(0)
    B, N, C, H, K = 3, 5, 4, 8, 6
    feat = rng.normal(size=(B, N, C))
    W1, b1 = rng.normal(size=(N * C, H)) / np.sqrt(N * C), rng.normal(size=H) * 0.1
    W2, b2 = rng.normal(size=(H, K)) / np.sqrt(H), rng.normal(size=K) * 0.1
    labels = rng.integers(0, K, size=B)
    mask = (rng.random((B, H)) < 0.5).astype(np.float64)
    r = ref.head_forward_backward(feat, W1, b1, W2, b2, labels, 0.5, mask)
    eps = 1e-6
    for name, arr, grad in (("W1", W1, r["dW1"]), ("b2", b2, r["db2"]), ("feat", feat, r["dfeat"])):
        idx = tuple(rng.integers(0, s) for s in arr.shape)
        arr[idx] += eps
        up = ref.head_forward_backward(feat, W1, b1, W2, b2, labels, 0.5, mask)["loss"]
        arr[idx] -= 2 * eps
        dn = ref.head_forward_backward(feat, W1, b1, W2, b2, labels, 0.5, mask)["loss"]
        arr[idx] += eps
        assert abs((up - dn) / (2 * eps) - grad[idx]) < 1e-6, name


@pytest.fixture(scope="module")
def dev():
    import torch
    from pointwise_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,act", [(32, 73728, 512, True), (32, 512, 40, True), (7, 1000, 24, False),
                                       (40, 3001, 520, True), (1, 64, 8, True), (128, 4096, 64, False)])
def test_fc_kernels_match_float64(dev, M, K, N, act):
    import torch
    from pointwise_amd import head
    rng = np.random.default_rng(M + K + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    W = (rng.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32)
    b = (rng.normal(size=N) * 0.1).astype(np.float32)
    dy = rng.normal(size=(M, N)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    y = head.fully_connected(t(x), t(W), t(b), selu=act)
    y_ref = ref.fully_connected(x, W, b, act)
    assert rel(y.cpu().numpy(), y_ref) <= TOL
    dx, dW, db = head.fully_connected_grad(t(x), t(W), y, t(dy), selu=act)
    dx_ref, dW_ref, db_ref = ref.fully_connected_grad(x, W, y.cpu().numpy(), dy, act)
    assert rel(dx.cpu().numpy(), dx_ref) <= TOL and rel(dW.cpu().numpy(), dW_ref) <= TOL
    assert rel(db.cpu().numpy(), db_ref) <= TOL
    y2 = head.fully_connected(t(x), t(W), t(b), selu=act)
    dx2, dW2, _ = head.fully_connected_grad(t(x), t(W), y2, t(dy), selu=act)
    assert torch.equal(y, y2) and torch.equal(dx, dx2) and torch.equal(dW, dW2)      # bitwise reproducible


@pytest.mark.gpu
def test_fc_kernels_match_torch_float64(dev):
    """Second opinion on the checker itself: y = selu(x W + b) and its gradients from torch's float64 autograd on the
    CPU (an implementation neither the kernels nor oracle/head_numpy.py share anything with)."""
    import torch
    from pointwise_amd import head
    M, K, N = 32, 4096, 512
    g = torch.Generator().manual_seed(11)
    x = torch.randn((M, K), generator=g)
    W = torch.randn((K, N), generator=g) / np.sqrt(K)
    b = torch.randn((N,), generator=g) * 0.1
    dy = torch.randn((M, N), generator=g)
    xd, Wd, bd = (v.double().requires_grad_(True) for v in (x, W, b))
    yd = torch.nn.functional.selu(xd @ Wd + bd)
    yd.backward(dy.double())
    y = head.fully_connected(x.to(dev), W.to(dev), b.to(dev), selu=True)
    dx, dW, db = head.fully_connected_grad(x.to(dev), W.to(dev), y, dy.to(dev), selu=True)
    assert rel(y.cpu().numpy(), yd.detach().numpy()) <= TOL
    assert rel(dx.cpu().numpy(), xd.grad.numpy()) <= TOL and rel(dW.cpu().numpy(), Wd.grad.numpy()) <= TOL
    assert rel(db.cpu().numpy(), bd.grad.numpy()) <= TOL


@pytest.mark.gpu
def test_fc_backward_says_unsupported_before_it_launches(dev):
    """32 < M <= 64 with N = 1024: the grad-weight kernel's LDS tile does not fit; the call must say so (status
    UNSUPPORTED) and leave the outputs untouched, not fail inside a launch after its first kernel ran."""
    import ctypes
    import torch
    from pointwise_amd import _lib
    lib = _lib.load()
    M, K, N = 64, 256, 1024
    x = torch.randn((M, K), device=dev)
    W = torch.randn((K, N), device=dev)
    y = torch.randn((M, N), device=dev)
    dy = torch.randn((M, N), device=dev)
    dW = torch.full((K, N), 7.0, device=dev)
    db = torch.full((N,), 7.0, device=dev)
    ws_bytes = lib.conv3p_fc_workspace_bytes(M, K, N)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
    rc = lib.conv3p_fc_backward_f32(x.data_ptr(), W.data_ptr(), y.data_ptr(), dy.data_ptr(), M, K, N, 1, None, dW.data_ptr(),
                                    db.data_ptr(), ws.data_ptr(), ws.numel(), None)
    torch.cuda.synchronize(dev)
    assert rc == _lib.ERR_UNSUPPORTED
    assert float(dW.min()) == 7.0 and float(db.min()) == 7.0          # nothing ran
    # the same M with a narrower layer is served
    N2 = 512
    dW2 = torch.empty((K, N2), device=dev)
    db2 = torch.empty((N2,), device=dev)
    ws2 = torch.empty(max(lib.conv3p_fc_workspace_bytes(M, K, N2), 256), dtype=torch.uint8, device=dev)
    rc = lib.conv3p_fc_backward_f32(x.data_ptr(), W[:, :N2].contiguous().data_ptr(), y[:, :N2].contiguous().data_ptr(),
                                    dy[:, :N2].contiguous().data_ptr(), M, K, N2, 0, None, dW2.data_ptr(), db2.data_ptr(),
                                    ws2.data_ptr(), ws2.numel(), None)
    torch.cuda.synchronize(dev)
    assert rc == _lib.OK
    ref_dW = x.double().T @ dy[:, :N2].double()
    assert rel(dW2.cpu().numpy(), ref_dW.cpu().numpy()) <= TOL


@pytest.mark.gpu
def test_classification_head_end_to_end(dev):
    """The model's head at ModelNet size (N = 2048 -> K = 73 728, 512 hidden, 40 classes) on the conv3p stack's
    output: logits, loss and every gradient against the float64 restatement, with the dropout mask made explicit."""
    import torch
    from pointwise_amd import head, stack, synth
    B, N = 32, 2048
    P = torch.from_numpy(synth.modelnet_like(B, N, seed=1400)).to(dev)
    st = stack.Conv3pStack(3, None, device=dev, seed=1401)
    acts = st.forward(P, P)
    feat = torch.cat([a for a in acts], dim=2).contiguous()                        # (B, N, 36)
    hd = head.ClassificationHead(N, num_class=40, device=dev, seed=5)
    rng = np.random.default_rng(3)
    mask = (rng.random((B, 512)) < 0.5).astype(np.float32)
    labels = rng.integers(0, 40, size=B)
    logits = hd.forward(feat, training=True, keep_mask=torch.from_numpy(mask).to(dev))
    loss, dlogits = hd.loss(logits, torch.from_numpy(labels).to(dev))
    dfeat = hd.backward(dlogits)
    r = ref.head_forward_backward(feat.cpu().numpy(), hd.W1.cpu().numpy(), hd.b1.cpu().numpy(), hd.W2.cpu().numpy(),
                                  hd.b2.cpu().numpy(), labels, 0.5, mask.astype(np.float64))
    assert rel(logits.cpu().numpy(), r["logits"]) <= TOL
    assert abs(float(loss) - r["loss"]) <= TOL * max(1.0, abs(r["loss"]))
    assert rel(dfeat.cpu().numpy(), r["dfeat"]) <= TOL
    assert rel(hd.dW1.cpu().numpy(), r["dW1"]) <= TOL and rel(hd.db1.cpu().numpy(), r["db1"]) <= TOL
    assert rel(hd.dW2.cpu().numpy(), r["dW2"]) <= TOL and rel(hd.db2.cpu().numpy(), r["db2"]) <= TOL
    # the gradient flows on into the conv3p stack: dL/dfeat is the stack's upstream gradient
    dx, fused = st.backward(dfeat)
    assert bool(torch.isfinite(dx).all()) and bool(torch.isfinite(fused).all()) and float(fused.abs().max()) > 0
