"""One training step of the reference's classification model (pointcnn2_acsd.py:33-90) assembled from this
repository's pieces -- provider pre-step (modelnet_provider.py:196-198), the four-layer conv3p stack with its concat,
the dense head, softmax cross-entropy -- forward and backward, against the CPU restatements composed the same way
(oracle conv3p + numpy head + numpy pre-step).  What is checked beyond the per-piece tests: the pieces agree on the
layout of the (B, N, 36) concat / its (B, N*36) view, and the gradient that leaves the head is the one the stack
expects.  Small size (the oracle runs in a second); tolerance of the head test."""
import numpy as np
import pytest

from oracle import head_numpy, oracle, prestep_numpy

TOL = 2e-4
VOX = 0.1


def rel(got, want):
    want = np.asarray(want)
    return float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / max(1.0, np.abs(want).max()))


@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from pointwise_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_classification_model_training_step(dev):
    import torch
    from pointwise_amd import head, prestep, stack, synth
    B, N, NCLS = 3, 192, 40
    rng = np.random.default_rng(21)
    raw = synth.modelnet_like(B, N, seed=1500)
    angles = rng.uniform(0, 2 * np.pi, size=B)
    noise = rng.standard_normal((B, N, 3))
    labels = rng.integers(0, NCLS, size=B)
    mask = (rng.random((B, 512)) < 0.5).astype(np.float32)

    # ---- device: pre-step -> stack -> head -> loss -> head backward -> stack backward
    P = prestep.rotate_and_jitter(torch.from_numpy(raw).to(dev), angles=angles, noise=torch.from_numpy(noise).to(dev))
    st = stack.Conv3pStack(3, None, device=dev, seed=1501)
    acts = st.forward(P, P)                                         # features == points (modelnet_provider.py:212-213)
    feat = torch.cat(list(acts), dim=2).contiguous()               # (B, N, 36)
    hd = head.ClassificationHead(N, num_class=NCLS, device=dev, seed=7)
    logits = hd.forward(feat, training=True, keep_mask=torch.from_numpy(mask).to(dev))
    loss, dlogits = hd.loss(logits, torch.from_numpy(labels).to(dev))
    dfeat = hd.backward(dlogits)
    dx, fused = st.backward(dfeat)

    # ---- CPU: the same graph from the restatements
    Pn = prestep_numpy.jitter_point_cloud(prestep_numpy.rotate_point_cloud_by_angles(raw, angles), noise)
    assert rel(P.cpu().numpy(), Pn) <= 1e-6
    Pn = P.cpu().numpy()                                            # identical geometry for both sides from here on
    filters = [f.cpu().numpy() for f in st.filters]
    x, ref_acts = Pn, []
    for li in range(4):
        s = st.layers[li][2]
        x = stack.selu_numpy(oracle.forward(Pn, x, filters[li], (s, s, s), VOX))
        ref_acts.append(x)
    concat = np.concatenate(ref_acts, axis=2)
    r = head_numpy.head_forward_backward(concat, hd.W1.cpu().numpy(), hd.b1.cpu().numpy(), hd.W2.cpu().numpy(),
                                         hd.b2.cpu().numpy(), labels, 0.5, mask.astype(np.float64))
    g = np.asarray(r["dfeat"], dtype=np.float32).reshape(B, N, 36)
    carry, dws = None, [None] * 4
    for li in (3, 2, 1, 0):
        s = st.layers[li][2]
        up = g[:, :, 9 * li:9 * li + 9]
        gi = stack.selu_grad_numpy(ref_acts[li], np.ascontiguousarray(up if carry is None else up + carry))
        carry, dws[li] = oracle.backward(gi, Pn, ref_acts[li - 1] if li > 0 else Pn, filters[li], (s, s, s), VOX)

    assert rel(feat.cpu().numpy(), concat) <= 2e-5
    assert rel(logits.cpu().numpy(), r["logits"]) <= TOL
    assert abs(float(loss) - r["loss"]) <= TOL * max(1.0, abs(r["loss"]))
    assert rel(dfeat.cpu().numpy().reshape(B, N, 36), g) <= TOL
    assert rel(hd.dW1.cpu().numpy(), r["dW1"]) <= TOL and rel(hd.dW2.cpu().numpy(), r["dW2"]) <= TOL
    assert rel(dx.cpu().numpy(), carry) <= TOL
    assert rel(fused.cpu().numpy(), np.concatenate([d.reshape(-1) for d in dws])) <= TOL


@pytest.mark.gpu
def test_bench_reducer_device_path_without_a_process_group(dev):
    """bench.Reducer as the N > 1 legs drive it, on HIP streams (the gloo tests cover CPU tensors only): the collective
    on its own stream behind the backward, the next backward waiting for it, HIP-event timing of the collective and of
    the main stream's wait.  No process group here, so the collective itself is a no-op -- stream / event logic only."""
    import torch
    import bench
    red = bench.Reducer(dev, 2)                    # world 2: takes the communication-stream path
    assert red.stream is not None
    buf = torch.zeros(7290, device=dev)
    red.timing = True
    for step in range(4):
        red.wait_previous()
        buf.add_(1.0)                              # "the backward writes the fused buffer"
        red.launch(buf)
    red.finish()
    torch.cuda.synchronize(dev)
    assert float(buf[0]) == 4.0
    assert len(red.pairs) == 4 and len(red.waits) == 3
    assert red.ms_per_step(4) >= 0.0 and red.exposed_ms_per_step(4) >= 0.0
    assert bench.rank_spread(0.012, 4, dev) == {"min": 3.0, "max": 3.0}
