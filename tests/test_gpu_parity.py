"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the golden
fixtures.  Bars: per-tap neighbour counts EXACT (integer equality); y / dX within 1e-5 * max(1, max|ref|),
dW within 2e-5 * max(1, max|dW_ref|) in fp32; 1e-12 in fp64 (tests/parity_util.TOL)."""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from pointwise_amd import _lib, conv3p_op as op, stack, synth
from tests.parity_util import TOL, exact_from_oracle_lists, make_case, rel_err

pytestmark = pytest.mark.gpu
VOX = 0.1
GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith("prestep_"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.load()                       # fails loudly if the native library is missing
    return torch.device("cuda:0")


def run_hip(dev, P, X, W, dY, s, vox=VOX):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp, tx, tw, tdy = t(P), t(X), t(W), t(dY)
    cnt = op.neighbor_count(tp, W.shape[:3], s, vox)
    y = op.conv3p(tp, tx, tw, s, vox)
    dx, dw = op.conv3p_grad(tdy, tp, tx, tw, s, vox)
    torch.cuda.synchronize()
    return cnt.cpu().numpy(), y.cpu().numpy(), dx.cpu().numpy(), dw.cpu().numpy()


def check_against(ref, got, dtype, dw_floor=0.0):
    """dw_floor: the reference's OWN fp32 rounding error on grad_filter for this input (f32 oracle vs f64
    oracle).  grad_filter sums B*N*K terms into one fp32 accumulator, so for degenerate dense inputs (all
    points identical: N^2 terms) the reference itself is further than 2e-5 from the exact value; the bar is
    then 4x its own error.  For every realistic case dw_floor is far below the stated tolerance and unused."""
    tol_y, tol_w = TOL[np.dtype(dtype)]
    tol_w = max(tol_w, 4.0 * dw_floor)
    cnt_ref, y_ref, dx_ref, dw_ref = ref
    cnt, y, dx, dw = got
    assert np.array_equal(cnt, cnt_ref), "neighbour / tap decisions differ from the CPU reference"
    assert rel_err(y, y_ref) <= tol_y, ("y", rel_err(y, y_ref))
    assert rel_err(dx, dx_ref) <= tol_y, ("dX", rel_err(dx, dx_ref))
    assert rel_err(dw, dw_ref) <= tol_w, ("dW", rel_err(dw, dw_ref))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixtures(dev, path):
    g = np.load(path)
    P, X, W, dY = g["points"], g["input"], g["filter"], g["grad_out"]
    s = tuple(int(v) for v in g["stride"])
    got = run_hip(dev, P, X, W, dY, s, float(g["voxel"]))
    check_against((g["ref_count"].astype(np.int32), g["y"], g["dX"], g["dW"]), got, P.dtype)


CASES = [
    # kind, B, N, Cin, Cout, filter zyx, stride xyz, dtype
    ("modelnet", 4, 2048, 3, 9, (3, 3, 3), (1, 1, 1), np.float32),
    ("modelnet", 4, 2048, 9, 9, (3, 3, 3), (2, 2, 2), np.float32),
    ("modelnet", 4, 2048, 9, 9, (3, 3, 3), (3, 3, 3), np.float32),
    ("modelnet", 4, 2048, 9, 9, (3, 3, 3), (4, 4, 4), np.float32),
    ("room", 2, 4096, 9, 9, (3, 3, 3), (1, 1, 1), np.float32),
    ("room", 2, 4096, 36, 13, (3, 3, 3), (1, 1, 1), np.float32),
    ("room", 1, 127, 36, 13, (3, 5, 3), (4, 1, 4), np.float64),   # 45 taps in fp64: not even the column blocks fit LDS (generic kernels)
    ("room", 1, 1024, 12, 9, (3, 3, 3), (2, 2, 2), np.float32),
    ("lattice", 2, 1024, 9, 9, (3, 3, 3), (2, 2, 2), np.float32),
    ("lattice", 2, 1024, 3, 9, (3, 3, 3), (1, 1, 1), np.float32),
    ("lattice", 1, 777, 9, 9, (3, 3, 3), (3, 3, 3), np.float64),
    ("cube", 3, 1000, 5, 7, (3, 3, 3), (1, 1, 1), np.float32),       # generic channel path
    ("cube", 2, 500, 4, 6, (2, 1, 3), (1, 2, 3), np.float32),        # anisotropic filter + stride
    ("cube", 2, 500, 3, 2, (2, 2, 2), (1, 1, 1), np.float32),        # even extent
    ("cube", 2, 500, 2, 2, (4, 4, 4), (2, 2, 2), np.float32),        # even extent, self point is a hole
    ("cube", 1, 400, 2, 3, (5, 5, 5), (1, 1, 1), np.float32),
    ("cube", 1, 300, 3, 4, (1, 1, 1), (1, 1, 1), np.float32),
    ("modelnet", 2, 512, 3, 9, (3, 3, 3), (1, 1, 1), np.float64),
    ("modelnet", 2, 512, 9, 9, (3, 3, 3), (4, 4, 4), np.float64),
    ("room", 1, 512, 32, 64, (3, 3, 3), (1, 1, 1), np.float32),      # deep channels (generic path today)
    ("room", 1, 256, 128, 256, (3, 3, 3), (1, 1, 1), np.float32),    # cfg5 channel shape, small N
    ("identical", 1, 200, 3, 9, (3, 3, 3), (1, 1, 1), np.float32),   # every point in every box
    ("isolated", 2, 70, 3, 9, (3, 3, 3), (2, 2, 2), np.float32),     # only self pairs
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-B%dN%d-%dto%d-f%s-s%s-%s" % (
    c[0], c[1], c[2], c[3], c[4], "x".join(map(str, c[5])), "x".join(map(str, c[6])), np.dtype(c[7]).name))
def test_hip_matches_oracle(dev, case):
    kind, B, N, ci, co, fzyx, s, dt = case
    P, X, W, dY = make_case(kind, B, N, ci, co, fzyx, seed=200, dtype=dt)
    ref = (oracle.neighbor_count(P, fzyx, s, VOX), oracle.forward(P, X, W, s, VOX)) + \
        oracle.backward(dY, P, X, W, s, VOX)
    floor = 0.0
    if kind == "identical":
        d = np.float64
        dw64 = oracle.backward(dY.astype(d), P.astype(d), X.astype(d), W.astype(d), s, VOX)[1]
        floor = rel_err(ref[3], dw64)
    check_against(ref, run_hip(dev, P, X, W, dY, s), dt, dw_floor=floor)


@pytest.mark.parametrize("fzyx,s,dt", [((2, 2, 2), (1, 1, 1), np.float32), ((4, 4, 4), (1, 1, 1), np.float32),
                                       ((2, 2, 2), (2, 2, 2), np.float32), ((2, 3, 2), (1, 2, 2), np.float64),
                                       ((4, 2, 3), (1, 1, 1), np.float64), ((2, 2, 2), (1, 1, 1), np.float64)])
def test_even_extents_on_voxel_aligned_clouds(dev, fzyx, s, dt):
    """Even dilated extents on voxel-aligned data: the reference's +-n cell window (.cpp:247-266) rejects some
    candidates that pass the inclusive box test.  Decisions must still be integer-exact -- through the pair lists
    and through the kernels' own search (pair buffer overflow fallback)."""
    B, N, ci, co = 2, 600, 3, 9
    P, X, W, dY = make_case("vlattice", B, N, ci, co, fzyx, seed=1200, dtype=dt)
    ref = (oracle.neighbor_count(P, fzyx, s, VOX), oracle.forward(P, X, W, s, VOX)) + oracle.backward(dY, P, X, W, s, VOX)
    check_against(ref, run_hip(dev, P, X, W, dY, s), dt)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    cache = op.NeighborCache(B, N, tdt, dev, slots=1, max_taps=int(np.prod(fzyx)), pairs_per_point=1, max_cin=ci,
                             max_cout=co)
    y, dx, dw = _both(dev, cache, P, X, W, dY, s)
    tol_y, tol_w = TOL[np.dtype(dt)]
    assert rel_err(y.cpu().numpy(), ref[1]) <= tol_y and rel_err(dx.cpu().numpy(), ref[2]) <= tol_y
    assert rel_err(dw.cpu().numpy(), ref[3]) <= tol_w


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 127, 129, 200])
def test_ragged_point_counts(dev, N):
    """Tiles are 64 points: sizes around the tile boundary, including a single point."""
    P, X, W, dY = make_case("modelnet", 3, N, 3, 9, seed=300 + N)
    s = (1, 1, 1)
    ref = (oracle.neighbor_count(P, (3, 3, 3), s, VOX), oracle.forward(P, X, W, s, VOX)) + \
        oracle.backward(dY, P, X, W, s, VOX)
    check_against(ref, run_hip(dev, P, X, W, dY, s), np.float32)


def test_empty_batches_and_clouds(dev):
    W = torch.from_numpy(synth.filter_weights(3, 3, 3, 3, 9, 1)).to(dev)
    for shape in [(0, 16), (2, 0)]:
        B, N = shape
        p = torch.zeros((B, N, 3), device=dev)
        y = op.conv3p(p, p.clone(), W, (1, 1, 1), VOX)
        assert tuple(y.shape) == (B, N, 9)
        dx, dw = op.conv3p_grad(torch.zeros((B, N, 9), device=dev), p, p.clone(), W, (1, 1, 1), VOX)
        assert tuple(dx.shape) == (B, N, 3) and float(dw.abs().max()) == 0.0      # zeroed like .cpp:590


def test_outputs_are_overwritten_not_accumulated(dev):
    """The op zero-initialises its outputs (.cpp:451, :580, :590): garbage in the buffers must not leak."""
    P, X, W, dY = make_case("cube", 2, 300, 5, 7, seed=7)          # generic path accumulates in global memory
    t = lambda a: torch.from_numpy(a).to(dev)
    for _ in range(2):
        y = op.conv3p(t(P), t(X), t(W), (1, 1, 1), VOX)
        dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), (1, 1, 1), VOX)
    assert rel_err(y.cpu().numpy(), oracle.forward(P, X, W, (1, 1, 1), VOX)) < 1e-5
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, (1, 1, 1), VOX)
    assert rel_err(dx.cpu().numpy(), dx_ref) < 1e-5 and rel_err(dw.cpu().numpy(), dw_ref) < 2e-5


def test_c_abi_status_codes(dev):
    lib = _lib.load()
    P, X, W, dY = make_case("cube", 1, 128, 3, 9, seed=1)
    tp, tx, tw = [torch.from_numpy(a).to(dev) for a in (P, X, W)]
    out = torch.empty((1, 128, 9), device=dev)
    s3 = (ctypes.c_int32 * 3)(1, 1, 1)
    need = lib.conv3p_workspace_bytes(_lib.PASS_FORWARD, 4, 1, 128, 3, 9, 3, 3, 3)
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    base = ws.data_ptr() + (-ws.data_ptr() % 256)
    call = lambda wptr, wbytes: lib.conv3p_forward_f32(
        tp.data_ptr(), tx.data_ptr(), tw.data_ptr(), ctypes.cast(s3, ctypes.c_void_p), ctypes.c_float(0.1),
        1, 128, 3, 9, 3, 3, 3, out.data_ptr(), wptr, wbytes, torch.cuda.current_stream().cuda_stream)
    assert call(base, need - 1) == _lib.ERR_WORKSPACE
    assert call(None, need) == _lib.ERR_WORKSPACE
    assert call(base + 8, need) == _lib.ERR_WORKSPACE             # misaligned
    assert call(base, need) == _lib.OK
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), oracle.forward(P, X, W, (1, 1, 1), VOX)) < 1e-5


def test_autograd_wiring(dev):
    """Gradients come back as [None, input_grad, filter_grad, None, None] (pointcnn2_acsd.py:31)."""
    P, X, W, dY = make_case("modelnet", 2, 256, 9, 9, seed=5)
    tp = torch.from_numpy(P).to(dev)
    tx = torch.from_numpy(X).to(dev).requires_grad_(True)
    tw = torch.from_numpy(W).to(dev).requires_grad_(True)
    y = op.conv3p_autograd(tp, tx, tw, (2, 2, 2), VOX)
    y.backward(torch.from_numpy(dY).to(dev))
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, (2, 2, 2), VOX)
    assert rel_err(tx.grad.cpu().numpy(), dx_ref) < 1e-5 and rel_err(tw.grad.cpu().numpy(), dw_ref) < 2e-5
    assert tp.grad is None


# ------------------------------------------------------------------ full-size property tests (BASELINE sizes)
@pytest.fixture(scope="module")
def cfg2(dev):
    P = synth.modelnet_like(32, 2048, seed=1236)
    W = synth.filter_weights(3, 3, 3, 9, 9, 3)
    X = synth.features(32, 2048, 9, 4, points=P)
    dY = synth.upstream_grad(32, 2048, 9, 5)
    return [torch.from_numpy(a).to(dev) for a in (P, X, W, dY)]


def test_full_size_counts_self_and_symmetry(dev, cfg2):
    """cfg2 size: centre tap always holds the point itself; total pairs are symmetric for stride 1."""
    tp = cfg2[0]
    cnt = op.neighbor_count(tp, (3, 3, 3), (1, 1, 1), VOX)
    assert int(cnt[:, :, 13].min()) >= 1
    # sub-sample check against the oracle (4 clouds)
    ref = oracle.neighbor_count(tp[:4].cpu().numpy(), (3, 3, 3), (1, 1, 1), VOX)
    assert np.array_equal(cnt[:4].cpu().numpy(), ref)


@pytest.mark.parametrize("stride", [1, 2, 4])
def test_full_size_linearity_and_adjoint(dev, cfg2, stride):
    """Size-independent properties at B=32, N=2048: linearity in input, batch independence, permutation
    equivariance, and <dY, conv(X)> = <dX, X> = <dW, W> (bilinearity + adjointness on generic data)."""
    tp, tx, tw, tdy = cfg2
    s = (stride,) * 3
    y = op.conv3p(tp, tx, tw, s, VOX)
    x2 = torch.randn_like(tx)
    lin = op.conv3p(tp, 2.0 * tx + x2, tw, s, VOX) - (2.0 * y + op.conv3p(tp, x2, tw, s, VOX))
    assert float(lin.abs().max()) <= 2e-5 * max(1.0, float(y.abs().max()))
    # batch independence: cloud 5 alone
    y5 = op.conv3p(tp[5:6].contiguous(), tx[5:6].contiguous(), tw, s, VOX)
    assert torch.equal(y5[0], y[5])
    # permutation equivariance within clouds
    perm = torch.randperm(2048, device=dev)
    yp = op.conv3p(tp[:, perm].contiguous(), tx[:, perm].contiguous(), tw, s, VOX)
    assert float((yp - y[:, perm]).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))
    # adjoint identities, float64 accumulation.  The reference backward is the exact adjoint of its forward
    # only when no pair sits on a tap boundary; among ~1e6 fp32 pairs a handful do (each worth ~0.05 here),
    # so this is a 2e-3 sanity bound -- the sharp full-size check is test_full_size_matches_oracle_cfg2_layer.
    dx, dw = op.conv3p_grad(tdy, tp, tx, tw, s, VOX)
    lhs = float((tdy.double() * y.double()).sum())
    assert abs(lhs - float((dx.double() * tx.double()).sum())) <= 2e-3 * max(1.0, abs(lhs))
    assert abs(lhs - float((dw.double() * tw.double()).sum())) <= 2e-3 * max(1.0, abs(lhs))


def test_full_size_matches_oracle_cfg2_layer(dev, cfg2):
    """One full cfg2 layer (B=32, N=2048, 9->9 stride 2) against the multi-threaded oracle."""
    tp, tx, tw, tdy = cfg2
    s = (2, 2, 2)
    P, X, W, dY = [t.cpu().numpy() for t in cfg2]
    nthr = min(32, os.cpu_count() or 1)
    y_ref = oracle.forward(P, X, W, s, VOX, nthreads=nthr)
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, VOX, nthreads=nthr)
    y = op.conv3p(tp, tx, tw, s, VOX)
    dx, dw = op.conv3p_grad(tdy, tp, tx, tw, s, VOX)
    assert rel_err(y.cpu().numpy(), y_ref) <= 1e-5
    assert rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5
    assert rel_err(dw.cpu().numpy(), dw_ref) <= 2e-5


# ------------------------------------------------------------------ BASELINE configs 4 and 5 at full size
@pytest.fixture(scope="module")
def cfg5(dev):
    """BASELINE config 5, one GPU's shard at FULL size: B=16 clouds of N=8192 SceneNN-shaped points, one conv3p layer
    128->256, stride 1 (matrix-core path): inputs (numpy + device) and the op's results on the whole shard."""
    B, N, ci, co = 16, 8192, 128, 256
    s = (1, 1, 1)
    P = synth.room_like(B, N, 7, extent=(2.4, 2.4, 3.0))
    X = synth.features(B, N, ci, 8, points=P)
    W = synth.filter_weights(3, 3, 3, ci, co, 5)
    dY = synth.upstream_grad(B, N, co, 9)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp, tx, tw, tdy = t(P), t(X), t(W), t(dY)
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
    y = op.conv3p(tp, tx, tw, s, VOX, cache=cache)
    dx, dw = op.conv3p_grad(tdy, tp, tx, tw, s, VOX, cache=cache)
    return dict(P=P, X=X, W=W, dY=dY, tp=tp, tx=tx, tw=tw, tdy=tdy, cache=cache, y=y, dx=dx, dw=dw, s=s)


def test_full_size_cfg5_every_channel_of_two_clouds(dev, cfg5):
    """Every one of the 256 output channels of y, the 128 input channels of dX and the 27 x 128 x 256 entries of dW, of
    two whole clouds of the cfg5 shard, against the oracle.  The reference's loops make every output channel an
    independent sum -- y over c, dX over k, dW over (k, c) -- so a slice of the filter (and of X for dW) gives exactly
    those channels: 16 forward calls on 16-channel slices of the output and 16 backward calls on 8-channel slices of the
    input per cloud cover everything; the calls (single-threaded C, the GIL released) run side by side on the host's
    cores.  dW is compared per cloud (the op called on that cloud alone): the sharp form of the one cross-cloud reduction.

    What it is compared WITH: at this depth (sums of ~10 000 terms per output) the reference's own fp32 loops are
    8e-6 / 5e-6 / 8e-6 (y / dX / dW, relative to the tensor's maximum) away from the exact sums -- measured,
    profiles/r05_deep_accuracy.txt -- which leaves no room under a 1e-5 bound against THEM (the matrix-core path itself
    is 2e-6 / 2e-6 / 2e-7 away).  The op's tolerance is therefore asserted against the exact sums over the oracle's own
    pair lists (parity_util.exact_from_oracle_lists: the reference's single-precision decisions, float64 accumulation),
    together with: no further from them than the reference's own single-precision loops are."""
    from concurrent.futures import ThreadPoolExecutor
    P, X, W, dY, s = cfg5["P"], cfg5["X"], cfg5["W"], cfg5["dY"], cfg5["s"]
    clouds = (3, 11)
    y_gpu = cfg5["y"].cpu().numpy()
    dx_gpu = cfg5["dx"].cpu().numpy()
    dw_gpu = {}
    for b in clouds:   # grad_filter of cloud b alone
        _, dwb = op.conv3p_grad(cfg5["tdy"][b:b + 1].contiguous(), cfg5["tp"][b:b + 1].contiguous(), cfg5["tx"][b:b + 1].contiguous(),
                                cfg5["tw"], s, VOX)
        dw_gpu[b] = dwb.cpu().numpy()

    def fwd_task(b, c0):
        cs = slice(c0, c0 + 16)
        return ("y", b, cs, oracle.forward(P[b:b + 1], X[b:b + 1], np.ascontiguousarray(W[..., cs]), s, VOX))

    def bwd_task(b, k0):
        ks = slice(k0, k0 + 8)
        dx_ref, dw_ref = oracle.backward(dY[b:b + 1], P[b:b + 1], np.ascontiguousarray(X[b:b + 1, :, ks]),
                                         np.ascontiguousarray(W[:, :, :, ks, :]), s, VOX)
        return ("d", b, ks, dx_ref, dw_ref)

    tasks = [(bwd_task, b, k0) for b in clouds for k0 in range(0, 128, 8)] + [(fwd_task, b, c0) for b in clouds for c0 in range(0, 256, 16)]
    single = dict(y=np.zeros((len(clouds),) + y_gpu.shape[1:], np.float32), dx=np.zeros((len(clouds),) + dx_gpu.shape[1:], np.float32),
                  dw={b: np.zeros(W.shape, np.float32) for b in clouds})
    with ThreadPoolExecutor(max_workers=max(1, min(64, os.cpu_count() or 1))) as ex:
        fut = ex.map(lambda a: a[0](a[1], a[2]), tasks)
        exact = {b: exact_from_oracle_lists(P[b], X[b], W, dY[b], s, VOX) for b in clouds}   # (BLAS, beside the C loops)
        for r in fut:
            i = clouds.index(r[1])
            if r[0] == "y":
                single["y"][i, :, r[2]] = r[3][0]
            else:
                single["dx"][i, :, r[2]] = r[3][0]
                single["dw"][r[1]][:, :, :, r[2], :] = r[4]
    for i, b in enumerate(clouds):
        ye, dxe, dwe = exact[b]
        for name, got, ref32, ref64, tol in (("y", y_gpu[b], single["y"][i], ye, 1e-5), ("dx", dx_gpu[b], single["dx"][i], dxe, 1e-5),
                                             ("dw", dw_gpu[b], single["dw"][b], dwe, 2e-5)):
            e_hip, e_ref = rel_err(got, ref64), rel_err(ref32, ref64)
            assert e_hip <= tol and e_hip <= e_ref, (name, b, e_hip, e_ref)


def test_full_size_cfg5_shard(dev, cfg5):
    """The whole cfg5 shard: neighbour counts exact vs the oracle on 2 clouds; batch independence, linearity and the
    adjoint identities over all 16 clouds (test_full_size_cfg5_every_channel_of_two_clouds holds the oracle comparison
    of every channel); one more cloud on channel slices."""
    P, X, W, dY, s = cfg5["P"], cfg5["X"], cfg5["W"], cfg5["dY"], cfg5["s"]
    tp, tx, tw, tdy, cache = cfg5["tp"], cfg5["tx"], cfg5["tw"], cfg5["tdy"], cfg5["cache"]
    y, dx, dw = cfg5["y"], cfg5["dx"], cfg5["dw"]
    cnt = op.neighbor_count(tp, (3, 3, 3), s, VOX)
    assert np.array_equal(cnt[:2].cpu().numpy(), oracle.neighbor_count(P[:2], (3, 3, 3), s, VOX))
    b = 7
    # a third cloud, every channel, against the exact sums over the oracle's pair lists (see the test above)
    ye, dxe, dwe = exact_from_oracle_lists(P[b], X[b], W, dY[b], s, VOX)
    assert rel_err(y[b].cpu().numpy(), ye) <= 1e-5
    assert rel_err(dx[b].cpu().numpy(), dxe) <= 1e-5
    # grad_filter of cloud b alone (the op called on that cloud)
    _, dw_b = op.conv3p_grad(tdy[b:b + 1].contiguous(), tp[b:b + 1].contiguous(), tx[b:b + 1].contiguous(), tw, s, VOX)
    assert rel_err(dw_b.cpu().numpy(), dwe) <= 2e-5
    # batch independence: cloud b inside the shard == cloud b alone; grad_filter of the shard == sum over clouds
    y_b = op.conv3p(tp[b:b + 1].contiguous(), tx[b:b + 1].contiguous(), tw, s, VOX)
    assert float((y_b[0] - y[b]).abs().max()) <= 1e-6 * max(1.0, float(y.abs().max()))
    # linearity in the input on the full shard
    x2 = torch.randn_like(tx)
    lin = op.conv3p(tp, 2.0 * tx + x2, tw, s, VOX, cache=cache) - (2.0 * y + op.conv3p(tp, x2, tw, s, VOX, cache=cache))
    assert float(lin.abs().max()) <= 5e-5 * max(1.0, float(y.abs().max()))
    # <dY, conv(X)> = <dW, W> = <dX, X> on generic data (float64 accumulation; 2e-3: a few pairs sit on tap edges)
    lhs = float((tdy.double() * y.double()).sum())
    assert abs(lhs - float((dw.double() * tw.double()).sum())) <= 2e-3 * max(1.0, abs(lhs))
    assert abs(lhs - float((dx.double() * tx.double()).sum())) <= 2e-3 * max(1.0, abs(lhs))


def test_full_size_cfg4_stack(dev):
    """BASELINE config 4 at FULL size: the 5-layer S3DIS scene_seg stack (9->9 s1..s4, 36->13 s1) on B=16 room
    blocks of N=4096.  ALL 16 clouds against the oracle stack (OpenMP over the batch): every activation, grad_input and
    the fused grad_filter of the whole batch per layer (the one cross-cloud reduction: tile schedule order, 1 024
    partials).  The first two clouds run as a batch of 2 must come out of the full batch bit-for-bit (activations and
    grad_input; the weight gradient is a sum over the batch)."""
    B, N, cin, ncls = 16, 4096, 9, 13
    P = synth.room_like(B, N, 40)
    X = synth.features(B, N, cin, 50, points=P)
    up = synth.upstream_grad(B, N, ncls, 60)
    st = stack.Conv3pStack(cin, ncls, device=dev, seed=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    acts = st.forward(t(P), t(X))
    dx, fused = st.backward([t(up)])
    acts = [a.clone() for a in acts]
    dx, fused = dx.clone(), fused.clone()
    st2 = stack.Conv3pStack(cin, ncls, device=dev, seed=3)
    acts2 = st2.forward(t(P[:2]), t(X[:2]))
    dx2, fused2 = st2.backward([t(up[:2])])
    for a, a2 in zip(acts, acts2):
        assert torch.equal(a[:2], a2)
    assert torch.equal(dx[:2], dx2)
    nthr = min(32, os.cpu_count() or 1)
    ref_acts, ref_dx, ref_fused = _oracle_stack(P, X, [f.cpu().numpy() for f in st.filters], st.layers, [up], ncls,
                                                nthreads=nthr)
    for a, r in zip(acts, ref_acts):
        assert rel_err(a.cpu().numpy(), r) <= 2e-5
    assert rel_err(dx.cpu().numpy(), ref_dx) <= 5e-5           # five chained layers
    got = fused.cpu().numpy()
    o = 0
    for f in st.filters:                                       # per layer: each has its own scale
        n = f.numel()
        assert rel_err(got[o:o + n], ref_fused[o:o + n]) <= 2e-5
        o += n


@pytest.mark.parametrize("ci,co,N", [(12, 9, 4096), (36, 41, 2048), (16, 9, 1024)])
def test_scenenn_model_shapes(dev, ci, co, N):
    """SceneNN scene_seg shapes (scene_seg/train_scene_seg_scenenn.py:42: 41 classes; scenenn_provider.py:47-59:
    >= 12 input channels): 12->9 first layer and the 36->41 head, against the oracle, bitwise reproducible."""
    B = 2
    P, X, W, dY = make_case("room", B, N, ci, co, seed=1300)
    s = (1, 1, 1)
    ref = (oracle.neighbor_count(P, (3, 3, 3), s, VOX), oracle.forward(P, X, W, s, VOX, nthreads=2)) + \
        oracle.backward(dY, P, X, W, s, VOX, nthreads=1)
    got = run_hip(dev, P, X, W, dY, s)
    check_against(ref, got, np.float32)
    again = run_hip(dev, P, X, W, dY, s)
    for a, b in zip(got, again):
        assert np.array_equal(a, b), "model shapes must be bitwise reproducible"


# ------------------------------------------------------------------ the models' layer stacks (row A8)
def _oracle_stack(P, X, filters, layers, ups, num_class, nthreads=1):
    fwd = lambda *a: oracle.forward(*a, nthreads=nthreads) if nthreads > 1 else oracle.forward(*a)
    acts, x = [], X
    for li in range(4):
        s = layers[li][2]
        x = stack.selu_numpy(fwd(P, x, filters[li], (s, s, s), VOX))
        acts.append(x)
    dws = [None] * len(layers)
    if num_class is not None:
        concat = np.concatenate(acts, axis=2)
        logits = stack.selu_numpy(fwd(P, concat, filters[4], (1, 1, 1), VOX))
        acts.append(logits)
        g = stack.selu_grad_numpy(logits, ups[0])
        dconcat, dws[4] = oracle.backward(g, P, concat, filters[4], (1, 1, 1), VOX, **({"nthreads": nthreads} if nthreads > 1 else {}))
        ext = [np.ascontiguousarray(dconcat[:, :, 9 * i:9 * i + 9]) for i in range(4)]
    else:
        ext = ups
    carry = None
    for li in (3, 2, 1, 0):
        s = layers[li][2]
        g = stack.selu_grad_numpy(acts[li], ext[li] if carry is None else ext[li] + carry)
        kw = {"nthreads": nthreads} if nthreads > 1 else {}
        carry, dws[li] = oracle.backward(g, P, acts[li - 1] if li > 0 else X, filters[li], (s, s, s), VOX, **kw)
    return acts, carry, np.concatenate([d.reshape(-1) for d in dws])


_REF_MEMO = {}


@pytest.mark.parametrize("tuned,prefetch", [(False, False), (True, False), (True, True), (False, True)])
def test_full_size_cfg2_stack_all_layers(dev, tuned, prefetch):
    """BASELINE config 2 at FULL size, the whole thing: B=32 clouds of N=2048, all four layers' activations, the
    stack's grad_input and the fused grad_filter of all layers against the oracle stack (OpenMP over the batch).
    grad_filter sums 65 536 points x ~10-18 pairs: its tolerance is the stated 2e-5 of max|dW| per layer.
    tuned + prefetch is EXACTLY what bench.py times: Conv3pStack.tune() (populated-rows backward for the dilated
    layers), this batch's geometry built by prefetch() on the side stream while another batch is between its forward
    and its backward, conv3p_stack_* entry points."""
    B, N = 32, 2048
    P = synth.modelnet_like(B, N, seed=1236 + 7)
    ups = [synth.upstream_grad(B, N, stack.HIDDEN, 77 + li) for li in range(4)]
    st = stack.Conv3pStack(3, None, device=dev, seed=1234)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp = t(P)
    if tuned:
        st.tune(tp)
        # (an untuned stack decides per launch from the lists' own statistics; tune() is the explicit override)
    if prefetch:
        other = t(synth.modelnet_like(B, N, seed=99))
        st.forward(other, other)                 # a different batch is in flight ...
        st.prefetch(tp)                          # ... while this batch's geometry is built on the side stream
        st.backward([t(u) for u in ups])
    acts = st.forward(tp, tp)
    dx, fused = st.backward([t(u) for u in ups])
    if "cfg2" not in _REF_MEMO:                  # the oracle stack is the same for every parametrisation
        nthr = min(32, os.cpu_count() or 1)
        _REF_MEMO["cfg2"] = _oracle_stack(P, P.copy(), [f.cpu().numpy() for f in st.filters], st.layers, ups, None,
                                          nthreads=nthr)
    ref_acts, ref_dx, ref_fused = _REF_MEMO["cfg2"]
    for a, r in zip(acts, ref_acts):
        assert rel_err(a.cpu().numpy(), r) <= 1e-5
    assert rel_err(dx.cpu().numpy(), ref_dx) <= 2e-5           # four chained layers
    got = fused.cpu().numpy()
    o = 0
    for f in st.filters:                                       # per layer: each has its own scale
        n = f.numel()
        assert rel_err(got[o:o + n], ref_fused[o:o + n]) <= 2e-5
        o += n


@pytest.mark.parametrize("ci,co,B,N,kind", [(32, 64, 1, 9000, "room"), (130, 20, 2, 300, "modelnet"), (200, 200, 1, 256, "cube"),
                                            (256, 128, 1, 300, "room")])
def test_shapes_at_the_edges_of_the_matrix_core_path(dev, ci, co, B, N, kind):
    """A matrix-core-path shape on a cloud larger than one group of tiles (N > 8192) and layers with more than 128
    channels other than cfg5's 128 -> 256: whatever kernel family takes them, the results are the reference's."""
    P, X, W, dY = make_case(kind, B, N, ci, co, seed=1700 + ci + N)
    s = (1, 1, 1)
    ref = (oracle.neighbor_count(P, (3, 3, 3), s, VOX), oracle.forward(P, X, W, s, VOX, nthreads=8)) + \
        oracle.backward(dY, P, X, W, s, VOX, nthreads=1)
    check_against(ref, run_hip(dev, P, X, W, dY, s), np.float32)


@pytest.mark.parametrize("num_class,cin,kind", [(None, 3, "modelnet"), (13, 9, "room")])
def test_layer_stacks_match_oracle(dev, num_class, cin, kind):
    B, N = 2, 256
    P = make_case(kind, B, N, 3, 3, seed=400)[0]
    X = synth.features(B, N, cin, 401, points=P)
    st = stack.Conv3pStack(cin, num_class, device=dev, seed=77)
    ups = [synth.upstream_grad(B, N, 9, 410 + i) for i in range(4)] if num_class is None else \
        [synth.upstream_grad(B, N, num_class, 420)]
    acts = st.forward(torch.from_numpy(P).to(dev), torch.from_numpy(X).to(dev))
    dx, fused = st.backward([torch.from_numpy(u).to(dev) for u in ups])
    ref_acts, ref_dx, ref_fused = _oracle_stack(P, X, [f.cpu().numpy() for f in st.filters], st.layers, ups,
                                                num_class)
    for a, r in zip(acts, ref_acts):
        assert rel_err(a.cpu().numpy(), r) <= 2e-5
    assert rel_err(dx.cpu().numpy(), ref_dx) <= 5e-5
    assert rel_err(fused.cpu().numpy(), ref_fused) <= 5e-5


# ------------------------------------------------------------------ neighbour cache (content-validated reuse)
def _both(dev, cache, P, X, W, dY, s):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp, tx, tw, tdy = t(P), t(X), t(W), t(dY)
    y = op.conv3p(tp, tx, tw, s, VOX, cache=cache)
    dx, dw = op.conv3p_grad(tdy, tp, tx, tw, s, VOX, cache=cache)
    return y, dx, dw


def test_cache_matches_stateless_bitwise_and_tracks_content(dev):
    """Same kernels, same order: cached results are bit-identical to the stateless ones; changing the clouds
    (same buffer shapes, even the same device pointers) is detected on the device and triggers a rebuild."""
    B, N = 4, 512
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=2, max_taps=27, max_cin=9, max_cout=9)
    P1, X, W, dY = make_case("modelnet", B, N, 9, 9, seed=800)
    P2 = make_case("modelnet", B, N, 9, 9, seed=801)[0]
    P3 = P1.copy()
    P3[2] = P2[2]                                       # only cloud 2 differs from P1
    tp = torch.from_numpy(P1).to(dev)                   # ONE device buffer, contents rewritten in place
    tx, tw, tdy = [torch.from_numpy(a).to(dev) for a in (X, W, dY)]
    for P in (P1, P1, P2, P3, P1):
        tp.copy_(torch.from_numpy(P))
        for s in ((2, 2, 2), (1, 1, 1), (2, 2, 2)):
            y_c = op.conv3p(tp, tx, tw, s, VOX, cache=cache)
            dx_c, dw_c = op.conv3p_grad(tdy, tp, tx, tw, s, VOX, cache=cache)
            y_s = op.conv3p(tp, tx, tw, s, VOX)
            dx_s, dw_s = op.conv3p_grad(tdy, tp, tx, tw, s, VOX)
            assert torch.equal(y_c, y_s) and torch.equal(dx_c, dx_s) and torch.equal(dw_c, dw_s)
    y_ref = oracle.forward(P1, X, W, (2, 2, 2), VOX)
    assert rel_err(y_c.cpu().numpy(), y_ref) <= 1e-5


def test_cache_slot_eviction(dev):
    """More stencils than slots: least-recently-used slots are rebuilt, results stay exact."""
    B, N = 2, 300
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=2, max_taps=27, max_cin=9, max_cout=9)
    P, X, W, dY = make_case("room", B, N, 9, 9, seed=810)
    for rep in range(2):
        for st in (1, 2, 3, 4, 2, 1):
            s = (st, st, st)
            y, dx, dw = _both(dev, cache, P, X, W, dY, s)
            assert rel_err(y.cpu().numpy(), oracle.forward(P, X, W, s, VOX)) <= 1e-5
            dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, VOX)
            assert rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5 and rel_err(dw.cpu().numpy(), dw_ref) <= 2e-5


@pytest.mark.parametrize("slots", [1, 3])
def test_cache_serves_several_voxel_sizes_over_the_same_points(dev, slots):
    """One cache, unchanged points, stencils of different VOXEL sizes back to back (a new tag, or -- one slot -- an
    evicted one): the window tables of the fused search are built for one voxel size and must be rebuilt when it
    changes (the tables' validity mark covered the cloud's content only)."""
    B, N = 3, 700
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=slots, max_taps=27, max_cin=9, max_cout=9)
    P, X, W, dY = make_case("modelnet", B, N, 9, 9, seed=815)
    t = lambda a: torch.from_numpy(a).to(dev)
    tp, tx, tw, tdy = t(P), t(X), t(W), t(dY)
    for vox, s in ((0.1, (1, 1, 1)), (0.2, (1, 1, 1)), (0.1, (2, 2, 2)), (0.05, (3, 3, 3)), (0.2, (2, 2, 2)), (0.1, (1, 1, 1))):
        cnt = op.neighbor_count(tp, (3, 3, 3), s, vox).cpu().numpy()
        y = op.conv3p(tp, tx, tw, s, vox, cache=cache)
        dx, dw = op.conv3p_grad(tdy, tp, tx, tw, s, vox, cache=cache)
        assert np.array_equal(cnt, oracle.neighbor_count(P, (3, 3, 3), s, vox))
        y_s = op.conv3p(tp, tx, tw, s, vox)
        assert torch.equal(y, y_s), ("cached != stateless", vox, s)
        assert rel_err(y.cpu().numpy(), oracle.forward(P, X, W, s, vox)) <= 1e-5, (vox, s)
        dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, vox)
        assert rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5 and rel_err(dw.cpu().numpy(), dw_ref) <= 2e-5, (vox, s)


def test_cache_that_trusts_tensor_identity_still_tracks_content(dev):
    """trust_tensor_identity (what the TF shim does with TensorFlow's tensors): calls on the very tensor the cache validated
    last -- same storage, address, shape, torch version counter -- skip the content hash; an in-place write through torch
    (version bump), another tensor, or a recycled address (the reference held by the cache makes that impossible while it
    is the held one; here: a new tensor after the old one was dropped) are all validated again.  Results bit for bit the
    stateless ones throughout."""
    B, N = 3, 600
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=2, max_taps=27, max_cin=9, max_cout=9, trust_tensor_identity=True)
    P1, X, W, dY = make_case("modelnet", B, N, 9, 9, seed=830)
    P2 = make_case("modelnet", B, N, 9, 9, seed=831)[0]
    tx, tw, tdy = [torch.from_numpy(a).to(dev) for a in (X, W, dY)]
    tp = torch.from_numpy(P1).to(dev)

    def check(tpts):
        for s in ((2, 2, 2), (1, 1, 1), (2, 2, 2)):
            y_c = op.conv3p(tpts, tx, tw, s, VOX, cache=cache)
            dx_c, dw_c = op.conv3p_grad(tdy, tpts, tx, tw, s, VOX, cache=cache)
            y_s = op.conv3p(tpts, tx, tw, s, VOX)
            dx_s, dw_s = op.conv3p_grad(tdy, tpts, tx, tw, s, VOX)
            assert torch.equal(y_c, y_s) and torch.equal(dx_c, dx_s) and torch.equal(dw_c, dw_s)
    check(tp)
    check(tp)                                            # the same tensor again: trusted (nothing to observe but the results)
    tp.copy_(torch.from_numpy(P2))                       # in-place write through torch: version bump -> validated again
    check(tp)
    other = torch.from_numpy(P1).to(dev)                 # another tensor
    check(other)
    del other
    torch.cuda.empty_cache()
    fresh = torch.from_numpy(P2).to(dev)                 # may well land on the address `other` had
    check(fresh)
    check(tp)


def test_cache_garbage_buffer_is_harmless(dev):
    """A cache whose bytes are garbage (never zero-filled, or recycled) can only cost a rebuild."""
    B, N = 2, 256
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=3, max_cout=9)
    cache.buf.random_(0, 255)
    P, X, W, dY = make_case("cube", B, N, 3, 9, seed=820)
    y, dx, dw = _both(dev, cache, P, X, W, dY, (1, 1, 1))
    assert rel_err(y.cpu().numpy(), oracle.forward(P, X, W, (1, 1, 1), VOX)) <= 1e-5
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, (1, 1, 1), VOX)
    assert rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5 and rel_err(dw.cpu().numpy(), dw_ref) <= 2e-5


def test_forward_only_cache_sized_for_narrow_layers_runs_the_segmentation_head(dev):
    """max_Cin / max_Cout size the BACKWARD's scratch.  A cache sized for narrow layers must still run the
    forward of 36 -> 13 (whose faster transform + gather variant wants a Z array the cache lacks): the plain forward
    kernel takes it, same results as a generously sized cache up to summation order, both the oracle's."""
    B, N = 2, 512
    P, X, W, dY = make_case("room", B, N, 36, 13, seed=825)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    small = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=3, max_cout=3)
    big = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=36, max_cout=13)
    assert small.nbytes < big.nbytes
    y_ref = oracle.forward(P, X, W, (1, 1, 1), VOX)
    for cache in (small, big):
        y = op.conv3p(t(P), t(X), t(W), (1, 1, 1), VOX, cache=cache)
        assert rel_err(y.cpu().numpy(), y_ref) <= 1e-5
    # the backward's scratch (per-workgroup partials) is mandatory: with a cache too small for it the call is refused
    # loudly; when the cache's scratch happens to be large enough it must simply be right
    try:
        dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), (1, 1, 1), VOX, cache=small)
    except op.Conv3pRuntimeError:
        return
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, (1, 1, 1), VOX)
    assert rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5 and rel_err(dw.cpu().numpy(), dw_ref) <= 2e-5


def test_cache_init_makes_a_recycled_buffer_safe(dev):
    """conv3p_cache_init (what the TF shim calls after allocate_persistent): a buffer that still holds a VALID cache's
    control words but whose lists were overwritten -- what a framework allocator can hand back -- is zero-filled on the
    stream, and the next call rebuilds."""
    B, N = 2, 256
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=3, max_cout=9)
    P, X, W, dY = make_case("cube", B, N, 3, 9, seed=821)
    y0, dx0, dw0 = _both(dev, cache, P, X, W, dY, (1, 1, 1))
    # overwrite everything behind the first 4 KiB (hashes, versions, slot marks survive; records and lists do not)
    cache.buf[4096:].random_(0, 255)
    lib = _lib.load()
    st = torch.cuda.current_stream(dev).cuda_stream
    assert lib.conv3p_cache_init(cache.buf.data_ptr(), cache.nbytes, st) == _lib.OK
    y1, dx1, dw1 = _both(dev, cache, P, X, W, dY, (1, 1, 1))
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1) and torch.equal(dw0, dw1)


def test_results_are_bitwise_reproducible(dev):
    """No floating-point atomics on the register-resident paths: two runs give identical bits."""
    P, X, W, dY = make_case("modelnet", 4, 1024, 9, 9, seed=830)
    a = _both(dev, None, P, X, W, dY, (2, 2, 2))
    b = _both(dev, None, P, X, W, dY, (2, 2, 2))
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("ci,co,filt,s,kind", [(9, 9, (4, 4, 4), (2, 2, 2), "modelnet"), (3, 9, (3, 3, 5), (2, 2, 2), "modelnet"),
                                               (9, 9, (3, 5, 3), (3, 2, 3), "room"), (9, 3, (4, 4, 4), (3, 3, 3), "lattice")])
def test_filters_of_33_to_64_taps_keep_the_populated_rows_backward(dev, ci, co, filt, s, kind):
    """The reference takes any filter extents from the tensor's shape (tf_conv3p_atrous.cpp:425-429).  Filters of 33 .. 64
    taps run the populated-rows backward with 64-bit tap sets (two planes of `qbm`): even dilated extents through the
    tile-pair search with the grid's candidate window, odd ones through the fused search.  Oracle parity, reproducible."""
    B, N = 3, 1500
    P, X, W, dY = make_case(kind, B, N, ci, co, filt, seed=1840)
    ntap = filt[0] * filt[1] * filt[2]
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=ntap, max_cin=ci, max_cout=co, sparse_neighbourhoods=True)
    a = _both(dev, cache, P, X, W, dY, s)
    b = _both(dev, cache, P, X, W, dY, s)
    c = _both(dev, None, P, X, W, dY, s)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    ry = oracle.forward(P, X, W, s, VOX)
    rdx, rdw = oracle.backward(dY, P, X, W, s, VOX)
    r64 = oracle.backward(dY.astype(np.float64), P.astype(np.float64), X.astype(np.float64), W.astype(np.float64), s, VOX)
    tol_y, tol_w = TOL[np.dtype(np.float32)]
    # (the cache carries the short-lists hint, so `a` is the populated-rows kernel's result; the stateless call `c` lets
    # the lists just built choose between it and the dense-G kernel: a different summation order, the same tolerance)
    for got in (a, c):
        assert rel_err(got[0].cpu().numpy(), ry) <= tol_y
        assert rel_err(got[1].cpu().numpy(), rdx) <= tol_y
        assert rel_err(got[2].cpu().numpy(), rdw) <= max(tol_w, 4.0 * rel_err(rdw, r64[1]))


@pytest.mark.parametrize("ci,co,filt,s,kind", [(9, 9, (5, 5, 5), (1, 1, 1), "modelnet"), (3, 9, (5, 5, 5), (2, 2, 2), "modelnet"),
                                               (9, 9, (5, 4, 5), (1, 2, 1), "room"), (9, 3, (3, 5, 7), (2, 1, 1), "lattice")])
def test_filters_of_65_to_128_taps_stay_on_the_deterministic_kernels(dev, ci, co, filt, s, kind):
    """Filters of 65 .. 128 taps (5 x 5 x 5 = 125): the dense G of backward_kernel no longer fits LDS, and until round 5 such
    layers took the thread-per-pair kernels with global float atomics (not reproducible).  Now: the populated-rows backward
    with 128-bit tap sets (four planes of `qbm`), its rounds by tap ranges, whatever the hint and for undilated stencils
    too; the forward was on the register path already.  No memset and no atomics kernel in the profile, bit-identical
    repeats (cached, stateless), oracle parity."""
    lib = _lib.load()
    B, N = 2, 700
    P, X, W, dY = make_case(kind, B, N, ci, co, filt, seed=1850)
    ntap = filt[0] * filt[1] * filt[2]
    assert 64 < ntap <= 128
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=ntap, max_cin=ci, max_cout=co)
    lib.conv3p_profile_reset()
    lib.conv3p_profile_enable(1)
    a = _both(dev, cache, P, X, W, dY, s)
    torch.cuda.synchronize()
    lib.conv3p_profile_enable(0)
    seen = {}
    for k in range(lib.conv3p_profile_kinds()):
        n = ctypes.c_uint64(0)
        lib.conv3p_profile_read(k, ctypes.byref(n), None)
        seen[lib.conv3p_profile_name(k).decode()] = n.value
    lib.conv3p_profile_reset()
    assert seen.get("memset", 0) == 0, seen        # (the atomics kernels accumulate into zeroed outputs)
    b = _both(dev, cache, P, X, W, dY, s)
    c = _both(dev, None, P, X, W, dY, s)
    d = _both(dev, None, P, X, W, dY, s)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    for u, v in zip(c, d):
        assert torch.equal(u, v)
    ry = oracle.forward(P, X, W, s, VOX)
    rdx, rdw = oracle.backward(dY, P, X, W, s, VOX)
    r64 = oracle.backward(dY.astype(np.float64), P.astype(np.float64), X.astype(np.float64), W.astype(np.float64), s, VOX)
    tol_y, tol_w = TOL[np.dtype(np.float32)]
    cnt = op.neighbor_count(torch.from_numpy(P).to(dev), filt, s, VOX).cpu().numpy()
    assert np.array_equal(cnt, oracle.neighbor_count(P, filt, s, VOX))
    for got in (a, c):
        assert rel_err(got[0].cpu().numpy(), ry) <= tol_y
        assert rel_err(got[1].cpu().numpy(), rdx) <= tol_y
        assert rel_err(got[2].cpu().numpy(), rdw) <= max(tol_w, 4.0 * rel_err(rdw, r64[1]))


def test_stack_with_cache_hints_over_changing_batches(dev):
    """What bench.py does: one Conv3pStack (neighbour cache + POINTS_UNCHANGED hints inside a step) fed a
    different batch every step must give exactly what a cache-less stack gives, step after step."""
    B, N = 4, 512
    cached = stack.Conv3pStack(3, None, device=dev, seed=5, use_cache=True)
    plain = stack.Conv3pStack(3, None, device=dev, seed=5, use_cache=False)
    ups = [torch.from_numpy(synth.upstream_grad(B, N, 9, 900 + i)).to(dev) for i in range(4)]
    batches = [torch.from_numpy(synth.modelnet_like(B, N, seed=910 + i)).to(dev) for i in range(3)]
    for step in range(7):
        P = batches[step % 3]
        a1 = cached.forward(P, P)
        if step % 2 == 0:
            cached.prefetch(batches[(step + 1) % 3])      # cross-step pipelining, every other step
        dx1, f1 = cached.backward(ups)
        a2 = plain.forward(P, P)
        dx2, f2 = plain.backward(ups)
        for u, v in zip(a1, a2):
            assert torch.equal(u, v)
        assert torch.equal(dx1, dx2) and torch.equal(f1, f2)


# ------------------------------------------------------------------ deep-channel (matrix-core) path
@pytest.mark.parametrize("ci,co", [(128, 256), (64, 128), (128, 128), (3, 64), (37, 2), (7, 43), (256, 256), (256, 128), (200, 130),
                                   (129, 250)])
def test_deep_channel_path_matches_oracle(dev, ci, co):
    """cfg5-shaped layers go through the factorised MFMA kernels (conv3p_deep.hpp); room-like data, several
    tiles per cloud, both ops.  The odd shapes exercise the padded instantiations: rows shorter than one 16-byte
    load (3 inputs / 2 outputs), row lengths that are not multiples of 4 (37, 7, 43)."""
    B, N = 2, 700
    P = synth.room_like(B, N, 950, extent=(1.0, 1.0, 1.5))
    X = synth.features(B, N, ci, 951, points=P)
    W = synth.filter_weights(3, 3, 3, ci, co, 952)
    dY = synth.upstream_grad(B, N, co, 953)
    s = (1, 1, 1)
    ref = (oracle.neighbor_count(P, (3, 3, 3), s, VOX), oracle.forward(P, X, W, s, VOX, nthreads=8)) + \
        oracle.backward(dY, P, X, W, s, VOX, nthreads=1)
    check_against(ref, run_hip(dev, P, X, W, dY, s), np.float32)


@pytest.mark.parametrize("ci,co,N", [(320, 320, 2048), (300, 70, 500), (40, 260, 500), (513, 9, 300)])
def test_more_than_256_channels_run_as_blocks_on_the_matrix_core_path(dev, ci, co, N):
    """Round-3 verdict, item 7 (fp32 half): layers with more than 256 channels on a side are cut into blocks of at most
    256 x 256 channels (zero-padded to 128 or 256), each on the matrix-core kernels, results added in a fixed order --
    instead of the global-atomics kernels.  Against the oracle; cached == stateless bit for bit; reproducible; and the
    profile says which kernels ran."""
    lib = _lib.load()
    B = 1 if N > 1000 else 2
    P = synth.room_like(B, N, 990, extent=(1.0, 1.0, 1.5)) if N <= 1000 else synth.modelnet_like(B, N, seed=990)
    X = synth.features(B, N, ci, 991, points=P)
    W = synth.filter_weights(3, 3, 3, ci, co, 992)
    dY = synth.upstream_grad(B, N, co, 993)
    s = (1, 1, 1)
    ref = (oracle.neighbor_count(P, (3, 3, 3), s, VOX), oracle.forward(P, X, W, s, VOX, nthreads=8)) + \
        oracle.backward(dY, P, X, W, s, VOX, nthreads=8)
    lib.conv3p_profile_reset()
    lib.conv3p_profile_enable(1)
    got = run_hip(dev, P, X, W, dY, s)
    torch.cuda.synchronize()
    lib.conv3p_profile_enable(0)
    seen = {}
    for k in range(lib.conv3p_profile_kinds()):
        n = ctypes.c_uint64(0)
        lib.conv3p_profile_read(k, ctypes.byref(n), None)
        seen[lib.conv3p_profile_name(k).decode()] = n.value
    lib.conv3p_profile_reset()
    check_against(ref, got, np.float32)
    nb = ((ci + 255) // 256) * ((co + 255) // 256)
    assert seen.get("deep_gemm_kernel", 0) >= 2 * nb and seen.get("deep_dw_kernel", 0) >= nb, seen
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
    a = _both(dev, cache, P, X, W, dY, s)
    b = _both(dev, cache, P, X, W, dY, s)
    c = _both(dev, None, P, X, W, dY, s)
    for u, v, w in zip(a, b, c):
        assert torch.equal(u, v) and torch.equal(u, w)


@pytest.mark.parametrize("ci,co,N", [(32, 64, 2048), (5, 7, 500), (17, 3, 300), (40, 9, 700)])
def test_fp64_outside_the_register_path_shapes_runs_as_blocks(dev, ci, co, N):
    """Round-3 verdict, item 7 (fp64 half): double precision is registered for every shape
    (tf_conv3p_atrous.cpp:511-517, :722-728).  Shapes outside the register-path list run as blocks of 16 input x 8 output
    channels (zero-padded) on the register-path kernels <double, 16, 8> -- fixed summation order -- instead of the
    global-atomics kernels: against the float64 oracle, cached == stateless bit for bit, reproducible."""
    B = 1 if N > 1000 else 2
    P = synth.room_like(B, N, 1190, extent=(1.0, 1.0, 1.5)) if N <= 1000 else synth.modelnet_like(B, N, seed=1190)
    X = synth.features(B, N, ci, 1191, points=P).astype(np.float64)
    W = synth.filter_weights(3, 3, 3, ci, co, 1192).astype(np.float64)
    dY = synth.upstream_grad(B, N, co, 1193).astype(np.float64)
    P = P.astype(np.float64)
    for s in ((1, 1, 1), (2, 2, 2)):
        ref = (oracle.neighbor_count(P, (3, 3, 3), s, VOX), oracle.forward(P, X, W, s, VOX, nthreads=8)) + \
            oracle.backward(dY, P, X, W, s, VOX, nthreads=8)
        check_against(ref, run_hip(dev, P, X, W, dY, s), np.float64)
        cache = op.NeighborCache(B, N, torch.float64, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
        a = _both(dev, cache, P, X, W, dY, s)
        b = _both(dev, cache, P, X, W, dY, s)
        c = _both(dev, None, P, X, W, dY, s)
        for u, v, w in zip(a, b, c):
            assert torch.equal(u, v) and torch.equal(u, w)
        if N > 1000:
            break


@pytest.mark.parametrize("ci,co", [(128, 256), (256, 256), (256, 128), (200, 200)])
def test_deep_channel_path_is_used_and_reproducible(dev, ci, co):
    """128 -> 256 (cfg5) and the 256-channel classes (256 -> 256, 256 -> 128, padded 200 -> 200) run on the matrix-core
    kernels -- not on the global-atomics kernels -- and are bitwise reproducible."""
    lib = _lib.load()
    P, X, W, dY = make_case("room", 1, 256, ci, co, seed=960)
    lib.conv3p_profile_reset()
    lib.conv3p_profile_enable(1)
    a = _both(dev, None, P, X, W, dY, (1, 1, 1))
    torch.cuda.synchronize()
    lib.conv3p_profile_enable(0)
    seen = {}
    for k in range(lib.conv3p_profile_kinds()):
        n = ctypes.c_uint64(0)
        lib.conv3p_profile_read(k, ctypes.byref(n), None)
        seen[lib.conv3p_profile_name(k).decode()] = n.value
    lib.conv3p_profile_reset()
    assert seen.get("deep_gemm_kernel", 0) == 2 and seen.get("deep_dw_kernel", 0) == 1, seen
    b = _both(dev, None, P, X, W, dY, (1, 1, 1))
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("N,kind", [(9000, "room"), (16384 + 70, "modelnet")])
def test_deep_channel_path_beyond_one_search_group(dev, N, kind):
    """Clouds of more than 8192 points are searched in several groups of candidate tiles (one pair segment per group);
    the matrix-core path reads them as one centre-major list.  32 -> 64 channels at N = 9000 (2 groups) and
    N = 16454 (3 groups, ragged): the deep kernels run (not the global-atomics ones), results match the oracle and
    are bitwise reproducible."""
    lib = _lib.load()
    ci, co = 32, 64
    P, X, W, dY = make_case(kind, 1, N, ci, co, seed=1400)
    lib.conv3p_profile_reset()
    lib.conv3p_profile_enable(1)
    a = _both(dev, None, P, X, W, dY, (1, 1, 1))
    torch.cuda.synchronize()
    lib.conv3p_profile_enable(0)
    seen = {}
    for k in range(lib.conv3p_profile_kinds()):
        n = ctypes.c_uint64(0)
        lib.conv3p_profile_read(k, ctypes.byref(n), None)
        seen[lib.conv3p_profile_name(k).decode()] = n.value
    lib.conv3p_profile_reset()
    assert seen.get("deep_gemm_kernel", 0) == 2 and seen.get("deep_dw_kernel", 0) == 1, seen
    b = _both(dev, None, P, X, W, dY, (1, 1, 1))
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert rel_err(a[0].cpu().numpy(), oracle.forward(P, X, W, (1, 1, 1), VOX, nthreads=8)) <= 1e-5
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, (1, 1, 1), VOX, nthreads=1)
    assert rel_err(a[1].cpu().numpy(), dx_ref) <= 1e-5 and rel_err(a[2].cpu().numpy(), dw_ref) <= 5e-5


def test_prefetched_record_orders_of_the_matrix_core_path(dev):
    """conv3p_cache_prepare with CONV3P_CACHE_PREPARE_DEEP_ORDERS leaves the deep path's forward / backward record orders in
    the cache: the layer's calls on the same points (hinted) skip deep_order_kernel and give bit-identical results; a
    narrow layer or another stencil in between, or new points, make them rebuild -- results unchanged."""
    lib = _lib.load()
    B, N, ci, co = 2, 700, 40, 72
    P, X, W, dY = make_case("room", B, N, ci, co, seed=1500)
    P2 = make_case("room", B, N, ci, co, seed=1501)[0]
    X9 = synth.features(B, N, 9, 1502, points=P)
    W9 = synth.filter_weights(3, 3, 3, 9, 9, 1503)
    t = lambda a: torch.from_numpy(a).to(dev)
    tP, tP2, tX, tW, tdY, tX9, tW9 = t(P), t(P2), t(X), t(W), t(dY), t(X9), t(W9)
    s = (1, 1, 1)

    def orders_run(fn):
        lib.conv3p_profile_reset()
        lib.conv3p_profile_enable(1)
        out = fn()
        torch.cuda.synchronize()
        lib.conv3p_profile_enable(0)
        n = 0
        for k in range(lib.conv3p_profile_kinds()):
            if lib.conv3p_profile_name(k).decode() == "deep_order_kernel":
                c = ctypes.c_uint64(0)
                lib.conv3p_profile_read(k, ctypes.byref(c), None)
                n = c.value
        lib.conv3p_profile_reset()
        return n, out

    mk = lambda: op.NeighborCache(B, N, torch.float32, dev, slots=2, max_taps=27, max_cin=ci, max_cout=co)
    ref_cache = mk()
    y_ref = op.conv3p(tP, tX, tW, s, VOX, cache=ref_cache)
    dx_ref, dw_ref = op.conv3p_grad(tdY, tP, tX, tW, s, VOX, cache=ref_cache, points_unchanged=True)

    cache = mk()
    n, _ = orders_run(lambda: op.cache_prepare(tP, (3, 3, 3), s, VOX, cache, deep_orders=True))
    assert n == 2                                                  # both orders built by the prepare call
    n, y = orders_run(lambda: op.conv3p(tP, tX, tW, s, VOX, cache=cache, points_unchanged=True))
    assert n == 0 and torch.equal(y, y_ref)
    n, (dx, dw) = orders_run(lambda: op.conv3p_grad(tdY, tP, tX, tW, s, VOX, cache=cache, points_unchanged=True))
    assert n == 0 and torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref)
    n, y = orders_run(lambda: op.conv3p(tP, tX, tW, s, VOX, cache=cache, points_unchanged=True))   # still in place
    assert n == 0 and torch.equal(y, y_ref)
    # a narrow layer uses the scratch region for its own purposes: the orders are rebuilt afterwards
    op.conv3p_grad(t(synth.upstream_grad(B, N, 9, 1504)), tP, tX9, tW9, s, VOX, cache=cache, points_unchanged=True)
    n, (dx, dw) = orders_run(lambda: op.conv3p_grad(tdY, tP, tX, tW, s, VOX, cache=cache, points_unchanged=True))
    assert n == 1 and torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref)
    # another stencil's wide layer overwrites them too
    op.cache_prepare(tP, (3, 3, 3), s, VOX, cache, points_unchanged=True, deep_orders=True)
    op.conv3p(tP, tX, tW, (2, 2, 2), VOX, cache=cache, points_unchanged=True)
    n, y = orders_run(lambda: op.conv3p(tP, tX, tW, s, VOX, cache=cache, points_unchanged=True))
    assert n == 1 and torch.equal(y, y_ref)
    # new points without the hint: everything is rebuilt, results follow the new clouds
    n, y2 = orders_run(lambda: op.conv3p(tP2, tX, tW, s, VOX, cache=cache))
    assert n == 1 and rel_err(y2.cpu().numpy(), oracle.forward(P2, X, W, s, VOX, nthreads=8)) <= 1e-5
    assert rel_err(y_ref.cpu().numpy(), oracle.forward(P, X, W, s, VOX, nthreads=8)) <= 1e-5


def test_pair_buffer_overflow_falls_back_correctly(dev):
    """A cache configured with a tiny pair capacity overflows: the small-channel kernels search the tile
    themselves, the deep path hands flagged tiles to the generic kernel.  Results must still be exact."""
    for ci, co in ((9, 9), (32, 64), (36, 13)):   # (36 -> 13: the transform + gather forward and the populated-rows backward)
        B, N = 2, 300
        P, X, W, dY = make_case("room", B, N, ci, co, seed=970)
        cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, pairs_per_point=4, max_cin=ci,
                                 max_cout=co)
        y, dx, dw = _both(dev, cache, P, X, W, dY, (1, 1, 1))
        assert rel_err(y.cpu().numpy(), oracle.forward(P, X, W, (1, 1, 1), VOX)) <= 1e-5
        dx_ref, dw_ref = oracle.backward(dY, P, X, W, (1, 1, 1), VOX)
        assert rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5 and rel_err(dw.cpu().numpy(), dw_ref) <= 2e-5


def test_transform_gather_forward_dilated_ragged_stateless(dev):
    """36 -> 13 (tap_transform_kernel + tap_gather_kernel): dilated and undilated stencils, N not a multiple of the
    tile or of the transform's 32-point blocks; against the oracle, and the stateless call (workspace sized by
    conv3p_workspace_bytes, Z included) bit for bit against the cached one."""
    B, N, ci, co = 3, 777, 36, 13
    P, X, W, dY = make_case("room", B, N, ci, co, seed=1201)
    t = lambda a: torch.from_numpy(a).to(dev)
    tP, tX, tW = t(P), t(X), t(W)
    for s in ((1, 1, 1), (2, 2, 2), (3, 1, 2)):
        cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
        y = op.conv3p(tP, tX, tW, s, VOX, cache=cache)
        assert rel_err(y.cpu().numpy(), oracle.forward(P, X, W, s, VOX)) <= 1e-5
        y3 = op.conv3p(tP, tX, tW, s, VOX)              # stateless workspace (sized by conv3p_workspace_bytes)
        assert torch.equal(y, y3)


def test_tile_schedule_is_a_permutation_and_results_do_not_depend_on_it(dev):
    """Clouds with a dense cluster (tiles of very different list lengths): every tile is processed exactly once whatever
    the launch order -- outputs equal the oracle's, two caches built independently agree bit for bit, also with more
    clouds than XCDs (several rounds per XCD) and with a single tile per cloud."""
    rng = np.random.default_rng(77)
    for B, N in ((11, 500), (3, 64), (1, 1000)):
        P = synth.room_like(B, N, 1300 + B)
        P[:, : N // 4] = P[:, :1] + 0.02 * rng.standard_normal((B, N // 4, 3)).astype(np.float32)   # a dense blob
        X = synth.features(B, N, 9, 1301, points=P)
        W = synth.filter_weights(3, 3, 3, 9, 9, 1302)
        dY = synth.upstream_grad(B, N, 9, 1303)
        a = _both(dev, op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=9, max_cout=9), P, X, W, dY, (1, 1, 1))
        b = _both(dev, op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=9, max_cout=9), P, X, W, dY, (1, 1, 1))
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        assert rel_err(a[0].cpu().numpy(), oracle.forward(P, X, W, (1, 1, 1), VOX)) <= 1e-5
        dx_ref, dw_ref = oracle.backward(dY, P, X, W, (1, 1, 1), VOX)
        assert rel_err(a[1].cpu().numpy(), dx_ref) <= 1e-5 and rel_err(a[2].cpu().numpy(), dw_ref) <= 2e-5


# ------------------------------------------------------------------ fused conv3p + SELU layer ops
@pytest.mark.parametrize("ci,co,dt", [(9, 9, np.float32), (3, 9, np.float32), (5, 7, np.float32), (32, 64, np.float32),
                                      (5, 7, np.float64), (36, 13, np.float64), (36, 13, np.float32), (260, 30, np.float32),
                                      (20, 6, np.float64)])
def test_layer_ops_equal_unfused_sequence(dev, ci, co, dt):
    """conv3p_layer == selu(conv3p); conv3p_layer_grad == selu_grad(input, dX + addend), on every kernel family
    (register path, generic path, deep path, fp64)."""
    B, N = 2, 400
    P, X, W, dY = make_case("room", B, N, ci, co, seed=990, dtype=dt)
    t = lambda a: torch.from_numpy(a).to(dev)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    tP, tW, tdY = t(P), t(W), t(dY)
    tX = op.selu(t(X))                                  # the layer's input is a SELU output
    add = t(synth.upstream_grad(B, N, ci, 991).astype(dt))
    cache = op.NeighborCache(B, N, tdt, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
    s = (2, 2, 2)
    y_ref = op.selu(op.conv3p(tP, tX, tW, s, VOX, cache=cache))
    dx_raw, dw_ref = op.conv3p_grad(tdY, tP, tX, tW, s, VOX, cache=cache)
    tol = 1e-6 if dt == np.float32 else 1e-13
    y = op.conv3p_layer(tP, tX, tW, s, VOX, cache)
    assert rel_err(y.cpu().numpy(), y_ref.cpu().numpy()) <= tol
    for a in (None, add):
        dx_ref = op.selu_grad(tX, dx_raw, a)
        dx, dw = op.conv3p_layer_grad(tdY, tP, tX, tW, s, VOX, cache, grad_addend=a)
        assert rel_err(dx.cpu().numpy(), dx_ref.cpu().numpy()) <= tol
        assert torch.equal(dw, dw_ref)   # (fp64 5 -> 7 too since round 4: blocks on the register-path kernels, fixed order)


def test_stack_fused_selu_matches_unfused(dev):
    B, N = 4, 512
    P = synth.modelnet_like(B, N, seed=995)
    tP = torch.from_numpy(P).to(dev)
    ups = [torch.from_numpy(synth.upstream_grad(B, N, stack.HIDDEN, 996 + i)).to(dev) for i in range(4)]
    res = []
    for fuse in (False, True):
        st = stack.Conv3pStack(3, None, device=dev, seed=7, fuse_selu=fuse)
        acts = st.forward(tP, tP.clone())
        dx, fused = st.backward(ups)
        res.append(([a.cpu().numpy() for a in acts], dx.cpu().numpy(), fused.cpu().numpy().copy()))
    for a, b in zip(res[0][0], res[1][0]):
        assert rel_err(b, a) <= 1e-6
    assert rel_err(res[1][1], res[0][1]) <= 1e-6
    assert rel_err(res[1][2], res[0][2]) <= 2e-6


@pytest.mark.parametrize("cin,ncls,B,N,kind", [(3, None, 11, 700, "modelnet"), (3, None, 32, 2048, "modelnet"), (9, 13, 3, 4096, "room"),
                                                (9, None, 2, 1000, "lattice")])
def test_fused_stack_launch_equals_the_per_layer_launches(dev, cin, ncls, B, N, kind):
    """CONV3P_CACHE_FUSED_STACK (opt-in): the hidden layers of a pass as ONE launch with per-cloud barriers between the
    layers (csrc/conv3p_stack_fused.hpp).  The tile passes are the per-layer kernels' own code: activations and grad_input
    bit for bit those of the per-layer launches, grad_filter within the op's tolerance (its partials are summed in another
    order); the launches are counted, no barrier wait gave up, every cloud's tiles shared an XCC; twice in a row (the
    counters are never reset) and against the oracle."""
    P = synth.modelnet_like(B, N, seed=5) if kind == "modelnet" else synth.room_like(B, N, 5) if kind == "room" else make_case("lattice", B, N, cin, 9, seed=5)[0]
    X = synth.features(B, N, cin, 6, points=P)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp, tx = t(P), t(X)
    ups = [synth.upstream_grad(B, N, ncls if ncls else stack.HIDDEN, 70 + i) for i in range(1 if ncls else 4)]
    res = []
    for fused in (True, False):
        st = stack.Conv3pStack(cin, ncls, device=dev, seed=3, fused_launch=fused)
        st.sparse_neighbourhoods = True if kind != "room" else None     # (the fused backward needs the SPARSE hint)
        for rep in range(2):
            acts = st.forward(tp, tx)
            dx, fg = st.backward([t(u) for u in ups])
        f, b, e = st.fused_status()
        assert e == 0, "fused launch error bits %d" % e
        assert (f, b) == ((2, 2 if kind != "room" else 0) if fused else (0, 0))
        res.append(([a.clone() for a in acts], dx.clone(), fg.clone()))
    for a, r in zip(res[0][0], res[1][0]):
        assert torch.equal(a, r)
    assert torch.equal(res[0][1], res[1][1])
    assert rel_err(res[0][2].cpu().numpy(), res[1][2].cpu().numpy()) <= 2e-6
    if N <= 1000:
        st = stack.Conv3pStack(cin, ncls, device=dev, seed=3)
        ref_acts, ref_dx, ref_fused = _oracle_stack(P, X, [f.cpu().numpy() for f in st.filters], st.layers, ups, ncls)
        for a, r in zip(res[0][0], ref_acts):
            assert rel_err(a.cpu().numpy(), r) <= 2e-5
        assert rel_err(res[0][1].cpu().numpy(), ref_dx) <= 5e-5
        assert rel_err(res[0][2].cpu().numpy(), ref_fused) <= 5e-5


def test_fused_stack_launch_with_overflowed_pair_buffers(dev):
    """Caches too small for the pair lists (pairs_per_point = 6: most tiles overflow and search themselves inside the
    kernels): the fused launches take the same fallback inside their tile passes -- equal to the per-layer launches and to
    the oracle."""
    B, N, cin = 5, 900, 3
    P = synth.modelnet_like(B, N, seed=15)
    X = synth.features(B, N, cin, 16, points=P)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp, tx = t(P), t(X)
    ups = [synth.upstream_grad(B, N, stack.HIDDEN, 170 + i) for i in range(4)]
    res = []
    for fused in (True, False):
        st = stack.Conv3pStack(cin, None, device=dev, seed=3, fused_launch=fused)
        st.sparse_neighbourhoods = True
        for i in (0, 1):
            st._caches[i] = op.NeighborCache(B, N, torch.float32, dev, slots=len(st.layers), max_taps=27, max_cin=9, max_cout=9,
                                             pairs_per_point=6, sparse_neighbourhoods=True, fused_stack=fused)
        acts = st.forward(tp, tx)
        dx, fg = st.backward([t(u) for u in ups])
        f, b, e = st.fused_status()
        assert e == 0 and (f, b) == ((1, 1) if fused else (0, 0))
        res.append(([a.clone() for a in acts], dx.clone(), fg.clone()))
    # (WHICH tiles overflow is decided by the order in which the tiles of a cloud reserve their slots -- a race between
    # workgroups: two independently built caches need not agree, and an overflowed tile sums in another order -- so the two
    # runs are compared within rounding, not bit for bit)
    for a, r in zip(res[0][0], res[1][0]):
        assert rel_err(a.cpu().numpy(), r.cpu().numpy()) <= 2e-6
    assert rel_err(res[0][1].cpu().numpy(), res[1][1].cpu().numpy()) <= 2e-6
    assert rel_err(res[0][2].cpu().numpy(), res[1][2].cpu().numpy()) <= 2e-6
    st = stack.Conv3pStack(cin, None, device=dev, seed=3)
    ref_acts, ref_dx, ref_fused = _oracle_stack(P, X, [f.cpu().numpy() for f in st.filters], st.layers, ups, None)
    for a, r in zip(res[0][0], ref_acts):
        assert rel_err(a.cpu().numpy(), r) <= 2e-5
    assert rel_err(res[0][1].cpu().numpy(), ref_dx) <= 5e-5
    assert rel_err(res[0][2].cpu().numpy(), ref_fused) <= 5e-5


def test_cache_prepare_multi_equals_per_stencil_prepare(dev):
    """One batched search launch for strides 1..4 gives the same lists (hence bitwise the same op results) as
    four separate prepares; a second multi call on unchanged points launches nothing new and stays valid."""
    B, N = 3, 700
    P, X, W, dY = make_case("modelnet", B, N, 9, 9, seed=1001)
    t = lambda a: torch.from_numpy(a).to(dev)
    tP, tX, tW, tdY = t(P), t(X), t(W), t(dY)
    strides = [(1, 1, 1), (2, 2, 2), (3, 3, 3), (4, 4, 4)]
    mk = lambda: op.NeighborCache(B, N, torch.float32, dev, slots=4, max_taps=27, max_cin=9, max_cout=9)
    ca, cb = mk(), mk()
    for s in strides:
        op.cache_prepare(tP, (3, 3, 3), s, VOX, ca, points_unchanged=s != strides[0])
    op.cache_prepare_multi(tP, (3, 3, 3), strides, VOX, cb)
    op.cache_prepare_multi(tP, (3, 3, 3), strides, VOX, cb, points_unchanged=True)
    for s in strides:
        ya = op.conv3p(tP, tX, tW, s, VOX, cache=ca, points_unchanged=True)
        yb = op.conv3p(tP, tX, tW, s, VOX, cache=cb, points_unchanged=True)
        assert torch.equal(ya, yb)
        da = op.conv3p_grad(tdY, tP, tX, tW, s, VOX, cache=ca, points_unchanged=True)
        db = op.conv3p_grad(tdY, tP, tX, tW, s, VOX, cache=cb, points_unchanged=True)
        assert torch.equal(da[0], db[0]) and torch.equal(da[1], db[1])
        assert rel_err(yb.cpu().numpy(), oracle.forward(P, X, W, s, VOX)) <= 1e-5
    # new points through the multi entry: everything is rebuilt
    P2 = synth.modelnet_like(B, N, seed=1002)
    tP2 = t(P2)
    op.cache_prepare_multi(tP2, (3, 3, 3), strides, VOX, cb)
    y2 = op.conv3p(tP2, tX, tW, (3, 3, 3), VOX, cache=cb, points_unchanged=True)
    assert rel_err(y2.cpu().numpy(), oracle.forward(P2, X, W, (3, 3, 3), VOX)) <= 1e-5


@pytest.mark.parametrize("batched", [False, True])
def test_stack_prefetch_orders_and_batched_search(dev, batched):
    """prefetch() right after forward() (the bench's order), prefetch() BEFORE the forward of the batch that was
    prefetched earlier (two prefetches outstanding), and the one-launch search all give the results of a stack
    that never prefetches."""
    B, N = 3, 320
    Ps = [torch.from_numpy(synth.modelnet_like(B, N, seed=1010 + i)).to(dev) for i in range(3)]
    ups = [torch.from_numpy(synth.upstream_grad(B, N, stack.HIDDEN, 1020 + i)).to(dev) for i in range(4)]

    def run(mode):
        st = stack.Conv3pStack(3, None, device=dev, seed=11)
        st.batched_prefetch = batched
        out = []
        if mode == "early":
            st.prefetch(Ps[0])
        for i in range(5):
            cur, nxt = Ps[i % 3], Ps[(i + 1) % 3]
            if mode == "early":
                st.prefetch(nxt)
            acts = st.forward(cur, cur)
            if mode == "late":
                st.prefetch(nxt)
            dx, fused = st.backward(ups)
            out.append((acts[3].clone(), dx.clone(), fused.clone()))
        torch.cuda.synchronize()
        return out

    ref = run("none")
    for mode in ("late", "early"):
        got = run(mode)
        for r, g in zip(ref, got):
            for a, b in zip(r, g):
                assert torch.equal(a, b), mode


def test_cloud_larger_than_the_lds_sort(dev):
    """N > 16384: the cloud does not fit the LDS sort, prep_kernel keeps the caller's order (loose tiles, same
    decisions).  Also exercises several 64-tile ballots of candidate tiles."""
    B, N = 1, 17000
    P = synth.room_like(B, N, 1030, extent=(3.0, 3.0, 2.0))
    X = synth.features(B, N, 3, 1031, points=P)
    W = synth.filter_weights(3, 3, 3, 3, 9, 1032)
    dY = synth.upstream_grad(B, N, 9, 1033)
    s = (2, 2, 2)
    ref = (oracle.neighbor_count(P, (3, 3, 3), s, VOX), oracle.forward(P, X, W, s, VOX, nthreads=8)) + \
        oracle.backward(dY, P, X, W, s, VOX, nthreads=1)
    check_against(ref, run_hip(dev, P, X, W, dY, s), np.float32)


def test_deep_channel_path_non_finite_values_reach_only_their_neighbours(dev):
    """The matrix-core path multiplies staged neighbour rows by 0 for the centres they do not belong to, which is
    only exact for finite rows: tiles that meet an Inf / NaN row are handed to the exact generic kernel, so the
    non-finite values land on exactly the outputs they reach in the reference."""
    B, N, ci, co = 1, 300, 32, 64
    P, X, W, dY = make_case("room", B, N, ci, co, seed=1040)
    X = X.copy(); dY = dY.copy()
    X[0, 17, 3] = np.inf
    X[0, 200, 0] = np.nan
    dY[0, 99, 5] = np.inf
    s = (1, 1, 1)
    with np.errstate(all="ignore"):
        y_ref = oracle.forward(P, X, W, s, VOX)
        dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, VOX)
    t = lambda a: torch.from_numpy(a).to(dev)
    y = op.conv3p(t(P), t(X), t(W), s, VOX).cpu().numpy()
    dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, VOX)
    for got, ref, tol in ((y, y_ref, 1e-5), (dx.cpu().numpy(), dx_ref, 1e-5), (dw.cpu().numpy(), dw_ref, 2e-5)):
        bad_ref = ~np.isfinite(ref)
        assert bad_ref.any() and not bad_ref.all()
        assert np.array_equal(~np.isfinite(got), bad_ref)
        ok = ~bad_ref
        assert np.max(np.abs(got[ok] - ref[ok])) <= tol * max(1.0, np.max(np.abs(ref[ok])))


def test_blocked_wide_layer_on_a_persistent_cache_with_prepared_orders_and_several_strides(dev):
    """A layer of more than 256 channels on one persistent cache: geometry and both record orders built ahead
    (conv3p_cache_prepare with CONV3P_CACHE_PREPARE_DEEP_ORDERS), then the layer's calls with the points-unchanged
    promise, a second stencil in between (the orders in the scratch are then rebuilt), and the first stencil again --
    every result bit for bit the stateless one."""
    B, N, ci, co = 2, 350, 270, 40
    P, X, W, dY = make_case("room", B, N, ci, co, seed=1460)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp, tx, tw, tdy = t(P), t(X), t(W), t(dY)
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=2, max_taps=27, max_cin=ci, max_cout=co)
    ref = {}
    for s in ((1, 1, 1), (2, 2, 2)):
        y = op.conv3p(tp, tx, tw, s, VOX)
        dx, dw = op.conv3p_grad(tdy, tp, tx, tw, s, VOX)
        ref[s] = (y, dx, dw)
    op.cache_prepare(tp, (3, 3, 3), (1, 1, 1), VOX, cache, deep_orders=True)
    for i, s in enumerate(((1, 1, 1), (1, 1, 1), (2, 2, 2), (1, 1, 1), (2, 2, 2))):
        y = op.conv3p(tp, tx, tw, s, VOX, cache=cache, points_unchanged=True)
        dx, dw = op.conv3p_grad(tdy, tp, tx, tw, s, VOX, cache=cache, points_unchanged=True)
        for u, v in zip((y, dx, dw), ref[s]):
            assert torch.equal(u, v), (i, s)


def test_deep_channel_path_forgets_the_non_finite_mark_of_an_earlier_call(dev):
    """The matrix-core path marks tiles that met Inf / NaN for the exact kernel; the mark lives with the record order,
    which a later call on the same points (CONV3P_CACHE_POINTS_UNCHANGED) reuses.  That call, with finite data, must
    not find the tiles still marked (the exact kernel would add its result to rows the matrix-core kernel has filled --
    found in round 4 through the channel-blocked path, where the next block is such a call)."""
    B, N, ci, co = 1, 400, 32, 64
    P, X, W, dY = make_case("room", B, N, ci, co, seed=1450)
    Xbad = X.copy(); Xbad[0, 50, 2] = np.nan
    dYbad = dY.copy(); dYbad[0, 60, 1] = np.inf
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    s = (1, 1, 1)
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
    op.conv3p(t(P), t(Xbad), t(W), s, VOX, cache=cache)
    y = op.conv3p(t(P), t(X), t(W), s, VOX, cache=cache, points_unchanged=True)
    op.conv3p_grad(t(dYbad), t(P), t(Xbad), t(W), s, VOX, cache=cache, points_unchanged=True)
    dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, VOX, cache=cache, points_unchanged=True)
    assert rel_err(y.cpu().numpy(), oracle.forward(P, X, W, s, VOX)) <= 1e-5
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, VOX)
    assert rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5 and rel_err(dw.cpu().numpy(), dw_ref) <= 2e-5


@pytest.mark.parametrize("ci,co,dt", [(300, 70, np.float32), (17, 3, np.float64), (40, 9, np.float64)])
def test_blocked_paths_non_finite_values_and_pair_buffer_overflow(dev, ci, co, dt):
    """The channel-blocked paths of round 4 (more than 256 channels on the matrix-core kernels, fp64 outside the
    register-path shapes on <double, 16, 8>): Inf / NaN inputs land on exactly the outputs they reach in the reference
    (zero padding of the blocks must not turn them into extra NaNs), and a cache whose pair buffer overflows still gives
    the reference's results."""
    B, N = 2, 300
    P, X, W, dY = make_case("room", B, N, ci, co, seed=1400, dtype=dt)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    tol_y, tol_w = (1e-5, 2e-5) if dt == np.float32 else (1e-12, 1e-12)
    s = (1, 1, 1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # pair-buffer overflow: tiles search themselves / are handed to the exact kernels
    cache = op.NeighborCache(B, N, tdt, dev, slots=1, max_taps=27, pairs_per_point=4, max_cin=ci, max_cout=co)
    y, dx, dw = _both(dev, cache, P, X, W, dY, s)
    assert rel_err(y.cpu().numpy(), oracle.forward(P, X, W, s, VOX)) <= tol_y
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, VOX)
    assert rel_err(dx.cpu().numpy(), dx_ref) <= tol_y and rel_err(dw.cpu().numpy(), dw_ref) <= tol_w
    # non-finite values
    X = X.copy(); dY = dY.copy()
    X[0, 17, 3] = np.inf
    X[1, 200, ci - 1] = np.nan
    dY[0, 99, co - 1] = np.inf
    with np.errstate(all="ignore"):
        y_ref = oracle.forward(P, X, W, s, VOX)
        dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, VOX)
    y = op.conv3p(t(P), t(X), t(W), s, VOX).cpu().numpy()
    dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, VOX)
    for name, got, ref, tol in (("y", y, y_ref, tol_y), ("dx", dx.cpu().numpy(), dx_ref, tol_y), ("dw", dw.cpu().numpy(), dw_ref, tol_w)):
        bad_ref = ~np.isfinite(ref)
        bad_got = ~np.isfinite(got)
        assert bad_ref.any() and not bad_ref.all()
        if name == "dw" and dt == np.float64:
            # the dense-G register kernels contract G (zero where a centre has no pair with the tap) with ALL 64 input
            # rows of the tile: a non-finite input channel k marks grad_filter[:, k, :] for every tap (0 x NaN), the
            # reference only for the taps the point takes part in -- a superset, nothing else
            assert np.all(bad_got[bad_ref]) and not bad_got.all()
            extra = bad_got & ~bad_ref
            keep = np.ones(ci, dtype=bool)
            keep[[3, ci - 1]] = False                      # the two input channels that hold the non-finite values
            assert not extra[..., keep, :].any()
        else:
            assert np.array_equal(bad_got, bad_ref)
        ok = ~bad_got
        assert np.max(np.abs(got[ok] - ref[ok])) <= 10 * tol * max(1.0, np.max(np.abs(ref[ok])))


def _fuzz_cases():
    rng = np.random.RandomState(20260927)
    shapes = [(3, 9), (9, 9), (36, 13), (5, 7), (1, 1), (32, 64), (64, 64), (9, 3), (12, 9), (2, 17)]
    kinds = ["modelnet", "room", "cube", "lattice"]
    out = []
    for i in range(28):
        ci, co = shapes[rng.randint(len(shapes))]
        B = int(rng.randint(1, 4))
        N = int(rng.choice([1, 5, 63, 64, 65, 130, 257, 500]))
        f = tuple(int(v) for v in rng.choice([1, 2, 3], size=3))
        st = tuple(int(v) for v in rng.choice([1, 2, 3, 4], size=3))
        dt = np.float64 if (ci, co) in ((5, 7), (1, 1)) and rng.rand() < 0.5 else np.float32
        out.append((kinds[rng.randint(len(kinds))], B, N, ci, co, f, st, dt, 3000 + i))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "%s-B%dN%d-%dto%d-f%s-s%s-%s" % (
    c[0], c[1], c[2], c[3], c[4], "".join(map(str, c[5])), "".join(map(str, c[6])), c[7].__name__))
def test_random_shapes_against_oracle(dev, case):
    """Seeded random mix of cloud kinds, ragged sizes, filter extents, anisotropic strides and channel shapes
    (register path, generic path, deep path, fp64): every kernel family against the oracle."""
    kind, B, N, ci, co, f, st, dt, seed = case
    P, X, W, dY = make_case(kind, B, N, ci, co, f, seed=seed, dtype=dt)
    ref = (oracle.neighbor_count(P, f, st, VOX), oracle.forward(P, X, W, st, VOX)) + oracle.backward(dY, P, X, W, st, VOX)
    check_against(ref, run_hip(dev, P, X, W, dY, st), dt)


def test_bench_prints_one_contract_json_line(dev):
    """bench.py's stdout is ONE JSON line with the driver's fields plus the roofline / cpu_baseline objects."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mpoints/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["unit"] in ("GB/s", "TFLOP/s")
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]


@pytest.mark.parametrize("ci,co,kind,N", [(9, 9, "modelnet", 600), (3, 9, "room", 300), (36, 13, "room", 200), (9, 3, "lattice", 256)])
def test_fp64_register_path_matches_oracle(dev, ci, co, kind, N):
    """The reference registers T in {float, double} (register_op.cpp:44-75): the models' shapes in fp64 take the
    same register-path kernels (templates on T; the 36->13 backward, whose G does not fit LDS in double, runs them in
    three passes over column blocks of the output channels), within the fp64 tolerance, and stay bitwise
    reproducible."""
    B = 2
    P, X, W, dY = make_case(kind, B, N, ci, co, seed=1100, dtype=np.float64)
    s = (2, 1, 2)
    ref = (oracle.neighbor_count(P, (3, 3, 3), s, VOX), oracle.forward(P, X, W, s, VOX)) + oracle.backward(dY, P, X, W, s, VOX)
    got = run_hip(dev, P, X, W, dY, s)
    check_against(ref, got, np.float64)
    again = run_hip(dev, P, X, W, dY, s)
    for a, b in zip(got, again):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ the window-table search at the edges of its tables
@pytest.mark.parametrize("name,shift,vox,kind,N", [
    ("far_from_origin", 5000.0, 0.1, "modelnet", 700),      # coordinates of 5e4 voxels: the look-up slack exceeds a bucket -> every point a candidate
    ("moderately_far", 40.0, 0.1, "modelnet", 700),         # slack of a few hundredths of a bucket: windows stay selective
    ("tiny_voxel", 0.0, 0.02, "modelnet", 600),             # a tile spans hundreds of buckets: coarsened tables (the oracle's grid refuses much smaller)
    ("huge_voxel", 0.0, 7.0, "room", 600),                  # the whole cloud in a fraction of a voxel: one bucket per tile
    ("negative_side", -3.0, 0.1, "room", 900)])
def test_window_tables_at_extreme_scales(dev, name, shift, vox, kind, N):
    """conv3p_search_fused.hpp is a superset filter whose fallbacks (coarser buckets for wide tiles, all candidates
    when fp32 rounding of the coordinates reaches a fraction of a bucket) must never lose a neighbour: populations
    integer-equal to the oracle's and results within tolerance at coordinate / voxel ratios far from the models'."""
    B = 2
    P, X, W, dY = make_case(kind, B, N, 3, 9, seed=2100 + N)
    P = (P.astype(np.float64) + shift).astype(np.float32)
    for s in ((1, 1, 1), (3, 3, 3)):
        ref = (oracle.neighbor_count(P, (3, 3, 3), s, vox), oracle.forward(P, X, W, s, vox)) + oracle.backward(dY, P, X, W, s, vox)
        check_against(ref, run_hip(dev, P, X, W, dY, s, vox), np.float32)


def test_non_finite_points_have_no_neighbours_and_disturb_nobody(dev):
    """Points with a NaN / Inf coordinate: the tables hold them in no window and give them empty masks as centres (the
    tile-pair scan's comparisons did the same); every other point's populations are what they are without them."""
    B, N = 1, 400
    P, X, W, dY = make_case("modelnet", B, N, 3, 9, seed=2200)
    bad = np.array([3, 77, 200, 399])
    P2 = P.copy()
    P2[0, bad[0], 0] = np.nan
    P2[0, bad[1], 1] = np.inf
    P2[0, bad[2], 2] = -np.inf
    P2[0, bad[3], :] = np.nan
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    s = (2, 2, 2)
    cnt = op.neighbor_count(t(P2), (3, 3, 3), s, VOX).cpu().numpy()
    keep = np.setdiff1d(np.arange(N), bad)
    ref = oracle.neighbor_count(np.ascontiguousarray(P[:, keep]), (3, 3, 3), s, VOX)
    assert np.array_equal(cnt[0, keep], ref[0])
    assert not cnt[0, bad].any()
    y = op.conv3p(t(P2), t(X), t(W), s, VOX).cpu().numpy()
    y_ref = oracle.forward(np.ascontiguousarray(P[:, keep]), np.ascontiguousarray(X[:, keep]), W, s, VOX)
    assert rel_err(y[0, keep], y_ref[0]) <= 1e-5 and not y[0, bad].any()


# ------------------------------------------------------------------ which backward kernel: decided on the device by default
@pytest.mark.parametrize("kind,N,expect_sparse", [("modelnet", 2048, True), ("room", 4096, False)])
def test_backward_kernel_choice_on_the_device_equals_the_matching_hint(dev, kind, N, expect_sparse):
    """Without a hint both backward kernels of a dilated 9 -> 9 layer are launched and the regime word the search left
    (short / long pair lists on average) lets exactly one run: the result is bit for bit the hinted kernel's of that
    regime -- ModelNet-shaped clouds: the populated-rows kernel, rooms: the dense-G kernel -- and the oracle's."""
    B, s = 2, (2, 2, 2)
    P, X, W, dY = make_case(kind, B, N, 9, 9, seed=1900 + N)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    res = {}
    for hint in (None, True, False):
        cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=9, max_cout=9, sparse_neighbourhoods=hint)
        res[hint] = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, VOX, cache=cache)
    same = res[True] if expect_sparse else res[False]
    other = res[False] if expect_sparse else res[True]
    assert torch.equal(res[None][0], same[0]) and torch.equal(res[None][1], same[1])
    assert not torch.equal(res[None][1], other[1])      # (the two kernels sum in different orders)
    stateless = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, VOX)     # per-call workspace: the same decision
    assert torch.equal(stateless[0], same[0]) and torch.equal(stateless[1], same[1])
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, VOX)
    assert rel_err(res[None][0].cpu().numpy(), dx_ref) <= 1e-5 and rel_err(res[None][1].cpu().numpy(), dw_ref) <= 2e-5


# ------------------------------------------------------------------ CONV3P_CACHE_SPARSE_NEIGHBOURHOODS (populated-rows backward)
@pytest.mark.parametrize("kind,B,N,ci,co,s,ppp", [
    ("modelnet", 3, 700, 9, 9, (2, 2, 2), 0), ("modelnet", 2, 2048, 9, 9, (4, 4, 4), 0), ("modelnet", 2, 600, 3, 9, (3, 3, 3), 0),
    ("modelnet", 2, 500, 12, 9, (2, 2, 2), 0), ("modelnet", 2, 500, 9, 3, (2, 2, 2), 0), ("modelnet", 2, 333, 6, 9, (1, 2, 3), 0),
    ("room", 2, 1024, 9, 9, (2, 2, 2), 0),            # dense neighbourhoods: tiles take several rounds of taps
    ("lattice", 2, 512, 9, 9, (2, 2, 2), 0),          # every pair on a tap boundary: forward and backward pair sets differ
    ("identical", 1, 300, 9, 9, (2, 2, 2), 0),        # one tap, 64 rows per tile
    ("modelnet", 2, 640, 9, 9, (2, 2, 2), 1),         # pair buffer too small: tiles search themselves inside the kernel
    ("modelnet", 1, 70, 9, 9, (3, 3, 3), 0)])         # a ragged last tile
def test_sparse_neighbourhoods_hint_matches_oracle(dev, kind, B, N, ci, co, s, ppp):
    """The backward kernel the hint selects (conv3p_backward_sparse.hpp) against the oracle, through the cached op
    entry points, plus bitwise reproducibility and the layer form with the fused SELU-gradient epilogue."""
    P, X, W, dY = make_case(kind, B, N, ci, co, seed=1800 + N + ci)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp, tx, tw, tdy = t(P), t(X), t(W), t(dY)
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=2, max_taps=27, pairs_per_point=ppp, max_cin=ci, max_cout=co,
                             sparse_neighbourhoods=True)
    y = op.conv3p(tp, tx, tw, s, VOX, cache=cache)
    dx, dw = op.conv3p_grad(tdy, tp, tx, tw, s, VOX, cache=cache)
    dx2, dw2 = op.conv3p_grad(tdy, tp, tx, tw, s, VOX, cache=cache, points_unchanged=True)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2)
    dx_ref, dw_ref = oracle.backward(dY, P, X, W, s, VOX)
    dw_floor = 0.0
    if kind == "identical":
        d64 = oracle.backward(dY.astype(np.float64), P.astype(np.float64), X.astype(np.float64), W.astype(np.float64), s, VOX)[1]
        dw_floor = rel_err(dw_ref, d64)
    assert rel_err(y.cpu().numpy(), oracle.forward(P, X, W, s, VOX)) <= 1e-5
    assert rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5
    assert rel_err(dw.cpu().numpy(), dw_ref) <= max(2e-5, 4.0 * dw_floor)
    # the hint changes kernels, not results beyond the op's tolerance: against the un-hinted cache
    plain = op.NeighborCache(B, N, torch.float32, dev, slots=2, max_taps=27, pairs_per_point=ppp, max_cin=ci, max_cout=co)
    dx0, dw0 = op.conv3p_grad(tdy, tp, tx, tw, s, VOX, cache=plain)
    assert rel_err(dx.cpu().numpy(), dx0.cpu().numpy()) <= 1e-5 and rel_err(dw.cpu().numpy(), dw0.cpu().numpy()) <= 2e-5
    if ci == 9 and co == 9:
        add = torch.randn_like(tx)
        act_in = op.selu(tx)                              # a SELU output as the layer's input
        g1, w1 = op.conv3p_layer_grad(tdy, tp, act_in, tw, s, VOX, cache, grad_addend=add, points_unchanged=True)
        g0, w0 = op.conv3p_layer_grad(tdy, tp, act_in, tw, s, VOX, plain, grad_addend=add, points_unchanged=True)
        assert rel_err(g1.cpu().numpy(), g0.cpu().numpy()) <= 1e-5 and rel_err(w1.cpu().numpy(), w0.cpu().numpy()) <= 2e-5


def test_stack_tune_sets_the_hint_by_measured_density(dev):
    """Conv3pStack.tune(): ModelNet-shaped clouds (7-11 neighbours at strides 2-4) get the hint, room blocks (~50) do
    not; the tuned classification stack matches the oracle stack."""
    P = synth.modelnet_like(3, 1024, seed=1900)
    st = stack.Conv3pStack(3, None, device=dev, seed=77)
    assert st.tune(torch.from_numpy(P).to(dev)) is True
    acts = st.forward(torch.from_numpy(P).to(dev), torch.from_numpy(P).to(dev))
    ups = [synth.upstream_grad(3, 1024, stack.HIDDEN, 1910 + li) for li in range(4)]
    dx, fused = st.backward([torch.from_numpy(u).to(dev) for u in ups])
    ref_acts, ref_dx, ref_fused = _oracle_stack(P, P.copy(), [f.cpu().numpy() for f in st.filters], st.layers, ups, None)
    for a, r in zip(acts, ref_acts):
        assert rel_err(a.cpu().numpy(), r) <= 1e-5
    assert rel_err(dx.cpu().numpy(), ref_dx) <= 2e-5 and rel_err(fused.cpu().numpy(), ref_fused) <= 2e-5
    rooms = stack.Conv3pStack(9, 13, device=dev, seed=78)
    assert rooms.tune(torch.from_numpy(synth.room_like(2, 4096, 1901)).to(dev)) is False


@pytest.mark.gpu
def test_random_configurations_match_the_oracle(dev):
    """150 random op configurations (fixed seed; tools/fuzz_gpu.py draws the same way, 2 800 of them in round 5): sizes
    that are not multiples of a tile, one-cloud batches, anisotropic strides, 1 / 2 / 3 / 5-tap axes (filters of up to 64
    taps), lattice / identical / isolated clouds, the models' and other channel shapes, fp32 and fp64, stateless and
    cached -- neighbour counts exact, y / dX / dW within the op's tolerance of the oracle.  Where a lattice cloud's many
    coincident points make the reference's own fp32 sums the looser side, both are judged against the exact sums over the
    oracle's pair lists."""
    rng = np.random.default_rng(20260929)
    shapes = [(3, 9), (9, 9), (6, 9), (12, 9), (36, 13), (3, 3), (9, 3), (5, 7), (16, 16), (33, 20)]
    kinds = ["modelnet", "room", "cube", "lattice", "vlattice", "identical", "isolated"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for it in range(150):
        kind = kinds[rng.integers(len(kinds))]
        ci, co = shapes[rng.integers(len(shapes))]
        B = int(rng.integers(1, 5))
        N = int(rng.choice([1, 2, 63, 64, 65, 100, 127, 129, 200, 500, 777, 1024, 1500, 2048, 2500]))
        if kind == "identical" and N > 300:
            N = 300
        if ci * co > 300 and N > 600:
            N = 600
        f = tuple(int(v) for v in rng.choice([1, 2, 3, 3, 3, 5], size=3))
        if f[0] * f[1] * f[2] > 64:
            f = (3, 3, 3)
        s = tuple(int(v) for v in rng.integers(1, 5, size=3)) if rng.random() < 0.4 else (int(rng.integers(1, 5)),) * 3
        dt = np.float64 if rng.random() < 0.15 else np.float32
        P, X, W, dY = make_case(kind, B, N, ci, co, f, seed=1000 + it, dtype=dt)
        cache = None
        if rng.random() < 0.5:
            cache = op.NeighborCache(B, N, torch.float32 if dt == np.float32 else torch.float64, dev, slots=2,
                                     max_taps=f[0] * f[1] * f[2], max_cin=ci, max_cout=co)
        what = (it, kind, B, N, ci, co, f, s, np.dtype(dt).name, "cached" if cache is not None else "stateless")
        cnt = op.neighbor_count(t(P), f, s, VOX).cpu().numpy()
        y = op.conv3p(t(P), t(X), t(W), s, VOX, cache=cache).cpu().numpy()
        dx, dw = op.conv3p_grad(t(dY), t(P), t(X), t(W), s, VOX, cache=cache)
        dx, dw = dx.cpu().numpy(), dw.cpu().numpy()
        assert np.array_equal(cnt, oracle.neighbor_count(P, f, s, VOX)), what
        ry = oracle.forward(P, X, W, s, VOX)
        rdx, rdw = oracle.backward(dY, P, X, W, s, VOX)
        ty, tw = TOL[np.dtype(dt)]
        if rel_err(y, ry) <= ty and rel_err(dx, rdx) <= ty and rel_err(dw, rdw) <= tw:
            continue
        assert dt == np.float32, what
        dwe = 0.0
        for b in range(B):
            ye, dxe, dwb = exact_from_oracle_lists(P[b], X[b], W, dY[b], s, VOX)
            assert rel_err(y[b], ye) <= ty and rel_err(dx[b], dxe) <= ty, what
            dwe = dwe + dwb
        assert rel_err(dw, dwe) <= tw or rel_err(dw, dwe) <= rel_err(rdw, dwe), what
