"""CPU tests of the oracle (not gpu): pinning against the reference Grid and the golden fixtures, the
independent numpy restatement, and the structural properties the reference algorithm has."""
import glob
import os

import numpy as np
import pytest

from oracle import conv3p_numpy as cn
from oracle import oracle
from pointwise_amd import synth
from tests.parity_util import make_case, rel_err

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith("prestep_"))
VOX = 0.1


def test_golden_fixtures_exist():
    assert len(GOLDEN) >= 24


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    """Neighbour lists / taps / counts produced by the REFERENCE Grid and y/dX/dW produced by the REFERENCE's
    own batch loops (both stored in the fixture, tests/golden/make_golden.py) must be reproduced exactly:
    lists in the reference's visit order, floating-point outputs bit-for-bit."""
    g = np.load(path)
    P, X, W, dY = g["points"], g["input"], g["filter"], g["grad_out"]
    s, vox = tuple(int(v) for v in g["stride"]), float(g["voxel"])
    fzyx = W.shape[:3]
    pos = 0
    for b in range(P.shape[0]):
        off, idx, tap = oracle.neighbor_lists(P[b], fzyx, s, vox)
        n = int(g["ref_pairs_per_cloud"][b])
        assert np.array_equal(off, g["ref_offsets"][b])
        assert np.array_equal(idx, g["ref_index"][pos:pos + n])
        assert np.array_equal(tap, g["ref_tap"][pos:pos + n])
        pos += n
    assert np.array_equal(oracle.neighbor_count(P, fzyx, s, vox), g["ref_count"])
    y = oracle.forward(P, X, W, s, vox)
    dx, dw = oracle.backward(dY, P, X, W, s, vox)
    assert np.array_equal(y, g["y"]) and np.array_equal(dx, g["dX"]) and np.array_equal(dw, g["dW"])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_reference_loops_reproduce_golden_live(path):
    """oracle/_ref/libref_compute_* (the reference's loop text compiled in place) still produces the stored
    y/dX/dW: the fixtures really are reference output.  Skipped where _ref is absent (never on the build box)."""
    g = np.load(path)
    kind = str(g["ref_kind"])
    if oracle.ref_compute(kind) is None:
        pytest.skip("oracle/_ref not built (no /root/reference at build time)")
    P, X, W, dY = g["points"], g["input"], g["filter"], g["grad_out"]
    s, vox = tuple(int(v) for v in g["stride"]), float(g["voxel"])
    y = oracle.reference_forward(P, X, W, s, vox, kind=kind)
    dx, dw = oracle.reference_backward(dY, P, X, W, s, vox, kind=kind)
    assert np.array_equal(y, g["y"]) and np.array_equal(dx, g["dX"]) and np.array_equal(dw, g["dW"])


@pytest.mark.parametrize("kind", ["modelnet", "lattice", "room", "cube", "identical"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("stride", [1, 2, 3, 4])
def test_oracle_matches_reference_loops_live(kind, dtype, stride):
    """Fresh seeds: oracle.forward / backward (serial) are BIT-EQUAL to the reference's own accumulation loops
    (tf_conv3p_atrous.cpp:451-504, :608-716) -- the per-term division, the c-outer / k-inner order, the
    re-binning without inclusion re-test, the count == 0 skip."""
    if oracle.ref_compute("atrous") is None:
        pytest.skip("oracle/_ref not built (no /root/reference at build time)")
    ci, co = (3, 9) if stride == 1 else (9, 9)
    P, X, W, dY = make_case(kind, 2, 200, ci, co, seed=2000 + stride, dtype=dtype)
    s = (stride, stride, stride)
    y, (dx, dw) = oracle.forward(P, X, W, s, VOX), oracle.backward(dY, P, X, W, s, VOX)
    ry = oracle.reference_forward(P, X, W, s, VOX)
    rdx, rdw = oracle.reference_backward(dY, P, X, W, s, VOX)
    assert np.array_equal(y, ry) and np.array_equal(dx, rdx) and np.array_equal(dw, rdw)


@pytest.mark.parametrize("fzyx,s,ci,co", [((2, 1, 3), (1, 2, 3), 4, 5), ((5, 5, 5), (1, 1, 1), 2, 3),
                                          ((2, 2, 2), (1, 1, 1), 3, 2), ((4, 4, 4), (2, 2, 2), 2, 2),
                                          ((1, 1, 1), (1, 1, 1), 3, 4), ((3, 3, 3), (1, 2, 4), 5, 7),
                                          ((3, 3, 3), (1, 1, 1), 36, 41), ((3, 3, 3), (1, 1, 1), 32, 64)])
def test_oracle_matches_reference_loops_odd_shapes(fzyx, s, ci, co):
    if oracle.ref_compute("atrous") is None:
        pytest.skip("oracle/_ref not built")
    for kind in ("cube", "lattice"):
        for dtype in (np.float32, np.float64):
            P, X, W, dY = make_case(kind, 1, 150, ci, co, fzyx, seed=17, dtype=dtype)
            y, (dx, dw) = oracle.forward(P, X, W, s, VOX), oracle.backward(dY, P, X, W, s, VOX)
            ry = oracle.reference_forward(P, X, W, s, VOX)
            rdx, rdw = oracle.reference_backward(dY, P, X, W, s, VOX)
            assert np.array_equal(y, ry) and np.array_equal(dx, rdx) and np.array_equal(dw, rdw)


def test_non_atrous_reference_loops_equal_stride_one():
    """tf_conv3p_grid.cpp's loops (the non-atrous op) give the atrous op's stride-(1,1,1) result bit-for-bit."""
    if oracle.ref_compute("plain") is None:
        pytest.skip("oracle/_ref not built")
    P, X, W, dY = make_case("modelnet", 2, 300, 3, 9, seed=5)
    a = oracle.reference_forward(P, X, W, (1, 1, 1), VOX, kind="plain")
    b = oracle.reference_forward(P, X, W, (1, 1, 1), VOX, kind="atrous")
    assert np.array_equal(a, b) and np.array_equal(a, oracle.forward(P, X, W, (1, 1, 1), VOX))
    da, db = oracle.reference_backward(dY, P, X, W, (1, 1, 1), VOX, kind="plain"), oracle.reference_backward(dY, P, X, W, (1, 1, 1), VOX, kind="atrous")
    assert np.array_equal(da[0], db[0]) and np.array_equal(da[1], db[1])


def test_openmp_reference_build_matches_serial():
    """The reference built like its CPU object (-fopenmp -DCONV_OPENMP; what bench.py times as the CPU
    baseline): y and dX bit-equal to the serial build, dW up to the order of the per-thread partials."""
    if oracle.ref_compute("atrous_omp") is None:
        pytest.skip("oracle/_ref not built")
    P, X, W, dY = make_case("modelnet", 4, 256, 9, 9, seed=61)
    s = (2, 2, 2)
    y = oracle.reference_forward(P, X, W, s, VOX, kind="atrous_omp")
    dx, dw = oracle.reference_backward(dY, P, X, W, s, VOX, kind="atrous_omp")
    assert np.array_equal(y, oracle.forward(P, X, W, s, VOX))
    odx, odw = oracle.backward(dY, P, X, W, s, VOX)
    assert np.array_equal(dx, odx) and rel_err(dw, odw) < 1e-5


def test_even_extent_cell_window_drops_box_edge_candidates():
    """On voxel-aligned clouds with EVEN dilated extents the reference's cell window (.cpp:247-266) rejects
    candidates that pass the inclusive box test: the fixtures must contain such cases, so that an
    all-pairs implementation cannot pass them."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "even_lattice_f32.npz"))
    P, vox = g["points"], float(np.float32(g["voxel"]))   # T voxel_size: the float nearest 0.1 (.cpp:444)
    ref_pairs = int(g["ref_pairs_per_cloud"].sum())
    brute = 0
    for b in range(P.shape[0]):
        p = P[b]
        lo = (p.astype(np.float64) - 1.0 * vox).astype(np.float32)     # full = 2: half extent = 1 voxel
        hi = (p.astype(np.float64) + 1.0 * vox).astype(np.float32)
        inside = np.all((p[None, :, :] >= lo[:, None, :]) & (p[None, :, :] <= hi[:, None, :]), axis=2)
        brute += int(inside.sum())
    assert brute > ref_pairs, (brute, ref_pairs)


@pytest.mark.parametrize("kind", ["modelnet", "lattice", "room", "cube", "identical"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("stride", [1, 2, 3, 4])
def test_oracle_matches_reference_grid_live(kind, dtype, stride):
    """Fresh seeds against oracle/_ref (the reference Grid compiled in place); skipped where _ref is absent."""
    if oracle.ref_grid("atrous") is None:
        pytest.skip("oracle/_ref not built (no /root/reference at build time)")
    P = make_case(kind, 2, 300, 3, 3, seed=1000 + stride, dtype=dtype)[0]
    s = (stride, stride, stride)
    for b in range(P.shape[0]):
        ours = oracle.neighbor_lists(P[b], (3, 3, 3), s, VOX)
        ref = oracle.reference_grid_lists(P[b], (3, 3, 3), s, VOX)
        for a, r in zip(ours, ref[:3]):
            assert np.array_equal(a, r)
        assert np.array_equal(oracle.neighbor_count(P[b:b + 1], (3, 3, 3), s, VOX)[0], ref[3])


@pytest.mark.parametrize("fzyx,s", [((2, 1, 3), (1, 2, 3)), ((5, 5, 5), (1, 1, 1)), ((2, 2, 2), (1, 1, 1)),
                                    ((4, 4, 4), (2, 2, 2)), ((1, 1, 1), (1, 1, 1)), ((3, 3, 3), (1, 2, 4))])
def test_oracle_matches_reference_grid_odd_filters(fzyx, s):
    if oracle.ref_grid("atrous") is None:
        pytest.skip("oracle/_ref not built")
    for kind in ("cube", "lattice"):
        P = make_case(kind, 1, 250, 3, 3, seed=7, dtype=np.float32)[0]
        ours = oracle.neighbor_lists(P[0], fzyx, s, VOX)
        ref = oracle.reference_grid_lists(P[0], fzyx, s, VOX)
        for a, r in zip(ours, ref[:3]):
            assert np.array_equal(a, r)
        assert np.array_equal(oracle.neighbor_count(P, fzyx, s, VOX)[0], ref[3])


def test_stride_one_equals_non_atrous_reference():
    """The non-atrous op (tf_conv3p_grid.cpp) is the stride (1,1,1) case of the atrous one."""
    if oracle.ref_grid("plain") is None:
        pytest.skip("oracle/_ref not built")
    P = make_case("modelnet", 1, 400, 3, 3, seed=3)[0]
    a = oracle.reference_grid_lists(P[0], (3, 3, 3), (1, 1, 1), VOX, kind="plain")
    b = oracle.reference_grid_lists(P[0], (3, 3, 3), (1, 1, 1), VOX, kind="atrous")
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    ours = oracle.neighbor_lists(P[0], (3, 3, 3), (1, 1, 1), VOX)
    for x, y in zip(ours, a[:3]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("kind", ["modelnet", "lattice", "room"])
@pytest.mark.parametrize("cfg", [((3, 3, 3), (1, 1, 1), 3, 9), ((3, 3, 3), (2, 2, 2), 9, 9),
                                 ((3, 3, 3), (4, 4, 4), 9, 9), ((2, 1, 3), (1, 2, 3), 4, 5),
                                 ((4, 4, 4), (2, 2, 2), 2, 2)])
def test_c_oracle_vs_numpy_restatement(kind, cfg):
    """Two structurally different restatements (grid search in C, brute force in numpy) agree to rounding
    in float64 -- including on lattice data where forward and backward pair sets differ."""
    fzyx, s, ci, co = cfg
    P, X, W, dY = make_case(kind, 2, 200, ci, co, fzyx, seed=11, dtype=np.float64)
    y, (dx, dw) = oracle.forward(P, X, W, s, VOX), oracle.backward(dY, P, X, W, s, VOX)
    y2, (dx2, dw2) = cn.forward(P, X, W, s, VOX), cn.backward(dY, P, X, W, s, VOX)
    assert rel_err(y, y2) < 1e-13 and rel_err(dx, dx2) < 1e-13 and rel_err(dw, dw2) < 1e-13


def test_backward_is_adjoint_on_generic_data():
    """<dY, conv(X, W)> is bilinear in (X, W); on generic data (no point on a tap boundary) the reference
    backward is its exact adjoint:  <dY, conv(dXdir, W)> = <dX, dXdir>  and  <dY, conv(X, dWdir)> = <dW, dWdir>."""
    P, X, W, dY = make_case("modelnet", 2, 256, 9, 9, seed=21, dtype=np.float64)
    s = (2, 2, 2)
    dx, dw = oracle.backward(dY, P, X, W, s, VOX)
    rng = np.random.default_rng(5)
    Xd, Wd = rng.normal(size=X.shape), rng.normal(size=W.shape)
    lhs_x = float((dY * oracle.forward(P, Xd, W, s, VOX)).sum())
    lhs_w = float((dY * oracle.forward(P, X, Wd, s, VOX)).sum())
    assert abs(lhs_x - float((dx * Xd).sum())) < 1e-10 * max(1.0, abs(lhs_x))
    assert abs(lhs_w - float((dw * Wd).sum())) < 1e-10 * max(1.0, abs(lhs_w))


def test_lattice_backward_pair_set_differs_from_forward():
    """On lattice-aligned clouds the reference backward visits a different pair set from its forward
    (no inclusion re-test, count==0 skip) -- the oracle must reproduce that."""
    P = synth.lattice(1, 1024, 4, voxel=VOX, span=8)
    s = (2, 2, 2)
    off, idx, tap = oracle.neighbor_lists(P[0], (3, 3, 3), s, VOX)
    i_of = np.repeat(np.arange(1024), np.diff(off))
    fwd = set(zip(i_of.tolist(), idx.tolist(), tap.tolist()))          # (centre, neighbour, tap)
    j, ii, f, cnt = oracle.backward_pairs(P[0], (3, 3, 3), s, VOX)
    bwd = set(zip(ii.tolist(), j.tolist(), f.tolist()))                # same orientation
    assert len(bwd - fwd) > 0 and len(fwd - bwd) > 0
    assert (cnt > 0).all()


def test_generic_backward_pair_set_equals_forward():
    P = synth.uniform_cube(1, 600, 9)
    for st in (1, 2, 3):
        s = (st, st, st)
        off, idx, tap = oracle.neighbor_lists(P[0], (3, 3, 3), s, VOX)
        i_of = np.repeat(np.arange(600), np.diff(off))
        fwd = set(zip(i_of.tolist(), idx.tolist(), tap.tolist()))
        j, ii, f, _ = oracle.backward_pairs(P[0], (3, 3, 3), s, VOX)
        assert fwd == set(zip(ii.tolist(), j.tolist(), f.tolist()))


def test_self_is_centre_tap_for_odd_filters():
    """Contract item 8: for odd extents every point is its own neighbour in the centre tap."""
    P = synth.modelnet_like(1, 300, 2)
    for st in (1, 4):
        cnt = oracle.neighbor_count(P, (3, 3, 3), (st, st, st), VOX)
        assert (cnt[0, :, 13] >= 1).all()


def test_permutation_equivariance_and_batch_independence():
    P, X, W, dY = make_case("modelnet", 3, 200, 3, 9, seed=31, dtype=np.float64)
    s = (1, 1, 1)
    y = oracle.forward(P, X, W, s, VOX)
    perm = np.random.default_rng(1).permutation(200)
    y_p = oracle.forward(P[:, perm], X[:, perm], W, s, VOX)
    assert rel_err(y_p, y[:, perm]) < 1e-13
    y_1 = oracle.forward(P[1:2], X[1:2], W, s, VOX)
    assert np.array_equal(y_1[0], y[1])


def test_linearity():
    P, X, W, dY = make_case("room", 1, 300, 9, 9, seed=41, dtype=np.float64)
    s = (2, 2, 2)
    X2 = np.random.default_rng(2).normal(size=X.shape)
    a = oracle.forward(P, 2.0 * X + X2, W, s, VOX)
    b = 2.0 * oracle.forward(P, X, W, s, VOX) + oracle.forward(P, X2, W, s, VOX)
    assert rel_err(a, b) < 1e-13


def test_openmp_variant_matches_serial():
    """Forward and grad_input do not depend on the thread count; grad_filter differs only by the
    order the per-thread partials are added (.cpp:709-716)."""
    P, X, W, dY = make_case("modelnet", 8, 256, 9, 9, seed=51)
    s = (2, 2, 2)
    y1, y4 = oracle.forward(P, X, W, s, VOX, nthreads=1), oracle.forward(P, X, W, s, VOX, nthreads=4)
    assert np.array_equal(y1, y4)
    dx1, dw1 = oracle.backward(dY, P, X, W, s, VOX, nthreads=1)
    dx4, dw4 = oracle.backward(dY, P, X, W, s, VOX, nthreads=4)
    assert np.array_equal(dx1, dx4)
    assert rel_err(dw4, dw1) < 1e-5


def test_empty_inputs():
    W = synth.filter_weights(3, 3, 3, 3, 9, 1)
    y = oracle.forward(np.zeros((0, 10, 3), np.float32), np.zeros((0, 10, 3), np.float32), W, (1, 1, 1), VOX)
    assert y.shape == (0, 10, 9)
    y = oracle.forward(np.zeros((2, 0, 3), np.float32), np.zeros((2, 0, 3), np.float32), W, (1, 1, 1), VOX)
    assert y.shape == (2, 0, 9)
    dx, dw = oracle.backward(np.zeros((2, 0, 9), np.float32), np.zeros((2, 0, 3), np.float32),
                             np.zeros((2, 0, 3), np.float32), W, (1, 1, 1), VOX)
    assert dx.shape == (2, 0, 3) and not dw.any()


def test_neighbour_statistics_match_survey():
    """Sanity anchor on the statistics SURVEY.md 8(a) measured with the reference Grid on unit-sphere surface
    clouds of 2048 points (mean neighbours/point 18.5 / 10.7 / 8.1 / 6.6 for stride 1-4)."""
    rng = np.random.default_rng(0)
    v = rng.normal(size=(1, 2048, 3))
    v /= np.linalg.norm(v, axis=2, keepdims=True)
    P = v.astype(np.float32)
    means = [oracle.neighbor_count(P, (3, 3, 3), (s, s, s), VOX).sum() / 2048 for s in (1, 2, 3, 4)]
    for got, want in zip(means, (18.5, 10.7, 8.1, 6.6)):
        assert abs(got - want) / want < 0.1, (means,)
