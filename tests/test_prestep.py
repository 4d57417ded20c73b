"""The providers' host pre-step: rotate / jitter / sort-by-xyz.

CPU: the numpy restatement (oracle/prestep_numpy.py) against fixtures produced by the reference's own functions
(tests/golden/make_prestep_golden.py).  GPU (-m gpu): the HIP kernels behind pointwise_amd/prestep.py against the
same fixtures and against the restatement on fresh seeds.
Bars: sorting bit-exact; augmentation within ONE float32 ulp of the reference (the reference's 3x3 product runs in
BLAS in float64, whose rounding in the 16th digit can move the float32 result by one ulp)."""
import os

import numpy as np
import pytest

from oracle import prestep_numpy as ref
from pointwise_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return int(np.abs(a - b).max())


def test_restatement_matches_reference_augmentation():
    g = np.load(os.path.join(G, "prestep_augment.npz"))
    rot = ref.rotate_point_cloud_by_angles(g["points"], g["angles"])
    assert np.array_equal(rot, g["rotated"])
    assert np.array_equal(ref.jitter_point_cloud(rot, g["noise"]), g["jittered"])
    fixed = ref.rotate_point_cloud_by_angles(g["points"], [float(g["fixed_angle"])] * g["points"].shape[0])
    assert np.array_equal(fixed, g["rotated_fixed"])


def test_restatement_matches_reference_sort():
    g = np.load(os.path.join(G, "prestep_sort.npz"))
    for name in ("generic", "lattice_unique"):
        assert np.array_equal(ref.sort_point_cloud_xyz(g[name + "_in"]), g[name + "_sorted"])
    s, a = ref.sort_point_cloud_xyz2(g["room9_labels_in"], g["room9_labels_attr"])
    assert np.array_equal(s, g["room9_labels_sorted"]) and np.array_equal(a, g["room9_labels_attr_sorted"])


def test_sorted_is_lexicographic():
    g = np.load(os.path.join(G, "prestep_sort.npz"))
    s = g["lattice_unique_sorted"]
    for b in range(s.shape[0]):
        keys = [tuple(r) for r in s[b]]
        assert keys == sorted(keys)


# ------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def dev():
    import torch
    from pointwise_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_hip_augmentation_matches_reference(dev):
    import torch
    from pointwise_amd import prestep
    g = np.load(os.path.join(G, "prestep_augment.npz"))
    P = torch.from_numpy(g["points"]).to(dev)
    rot = prestep.rotate_point_cloud(P, g["angles"])
    assert ulp_diff(rot.cpu().numpy(), g["rotated"]) <= 1
    assert ulp_diff(prestep.rotate_point_cloud_by_angle(P, float(g["fixed_angle"])).cpu().numpy(), g["rotated_fixed"]) <= 1
    noise = torch.from_numpy(g["noise"]).to(dev)
    jit = prestep.jitter_point_cloud(torch.from_numpy(g["rotated"]).to(dev), noise=noise)
    assert np.array_equal(jit.cpu().numpy(), g["fed"])                     # no matrix product here: exact
    both = prestep.rotate_and_jitter(P, g["angles"], noise=noise)
    assert ulp_diff(both.cpu().numpy(), g["fed"]) <= 1
    assert float((both - P).abs().max()) > 0.01                            # it did something


@pytest.mark.gpu
def test_hip_sort_matches_reference(dev):
    import torch
    from pointwise_amd import prestep
    g = np.load(os.path.join(G, "prestep_sort.npz"))
    for name in ("generic", "lattice_unique"):
        got = prestep.sort_point_cloud_xyz(torch.from_numpy(g[name + "_in"]).to(dev))
        assert np.array_equal(got.cpu().numpy(), g[name + "_sorted"])
    s, a = prestep.sort_point_cloud_xyz2(torch.from_numpy(g["room9_labels_in"]).to(dev),
                                         torch.from_numpy(g["room9_labels_attr"]).to(dev))
    assert np.array_equal(s.cpu().numpy(), g["room9_labels_sorted"])
    assert np.array_equal(a.cpu().numpy(), g["room9_labels_attr_sorted"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,K", [(1, 1, 3), (3, 63, 3), (2, 2048, 3), (2, 4096, 9), (1, 8192, 12)])
def test_hip_sort_fresh_seeds(dev, B, N, K):
    """Model sizes (ModelNet 2048, S3DIS 4096 x 9 channels, SceneNN 8192 x 12) against the restatement; int64 labels."""
    import torch
    from pointwise_amd import prestep
    P = synth.room_like(B, N, 600 + N)
    F = synth.features(B, N, K, 601 + N, points=P)
    lab = np.random.default_rng(N).integers(0, 41, size=(B, N)).astype(np.int64)
    s_ref, l_ref = ref.sort_point_cloud_xyz2(F, lab)
    s, l = prestep.sort_point_cloud_xyz2(torch.from_numpy(F).to(dev), torch.from_numpy(lab).to(dev))
    assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(l.cpu().numpy(), l_ref)
    again = prestep.sort_point_cloud_xyz(s)                                  # idempotent
    assert torch.equal(again, s)


@pytest.mark.gpu
def test_hip_sort_puts_every_nan_last_like_numpy(dev):
    """numpy.argsort (util.py:66-68) orders NaN after +inf whatever its sign bit; so does the key of the HIP sort."""
    import torch
    from pointwise_amd import prestep
    rng = np.random.default_rng(5)
    P = rng.uniform(-1, 1, size=(1, 64, 3)).astype(np.float32)
    neg_nan = np.frombuffer(np.uint32(0xFFC00000).tobytes(), dtype=np.float32)[0]
    P[0, 3, 0] = neg_nan
    P[0, 10, 0] = np.nan
    P[0, 20, 0] = np.inf
    want = ref.sort_point_cloud_xyz(P)
    got = prestep.sort_point_cloud_xyz(torch.from_numpy(P).to(dev)).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got, nan=123.0), np.nan_to_num(want, nan=123.0))
    assert np.isnan(got[0, -1, 0]) and np.isnan(got[0, -2, 0]) and np.isinf(got[0, -3, 0])


@pytest.mark.gpu
def test_hip_prestep_rejects_bad_input(dev):
    import torch
    from pointwise_amd import prestep
    from pointwise_amd.conv3p_op import Conv3pInvalidArgument, Conv3pRuntimeError
    P = torch.zeros((2, 16, 3), device=dev)
    with pytest.raises(Conv3pInvalidArgument):
        prestep.jitter_point_cloud(P, clip=0.0)
    with pytest.raises(Conv3pRuntimeError):
        prestep.sort_point_cloud_xyz(torch.zeros((1, 9000, 3), device=dev))   # N > 8192: unsupported, says so
    assert prestep.sort_point_cloud_xyz(torch.zeros((0, 5, 3), device=dev)).shape == (0, 5, 3)
