"""Synthetic point clouds shaped like the reference's datasets (no dataset, h5py or network here).

The reference feeds ModelNet40 clouds normalised to the unit ball with points == features
(/root/reference/modelnet_provider.py:212-213, jitter :64-75) and S3DIS / SceneNN room blocks
whose first three feature channels are xyz (/root/reference/scene_seg/s3dis_provider.py:111-112).
These generators reproduce those *shapes and densities* with seeded numpy so that the CPU oracle,
the HIP path, the parity tests and bench.py all see identical inputs.
"""
import numpy as np


def _unit_ball(pts):
    pts = pts - pts.mean(axis=0, keepdims=True)
    r = np.sqrt((pts ** 2).sum(axis=1)).max()
    return pts / max(r, 1e-12)


def modelnet_like(B, N, seed, jitter=True, dtype=np.float32):
    """Surface samples of random sphere / box / cylinder mixtures, normalised to the unit ball."""
    rng = np.random.default_rng(seed)
    out = np.empty((B, N, 3), dtype=np.float64)
    for b in range(B):
        kind = rng.integers(0, 3)
        if kind == 0:  # sphere surface, anisotropically scaled
            v = rng.normal(size=(N, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            v *= rng.uniform(0.4, 1.0, size=(1, 3))
        elif kind == 1:  # box surface
            v = rng.uniform(-1, 1, size=(N, 3))
            face = rng.integers(0, 3, size=N)
            sign = rng.choice([-1.0, 1.0], size=N)
            v[np.arange(N), face] = sign
            v *= rng.uniform(0.3, 1.0, size=(1, 3))
        else:  # cylinder: side + caps
            th = rng.uniform(0, 2 * np.pi, size=N)
            h = rng.uniform(-1, 1, size=N)
            rad = np.ones(N)
            cap = rng.random(N) < 0.25
            rad[cap] = np.sqrt(rng.random(cap.sum()))
            h[cap] = rng.choice([-1.0, 1.0], size=cap.sum())
            v = np.stack([rad * np.cos(th), rad * np.sin(th), h * rng.uniform(0.5, 1.5)], axis=1)
        v = _unit_ball(v)
        if jitter:  # modelnet_provider.py:64-75: N(0, 0.01) clipped to +-0.05
            v = v + np.clip(rng.normal(0, 0.01, size=v.shape), -0.05, 0.05)
        out[b] = v
    return out.astype(dtype)


def room_like(B, N, seed, extent=(1.0, 1.0, 3.0), dtype=np.float32):
    """S3DIS / SceneNN-like block: floor, ceiling, two walls and clutter, xyz in metres."""
    rng = np.random.default_rng(seed)
    ex = np.asarray(extent, dtype=np.float64)
    out = np.empty((B, N, 3), dtype=np.float64)
    for b in range(B):
        v = rng.uniform(0, 1, size=(N, 3)) * ex
        which = rng.random(N)
        floor = which < 0.25
        ceil = (which >= 0.25) & (which < 0.40)
        wall_x = (which >= 0.40) & (which < 0.60)
        wall_y = (which >= 0.60) & (which < 0.75)
        v[floor, 2] = rng.normal(0.0, 0.004, floor.sum())
        v[ceil, 2] = ex[2] + rng.normal(0.0, 0.004, ceil.sum())
        v[wall_x, 0] = rng.normal(0.0, 0.004, wall_x.sum())
        v[wall_y, 1] = ex[1] + rng.normal(0.0, 0.004, wall_y.sum())
        clutter = which >= 0.75
        nclut = int(clutter.sum())
        centres = rng.uniform(0.2, 0.8, size=(4, 3)) * ex * np.array([1, 1, 0.4])
        pick = rng.integers(0, 4, size=nclut)
        v[clutter] = centres[pick] + rng.normal(0, 0.08, size=(nclut, 3))
        out[b] = v
    return out.astype(dtype)


def uniform_cube(B, N, seed, half=0.5, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return rng.uniform(-half, half, size=(B, N, 3)).astype(dtype)


def lattice(B, N, seed, voxel=0.1, span=12, dtype=np.float32, div=2):
    """Coordinates on integer multiples of voxel/div.  div=2: every point sits on box edges / tap
    boundaries of its neighbours, which exercises the inclusive test, the clamp, the hole test
    and the backward's count==0 skip.  div=1 (voxel-aligned): with EVEN
    dilated extents candidates lie exactly on the box edge AND on the border of the reference's
    cell window (tf_conv3p_atrous.cpp:247-266), which then drops some of them by rounding."""
    rng = np.random.default_rng(seed)
    k = rng.integers(-span, span + 1, size=(B, N, 3))
    return (k.astype(np.float64) * (np.float64(np.float32(voxel)) / float(div))).astype(dtype)


def features(B, N, C, seed, points=None, dtype=np.float32):
    """Per-point features: xyz first (as the providers do) when points is given, U[0,1) after."""
    rng = np.random.default_rng(seed)
    f = rng.uniform(0, 1, size=(B, N, C))
    if points is not None:
        m = min(3, C)
        f[:, :, :m] = points[:, :, :m]
    return f.astype(dtype)


def filter_weights(fz, fy, fx, Cin, Cout, seed, dtype=np.float32):
    """U(-a, a), a = sqrt(3 / (taps * Cin))  (variance-preserving; the reference uses tf.get_variable
    defaults, /root/reference/pointcnn2_acsd.py:50)."""
    rng = np.random.default_rng(seed)
    a = np.sqrt(3.0 / (fz * fy * fx * max(Cin, 1)))
    return rng.uniform(-a, a, size=(fz, fy, fx, Cin, Cout)).astype(dtype)


def upstream_grad(B, N, C, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, size=(B, N, C)).astype(dtype)
