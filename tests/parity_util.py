"""Shared helpers of the parity tests: tolerances, case generation, HIP-vs-oracle comparison."""
import numpy as np

from pointwise_amd import synth

# Stated fp32 tolerance of the HIP path against the CPU reference restatement:
#   |dy|, |ddX| <= 1e-5 * max(1, max|ref|)   elementwise
#   |ddW|       <= 2e-5 * max(1, max|dW_ref|)
# fp64: 1e-12 on the same scales.  Neighbour / tap decisions (the int32 count tensor): exact.
TOL = {np.dtype(np.float32): (1e-5, 2e-5), np.dtype(np.float64): (1e-12, 1e-12)}


def rel_err(got, ref):
    ref = np.asarray(ref)
    scale = max(1.0, float(np.abs(ref).max())) if ref.size else 1.0
    return float(np.abs(np.asarray(got) - ref).max() / scale) if ref.size else 0.0


def make_case(kind, B, N, Cin, Cout, filter_zyx=(3, 3, 3), seed=0, dtype=np.float32, voxel=0.1):
    if kind == "modelnet":
        P = synth.modelnet_like(B, N, seed)
    elif kind == "room":
        P = synth.room_like(B, N, seed)
    elif kind == "cube":
        P = synth.uniform_cube(B, N, seed)
    elif kind == "lattice":
        P = synth.lattice(B, N, seed, voxel=voxel, span=6)
    elif kind == "vlattice":   # voxel-aligned: multiples of the voxel itself
        P = synth.lattice(B, N, seed, voxel=voxel, span=8, div=1)
    elif kind == "identical":
        P = np.full((B, N, 3), 0.25, dtype=np.float32)
    elif kind == "isolated":
        i = np.arange(N)
        P = np.stack([i % 5, (i // 5) % 5, i // 25], axis=1)[None].repeat(B, 0).astype(np.float64) * 0.7
        P = P.astype(np.float32)   # spacing 0.7 > any tested box half-width: only self-pairs
    else:
        raise ValueError(kind)
    P = P.astype(dtype)
    X = synth.features(B, N, Cin, seed + 1, points=P if Cin >= 3 else None, dtype=dtype)
    W = synth.filter_weights(*filter_zyx, Cin, Cout, seed + 2, dtype=dtype)
    dY = synth.upstream_grad(B, N, Cout, seed + 3, dtype=dtype)
    return P, X, W, dY


def exact_from_oracle_lists(cloud, X, W, dY, stride, voxel):
    """y, dX, dW of ONE cloud accumulated in float64 over the ORACLE's pair lists -- the reference's single-precision
    neighbour / tap / count decisions (oracle.neighbor_lists, oracle.backward_pairs: bit-pinned to the reference Grid),
    its sums without their single-precision rounding.  (The double-precision op is not that: run on the same float32
    coordinates it re-decides the pairs in double, and a pair within an ulp of a tap boundary may fall on the other side
    -- one such pair in cloud 7 of the cfg5 shard moves an output by 1.3e-2.)  Per tap: a sparse selection matrix with
    the 1 / count weights times the feature rows, then one dense product with the tap's filter block."""
    import scipy.sparse as sp
    from oracle import oracle
    fz, fy, fx, Cin, Cout = W.shape
    F, N = fz * fy * fx, cloud.shape[0]
    X64, W64, dY64 = X.astype(np.float64), W.reshape(F, Cin, Cout).astype(np.float64), dY.astype(np.float64)
    off, idx, tap = oracle.neighbor_lists(np.ascontiguousarray(cloud, dtype=np.float32), (fz, fy, fx), stride, voxel)
    ci = np.repeat(np.arange(N, dtype=np.int64), np.diff(off))
    cnt = np.bincount(ci * F + tap, minlength=N * F).reshape(N, F).astype(np.float64)
    y = np.zeros((N, Cout))
    for f in range(F):
        m = tap == f
        if not m.any():
            continue
        S = sp.csr_matrix((1.0 / cnt[ci[m], f], (ci[m], idx[m])), shape=(N, N))
        y += (S @ X64) @ W64[f]
    j, ii, fb, count = oracle.backward_pairs(np.ascontiguousarray(cloud, dtype=np.float32), (fz, fy, fx), stride, voxel)
    dX, dW = np.zeros((N, Cin)), np.zeros((F, Cin, Cout))
    for f in range(F):
        m = (fb == f) & (count > 0)
        if not m.any():
            continue
        T = sp.csr_matrix((1.0 / count[m].astype(np.float64), (j[m], ii[m])), shape=(N, N))
        G = T @ dY64                       # G_f'[j] = sum over the pairs of dY[ii] / count
        dX += G @ W64[f].T
        dW[f] = X64.T @ G
    return y, dX, dW.reshape(W.shape)
