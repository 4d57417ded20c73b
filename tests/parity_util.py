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
