"""Generate the golden fixtures under tests/golden/ (run in the BUILD container, where /root/reference exists).

    python tests/golden/make_golden.py

What a fixture holds, and where each part comes from -- read this before trusting a number:

  inputs           points / input / filter / grad_out / stride / voxel: seeded synthetic data
                   (pointwise_amd.synth; seeds in CASES below).
  ref_offsets,     OUTPUT OF THE REFERENCE'S OWN CODE: the neighbour lists (visit order preserved), tap
  ref_index,       indices and per-tap populations produced by the reference Grid template
  ref_tap,         (/root/reference/tf_ops/conv3p/tf_conv3p_atrous.cpp:138-388, or the non-atrous twin
  ref_count        tf_conv3p_grid.cpp:154-381 for kind="plain"), compiled in place by oracle/Makefile into
                   oracle/_ref/.  This is the part of the op where a one-ulp difference flips a result.
  y, dX, dW        OUTPUT OF THE REFERENCE'S OWN CODE: the batch loops of Conv3pOp / Conv3pGradOp::Compute
                   (tf_conv3p_atrous.cpp:451-504, :608-716, serial branch; tf_conv3p_grid.cpp:438-491, :585-689
                   for kind="plain"), compiled in place by oracle/Makefile into oracle/_ref/libref_compute_*.so.
                   The loop text is the reference's, unmodified; the locals it reads (sizes, pointers, the
                   `*_flat` views) are supplied by oracle/ref_compute_driver.cpp instead of by TensorFlow
                   tensors -- see that file.  The generator asserts that the oracle (oracle/conv3p_oracle.c)
                   reproduces these arrays bit-for-bit before writing them.

The reference repository has no tests, fixtures or golden vectors of its own.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from pointwise_amd import synth  # noqa: E402

VOXEL = 0.1

# name, cloud kind, B, N, Cin, Cout, (fz,fy,fx), (sx,sy,sz), dtype, seed, ref kind
CASES = [
    ("generic_s1_f32", "cube", 2, 256, 3, 9, (3, 3, 3), (1, 1, 1), "float32", 101, "atrous"),
    ("generic_s1_plain_f32", "cube", 2, 256, 3, 9, (3, 3, 3), (1, 1, 1), "float32", 101, "plain"),
    ("modelnet_s2_f32", "modelnet", 2, 512, 9, 9, (3, 3, 3), (2, 2, 2), "float32", 102, "atrous"),
    ("modelnet_s3_f32", "modelnet", 2, 512, 9, 9, (3, 3, 3), (3, 3, 3), "float32", 103, "atrous"),
    ("modelnet_s4_f32", "modelnet", 2, 512, 9, 9, (3, 3, 3), (4, 4, 4), "float32", 104, "atrous"),
    ("modelnet_s2_f64", "modelnet", 2, 512, 9, 9, (3, 3, 3), (2, 2, 2), "float64", 102, "atrous"),
    ("lattice_s1_f32", "lattice", 1, 512, 3, 9, (3, 3, 3), (1, 1, 1), "float32", 105, "atrous"),
    ("lattice_s2_f32", "lattice", 1, 512, 9, 9, (3, 3, 3), (2, 2, 2), "float32", 106, "atrous"),
    ("lattice_s3_f64", "lattice", 1, 512, 9, 9, (3, 3, 3), (3, 3, 3), "float64", 107, "atrous"),
    ("room_head_f32", "room", 1, 768, 36, 13, (3, 3, 3), (1, 1, 1), "float32", 108, "atrous"),
    ("aniso_f32", "cube", 2, 200, 4, 5, (2, 1, 3), (1, 2, 3), "float32", 109, "atrous"),   # (fz,fy,fx)=(2,1,3)
    ("even_f32", "cube", 1, 200, 3, 2, (2, 2, 2), (1, 1, 1), "float32", 110, "atrous"),
    ("even_hole_f32", "cube", 1, 200, 2, 2, (4, 4, 4), (2, 2, 2), "float32", 111, "atrous"),
    ("five_f32", "cube", 1, 200, 2, 3, (5, 5, 5), (1, 1, 1), "float32", 112, "atrous"),
    ("one_tap_f32", "cube", 1, 100, 3, 4, (1, 1, 1), (1, 1, 1), "float32", 113, "atrous"),
    ("identical_f32", "identical", 1, 70, 3, 9, (3, 3, 3), (1, 1, 1), "float32", 114, "atrous"),
    ("isolated_f32", "isolated", 1, 65, 3, 9, (3, 3, 3), (2, 2, 2), "float32", 115, "atrous"),
    ("single_point_f32", "cube", 3, 1, 3, 9, (3, 3, 3), (1, 1, 1), "float32", 116, "atrous"),
    ("deep_f32", "room", 1, 256, 32, 64, (3, 3, 3), (1, 1, 1), "float32", 117, "atrous"),
    # voxel-aligned clouds with EVEN dilated extents: candidates sit exactly on the box edge and the reference's
    # cell window (+-n cells, n = (int)((full+1)*0.5), .cpp:247-266) drops some that pass the box test
    ("even_lattice_f32", "vlattice", 2, 384, 3, 4, (2, 2, 2), (1, 1, 1), "float32", 118, "atrous"),
    ("even4_lattice_f32", "vlattice", 1, 384, 2, 3, (4, 4, 4), (1, 1, 1), "float32", 119, "atrous"),
    ("even_mixed_lattice_f64", "vlattice", 1, 300, 2, 2, (2, 3, 2), (1, 2, 2), "float64", 120, "atrous"),
    ("scenenn_in_f32", "room", 1, 512, 12, 9, (3, 3, 3), (1, 1, 1), "float32", 121, "atrous"),
    ("scenenn_head_f32", "room", 1, 512, 36, 41, (3, 3, 3), (1, 1, 1), "float32", 122, "atrous"),
]


def make_points(kind, B, N, seed, dtype):
    if kind == "cube":
        P = synth.uniform_cube(B, N, seed)
    elif kind == "modelnet":
        P = synth.modelnet_like(B, N, seed)
    elif kind == "room":
        P = synth.room_like(B, N, seed)
    elif kind == "lattice":
        P = synth.lattice(B, N, seed, voxel=VOXEL, span=6)
    elif kind == "vlattice":   # voxel-aligned: multiples of the voxel itself
        P = synth.lattice(B, N, seed, voxel=VOXEL, span=8, div=1)
    elif kind == "identical":
        P = np.full((B, N, 3), 0.25, dtype=np.float32)
    elif kind == "isolated":
        i = np.arange(N)
        P = np.stack([i % 5, (i // 5) % 5, i // 25], axis=1)[None].repeat(B, 0).astype(np.float64) * 0.7
        P = P.astype(np.float32)   # spacing 0.7 > any tested box half-width: only self-pairs
    else:
        raise ValueError(kind)
    return P.astype(dtype)


def main():
    oracle.build()
    if (oracle.ref_grid("atrous") is None or oracle.ref_grid("plain") is None or
            oracle.ref_compute("atrous") is None or oracle.ref_compute("plain") is None):
        raise SystemExit("oracle/_ref missing: run in the container that has /root/reference")
    for name, kind, B, N, Cin, Cout, fzyx, s, dts, seed, refkind in CASES:
        dt = np.dtype(dts)
        P = make_points(kind, B, N, seed, dt)
        X = synth.features(B, N, Cin, seed + 1, points=P if Cin >= 3 else None, dtype=dt)
        W = synth.filter_weights(*fzyx, Cin, Cout, seed + 2, dtype=dt)
        dY = synth.upstream_grad(B, N, Cout, seed + 3, dtype=dt)
        offs, idxs, taps, cnts = [], [], [], []
        for b in range(B):
            off, idx, tap, cnt = oracle.reference_grid_lists(P[b], fzyx, s, VOXEL, kind=refkind)
            offs.append(off); idxs.append(idx); taps.append(tap); cnts.append(cnt)
        y = oracle.reference_forward(P, X, W, s, VOXEL, kind=refkind)            # the reference's own loops
        dx, dw = oracle.reference_backward(dY, P, X, W, s, VOXEL, kind=refkind)
        oy = oracle.forward(P, X, W, s, VOXEL)
        odx, odw = oracle.backward(dY, P, X, W, s, VOXEL)
        assert np.array_equal(y, oy) and np.array_equal(dx, odx) and np.array_equal(dw, odw), name
        small = np.int16 if N < 32768 else np.int32
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            points=P, input=X, filter=W, grad_out=dY, stride=np.asarray(s, np.int32), voxel=np.float64(VOXEL),
            ref_kind=refkind,
            ref_offsets=np.stack(offs).astype(np.int32),
            ref_index=np.concatenate(idxs).astype(small), ref_tap=np.concatenate(taps).astype(np.int16),
            ref_pairs_per_cloud=np.asarray([len(i) for i in idxs], np.int64),
            ref_count=np.stack(cnts).astype(small),
            y=y, dX=dx, dW=dw)
        print("%-22s pairs/cloud %s  |y|max %.3f" % (name, [len(i) for i in idxs], float(np.abs(y).max())))


if __name__ == "__main__":
    main()
