"""ctypes front-end of the conv3p CPU oracle -- TEST INFRASTRUCTURE ONLY.

Wraps
  * oracle/libconv3p_oracle.so   the from-scratch C restatement of the reference CPU op
                                 (/root/reference/tf_ops/conv3p/tf_conv3p_atrous.cpp);
  * oracle/_ref/libref_grid_*.so the reference's own Grid template compiled in place
                                 (oracle/Makefile), when present.

  * oracle/_ref/libref_compute_*.so  the reference's own Conv3pOp / Conv3pGradOp batch loops,
                                 spliced in place around oracle/ref_compute_driver.cpp (which
                                 supplies the locals those lines use), when present.

PARITY STATUS: pinned.  Neighbour search against the reference Grid; the accumulation loops
(y, dX, dW) bit-for-bit against the reference's own loop text (tests/test_oracle.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = {}

_CT = {np.dtype(np.float32): (ctypes.c_float, "f32"), np.dtype(np.float64): (ctypes.c_double, "f64")}


def build(quiet=True):
    """Compile the oracle (and, when /root/reference exists, oracle/_ref)."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libconv3p_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        for sfx in ("f32", "f64"):
            getattr(_LIB, "conv3p_oracle_neighbor_lists_" + sfx).restype = ctypes.c_long
            getattr(_LIB, "conv3p_oracle_backward_pairs_" + sfx).restype = ctypes.c_long
    return _LIB


def ref_grid(kind="atrous"):
    """The reference Grid shared object ('atrous' or 'plain'), or None if it was never built."""
    if kind not in _REF:
        path = os.path.join(_HERE, "_ref", "libref_grid_%s.so" % kind)
        if not os.path.exists(path):
            _REF[kind] = None
        else:
            h = ctypes.CDLL(path)
            h.ref_grid_lists_f32.restype = ctypes.c_long
            h.ref_grid_lists_f64.restype = ctypes.c_long
            _REF[kind] = h
    return _REF[kind]


def ref_compute(kind="atrous"):
    """The reference accumulation-loop shared object: 'atrous' / 'plain' (serial, deterministic) or
    'atrous_omp' (built like the reference CPU object, OpenMP over the batch).  None if never built."""
    key = "compute_" + kind
    if key not in _REF:
        path = os.path.join(_HERE, "_ref", "libref_compute_%s.so" % kind)
        _REF[key] = ctypes.CDLL(path) if os.path.exists(path) else None
    return _REF[key]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(dtype, *arrays):
    dt = np.dtype(dtype)
    if dt not in _CT:
        raise TypeError("oracle supports float32/float64 only")
    return [np.ascontiguousarray(a, dtype=dt) for a in arrays]


def _dims(points, inp, filt):
    B, N, three = points.shape
    assert three == 3
    fz, fy, fx, Cin, Cout = filt.shape
    assert inp.shape == (B, N, Cin), (inp.shape, (B, N, Cin))
    return B, N, Cin, Cout, fz, fy, fx


def _stride(stride):
    s = np.ascontiguousarray(stride, dtype=np.int32)
    assert s.shape == (3,)
    return s


def forward(points, inp, filt, stride, voxel, nthreads=1):
    """Conv3p forward on the CPU.  Shapes as the reference op (tf_conv3p_atrous.cpp:409-451)."""
    dt = np.dtype(points.dtype)
    points, inp, filt = _prep(dt, points, inp, filt)
    B, N, Cin, Cout, fz, fy, fx = _dims(points, inp, filt)
    s = _stride(stride)
    out = np.empty((B, N, Cout), dtype=dt)
    ct, sfx = _CT[dt]
    rc = getattr(lib(), "conv3p_oracle_forward_" + sfx)(
        _p(points), _p(inp), _p(filt), _p(s), ct(float(voxel)), B, N, Cin, Cout, fz, fy, fx, _p(out),
        int(nthreads))
    if rc != 0:
        raise RuntimeError("conv3p_oracle_forward rc=%d" % rc)
    return out


def backward(grad_out, points, inp, filt, stride, voxel, nthreads=1):
    """Conv3pGrad on the CPU -> (grad_input, grad_filter) (tf_conv3p_atrous.cpp:526-720)."""
    dt = np.dtype(points.dtype)
    grad_out, points, inp, filt = _prep(dt, grad_out, points, inp, filt)
    B, N, Cin, Cout, fz, fy, fx = _dims(points, inp, filt)
    assert grad_out.shape == (B, N, Cout)
    s = _stride(stride)
    dx = np.empty((B, N, Cin), dtype=dt)
    dw = np.empty(filt.shape, dtype=dt)
    ct, sfx = _CT[dt]
    rc = getattr(lib(), "conv3p_oracle_backward_" + sfx)(
        _p(grad_out), _p(points), _p(inp), _p(filt), _p(s), ct(float(voxel)), B, N, Cin, Cout, fz, fy,
        fx, _p(dx), _p(dw), int(nthreads))
    if rc != 0:
        raise RuntimeError("conv3p_oracle_backward rc=%d" % rc)
    return dx, dw


def neighbor_count(points, filter_zyx, stride, voxel):
    """int32 (B,N,F) per-tap neighbour populations (Grid::neighbor_count, .cpp:306-379)."""
    dt = np.dtype(points.dtype)
    (points,) = _prep(dt, points)
    B, N, _ = points.shape
    fz, fy, fx = filter_zyx
    s = _stride(stride)
    cnt = np.zeros((B, N, fz * fy * fx), dtype=np.int32)
    ct, sfx = _CT[dt]
    rc = getattr(lib(), "conv3p_oracle_neighbor_count_" + sfx)(
        _p(points), _p(s), ct(float(voxel)), B, N, fz, fy, fx, _p(cnt))
    if rc != 0:
        raise RuntimeError("conv3p_oracle_neighbor_count rc=%d" % rc)
    return cnt


def neighbor_lists(cloud, filter_zyx, stride, voxel):
    """CSR (offsets, index, tap) of one (N,3) cloud in the reference's visit order."""
    dt = np.dtype(cloud.dtype)
    (cloud,) = _prep(dt, cloud)
    N = cloud.shape[0]
    fz, fy, fx = filter_zyx
    s = _stride(stride)
    cap = max(1, N * N)
    off = np.zeros(N + 1, dtype=np.int64)
    idx = np.empty(cap, dtype=np.int32)
    tap = np.empty(cap, dtype=np.int32)
    ct, sfx = _CT[dt]
    tot = getattr(lib(), "conv3p_oracle_neighbor_lists_" + sfx)(
        _p(cloud), N, _p(s), ct(float(voxel)), fz, fy, fx, _p(off), _p(idx), _p(tap), ctypes.c_long(cap))
    if tot < 0:
        raise RuntimeError("conv3p_oracle_neighbor_lists rc=%d" % tot)
    return off, idx[:tot].copy(), tap[:tot].copy()


def backward_pairs(cloud, filter_zyx, stride, voxel):
    """(j, ii, tap, count) of every pair Conv3pGrad accumulates for one cloud (.cpp:647-700)."""
    dt = np.dtype(cloud.dtype)
    (cloud,) = _prep(dt, cloud)
    N = cloud.shape[0]
    fz, fy, fx = filter_zyx
    s = _stride(stride)
    cap = max(1, N * N)
    arrs = [np.empty(cap, dtype=np.int32) for _ in range(4)]
    ct, sfx = _CT[dt]
    tot = getattr(lib(), "conv3p_oracle_backward_pairs_" + sfx)(
        _p(cloud), N, _p(s), ct(float(voxel)), fz, fy, fx, *[_p(a) for a in arrs], ctypes.c_long(cap))
    if tot < 0:
        raise RuntimeError("conv3p_oracle_backward_pairs rc=%d" % tot)
    return tuple(a[:tot].copy() for a in arrs)


def reference_grid_lists(cloud, filter_zyx, stride, voxel, kind="atrous"):
    """Same CSR + (N,F) counts, produced by the REFERENCE Grid template (oracle/_ref).

    Returns None when oracle/_ref was never built (no /root/reference at build time)."""
    h = ref_grid(kind)
    if h is None:
        return None
    dt = np.dtype(cloud.dtype)
    (cloud,) = _prep(dt, cloud)
    N = cloud.shape[0]
    fz, fy, fx = filter_zyx
    s = _stride(stride)
    cap = max(1, N * N)
    off = np.zeros(N + 1, dtype=np.int64)
    idx = np.empty(cap, dtype=np.int32)
    tap = np.empty(cap, dtype=np.int32)
    cnt = np.zeros((N, fz * fy * fx), dtype=np.int32)
    ct, sfx = _CT[dt]
    tot = getattr(h, "ref_grid_lists_" + sfx)(
        _p(cloud), N, ct(float(voxel)), fx, fy, fz, int(s[0]), int(s[1]), int(s[2]), _p(off), _p(idx),
        _p(tap), ctypes.c_long(cap), _p(cnt))
    if tot < 0:
        raise RuntimeError("ref_grid_lists rc=%d" % tot)
    return off, idx[:tot].copy(), tap[:tot].copy(), cnt


def reference_forward(points, inp, filt, stride, voxel, kind="atrous"):
    """Conv3p forward computed by the REFERENCE's own batch loop (tf_conv3p_atrous.cpp:451-504, or the
    non-atrous twin for kind='plain', which ignores stride).  None when oracle/_ref was never built."""
    h = ref_compute(kind)
    if h is None:
        return None
    dt = np.dtype(points.dtype)
    points, inp, filt = _prep(dt, points, inp, filt)
    B, N, Cin, Cout, fz, fy, fx = _dims(points, inp, filt)
    s = _stride(stride)
    out = np.empty((B, N, Cout), dtype=dt)
    ct, sfx = _CT[dt]
    getattr(h, "ref_compute_forward_" + sfx)(_p(points), _p(inp), _p(filt), _p(s), ct(float(voxel)), B, N, Cin,
                                             Cout, fz, fy, fx, _p(out))
    return out


def reference_backward(grad_out, points, inp, filt, stride, voxel, kind="atrous"):
    """Conv3pGrad computed by the REFERENCE's own batch loop (tf_conv3p_atrous.cpp:608-716)."""
    h = ref_compute(kind)
    if h is None:
        return None
    dt = np.dtype(points.dtype)
    grad_out, points, inp, filt = _prep(dt, grad_out, points, inp, filt)
    B, N, Cin, Cout, fz, fy, fx = _dims(points, inp, filt)
    assert grad_out.shape == (B, N, Cout)
    s = _stride(stride)
    dx = np.empty((B, N, Cin), dtype=dt)
    dw = np.empty(filt.shape, dtype=dt)
    ct, sfx = _CT[dt]
    getattr(h, "ref_compute_backward_" + sfx)(_p(grad_out), _p(points), _p(inp), _p(filt), _p(s), ct(float(voxel)),
                                              B, N, Cin, Cout, fz, fy, fx, _p(dx), _p(dw))
    return dx, dw


def reference_threads(kind="atrous_omp"):
    h = ref_compute(kind)
    return int(h.ref_compute_threads()) if h is not None else 0


def max_threads():
    return int(lib().conv3p_oracle_max_threads())
