"""CPU tests of the host side (not gpu): the C-ABI library loads and exports every symbol include/conv3p.h
declares, workspace sizing, the operator mirror's argument validation (same messages as the reference's
OP_REQUIRES), batch sharding.  No compute call is made here -- there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from pointwise_amd import _lib, conv3p_op as op, distributed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "conv3p.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(conv3p_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libconv3p_hip.so does not export " + n
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)


def test_load_binds_and_reports_version():
    lib = _lib.load()
    assert lib.conv3p_abi_version() == _lib.ABI_VERSION == 5
    assert _lib.status_string(0) == "ok"
    assert "workspace" in _lib.status_string(_lib.ERR_WORKSPACE)
    assert lib.conv3p_profile_kinds() >= 5
    names = [lib.conv3p_profile_name(k).decode() for k in range(lib.conv3p_profile_kinds())]
    assert {"prep_kernel", "search_kernel", "forward_kernel", "backward_kernel"} <= set(names)


def test_workspace_bytes():
    lib = _lib.load()
    f = lib.conv3p_workspace_bytes
    fwd = f(_lib.PASS_FORWARD, 4, 32, 2048, 9, 9, 3, 3, 3)
    bwd = f(_lib.PASS_BACKWARD, 4, 32, 2048, 9, 9, 3, 3, 3)
    cnt = f(_lib.PASS_NEIGHBOR_COUNT, 4, 32, 2048, 0, 0, 3, 3, 3)
    assert 0 < cnt < fwd < bwd
    assert fwd % 256 == 0 and bwd % 256 == 0
    assert fwd >= 32 * 2048 * (16 + 27 * 4)                    # staged points + per-tap counts
    assert f(_lib.PASS_FORWARD, 8, 32, 2048, 9, 9, 3, 3, 3) > fwd
    assert f(_lib.PASS_FORWARD, 4, -1, 2048, 9, 9, 3, 3, 3) == 0   # invalid shapes
    assert f(_lib.PASS_FORWARD, 4, 1, 1, 1, 1, 0, 3, 3) == 0
    assert f(7, 4, 1, 1, 1, 1, 3, 3, 3) == 0
    assert f(_lib.PASS_FORWARD, 2, 1, 1, 1, 1, 3, 3, 3) == 0
    # cfg5 per GPU (B=16, N=8192, 128->256): sizes are 64-bit, nothing overflows
    big = f(_lib.PASS_BACKWARD, 4, 16, 8192, 128, 256, 3, 3, 3)
    assert big > 16 * 8192 * 27 * 4


def test_cache_bytes():
    lib = _lib.load()
    cfg = _lib.CacheConfig(4, 27, 0, 9, 9)
    one = lib.conv3p_cache_bytes(4, 32, 2048, ctypes.byref(_lib.CacheConfig(1, 27, 0, 9, 9)))
    four = lib.conv3p_cache_bytes(4, 32, 2048, ctypes.byref(cfg))
    assert 0 < one < four and four % 256 == 0
    assert four >= 4 * 32 * 2048 * (2 * 27 * 4 + 256 * 8)           # populations (two orders) + 8-byte pair records per slot
    assert lib.conv3p_cache_bytes(4, 32, 2048, ctypes.byref(_lib.CacheConfig(0, 27, 0, 9, 9))) == 0
    assert lib.conv3p_cache_bytes(4, 32, 2048, ctypes.byref(_lib.CacheConfig(4, 5000, 0, 9, 9))) == 0
    assert lib.conv3p_cache_bytes(2, 32, 2048, ctypes.byref(cfg)) == 0
    assert lib.conv3p_cache_forget(None) == _lib.OK
    assert lib.conv3p_cache_init(None, 0, None) == _lib.OK
    assert lib.conv3p_cache_init(None, 16, None) == _lib.ERR_INVALID_ARGUMENT


def test_invalid_arguments_without_touching_the_gpu():
    """The C entry points validate before any HIP call; NULL pointers are fine for this."""
    lib = _lib.load()
    s_ok = (ctypes.c_int32 * 3)(1, 1, 1)
    s_bad = (ctypes.c_int32 * 3)(1, 0, 1)
    call = lambda s, vox, fz=3: lib.conv3p_forward_f32(None, None, None, ctypes.cast(s, ctypes.c_void_p),
                                                       ctypes.c_float(vox), 2, 64, 3, 9, fz, 3, 3, None, None, 0, None)
    assert call(s_bad, 0.1) == _lib.ERR_INVALID_ARGUMENT
    assert call(s_ok, 0.0) == _lib.ERR_INVALID_ARGUMENT
    assert call(s_ok, float("nan")) == _lib.ERR_INVALID_ARGUMENT
    assert call(s_ok, 0.1, fz=0) == _lib.ERR_INVALID_ARGUMENT
    assert call(None, 0.1) == _lib.ERR_INVALID_ARGUMENT
    assert call(s_ok, 0.1) == _lib.ERR_INVALID_ARGUMENT          # NULL tensors
    # empty problems succeed without a device
    assert lib.conv3p_forward_f32(None, None, None, ctypes.cast(s_ok, ctypes.c_void_p), ctypes.c_float(0.1),
                                  0, 64, 3, 9, 3, 3, 3, None, None, 0, None) == _lib.OK
    assert lib.conv3p_forward_f32(None, None, None, ctypes.cast(s_ok, ctypes.c_void_p), ctypes.c_float(0.1),
                                  2, 0, 3, 9, 3, 3, 3, None, None, 0, None) == _lib.OK


def _t(*shape, dtype=torch.float32):
    return torch.zeros(*shape, dtype=dtype)


@pytest.mark.parametrize("args,msg", [
    ((_t(2, 8), _t(2, 8, 3), _t(3, 3, 3, 3, 9)), "Conv3p expects (batch_size, num_points, 3) points shape"),
    ((_t(2, 8, 3), _t(3, 8, 3), _t(3, 3, 3, 3, 9)), "Conv3p expects points and input tensor to have the same batch size"),
    ((_t(2, 8, 3), _t(2, 9, 3), _t(3, 3, 3, 3, 9)), "Conv3p expects points and input tensor to have the same number of points"),
    ((_t(2, 8, 3), _t(2, 8, 4), _t(3, 3, 3, 3, 9)), "Conv3p expects filter channels to be matched with input channels"),
])
def test_shape_checks_mirror_op_requires(args, msg):
    """Messages are the reference's (tf_conv3p_atrous.cpp:410, :417, :418, :430)."""
    with pytest.raises(op.Conv3pInvalidArgument) as e:
        op.conv3p(*args, [1, 1, 1], [0.1])
    assert msg in str(e.value)


def test_stride_and_voxel_checks():
    p, x, w = _t(2, 8, 3), _t(2, 8, 3), _t(3, 3, 3, 3, 9)
    with pytest.raises(op.Conv3pInvalidArgument, match="stride tensor to have size 3"):      # .cpp:437
        op.conv3p(p, x, w, [1, 1], [0.1])
    with pytest.raises(op.Conv3pInvalidArgument, match="voxel tensor to have dimension 1"):  # .cpp:443
        op.conv3p(p, x, w, [1, 1, 1], [0.1, 0.2])
    with pytest.raises(op.Conv3pInvalidArgument, match="float32 or float64"):                # Attr T
        op.conv3p(p.half(), x.half(), w.half(), [1, 1, 1], [0.1])
    with pytest.raises(op.Conv3pInvalidArgument, match="wrong size for dim 2"):              # .cpp:585
        op.conv3p_grad(_t(2, 8, 7), p, x, w, [1, 1, 1], [0.1])
    with pytest.raises(op.Conv3pInvalidArgument, match="wrong size for dim 1"):              # .cpp:584
        op.conv3p_grad(_t(2, 7, 9), p, x, w, [1, 1, 1], [0.1])


def test_cpu_tensors_are_rejected_loudly():
    """No CPU fallback in the product path."""
    p, x, w = _t(2, 8, 3), _t(2, 8, 3), _t(3, 3, 3, 3, 9)
    with pytest.raises(op.Conv3pRuntimeError, match="HIP device"):
        op.conv3p(p, x, w, [1, 1, 1], [0.1])
    with pytest.raises(op.Conv3pRuntimeError, match="HIP device"):
        op.conv3p_grad(_t(2, 8, 9), p, x, w, torch.tensor([1, 1, 1], dtype=torch.int32), torch.tensor([0.1]))


def test_product_package_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import pointwise_amd, pointwise_amd.stack, "
            "pointwise_amd.distributed; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), "
            "'product imports oracle'" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)
    for fn in os.listdir(os.path.join(ROOT, "pointwise_amd")):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "pointwise_amd", fn)).read().replace(
                "the oracle side", "").replace("CPU oracle", ""), fn


def test_shard_bounds():
    assert [distributed.shard_bounds(256, 8, r) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    b = [distributed.shard_bounds(10, 4, r) for r in range(4)]
    assert b == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert distributed.shard_bounds(2, 4, 3) == (2, 2)
    with pytest.raises(ValueError):
        distributed.shard_bounds(8, 2, 2)


def test_tf_shim_is_well_formed_against_the_declared_api():
    """integration/tf_conv3p_shim.cc type-checks (-fsyntax-only) against include/conv3p.h and the declared subset of
    the TensorFlow op API (integration/tf_decl).  Not a TensorFlow build -- there is none in this image -- but it
    keeps the shim's C-ABI calls in step with the header."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(["make", "-C", os.path.join(root, "integration"), "check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_no_kernel_of_the_library_spills_vector_registers():
    """A spilled vector register costs a scratch access inside loops that live on their gather queues (one spill drained
    the populated-rows backward's pipeline): the code object's notes must show none -- for the
    models' shapes (3 / 6 / 9 / 12 -> 9, 36 -> 13), the matrix-core classes (deep_gemm / deep_dw / deep_order), the
    search and every other kernel of the library."""
    from pointwise_amd import build
    res = build.kernel_resources()
    names = [r[0] for r in res]
    for must in ("forward_kernelIfLi9ELi9E", "forward_kernelIfLi12ELi9E", "backward_sparse_kernelIfLi9ELi9E",
                 "backward_sparse_kernelIfLi36ELi13E", "backward_kernelIfLi3ELi9E", "deep_dw_kernelILi256ELi256E",
                 "deep_gemm_kernelILi256ELi128ELb1E", "search_fused_kernelIfLb1E"):
        assert any(must in n for n in names), must + ": kernel not found in the code object"
    spilled = [(n, vs) for n, _, vs, _, _ in res if vs != 0]
    assert not spilled, spilled
    # ... and no private segment at all (no dynamically indexed local array, no struct that went through memory) in the
    # models' forward kernels, the dense backward, the fused stack kernels and the matrix-core kernels.  (Round 6: the
    # forward kernels' 28 bytes were a Window struct behind a pointer that may be null; what is left, and listed here so
    # that it cannot grow unnoticed: 16 bytes of dead stores -- a uint2 member written on two paths, never re-loaded --
    # in the populated-rows backward, 36 in the tile-pair search of even extents.)
    allowed = {"backward_sparse_kernel": 16, "search_kernel": 36}
    private = [(n, sc) for n, _, _, _, sc in res if sc > max([v for k, v in allowed.items() if k in n] + [0])]
    assert not private, private
    for must0 in ("forward_kernelIfLi3ELi9E", "forward_kernelIfLi9ELi9E", "forward_kernelIfLi12ELi9E", "forward_kernelIfLi36ELi13E",
                  "backward_kernelIfLi3ELi9E", "backward_kernelIfLi9ELi9E", "stack_forward_kernelIfLi3ELi9E", "stack_backward_kernelIfLi9E"):
        hit = [sc for n, _, _, _, sc in res if must0 in n]
        assert hit and max(hit) == 0, (must0, hit)
