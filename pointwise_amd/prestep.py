"""Device-side mirror of the reference providers' host pre-step.

Names, argument meaning and results follow the reference functions
    rotate_point_cloud, rotate_point_cloud_by_angle, jitter_point_cloud    /root/reference/modelnet_provider.py:23-75
    sort_point_cloud_xyz, sort_point_cloud_xyz2                            /root/reference/util.py:55-109
with torch tensors on a HIP device in place of numpy arrays: the per-cloud Python loops of the reference become one
kernel launch per batch (include/conv3p.h: conv3p_augment_f32, conv3p_sort_xyz_order_f32, conv3p_gather_rows).
Random numbers are drawn by the caller's generator (numpy on the host for the B angles, exactly as the reference
does; torch on the device for the B x N x 3 Gaussian noise) and can be passed in explicitly, which is how the tests
compare with the reference functions.  sort_point_cloud_morton (modelnet_provider.py:100-109) needs the third-party
`libpluie`, which is not part of the reference tree, and is not provided; the default sort_method is "xyz".
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .conv3p_op import Conv3pInvalidArgument, Conv3pRuntimeError, _call, _check_device


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _augment(batch_data, cos_sin, noise, sigma, clip):
    lib = _lib.load()
    dev = _check_device(batch_data)
    if batch_data.dim() != 3 or batch_data.shape[2] != 3 or batch_data.dtype != torch.float32:
        raise Conv3pInvalidArgument("expected a float32 BxNx3 batch of point clouds")
    B, N, _ = batch_data.shape
    x = batch_data.contiguous()
    out = torch.empty_like(x)
    cs = None
    if cos_sin is not None:
        cs = torch.as_tensor(np.ascontiguousarray(cos_sin, dtype=np.float64)).to(dev)
        if tuple(cs.shape) != (B, 2):
            raise Conv3pInvalidArgument("one rotation per cloud expected")
    if noise is not None:
        if tuple(noise.shape) != (B, N, 3) or noise.dtype != torch.float64 or noise.device != dev:
            raise Conv3pInvalidArgument("noise must be a float64 BxNx3 tensor on the batch's device")
        noise = noise.contiguous()
    with torch.cuda.device(dev):
        _call(lib.conv3p_augment_f32, x.data_ptr(), cs.data_ptr() if cs is not None else None,
              noise.data_ptr() if noise is not None else None, float(sigma), float(clip), B, N, out.data_ptr(), _stream(dev))
    return out


def rotate_point_cloud(batch_data, angles=None):
    """Random rotation of every cloud about the up axis (modelnet_provider.py:23-41).  angles (B,) in radians;
    None draws np.random.uniform() * 2 * pi per cloud, in cloud order, like the reference."""
    B = batch_data.shape[0]
    if angles is None:
        angles = [np.random.uniform() * 2 * np.pi for _ in range(B)]
    angles = np.asarray(angles, dtype=np.float64).reshape(B)
    return _augment(batch_data, np.stack([np.cos(angles), np.sin(angles)], axis=1), None, 0.0, 1.0)


def rotate_point_cloud_by_angle(batch_data, rotation_angle):
    """modelnet_provider.py:44-61: the same angle for every cloud."""
    return rotate_point_cloud(batch_data, [rotation_angle] * batch_data.shape[0])


def jitter_point_cloud(batch_data, sigma=0.01, clip=0.05, noise=None):
    """Per-point Gaussian jitter clipped to +-clip (modelnet_provider.py:64-75).  noise: float64 (B,N,3) standard
    normal samples on the device; None draws them with torch.randn."""
    if not clip > 0:
        raise Conv3pInvalidArgument("clip must be positive")          # assert(clip > 0), :72
    if noise is None:
        noise = torch.randn(tuple(batch_data.shape), dtype=torch.float64, device=batch_data.device)
    return _augment(batch_data, None, noise, sigma, clip)


def rotate_and_jitter(batch_data, angles=None, sigma=0.01, clip=0.05, noise=None):
    """jitter_point_cloud(rotate_point_cloud(batch)) as the training provider applies them
    (modelnet_provider.py:196-198), in one launch."""
    B = batch_data.shape[0]
    if angles is None:
        angles = [np.random.uniform() * 2 * np.pi for _ in range(B)]
    angles = np.asarray(angles, dtype=np.float64).reshape(B)
    if noise is None:
        noise = torch.randn(tuple(batch_data.shape), dtype=torch.float64, device=batch_data.device)
    return _augment(batch_data, np.stack([np.cos(angles), np.sin(angles)], axis=1), noise, sigma, clip)


def sort_order_xyz(batch_data):
    """int32 (B, N): for every cloud the permutation that sorts its points by x, then y, then z."""
    lib = _lib.load()
    dev = _check_device(batch_data)
    if batch_data.dim() != 3 or batch_data.shape[2] < 3 or batch_data.dtype != torch.float32:
        raise Conv3pInvalidArgument("expected a float32 BxNxK batch whose first three channels are XYZ")
    B, N, K = batch_data.shape
    x = batch_data.contiguous()
    order = torch.empty((B, N), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _call(lib.conv3p_sort_xyz_order_f32, x.data_ptr(), B, N, K, order.data_ptr(), _stream(dev))
    return order


def _gather(t, order):
    lib = _lib.load()
    dev = _check_device(t, order)
    B, N = order.shape
    if t.shape[0] != B or t.shape[1] != N:
        raise Conv3pInvalidArgument("attributes must be BxNx...")
    src = t.contiguous()
    dst = torch.empty_like(src)
    row_bytes = src.element_size() * int(np.prod(src.shape[2:])) if src.dim() > 2 else src.element_size()
    with torch.cuda.device(dev):
        _call(lib.conv3p_gather_rows, src.data_ptr(), order.data_ptr(), B, N, row_bytes, dst.data_ptr(), _stream(dev))
    return dst


def sort_point_cloud_xyz(batch_data):
    """util.py:55-74: every cloud sorted by coordinate with priority x -> y -> z (BxNxK, XYZ first)."""
    return _gather(batch_data, sort_order_xyz(batch_data))


def sort_point_cloud_xyz2(batch_data, batch_attributes):
    """util.py:76-109: the same, the per-point attributes (any dtype, BxN or BxNxM) permuted accordingly."""
    order = sort_order_xyz(batch_data)
    return _gather(batch_data, order), _gather(batch_attributes, order)
