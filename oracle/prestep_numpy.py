"""CPU restatement of the reference providers' host pre-step -- TEST INFRASTRUCTURE ONLY.

  rotate_point_cloud_by_angles / jitter   /root/reference/modelnet_provider.py:23-75
  sort_point_cloud_xyz / _xyz2            /root/reference/util.py:55-109

PARITY STATUS: pinned.  The reference modules cannot be imported here (util.py imports tensorflow, the providers
import h5py; neither is in this image), but these functions only need numpy: tests/golden/make_prestep_golden.py
extracts their own source text from the reference files with `ast`, executes it unmodified with numpy, and stores
inputs and outputs under tests/golden/prestep_*.npz.  tests/test_prestep.py checks this restatement against those
fixtures; the HIP kernels (pointwise_amd/prestep.py) are checked against both.
Random draws are made explicit (angles, noise) so that the functions are deterministic.
"""
import numpy as np


def rotate_point_cloud_by_angles(batch_data, angles):
    """modelnet_provider.py:23-41 with the per-cloud angle given instead of drawn (:33)."""
    rotated = np.zeros(batch_data.shape, dtype=np.float32)                      # :32
    for k in range(batch_data.shape[0]):
        c, s = np.cos(angles[k]), np.sin(angles[k])                             # :34-35
        R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])                         # :36-38
        rotated[k, ...] = np.dot(batch_data[k, ...].reshape((-1, 3)), R)         # :40
    return rotated


def jitter_point_cloud(batch_data, noise, sigma=0.01, clip=0.05):
    """modelnet_provider.py:64-75 with the standard-normal samples given instead of drawn (:73)."""
    assert clip > 0                                                             # :72
    jittered = np.clip(sigma * noise, -1 * clip, clip)                           # :73
    jittered += batch_data                                                       # :74
    return jittered


def sort_point_cloud_xyz(batch_data):
    """util.py:55-74."""
    out = np.zeros(batch_data.shape, dtype=np.float32)
    K = batch_data.shape[2]
    for k in range(batch_data.shape[0]):
        pc = batch_data[k, ...].reshape((-1, K))
        pc = pc[pc[:, 2].argsort()]                                             # :66 least significant field first
        pc = pc[pc[:, 1].argsort(kind="mergesort")]                             # :67 stable from here on
        pc = pc[pc[:, 0].argsort(kind="mergesort")]                             # :68
        out[k, ...] = pc
    return out


def sort_point_cloud_xyz2(batch_data, batch_attributes):
    """util.py:76-109."""
    out = np.zeros(batch_data.shape, dtype=batch_data.dtype)
    attr = np.zeros(batch_attributes.shape, dtype=batch_attributes.dtype)
    for k in range(batch_data.shape[0]):
        pc, at = batch_data[k, ...], batch_attributes[k, ...]
        for col, kind in ((2, None), (1, "mergesort"), (0, "mergesort")):        # :91-103
            idx = pc[:, col].argsort() if kind is None else pc[:, col].argsort(kind=kind)
            pc, at = pc[idx], at[idx]
        out[k, ...], attr[k, ...] = pc, at
    return out, attr
