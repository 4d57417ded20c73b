"""Independent numpy restatement of conv3p (brute force over all pairs) -- TEST INFRASTRUCTURE ONLY.

A second, structurally different statement of the reference algorithm
(/root/reference/tf_ops/conv3p/tf_conv3p_atrous.cpp) used to cross-check the C oracle:
no grid, no candidate windows -- every (centre, candidate) pair of a cloud is tested with the
reference's own predicates.  This is also how the reference's CUDA path enumerates candidates
(tf_conv3p_atrous.cu:262-285); SURVEY.md section 8(a) shows both accept the same set.

Small clouds only (O(N^2) memory).  Summation order differs from the reference (pairs are
accumulated tap-major with np.add.at), so compare in float64 or with an fp32 tolerance.
PARITY STATUS: see oracle/conv3p_oracle.h.
"""
import numpy as np


def _edges(c, full, voxel, dt):
    # .cpp:240-245: double arithmetic, one rounding to T
    h = (np.float64(full) * 0.5) * np.float64(dt.type(voxel))
    return (c.astype(np.float64) - h).astype(dt), (c.astype(np.float64) + h).astype(dt)


def _taps(v, lo, voxel, full, step, dt):
    # .cpp:280-288: IEEE divide in T, truncate, clamp, hole test, divide by stride
    q = ((v - lo) / dt.type(voxel)).astype(dt)
    t = np.minimum(np.trunc(q).astype(np.int64), full - 1)
    hole = np.fmod(t, step) != 0
    return t // step, hole


def pair_table(cloud, filter_zyx, stride_xyz, voxel):
    """Forward pair set of one cloud: boolean (N,N) `accept[i,j]` and int (N,N) `tap[i,j]`
    (j is a neighbour of centre i in tap tap[i,j]); plus count (N,F)."""
    dt = np.dtype(cloud.dtype)
    N = cloud.shape[0]
    fz, fy, fx = filter_zyx
    ext = (fx, fy, fz)
    accept = np.ones((N, N), dtype=bool)
    taps = []
    for a in range(3):
        full = (ext[a] - 1) * stride_xyz[a] + 1
        lo, hi = _edges(cloud[:, a], full, voxel, dt)
        v = cloud[None, :, a]
        accept &= ~((v < lo[:, None]) | (v > hi[:, None]))           # .cpp:277
        t, hole = _taps(np.broadcast_to(v, (N, N)), lo[:, None], voxel, full, stride_xyz[a], dt)
        accept &= ~hole                                               # .cpp:285
        taps.append(t)
    tap = (taps[2] * fy + taps[1]) * fx + taps[0]                     # .cpp:290
    F = fx * fy * fz
    count = np.zeros((N, F), dtype=np.int64)
    ii, jj = np.nonzero(accept)
    np.add.at(count, (ii, tap[ii, jj]), 1)
    return accept, tap, count


def forward(points, inp, filt, stride_xyz, voxel):
    dt = np.dtype(points.dtype)
    B, N, _ = points.shape
    fz, fy, fx, Cin, Cout = filt.shape
    W = filt.reshape(fz * fy * fx, Cin, Cout)
    out = np.zeros((B, N, Cout), dtype=dt)
    for b in range(B):
        accept, tap, count = pair_table(points[b], (fz, fy, fx), stride_xyz, voxel)
        ii, jj = np.nonzero(accept)
        f = tap[ii, jj]
        # .cpp:492  out[i,c] += w[f,k,c] * in[j,k] / count[i,f]
        terms = np.einsum("pkc,pk->pc", W[f], inp[b][jj]) / count[ii, f][:, None].astype(dt)
        np.add.at(out[b], ii, terms.astype(dt))
    return out


def backward(grad_out, points, inp, filt, stride_xyz, voxel):
    dt = np.dtype(points.dtype)
    B, N, _ = points.shape
    fz, fy, fx, Cin, Cout = filt.shape
    ext = (fx, fy, fz)
    F = fz * fy * fx
    W = filt.reshape(F, Cin, Cout)
    dx = np.zeros((B, N, Cin), dtype=dt)
    dw = np.zeros((F, Cin, Cout), dtype=dt)
    for b in range(B):
        cloud = points[b]
        accept, _, count = pair_table(cloud, (fz, fy, fx), stride_xyz, voxel)
        # .cpp:652: for centre j the candidate set is j's own accepted list -> pairs (j, ii)
        jj, ii = np.nonzero(accept)
        ok = np.ones(jj.shape, dtype=bool)
        taps = []
        for a in range(3):
            full = (ext[a] - 1) * stride_xyz[a] + 1
            lo, _ = _edges(cloud[ii, a], full, voxel, dt)             # .cpp:662-664 box of ii
            t, hole = _taps(cloud[jj, a], lo, voxel, full, stride_xyz[a], dt)  # .cpp:667-669, no inclusion test
            ok &= ~hole                                               # .cpp:672
            taps.append(t)
        f = (taps[2] * fy + taps[1]) * fx + taps[0]
        ok &= (f >= 0) & (f < F)
        f = np.where(ok, f, 0)
        cnt = count[ii, f]
        ok &= cnt != 0                                                # .cpp:679
        jj, ii, f, cnt = jj[ok], ii[ok], f[ok], cnt[ok].astype(dt)
        g = grad_out[b][ii] / cnt[:, None]                            # dY[ii,c]/count
        np.add.at(dx[b], jj, np.einsum("pc,pkc->pk", g, W[f]).astype(dt))   # .cpp:692
        np.add.at(dw, f, np.einsum("pk,pc->pkc", inp[b][jj], g).astype(dt))  # .cpp:696
    return dx, dw.reshape(filt.shape)
