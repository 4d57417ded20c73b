// conv3p_kernels.hpp -- the gfx950 kernels of the conv3p operator pair.
// See conv3p_device.hpp for the search structure; this file holds
//   prep_kernel            stage points as 16-byte records + per-tile bounding boxes
//   count_kernel           per-point, per-tap neighbour populations (Grid::neighbor_count)
//   forward_kernel         Conv3p           (reference tf_conv3p_atrous.cpp:453-504)
//   backward_kernel        Conv3pGrad       (reference tf_conv3p_atrous.cpp:608-716)
//   reduce_partials_kernel deterministic second stage of grad_filter
//   selu kernels           the activation between the stack's layers (selu.py:22-26)
#pragma once

#include "conv3p_device.hpp"

namespace conv3p {

// ---------------------------------------------------------------------------------
// prep: one wavefront per tile.  Identity order (tile t = points [64t, 64t+64)).
// Padding lanes of the last tile get +inf coordinates (rejected by every finite box)
// and idx = -1.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void prep_kernel(const T *__restrict__ points, int N, int ntiles,
                                                   PointRec<T> *__restrict__ pts,
                                                   T *__restrict__ boxes)
{
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (tile >= ntiles) return;
    const int i = tile * kTile + lane;
    PointRec<T> r;
    const T inf = Limits<T>::inf();
    if (i < N) {
        const T *p = points + ((size_t)b * N + i) * 3;
        r.x = p[0];
        r.y = p[1];
        r.z = p[2];
        r.idx = i;
    } else {
        r.x = r.y = r.z = inf;
        r.idx = -1;
    }
    pts[((size_t)b * ntiles + tile) * kTile + lane] = r;
    const bool v = i < N;
    T mn[3] = {wave_min(v ? r.x : inf), wave_min(v ? r.y : inf), wave_min(v ? r.z : inf)};
    T mx[3] = {wave_max(v ? r.x : -inf), wave_max(v ? r.y : -inf), wave_max(v ? r.z : -inf)};
    if (lane == 0) {
        T *bb = boxes + ((size_t)b * ntiles + tile) * 6;
        bb[0] = mn[0]; bb[1] = mn[1]; bb[2] = mn[2];
        bb[3] = mx[0]; bb[4] = mx[1]; bb[5] = mx[2];
    }
}

// LDS carve helpers (all offsets multiples of 16 B; one extern array per kernel).
__device__ __forceinline__ size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// ---------------------------------------------------------------------------------
// count: count[(b*N + i)*F + f] = neighbours of i in tap f   (.cpp:306-379)
// LDS: tapmap | per wave { tile records | [F][65] u32 }
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void count_kernel(const PointRec<T> *__restrict__ pts,
                                                    const T *__restrict__ boxes, Stencil<T> st,
                                                    int N, int ntiles, BlockMap bm,
                                                    int32_t *__restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    const size_t per_wave = align16(sizeof(PointRec<T>) * kTile) + align16((size_t)st.ntap * kCntStride * 4);
    char *wave_base = smem + align16((size_t)3 * st.maxfull * 2) + per_wave * (threadIdx.x >> 6);
    PointRec<T> *tile_lds = reinterpret_cast<PointRec<T> *>(wave_base);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(wave_base + align16(sizeof(PointRec<T>) * kTile));

    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    __syncthreads();

    int b, blk;
    if (!block_to_cloud(bm, b, blk)) return;
    const int lane = threadIdx.x & 63;
    const int qt = blk * kWavesPerBlock + (threadIdx.x >> 6);
    if (qt >= ntiles) return;

    const PointRec<T> *cloud_pts = pts + (size_t)b * ntiles * kTile;
    const T *cloud_box = boxes + (size_t)b * ntiles * 6;
    Query<T> q;
    make_query(q, cloud_pts[(size_t)qt * kTile + lane], st);
    for (int f = 0; f < st.ntap; ++f) cnt[f * kCntStride + lane] = 0;

    for_each_box_hit(cloud_pts, cloud_box, ntiles, q, tile_lds, [&](const PointRec<T> &v) {
        const int tx = axis_tap(v.x, q.lo[0], st.voxel, st.full[0], tapmap);
        const int ty = axis_tap(v.y, q.lo[1], st.voxel, st.full[1], tapmap + st.maxfull);
        const int tz = axis_tap(v.z, q.lo[2], st.voxel, st.full[2], tapmap + 2 * st.maxfull);
        if ((tx | ty | tz) >= 0) {
            const int f = (tz * st.ext[1] + ty) * st.ext[0] + tx;
            cnt[f * kCntStride + lane] += 1;
        }
    });
    __builtin_amdgcn_wave_barrier();
    // row-wise write-out: lanes = taps, one query per step (coalesced F*4-byte rows)
    for (int qq = 0; qq < kTile; ++qq) {
        const int orig = __shfl(q.orig, qq);
        if (orig < 0) continue;
        int32_t *row = count + ((size_t)b * N + orig) * st.ntap;
        for (int f = lane; f < st.ntap; f += 64) row[f] = (int32_t)cnt[f * kCntStride + qq];
    }
}

// ---------------------------------------------------------------------------------
// forward.  CIN/COUT > 0: channel counts are compile-time, the output row lives in
// registers and the filter in LDS.  CIN == 0: generic shapes -- the output row is
// accumulated in (pre-zeroed) global memory, owned by the lane, filter read through L1/L2.
// LDS: tapmap | filter (small path) | per wave { tile records | [F][65] u32 own counts }
// ---------------------------------------------------------------------------------
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void forward_kernel(
    const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes,
    const int32_t *__restrict__ count, const T *__restrict__ input, const T *__restrict__ filter,
    Stencil<T> st, int N, int ntiles, int cin_rt, int cout_rt, BlockMap bm, T *__restrict__ output)
{
    constexpr bool kSmall = CIN > 0;
    const int cin = kSmall ? CIN : cin_rt;
    const int cout = kSmall ? COUT : cout_rt;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    T *w_lds = reinterpret_cast<T *>(smem + off);
    const size_t nw = (size_t)st.ntap * cin * cout;
    if (kSmall) off += align16(nw * sizeof(T));
    const size_t per_wave = align16(sizeof(PointRec<T>) * kTile) + align16((size_t)st.ntap * kCntStride * 4);
    char *wave_base = smem + off + per_wave * (threadIdx.x >> 6);
    PointRec<T> *tile_lds = reinterpret_cast<PointRec<T> *>(wave_base);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(wave_base + align16(sizeof(PointRec<T>) * kTile));

    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    if (kSmall)
        for (size_t e = threadIdx.x; e < nw; e += blockDim.x) w_lds[e] = filter[e];
    __syncthreads();

    int b, blk;
    if (!block_to_cloud(bm, b, blk)) return;
    const int lane = threadIdx.x & 63;
    const int qt = blk * kWavesPerBlock + (threadIdx.x >> 6);
    if (qt >= ntiles) return;

    const PointRec<T> *cloud_pts = pts + (size_t)b * ntiles * kTile;
    const T *cloud_box = boxes + (size_t)b * ntiles * 6;
    Query<T> q;
    make_query(q, cloud_pts[(size_t)qt * kTile + lane], st);

    // own tap populations -> LDS, [tap][lane]
    for (int qq = 0; qq < kTile; ++qq) {
        const int orig = __shfl(q.orig, qq);
        if (orig < 0) continue;
        const int32_t *row = count + ((size_t)b * N + orig) * st.ntap;
        for (int f = lane; f < st.ntap; f += 64) cnt[f * kCntStride + qq] = (uint32_t)row[f];
    }
    __builtin_amdgcn_wave_barrier();

    const T *in_cloud = input + (size_t)b * N * cin;
    T *out_row = output + ((size_t)b * N + (q.orig < 0 ? 0 : q.orig)) * cout;

    T acc[kSmall ? COUT : 1];
    if (kSmall) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = (T)0;
    }

    for_each_box_hit(cloud_pts, cloud_box, ntiles, q, tile_lds, [&](const PointRec<T> &v) {
        const int tx = axis_tap(v.x, q.lo[0], st.voxel, st.full[0], tapmap);
        const int ty = axis_tap(v.y, q.lo[1], st.voxel, st.full[1], tapmap + st.maxfull);
        const int tz = axis_tap(v.z, q.lo[2], st.voxel, st.full[2], tapmap + 2 * st.maxfull);
        if ((tx | ty | tz) < 0) return;                                  // hole (.cpp:285)
        const int f = (tz * st.ext[1] + ty) * st.ext[0] + tx;            // .cpp:290
        const T denom = (T)cnt[f * kCntStride + lane];                   // (T)fsize, .cpp:483
        const T *xr = in_cloud + (size_t)v.idx * cin;
        if constexpr (kSmall) {
            T xs[CIN];
#pragma unroll
            for (int k = 0; k < CIN; ++k) xs[k] = xr[k] / denom;         // x / count, .cpp:492
            const T *wf = w_lds + (size_t)f * CIN * COUT;
#pragma unroll
            for (int k = 0; k < CIN; ++k)
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] = __builtin_fma(wf[k * COUT + c], xs[k], acc[c]);
        } else {
            const T *wf = filter + (size_t)f * cin * cout;
            for (int c0 = 0; c0 < cout; c0 += 4) {
                T a[4] = {(T)0, (T)0, (T)0, (T)0};
                for (int k = 0; k < cin; ++k) {
                    const T xs = xr[k] / denom;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (c0 + u < cout) a[u] = __builtin_fma(wf[(size_t)k * cout + c0 + u], xs, a[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (c0 + u < cout) out_row[c0 + u] += a[u];
            }
        }
    });

    if (kSmall && q.orig >= 0) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) out_row[c] = acc[c];
    }
}

// ---------------------------------------------------------------------------------
// backward.  For centre j (the lane) and every ii in j's accepted set (.cpp:652):
//   f' = tap of j inside ii's box, clamp, NO inclusion re-test (.cpp:658-677),
//   count = population of tap f' of ii, skipped when 0 (.cpp:678-679),
//   g[c] = dY[ii,c] / count,  dX[j,k] += g[c] W[f',k,c],  dW[f',k,c] += g[c] X[j,k].
// Small path: dX row and X row in registers, filter in LDS, dW accumulated in an LDS copy
// shared by the workgroup (ds_add), written out as one partial per workgroup.
// Generic path: dX row in pre-zeroed global memory (lane-owned), dW through global atomics
// into partial slot 0 (the reduce kernel then just copies it).
// ---------------------------------------------------------------------------------
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void backward_kernel(
    const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes,
    const int32_t *__restrict__ count, const T *__restrict__ grad_out,
    const T *__restrict__ input, const T *__restrict__ filter, Stencil<T> st, int N, int ntiles,
    int cin_rt, int cout_rt, BlockMap bm, T *__restrict__ grad_input, T *__restrict__ partials)
{
    constexpr bool kSmall = CIN > 0;
    const int cin = kSmall ? CIN : cin_rt;
    const int cout = kSmall ? COUT : cout_rt;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    const size_t nw = (size_t)st.ntap * cin * cout;
    T *w_lds = reinterpret_cast<T *>(smem + off);
    if (kSmall) off += align16(nw * sizeof(T));
    T *dw_lds = reinterpret_cast<T *>(smem + off);
    if (kSmall) off += align16(nw * sizeof(T));
    const size_t per_wave = align16(sizeof(PointRec<T>) * kTile);
    PointRec<T> *tile_lds = reinterpret_cast<PointRec<T> *>(smem + off + per_wave * (threadIdx.x >> 6));

    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    if (kSmall)
        for (size_t e = threadIdx.x; e < nw; e += blockDim.x) {
            w_lds[e] = filter[e];
            dw_lds[e] = (T)0;
        }
    __syncthreads();

    int b, blk;
    const bool live = block_to_cloud(bm, b, blk);
    const int lane = threadIdx.x & 63;
    const int qt = blk * kWavesPerBlock + (threadIdx.x >> 6);

    if (live && qt < ntiles) {
        const PointRec<T> *cloud_pts = pts + (size_t)b * ntiles * kTile;
        const T *cloud_box = boxes + (size_t)b * ntiles * 6;
        Query<T> q;
        make_query(q, cloud_pts[(size_t)qt * kTile + lane], st);
        const size_t jrow = (size_t)b * N + (q.orig < 0 ? 0 : q.orig);
        const int32_t *cnt_cloud = count + (size_t)b * N * st.ntap;
        const T *dy_cloud = grad_out + (size_t)b * N * cout;
        const T *x_row = input + jrow * cin;
        T *dx_row = grad_input + jrow * cin;

        T xj[kSmall ? CIN : 1], dx[kSmall ? CIN : 1];
        if (kSmall) {
#pragma unroll
            for (int k = 0; k < CIN; ++k) {
                xj[k] = q.orig >= 0 ? x_row[k] : (T)0;
                dx[k] = (T)0;
            }
        }

        for_each_box_hit(cloud_pts, cloud_box, ntiles, q, tile_lds, [&](const PointRec<T> &v) {
            // membership of ii in j's set includes j's own hole test (.cpp:285 via :652)
            const int sx = axis_tap(v.x, q.lo[0], st.voxel, st.full[0], tapmap);
            const int sy = axis_tap(v.y, q.lo[1], st.voxel, st.full[1], tapmap + st.maxfull);
            const int sz = axis_tap(v.z, q.lo[2], st.voxel, st.full[2], tapmap + 2 * st.maxfull);
            if ((sx | sy | sz) < 0) return;
            // tap of j inside the box centred on ii (.cpp:662-677)
            const T lx = (T)((double)v.x - st.half[0]);
            const T ly = (T)((double)v.y - st.half[1]);
            const T lz = (T)((double)v.z - st.half[2]);
            const int tx = axis_tap(q.p[0], lx, st.voxel, st.full[0], tapmap);
            const int ty = axis_tap(q.p[1], ly, st.voxel, st.full[1], tapmap + st.maxfull);
            const int tz = axis_tap(q.p[2], lz, st.voxel, st.full[2], tapmap + 2 * st.maxfull);
            if ((tx | ty | tz) < 0) return;                                   // .cpp:672
            const int f = (tz * st.ext[1] + ty) * st.ext[0] + tx;             // .cpp:677
            const int cn = cnt_cloud[(size_t)v.idx * st.ntap + f];
            if (cn == 0) return;                                              // .cpp:679
            const T denom = (T)cn;
            const T *dyr = dy_cloud + (size_t)v.idx * cout;
            if constexpr (kSmall) {
                T g[COUT];
#pragma unroll
                for (int c = 0; c < COUT; ++c) g[c] = dyr[c] / denom;
                const T *wf = w_lds + (size_t)f * CIN * COUT;
                T *dwf = dw_lds + (size_t)f * CIN * COUT;
#pragma unroll
                for (int k = 0; k < CIN; ++k) {
#pragma unroll
                    for (int c = 0; c < COUT; ++c) {
                        dx[k] = __builtin_fma(g[c], wf[k * COUT + c], dx[k]);           // .cpp:692
                        __hip_atomic_fetch_add(&dwf[k * COUT + c], g[c] * xj[k], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);           // .cpp:696
                    }
                }
            } else {
                const T *wf = filter + (size_t)f * cin * cout;
                T *dwf = partials + (size_t)f * cin * cout;
                for (int k = 0; k < cin; ++k) {
                    const T xk = x_row[k];
                    T a = (T)0;
                    for (int c = 0; c < cout; ++c) {
                        const T g = dyr[c] / denom;
                        a = __builtin_fma(g, wf[(size_t)k * cout + c], a);
                        __hip_atomic_fetch_add(&dwf[(size_t)k * cout + c], g * xk, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    }
                    dx_row[k] += a;
                }
            }
        });

        if (kSmall && q.orig >= 0) {
#pragma unroll
            for (int k = 0; k < CIN; ++k) dx_row[k] = dx[k];
        }
    }

    if (kSmall) {
        __syncthreads();
        T *slot = partials + (size_t)blockIdx.x * nw;
        for (size_t e = threadIdx.x; e < nw; e += blockDim.x) slot[e] = dw_lds[e];
    }
}

// grad_filter[e] = sum over partial slots, fixed order (slot index ascending within a
// thread's stripe, stripes combined in a fixed tree) -> run-to-run deterministic given
// deterministic partials.
template <typename T>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const T *__restrict__ partials,
                                                              int nslots, size_t nw,
                                                              T *__restrict__ grad_filter)
{
    __shared__ T part[kWavesPerBlock][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t e = (size_t)blockIdx.x * 64 + lane;
    T s = (T)0;
    if (e < nw)
        for (int p = wave; p < nslots; p += kWavesPerBlock) s += partials[(size_t)p * nw + e];
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && e < nw) grad_filter[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// SELU (selu.py:22-26): scale * (x >= 0 ? x : alpha * (exp(x) - 1)).
template <typename T> struct SeluConst {
    static constexpr double alpha = 1.6732632423543772848170429916717;
    static constexpr double scale = 1.0507009873554804934193349852946;
};
template <typename T>
__global__ __launch_bounds__(256) void selu_kernel(const T *__restrict__ x, T *__restrict__ y, size_t n)
{
    const T alpha = (T)SeluConst<T>::alpha, scale = (T)SeluConst<T>::scale;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = x[i];
        y[i] = scale * (v >= (T)0 ? v : alpha * (T)expm1((double)v));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void selu_grad_kernel(const T *__restrict__ y, const T *__restrict__ dy,
                                                        const T *__restrict__ dy_b, T *__restrict__ dx, size_t n)
{
    const T alpha = (T)SeluConst<T>::alpha, scale = (T)SeluConst<T>::scale;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = y[i];
        const T g = dy_b ? dy[i] + dy_b[i] : dy[i];
        dx[i] = g * (v >= (T)0 ? scale : v + scale * alpha);
    }
}

}  // namespace conv3p
