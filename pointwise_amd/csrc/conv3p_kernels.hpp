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

// ---------------------------------------------------------------------------------
// prep with spatial sort: one workgroup per cloud.  Points are ordered by the Morton code of
// their position inside the cloud's bounding cube (10 bits per axis), sorted in LDS with a
// bitonic network on 64-bit (code << 32 | index) keys, then staged as records; every run of 64
// sorted points is a tile with a tight bounding box, which is what makes the candidate-tile
// culling of for_each_box_hit effective.  The order only affects speed: neighbour decisions
// are taken per pair with the reference's arithmetic, whatever the tiling.
// Requires npad (power of two >= N) * 8 bytes of LDS: N <= 16384.  Larger clouds use
// prep_kernel (identity order).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

template <typename T>
__global__ __launch_bounds__(1024) void prep_sort_kernel(const T *__restrict__ points, int N, int ntiles,
                                                         int npad, PointRec<T> *__restrict__ pts,
                                                         T *__restrict__ boxes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    __shared__ float red[6][16];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const T *cloud = points + (size_t)b * N * 3;

    // bounding cube (float precision is enough: the order is a performance hint only)
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = tid; i < N; i += nthr)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = (float)cloud[(size_t)i * 3 + a];
            mn[a] = v < mn[a] ? v : mn[a];
            mx[a] = v > mx[a] ? v : mx[a];
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_min(mn[a]);
        mx[a] = wave_max(mx[a]);
        if (lane == 0) {
            red[a][wave] = mn[a];
            red[3 + a][wave] = mx[a];
        }
    }
    __syncthreads();
    const int nwaves = nthr >> 6;
    float ext = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = red[a][0], hi = red[3 + a][0];
        for (int w = 1; w < nwaves; ++w) {
            lo = red[a][w] < lo ? red[a][w] : lo;
            hi = red[3 + a][w] > hi ? red[3 + a][w] : hi;
        }
        mn[a] = lo;
        ext = (hi - lo) > ext ? (hi - lo) : ext;
    }
    const float scale = ext > 0.f ? 1023.0f / ext : 0.f;

    for (int i = tid; i < npad; i += nthr) {
        uint64_t k = ~0ull;   // padding sorts last
        if (i < N) {
            uint32_t q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float f = ((float)cloud[(size_t)i * 3 + a] - mn[a]) * scale;
                f = f < 0.f ? 0.f : (f > 1023.f ? 1023.f : f);
                q[a] = (uint32_t)f;
            }
            const uint32_t code = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
            k = ((uint64_t)code << 32) | (uint32_t)i;
        }
        keys[i] = k;
    }
    __syncthreads();

    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (npad >> 1); t += nthr) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const uint64_t a = keys[i], c = keys[l];
                const bool up = (i & k) == 0;
                if ((a > c) == up) {
                    keys[i] = c;
                    keys[l] = a;
                }
            }
            __syncthreads();
        }

    const T inf = Limits<T>::inf();
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int p = tile * kTile + lane;
        PointRec<T> r;
        const bool v = p < N;
        if (v) {
            const int i = (int)(uint32_t)keys[p];
            r.x = cloud[(size_t)i * 3 + 0];
            r.y = cloud[(size_t)i * 3 + 1];
            r.z = cloud[(size_t)i * 3 + 2];
            r.idx = i;
        } else {
            r.x = r.y = r.z = inf;
            r.idx = -1;
        }
        pts[((size_t)b * ntiles + tile) * kTile + lane] = r;
        T bmn[3] = {wave_min(v ? r.x : inf), wave_min(v ? r.y : inf), wave_min(v ? r.z : inf)};
        T bmx[3] = {wave_max(v ? r.x : -inf), wave_max(v ? r.y : -inf), wave_max(v ? r.z : -inf)};
        if (lane == 0) {
            T *bb = boxes + ((size_t)b * ntiles + tile) * 6;
            bb[0] = bmn[0]; bb[1] = bmn[1]; bb[2] = bmn[2];
            bb[3] = bmx[0]; bb[4] = bmx[1]; bb[5] = bmx[2];
        }
    }
}

// LDS carve helpers (all offsets multiples of 16 B; one extern array per kernel).
__device__ __forceinline__ size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// Workgroup geometry: WAVES waves share one query tile and split its candidate tiles.
// WAVES = 4 for the register-resident ("small") paths; WAVES = 1 for the generic channel
// path, whose output rows are accumulated in global memory and must have a single owner.

// ---------------------------------------------------------------------------------
// count: count[(b*N + i)*F + f] = neighbours of i in tap f   (.cpp:306-379)
// LDS: tapmap | [F][65] u32 (shared by the workgroup, ds_add) | per wave SoA slot
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void count_kernel(const PointRec<T> *__restrict__ pts,
                                                    const T *__restrict__ boxes, Stencil<T> st,
                                                    int N, int ntiles, BlockMap bm,
                                                    int32_t *__restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + off);
    off += align16((size_t)st.ntap * kCntStride * 4);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *soa = reinterpret_cast<float *>(smem + off) + wave * 192;

    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    for (int e = threadIdx.x; e < st.ntap * kCntStride; e += blockDim.x) cnt[e] = 0;
    __syncthreads();

    int b, qt;
    if (!block_to_cloud(bm, b, qt)) return;
    const PointRec<T> *cloud_pts = pts + (size_t)b * ntiles * kTile;
    const T *cloud_box = boxes + (size_t)b * ntiles * 6;
    Query<T> q;
    make_query(q, cloud_pts[(size_t)qt * kTile + lane], st);

    for_each_neighbor(cloud_pts, cloud_box, ntiles, q, st, tapmap, soa, wave, kWavesPerBlock,
                      [&](const PointRec<T> &, int f) { atomicAdd(&cnt[f * kCntStride + lane], 1u); });
    __syncthreads();
    // row-wise write-out: lanes = taps, waves take every 4th query (coalesced F*4-byte rows)
    for (int qq = wave; qq < kTile; qq += kWavesPerBlock) {
        const int orig = __shfl(q.orig, qq);
        if (orig < 0) continue;
        int32_t *row = count + ((size_t)b * N + orig) * st.ntap;
        for (int f = lane; f < st.ntap; f += 64) row[f] = (int32_t)cnt[f * kCntStride + qq];
    }
}

// ---------------------------------------------------------------------------------
// forward, fused with its own population count.  One workgroup = one query tile.
//   pass 1  every wave pre-filters its share of the candidate tiles, resolves the hits
//           exactly, adds them to the shared [tap][lane] populations and keeps the hit masks
//           of exact neighbours in LDS (mask slot = position in the wave's tile sequence);
//   pass 2  after a barrier the populations are final: the waves walk their saved masks again
//           and accumulate  out[i,c] += W[f,k,c] * (x[j,k] / count[i,f])      (.cpp:480-494)
//           tiles beyond the mask capacity are simply searched again.
// CIN/COUT > 0: output row in registers (one partial per wave, summed through LDS), filter in
// LDS.  CIN == 0: generic shapes, single-wave workgroups, output row accumulated in
// pre-zeroed global memory, filter read through L1/L2.
// LDS: tapmap | filter | [F][65] u32 | masks [WAVES][cap][64] u64 | per wave SoA | reduce
// ---------------------------------------------------------------------------------
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void forward_kernel(
    const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes, const T *__restrict__ input,
    const T *__restrict__ filter, Stencil<T> st, int N, int ntiles, int cin_rt, int cout_rt, int mask_cap,
    BlockMap bm, T *__restrict__ output)
{
    constexpr bool kSmall = CIN > 0;
    const int cin = kSmall ? CIN : cin_rt;
    const int cout = kSmall ? COUT : cout_rt;
    const int nwaves = blockDim.x >> 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    T *w_lds = reinterpret_cast<T *>(smem + off);
    const size_t nw = (size_t)st.ntap * cin * cout;
    if (kSmall) off += align16(nw * sizeof(T));
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + off);
    off += align16((size_t)st.ntap * kCntStride * 4);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint64_t *masks = reinterpret_cast<uint64_t *>(smem + off) + (size_t)wave * mask_cap * 64;
    off += align16((size_t)nwaves * mask_cap * 64 * 8);
    float *soa = reinterpret_cast<float *>(smem + off) + wave * 192;
    off += align16((size_t)nwaves * 192 * 4);
    T *red = reinterpret_cast<T *>(smem + off);   // [nwaves][COUT][64], small path only

    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    if (kSmall)
        for (size_t e = threadIdx.x; e < nw; e += blockDim.x) w_lds[e] = filter[e];
    for (int e = threadIdx.x; e < st.ntap * kCntStride; e += blockDim.x) cnt[e] = 0;
    __syncthreads();

    int b, qt;
    const bool live = block_to_cloud(bm, b, qt);
    if (!live) return;   // uniform for the whole workgroup
    const PointRec<T> *cloud_pts = pts + (size_t)b * ntiles * kTile;
    const T *cloud_box = boxes + (size_t)b * ntiles * 6;
    Query<T> q;
    make_query(q, cloud_pts[(size_t)qt * kTile + lane], st);
    const bool qvalid = q.orig >= 0;

    // ---- pass 1: populations + exact hit masks
    {
        int seen = 0, slot = 0;
        for (int base = 0; base < ntiles; base += 64) {
            uint64_t tiles = overlapping_tiles(cloud_box, ntiles, base, q);
            while (tiles) {
                const int ct = base + __builtin_ctzll(tiles);
                tiles &= tiles - 1;
                if ((seen++ & (nwaves - 1)) != wave) continue;
                const PointRec<T> *tile = cloud_pts + (size_t)ct * kTile;
                stage_tile(soa, tile[lane]);
                __builtin_amdgcn_wave_barrier();
                uint32_t m0, m1;
                scan_tile(soa, q, st, m0, m1);
                __builtin_amdgcn_wave_barrier();
                if (!qvalid) m0 = m1 = 0;
                uint32_t k0 = 0, k1 = 0;   // exact neighbours, same bit layout
                for_each_bit(m0, m1, [&](int c) {
                    const int f = exact_tap(tile[c], q, st, tapmap);
                    if (f >= 0) {
                        atomicAdd(&cnt[f * kCntStride + lane], 1u);
                        if (c < 32) k0 |= 1u << (31 - c); else k1 |= 1u << (63 - c);
                    }
                });
                if (slot < mask_cap) masks[(size_t)slot * 64 + lane] = ((uint64_t)k1 << 32) | k0;
                ++slot;
            }
        }
    }
    __syncthreads();

    // ---- pass 2: accumulate
    const T *in_cloud = input + (size_t)b * N * cin;
    T *out_row = output + ((size_t)b * N + (qvalid ? q.orig : 0)) * cout;
    T acc[kSmall ? COUT : 1];
    if (kSmall) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = (T)0;
    }
    auto accumulate = [&](const PointRec<T> &v, int f) {
        const T denom = (T)cnt[f * kCntStride + lane];                   // (T)fsize, .cpp:483
        const T *xr = in_cloud + (size_t)v.idx * cin;
        if constexpr (kSmall) {
            const T rcp = (T)1 / denom;                                  // one IEEE division per pair
            T xs[CIN];
#pragma unroll
            for (int k = 0; k < CIN; ++k) xs[k] = xr[k] * rcp;           // x / count, .cpp:492
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
    };
    {
        int seen = 0, slot = 0;
        for (int base = 0; base < ntiles; base += 64) {
            uint64_t tiles = overlapping_tiles(cloud_box, ntiles, base, q);
            while (tiles) {
                const int ct = base + __builtin_ctzll(tiles);
                tiles &= tiles - 1;
                if ((seen++ & (nwaves - 1)) != wave) continue;
                const PointRec<T> *tile = cloud_pts + (size_t)ct * kTile;
                if (slot < mask_cap) {
                    const uint64_t m = masks[(size_t)slot * 64 + lane];
                    for_each_bit((uint32_t)m, (uint32_t)(m >> 32), [&](int c) {
                        const PointRec<T> v = tile[c];
                        const int f = exact_tap(v, q, st, tapmap);       // >= 0 by construction
                        accumulate(v, f);
                    });
                } else {
                    stage_tile(soa, tile[lane]);
                    __builtin_amdgcn_wave_barrier();
                    uint32_t m0, m1;
                    scan_tile(soa, q, st, m0, m1);
                    __builtin_amdgcn_wave_barrier();
                    if (!qvalid) m0 = m1 = 0;
                    for_each_bit(m0, m1, [&](int c) {
                        const PointRec<T> v = tile[c];
                        const int f = exact_tap(v, q, st, tapmap);
                        if (f >= 0) accumulate(v, f);
                    });
                }
                ++slot;
            }
        }
    }

    if constexpr (kSmall) {
        // fixed-order sum of the per-wave partial rows
#pragma unroll
        for (int c = 0; c < COUT; ++c) red[((size_t)wave * COUT + c) * 64 + lane] = acc[c];
        __syncthreads();
        for (int e = threadIdx.x; e < COUT * 64; e += blockDim.x) {
            const int c = e >> 6;   // e & 63 == lane: every wave holds the same 64 queries
            T s = red[((size_t)0 * COUT + c) * 64 + lane];
            for (int w = 1; w < nwaves; ++w) s += red[((size_t)w * COUT + c) * 64 + lane];
            if (qvalid) output[((size_t)b * N + q.orig) * COUT + c] = s;
        }
    }
}

// ---------------------------------------------------------------------------------
// backward.  For centre j (the lane) and every ii in j's accepted set (.cpp:652):
//   f' = tap of j inside ii's box, clamp, NO inclusion re-test (.cpp:658-677),
//   count = population of tap f' of ii, skipped when 0 (.cpp:678-679),
//   g[c] = dY[ii,c] / count,  dX[j,k] += g[c] W[f',k,c],  dW[f',k,c] += g[c] X[j,k].
//
// Small path (one workgroup = one query tile, 4 waves split the candidate tiles):
//   phase A  the hit loop only gathers  G[(f',c)][j] += dY[ii,c] * (1/count)  into LDS
//            ([row][lane], stride 65: lane-private columns, conflict-free ds_add);
//   phase B  lanes = rows (f',c):  dW[f',k,c] = sum_j G[row][j] * X[j,k]   (X tile broadcast
//            from LDS), written straight to this workgroup's partial slot;
//   phase C  lanes = centres j, waves split the rows:  dX[j,k] = sum_row G[row][j] * W[row][k]
//            (W broadcast from LDS), per-wave partial rows summed through LDS in fixed order.
//   Both contractions are dense and divergence-free; no float atomics leave the workgroup, and
//   the only LDS atomics are phase A's (summation order inside a workgroup is the only
//   non-fixed order in the op).
// Generic path: single-wave workgroups, dX row in pre-zeroed global memory (lane-owned), dW
// through global atomics into partial slot 0.
// LDS small: tapmap | Wt [F*COUT][CIN] | G [F*COUT][65] | X tile [64][CIN] | per wave SoA | reduce
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
    const int nwaves = blockDim.x >> 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    const size_t nw = (size_t)st.ntap * cin * cout;
    const int nrows = st.ntap * cout;
    T *wt = reinterpret_cast<T *>(smem + off);        // Wt[row][k], row = f*COUT + c
    if (kSmall) off += align16(nw * sizeof(T));
    T *G = reinterpret_cast<T *>(smem + off);         // G[row][65]
    if (kSmall) off += align16((size_t)nrows * kCntStride * sizeof(T));
    T *xt = reinterpret_cast<T *>(smem + off);        // X tile [64][CIN]
    if (kSmall) off += align16((size_t)64 * cin * sizeof(T));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *soa = reinterpret_cast<float *>(smem + off) + wave * 192;
    off += align16((size_t)nwaves * 192 * 4);
    T *red = reinterpret_cast<T *>(smem + off);       // [nwaves][CIN][64], small path only

    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    if (kSmall) {
        for (int e = threadIdx.x; e < (int)nw; e += blockDim.x) {
            const int row = e / CIN, k = e - row * CIN;
            const int f = row / COUT, c = row - f * COUT;
            wt[e] = filter[((size_t)f * CIN + k) * COUT + c];
        }
        for (int e = threadIdx.x; e < nrows * kCntStride; e += blockDim.x) G[e] = (T)0;
    }

    int b, qt;
    const bool live = block_to_cloud(bm, b, qt);   // uniform for the workgroup
    const PointRec<T> *cloud_pts = pts + (size_t)(live ? b : 0) * ntiles * kTile;
    const T *cloud_box = boxes + (size_t)(live ? b : 0) * ntiles * 6;
    Query<T> q;
    make_query(q, cloud_pts[(size_t)(live ? qt : 0) * kTile + lane], st);
    if (!live) q.orig = -1;
    const size_t jrow = (size_t)(live ? b : 0) * N + (q.orig < 0 ? 0 : q.orig);
    if (kSmall && wave == 0) {
#pragma unroll
        for (int k = 0; k < (kSmall ? CIN : 1); ++k) xt[lane * cin + k] = q.orig >= 0 ? input[jrow * cin + k] : (T)0;
    }
    __syncthreads();

    if (live) {
        const int32_t *cnt_cloud = count + (size_t)b * N * st.ntap;
        const T *dy_cloud = grad_out + (size_t)b * N * cout;
        const T *x_row = input + jrow * cin;
        T *dx_row = grad_input + jrow * cin;

        // membership of ii in j's set (incl. j's own hole test, .cpp:285 via :652) is what
        // for_each_neighbor delivers; the tap it reports (of ii inside j's box) is not used.
        for_each_neighbor(cloud_pts, cloud_box, ntiles, q, st, tapmap, soa, wave, nwaves,
                          [&](const PointRec<T> &v, int) {
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
            const T *dyr = dy_cloud + (size_t)v.idx * cout;
            if constexpr (kSmall) {
                const T rcp = (T)1 / (T)cn;
                T *grow = G + ((size_t)f * COUT) * kCntStride + lane;
#pragma unroll
                for (int c = 0; c < COUT; ++c)
                    __hip_atomic_fetch_add(&grow[c * kCntStride], dyr[c] * rcp, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                const T denom = (T)cn;
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
    }

    if constexpr (kSmall) {
        __syncthreads();
        // ---- phase B: dW rows.  thread = row (f,c); X tile read with wave-uniform addresses.
        T *slot = partials + (size_t)blockIdx.x * nw;
        for (int row = threadIdx.x; row < nrows; row += blockDim.x) {
            T acc[CIN];
#pragma unroll
            for (int k = 0; k < CIN; ++k) acc[k] = (T)0;
            const T *grow = G + (size_t)row * kCntStride;
            for (int j = 0; j < 64; ++j) {
                const T g = grow[j];
#pragma unroll
                for (int k = 0; k < CIN; ++k) acc[k] = __builtin_fma(g, xt[j * CIN + k], acc[k]);
            }
            const int f = row / COUT, c = row - f * COUT;
#pragma unroll
            for (int k = 0; k < CIN; ++k) slot[((size_t)f * CIN + k) * COUT + c] = acc[k];
        }
        // ---- phase C: dX rows.  lane = centre j, waves split the rows.
        T dx[CIN];
#pragma unroll
        for (int k = 0; k < CIN; ++k) dx[k] = (T)0;
        for (int row = wave; row < nrows; row += nwaves) {
            const T g = G[(size_t)row * kCntStride + lane];
            if (!__any(g != (T)0)) continue;
            const T *wr = wt + (size_t)row * CIN;
#pragma unroll
            for (int k = 0; k < CIN; ++k) dx[k] = __builtin_fma(g, wr[k], dx[k]);
        }
#pragma unroll
        for (int k = 0; k < CIN; ++k) red[((size_t)wave * CIN + k) * 64 + lane] = dx[k];
        __syncthreads();
        if (live)
            for (int e = threadIdx.x; e < CIN * 64; e += blockDim.x) {
                const int k = e >> 6;   // e & 63 == lane
                T sum = red[((size_t)0 * CIN + k) * 64 + lane];
                for (int w = 1; w < nwaves; ++w) sum += red[((size_t)w * CIN + k) * 64 + lane];
                if (q.orig >= 0) grad_input[((size_t)b * N + q.orig) * CIN + k] = sum;
            }
    }
}

// grad_filter[e] = sum over partial slots, fixed order (slot index ascending within a wave's
// stripe, stripes combined in ascending wave order) -> run-to-run deterministic given
// deterministic partials.  16 waves per workgroup stride over the slots; 64 weights per workgroup.
template <typename T>
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const T *__restrict__ partials,
                                                               int nslots, size_t nw,
                                                               T *__restrict__ grad_filter)
{
    __shared__ T part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t e = (size_t)blockIdx.x * 64 + lane;
    T s0 = (T)0, s1 = (T)0, s2 = (T)0, s3 = (T)0;
    if (e < nw) {
        int p = wave;
        for (; p + 48 < nslots; p += 64) {
            s0 += partials[(size_t)p * nw + e];
            s1 += partials[(size_t)(p + 16) * nw + e];
            s2 += partials[(size_t)(p + 32) * nw + e];
            s3 += partials[(size_t)(p + 48) * nw + e];
        }
        for (; p < nslots; p += 16) s0 += partials[(size_t)p * nw + e];
    }
    part[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && e < nw) {
        T s = part[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) s += part[w][lane];
        grad_filter[e] = s;
    }
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
