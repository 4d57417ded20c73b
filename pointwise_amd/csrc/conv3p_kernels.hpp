// conv3p_kernels.hpp -- the gfx950 kernels of the conv3p operator pair.
// See conv3p_device.hpp for the search structure; this file holds
//   prep_kernel            stage points as 16-byte records + per-tile bounding boxes
//   search_kernel          neighbour search: per-tap populations (Grid::neighbor_count) + pair lists
//   forward_kernel         Conv3p accumulate     (reference tf_conv3p_atrous.cpp:453-504)
//   backward_kernel        Conv3pGrad accumulate (reference tf_conv3p_atrous.cpp:608-716)
//   reduce_partials_kernel deterministic second stage of grad_filter
//   selu kernels           the activation between the stack's layers (selu.py:22-26)
#pragma once

#include "conv3p_dev.hpp"      // developer instruments (ablation switches, per-phase stamps): nothing in product builds
#include "conv3p_device.hpp"
#include "conv3p_deep.hpp"

#ifndef CONV3P_BWD_MFMA
#define CONV3P_BWD_MFMA 1   // backward phases B and C on the matrix cores (fp32); 0: the vector-ALU version (A/B builds)
#endif

namespace conv3p {

// ---------------------------------------------------------------------------------
// Neighbour cache control (device side).  A cache buffer holds, per cloud b,
//   hash[b]     64-bit content hash of the cloud's raw coordinates (prep_sort_kernel)
//   version[b]  bumped by prep_sort_kernel whenever hash[b] changes (single writer: one
//               workgroup per cloud)
// and per search slot (one slot = one stencil: filter extents, stride, voxel)
//   built_version[b], built_tag[b]   what the slot's pair lists of cloud b were built from
//   cursor[b]                        pair-slot allocator of cloud b's region
//   ticket[b]                        query tiles of cloud b that have finished rebuilding in the running search
//                                    launch; the last one commits built_version / built_tag and resets it to 0
// A slot is valid for cloud b iff built_version[b] == version[b] && built_tag[b] == tag.
// Nothing is ever read back by the host: every decision is taken on the device, so a stale
// or recycled buffer can only cost a rebuild, never a wrong result.
// force != 0 (the stateless entry points): ignore the buffer's history and rebuild.
// ---------------------------------------------------------------------------------
// Row strides (in elements) of the feature tensors of one op call; the reference's tensors are dense (stride =
// channel count), the stack-level entry points let a layer read / write a column block of a wider buffer (the
// (B, N, 36) concat of the models, pointcnn2_acsd.py:68) without a copy.
struct RowLd {
    int in;    // input rows                      (forward, backward)
    int out;   // output rows                     (forward)
    int dy;    // grad_out rows                   (backward)
    int dx;    // grad_input rows                 (backward)
    int add;   // grad_addend rows                (backward, fused SELU-gradient epilogue)
};

struct CacheCtl {
    unsigned long long *hash;        // [B]
    uint32_t *version;               // [B]
    uint32_t *built_version;         // [B] of the slot in use
    unsigned long long *built_tag;   // [B]
    uint32_t *cursor;                // [B] of the slot in use
    uint32_t *ticket;                // [B] of the slot in use
    uint32_t *cursor_all;            // [nslots][2][B]: all slots' allocators and tickets (contiguous)
    uint32_t *tab_version, *tab_ticket;   // [B] the window tables' validity mark and completion count (tile_tables_kernel)
    int nslots, nclouds;
    unsigned long long tag;
    uint32_t epoch;
    uint32_t pairs_per_cloud;
    int force;
};

__device__ __forceinline__ bool slot_valid(const CacheCtl &cc, int b)
{
    return !cc.force && cc.built_version[b] == cc.version[b] && cc.built_tag[b] == cc.tag;
}

// SELU (selu.py:22-26): scale * (x >= 0 ? x : alpha * (exp(x) - 1)); its derivative expressed through the
// OUTPUT y (what TensorFlow's SeluGrad takes): y >= 0 ? scale : y + scale * alpha.
template <typename T> struct SeluConst {
    static constexpr double alpha = 1.6732632423543772848170429916717;
    static constexpr double scale = 1.0507009873554804934193349852946;
};
__device__ __forceinline__ float expm1_t(float v) { return expm1f(v); }
__device__ __forceinline__ double expm1_t(double v) { return expm1(v); }
template <typename T> __device__ __forceinline__ T selu_value(T v)
{
    return (T)SeluConst<T>::scale * (v >= (T)0 ? v : (T)SeluConst<T>::alpha * expm1_t(v));
}
template <typename T> __device__ __forceinline__ T selu_slope(T y)
{
    return y >= (T)0 ? (T)SeluConst<T>::scale : y + (T)(SeluConst<T>::scale * SeluConst<T>::alpha);
}

// ---------------------------------------------------------------------------------
// prep: one wavefront per tile.  Identity order (tile t = points [64t, 64t+64)).
// Padding lanes of the last tile get +inf coordinates (rejected by every finite box)
// and idx = -1.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void prep_kernel(const T *__restrict__ points, int N, int ntiles,
                                                   PointRec<T> *__restrict__ pts,
                                                   T *__restrict__ boxes, CacheCtl cc)
{
    // clouds too large for the LDS sort are never cached: bump the version every call
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cc.version[blockIdx.y] += 1;
        cc.hash[blockIdx.y] = 0;
        for (int sl = 0; sl < 2 * cc.nslots; ++sl) cc.cursor_all[(size_t)sl * cc.nclouds + blockIdx.y] = 0;
    }
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
// prep with spatial sort: one workgroup per cloud.  Points are ordered by the Hilbert index of
// their position inside the cloud's bounding cube (10 bits per axis), sorted in LDS with a
// bitonic network on 64-bit (code << 32 | index) keys, then staged as records; every run of 64
// sorted points is a tile with a tight bounding box, which is what makes the candidate-tile
// culling of for_each_box_hit effective.  The order only affects speed: neighbour decisions
// are taken per pair with the reference's arithmetic, whatever the tiling.
// Requires npad (power of two >= N) * 8 + 256 bytes of LDS: N <= 16384.  Larger clouds use
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
// 30-bit Hilbert index of a 10-bit lattice point (Skilling's transpose algorithm).  Consecutive indices are
// always lattice neighbours -- a Morton curve jumps across the cube at every power-of-two boundary -- so
// runs of 64 (tiles) and of 4 (quads) sorted points have ~25 % smaller boxes and the search tests ~20 %
// fewer candidates (measured on the bench clouds).
__device__ __forceinline__ uint32_t hilbert30(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t X[3] = {x & 0x3ffu, y & 0x3ffu, z & 0x3ffu};
    for (uint32_t Q = 1u << 9; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const uint32_t t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    uint32_t t = 0;
    for (uint32_t Q = 1u << 9; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    return (spread10(X[0] ^ t) << 2) | (spread10(X[1] ^ t) << 1) | spread10(X[2] ^ t);
}

// Round 5: the keys are 32-bit -- the top cb bits of the Hilbert index (cb = the largest multiple of 3 that leaves room
// for the point's index: 21 bits = 7 per axis at N <= 2048, 18 above; the curve is hierarchical, so its leading bits are
// the index of the coarser cell) over the index -- and live in REGISTERS, KPT per thread (element e = s * threads + t):
// compare-exchanges between a thread's own keys need nothing, those inside a wave one lane exchange (ds_bpermute) and
// v_min / v_max, and only the stages whose partners sit in different waves (14 of the 66 at N = 2048) go through LDS,
// one barrier each (two buffers).  Before: every stage read and wrote 64-bit keys in LDS behind a workgroup barrier,
// ~30 instructions per compare-exchange on 16 waves: 18.6 us of the kernel's 36 at the cfg2 size.  The hash and the
// bounding cube share one pass over the cloud.
template <typename T, int KPT>
__global__ __launch_bounds__(1024) void prep_sort_kernel(const T *__restrict__ points, int N, int ntiles,
                                                         int npad, PointRec<T> *__restrict__ pts,
                                                         T *__restrict__ boxes, CacheCtl cc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *xbuf = reinterpret_cast<uint32_t *>(smem);   // [2][npad]: the cross-wave stages' exchange buffers
    __shared__ float red[6][16];
    const int tid = threadIdx.x, nthr = blockDim.x;   // nthr * KPT == npad
    const int lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const T *cloud = points + (size_t)b * N * 3;

    // ---- one pass: content hash of the raw coordinates (order-sensitive per element, order-free combination) and
    //      the bounding cube (float precision is enough: the order is a performance hint only)
    unsigned long long *hred = reinterpret_cast<unsigned long long *>(smem + (size_t)npad * 8);   // [16] + flag
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    {
        unsigned long long h = 0;
        constexpr int WPP = 3 * (int)(sizeof(T) / 4);   // 32-bit words per point
        const uint32_t *raw = reinterpret_cast<const uint32_t *>(cloud);
        for (int i = tid; i < N; i += nthr) {
#pragma unroll
            for (int w = 0; w < WPP; ++w) {
                unsigned long long x = ((unsigned long long)(uint32_t)(i * WPP + w) << 32) | raw[(size_t)i * WPP + w];
                x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
                h += x;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = (float)cloud[(size_t)i * 3 + a];
                mn[a] = v < mn[a] ? v : mn[a];
                mx[a] = v > mx[a] ? v : mx[a];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = wave_min(mn[a]);
            mx[a] = wave_max(mx[a]);
        }
        if (lane == 0) {
            hred[wave] = h;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                red[a][wave] = mn[a];
                red[3 + a][wave] = mx[a];
            }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < (nthr >> 6); ++w) t += hred[w];
            t |= 1ull;   // never 0: a zero-filled buffer is always stale
            const bool same = !cc.force && cc.hash[b] == t;
            if (!same) {
                cc.hash[b] = t;
                cc.version[b] += 1;
                // every slot's lists of this cloud are stale now: their allocators restart from empty
                for (int sl = 0; sl < 2 * cc.nslots; ++sl) cc.cursor_all[(size_t)sl * cc.nclouds + b] = 0;
                // ... and so are its window tables (a never-initialised or re-shaped buffer may hold anything in the count)
                cc.tab_version[b] = cc.version[b] - 1u;
                cc.tab_ticket[b] = 0;
            }
            // the slot about to be used must allocate from an empty region if it is going to rebuild
            const bool valid = same && cc.built_version[b] == cc.version[b] && cc.built_tag[b] == cc.tag;
            if (!valid) {
                cc.cursor[b] = 0;
                cc.ticket[b] = 0;   // (a never-initialised buffer may hold anything here)
            }
            hred[16] = same ? 1ull : 0ull;
        }
        __syncthreads();
        if (hred[16] != 0) return;   // sorted records and tile boxes of this cloud are still current
    }
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
    const int ib = 31 - __builtin_clz((unsigned)npad);   // bits of a point's index (npad is a power of two)
    const int cb = ((32 - ib) / 3) * 3 > 30 ? 30 : ((32 - ib) / 3) * 3;
    uint32_t key[KPT];
#pragma unroll
    for (int s = 0; s < KPT; ++s) {
        const int i = s * nthr + tid;
        uint32_t k = 0xFFFFFFFFu;   // padding sorts last
        if (i < N) {
            uint32_t q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float f = ((float)cloud[(size_t)i * 3 + a] - mn[a]) * scale;
                f = f < 0.f ? 0.f : (f > 1023.f ? 1023.f : f);
                q[a] = (uint32_t)f;
            }
            k = ((hilbert30(q[0], q[1], q[2]) >> (30 - cb)) << ib) | (uint32_t)i;
        }
        key[s] = k;
    }

    // ---- bitonic network on element e = s * nthr + tid; ascending blocks where (e & k) == 0
    int cur = 0;
    for (int k = 2; k <= npad; k <<= 1) {
        // partners in the same thread: distance nthr << m
#pragma unroll
        for (int m = KPT / 2; m >= 1; m >>= 1) {
            if (m * nthr < k) {
#pragma unroll
                for (int s = 0; s < KPT; ++s)
                    if ((s & m) == 0) {
                        const bool up = ((s * nthr + tid) & k) == 0;
                        const uint32_t a = key[s], c = key[s | m];
                        const uint32_t lo = a < c ? a : c, hi = a < c ? c : a;
                        key[s] = up ? lo : hi;
                        key[s | m] = up ? hi : lo;
                    }
            }
        }
        // partners in other waves: through LDS, one barrier per stage (alternating buffers)
        for (int j = (k >> 1) < (nthr >> 1) ? (k >> 1) : (nthr >> 1); j >= 64; j >>= 1) {
            uint32_t *xb = xbuf + (size_t)cur * npad;
#pragma unroll
            for (int s = 0; s < KPT; ++s) xb[s * nthr + tid] = key[s];
            __syncthreads();
#pragma unroll
            for (int s = 0; s < KPT; ++s) {
                const uint32_t pk = xb[s * nthr + (tid ^ j)];
                const bool up = ((s * nthr + tid) & k) == 0, lower = (tid & j) == 0;
                const uint32_t lo = key[s] < pk ? key[s] : pk, hi = key[s] < pk ? pk : key[s];
                key[s] = (lower == up) ? lo : hi;
            }
            cur ^= 1;
        }
        // partners in the same wave: one lane exchange
        for (int j = (k >> 1) < 32 ? (k >> 1) : 32; j >= 1; j >>= 1) {
#pragma unroll
            for (int s = 0; s < KPT; ++s) {
                const uint32_t pk = (uint32_t)__shfl_xor((int)key[s], j);
                const bool up = ((s * nthr + tid) & k) == 0, lower = (tid & j) == 0;
                const uint32_t lo = key[s] < pk ? key[s] : pk, hi = key[s] < pk ? pk : key[s];
                key[s] = (lower == up) ? lo : hi;
            }
        }
    }

    // ---- records and tile boxes: position p = s * nthr + tid is held by this thread, so a wave holds whole tiles
    const T inf = Limits<T>::inf();
    const uint32_t imask = (1u << ib) - 1u;
#pragma unroll
    for (int s = 0; s < KPT; ++s) {
        const int p = s * nthr + tid;
        const int tile = p >> 6;
        if (tile >= ntiles) continue;   // (wave-uniform)
        PointRec<T> r;
        const bool v = p < N;
        if (v) {
            const int i = (int)(key[s] & imask);
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


// ---------------------------------------------------------------------------------
// Origin of the reference's uniform grid, per cloud: vmin = min(1e6f, min over the points) per axis, evaluated
// in T with std::min's "keep unless smaller" rule, so a NaN coordinate is ignored (.cpp:163-177).  Needed only by
// stencils with an even dilated extent (Stencil::window); launched with those searches only.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cloud_min_kernel(const T *__restrict__ points, int N, T *__restrict__ cmin)
{
    __shared__ T red[3][4];
    const T *cloud = points + (size_t)blockIdx.x * N * 3;
    T m[3] = {(T)1e6f, (T)1e6f, (T)1e6f};
    for (int i = threadIdx.x; i < N; i += blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const T v = cloud[(size_t)i * 3 + a];
            m[a] = v < m[a] ? v : m[a];
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        m[a] = wave_min(m[a]);
        if ((threadIdx.x & 63) == 0) red[a][threadIdx.x >> 6] = m[a];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        T r = red[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) r = red[threadIdx.x][w] < r ? red[threadIdx.x][w] : r;
        cmin[(size_t)blockIdx.x * 3 + threadIdx.x] = r;
    }
}

// ---------------------------------------------------------------------------------
// search: the geometry of the op, done ONCE per (points, filter extents, stride, voxel).
// One workgroup (4 waves) per query tile, candidate tiles in groups of at most `gtiles`:
//   P1  pre-filter: waves split the candidate tiles; 64-bit hit masks + per-tile totals -> LDS
//   P2  one reservation of `sum of totals` pair slots for the whole query tile (one global
//       atomic per workgroup and group); inside it the pairs are laid out CENTRE-MAJOR (CSR):
//       centre q owns slots [start_q, start_q + n_q), published in qsegs -> consumers run
//       lane = centre with register accumulators and need no floating-point atomics
//   P3  waves turn their masks into a dense stream of (centre, candidate) pairs, 64 at a time
//       (lane = pair, all lanes busy): exact membership + forward tap with the reference's
//       arithmetic (.cpp:277-290), population update (LDS ds_add), backward tap (.cpp:662-677),
//       one PairEntry per pre-filter hit (false positives are stored as kNoTap entries)
//   end populations -> count[(b*N + orig)*F + f]   (Grid::neighbor_count, .cpp:306-379)
// pairs == nullptr: populations only (the public neighbour-count entry point).
// If the pair buffer is full the segment is marked kSegOverflow; consumers then fall back to
// searching that query tile themselves (slow, correct).
// LDS: tapmap | cnt [F][65] | centres [64] | masks [gtiles][64] u64 | tot [gtiles] | misc |
//      per wave: SoA slot [192] f32 + pair stream [128] u32
// ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load_centres(CentreRec<T> *centres, const Query<T> &q)
{
    if ((threadIdx.x >> 6) == 0) {
        CentreRec<T> r;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            r.p[a] = q.p[a];
            r.lo[a] = q.lo[a];
            r.hi[a] = q.hi[a];
        }
        r.orig = q.orig;
        centres[threadIdx.x & 63] = r;
    }
}

// backward tap of centre p inside the box of candidate v; kNoTap for a hole (.cpp:662-677)
template <typename T>
__device__ __forceinline__ uint32_t backward_tap(const T *p, const PointRec<T> &v, const Stencil<T> &st,
                                                 const int16_t *tapmap)
{
    const T lx = (T)((double)v.x - st.half[0]);
    const T ly = (T)((double)v.y - st.half[1]);
    const T lz = (T)((double)v.z - st.half[2]);
    const int tx = axis_tap(p[0], lx, st.voxel, st.full[0], tapmap);
    const int ty = axis_tap(p[1], ly, st.voxel, st.full[1], tapmap + st.maxfull);
    const int tz = axis_tap(p[2], lz, st.voxel, st.full[2], tapmap + 2 * st.maxfull);
    if ((tx | ty | tz) < 0) return kNoTap;                              // .cpp:672
    return (uint32_t)((tz * st.ext[1] + ty) * st.ext[0] + tx);          // .cpp:677
}

template <typename T, bool WIN>   // WIN: replicate the reference grid's candidate window (even dilated extents only)
__device__ __forceinline__ void search_tile(const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes,
                                            const Stencil<T> &st, int N, int ntiles, int gtiles, int ngroups,
                                            const BlockMap &bm, int32_t *__restrict__ count,
                                            PairEntry *__restrict__ pairs, const CacheCtl &cc,
                                            uint2 *__restrict__ segs, uint2 *__restrict__ qsegs,
                                            const T *__restrict__ cmin, int32_t *__restrict__ tcount,
                                            uint32_t *__restrict__ qbm,   // [tile][64] set of backward taps of every centre's list: taps 0 .. 31
                                            uint32_t *__restrict__ qbm_hi = nullptr)   // ... taps 32 .. 63 (filters of 33 .. 64 taps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t bmk[256];   // backward taps met by each centre of the tile (bit f' & 31 of word [f' >> 5][centre], up to 128 taps), see backward_sparse_kernel
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + off);
    off += align16((size_t)st.ntap * kCntStride * 4);
    CentreRec<T> *centres = reinterpret_cast<CentreRec<T> *>(smem + off);
    off += align16(sizeof(CentreRec<T>) * 64);
    int32_t *ccell = reinterpret_cast<int32_t *>(smem + off);   // [64][3] grid cells of the centres (window mode)
    off += 64 * 3 * 4;
    uint64_t *masks = reinterpret_cast<uint64_t *>(smem + off);
    off += align16((size_t)gtiles * 64 * 8);
    uint32_t *tot = reinterpret_cast<uint32_t *>(smem + off);
    off += align16((size_t)gtiles * 4);
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + off);   // [4] base, [5] ok
    off += 32;
    uint32_t *nqw = reinterpret_cast<uint32_t *>(smem + off);    // [wave][centre] pre-filter hits, then slot offsets
    off += kWavesPerBlock * 64 * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *soa = reinterpret_cast<float *>(smem + off) + wave * 192;
    off += align16((size_t)kWavesPerBlock * 192 * 4);
    uint32_t *stream = reinterpret_cast<uint32_t *>(smem + off) + wave * 256;   // [0..127] pair, [128..255] slot

    int b, qt;
    if (!block_to_cloud(bm, b, qt)) return;   // uniform for the workgroup
    if (pairs != nullptr && slot_valid(cc, b)) return;   // this cloud's lists are current (uniform)
    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    for (int e = threadIdx.x; e < st.ntap * kCntStride; e += blockDim.x) cnt[e] = 0;
    bmk[threadIdx.x] = 0;   // (256 threads)
    const bool want_bm = qbm != nullptr && (st.ntap <= 32 || (st.ntap <= 128 && qbm_hi != nullptr));
    const uint32_t cap = cc.pairs_per_cloud;
    const uint32_t region = (uint32_t)b * cap;   // first pair slot of this cloud's region
    const PointRec<T> *cloud_pts = pts + (size_t)b * ntiles * kTile;
    const T *cloud_box = boxes + (size_t)b * ntiles * 6;
    Query<T> q;
    make_query(q, cloud_pts[(size_t)qt * kTile + lane], st);
    const bool qvalid = q.orig >= 0;
    load_centres(centres, q);
    T vmin[3] = {(T)0, (T)0, (T)0};
    if constexpr (WIN) {
        Window<T> w;
        make_window(w, cloud_pts[(size_t)qt * kTile + lane], st, cmin + (size_t)b * 3);
#pragma unroll
        for (int a = 0; a < 3; ++a) vmin[a] = w.vmin[a];
        if (wave == 0) {
            ccell[lane * 3 + 0] = w.cell[0];
            ccell[lane * 3 + 1] = w.cell[1];
            ccell[lane * 3 + 2] = w.cell[2];
        }
    }
    __syncthreads();

    for (int g = 0; g < ngroups; ++g) {
        const int ct0 = g * gtiles;
        const int ct1 = min(ntiles, ct0 + gtiles);
        // ---- P1: pre-filter
        uint32_t mine = 0;   // this wave's pre-filter hits of centre `lane`
        for (int base = ct0; base < ct1; base += 64) {
            const uint64_t live = overlapping_tiles(cloud_box, ct1, base, q);
            // this wave's share: candidate tiles i == wave (mod 4) of the group
            for (int i = wave + lane * kWavesPerBlock; i < 64 && base + i < ct1; i += 64 * kWavesPerBlock)
                tot[base + i - ct0] = 0;
            uint64_t todo = live & (0x1111111111111111ull << wave);
            // (prefetches are unconditional: with nothing left to do they re-read the current tile)
            PointRec<T> rec = cloud_pts[(size_t)(base + (todo ? __builtin_ctzll(todo) : 0)) * kTile + lane];
            while (todo) {
                const int ct = base + __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint64_t quads = stage_tile(soa, rec, st, q);
                rec = cloud_pts[(size_t)(todo ? base + __builtin_ctzll(todo) : ct) * kTile + lane];   // prefetch
                __builtin_amdgcn_wave_barrier();
                uint32_t m0, m1;
                if (CONV3P_ABLATE & 32) { m0 = m1 = 0; } else scan_tile(soa, q, st, quads, m0, m1);
                __builtin_amdgcn_wave_barrier();
                if (!qvalid) m0 = m1 = 0;
                masks[(size_t)(ct - ct0) * 64 + lane] = ((uint64_t)m1 << 32) | m0;
                mine += __popc(m0) + __popc(m1);
                if (__any((m0 | m1) != 0) && lane == 0) tot[ct - ct0] = 1;
            }
        }
        nqw[wave * 64 + lane] = mine;
        __syncthreads();
        // ---- P2: one reservation for the whole query tile, centre-major slots inside it
        if (wave == 0) {
            const uint32_t n0 = nqw[lane], n1 = nqw[64 + lane], n2 = nqw[128 + lane], n3 = nqw[192 + lane];
            const uint32_t nq = n0 + n1 + n2 + n3;
            int L;
            const uint32_t offq = (uint32_t)wave_excl_scan((int)nq, L);
            uint32_t base = 0, ok = 0;
            if (pairs != nullptr) {
                if (lane == 0) {
                    // reserve L slots of the cloud's region.  A reservation that does not fit is taken back at once,
                    // so the 32-bit cursor never holds more than cap (<= 2^31, see carve()) plus the failed requests
                    // in flight (<= resident workgroups x 2^19 slots < 2^31): it cannot wrap back into the valid
                    // range however many pre-filter hits a degenerate cloud produces
                    base = L ? atomicAdd(&cc.cursor[b], (uint32_t)L) : 0u;
                    ok = (base <= cap && (uint32_t)L <= cap - base) ? 1u : 0u;
                    if (!ok) atomicSub(&cc.cursor[b], (uint32_t)L);
                    base += region;
                    segs[((size_t)b * ntiles + qt) * ngroups + g] =
                        ok ? make_uint2(base, (uint32_t)L) : make_uint2(0u, kSegOverflow);
                    misc[4] = base;
                    misc[5] = ok;
                }
                base = __shfl(base, 0);
                ok = __shfl(ok, 0);
                qsegs[(((size_t)b * ntiles + qt) * ngroups + g) * 64 + lane] =
                    ok ? make_uint2(base + offq, nq) : make_uint2(0u, kSegOverflow);
            } else if (lane == 0) {
                misc[4] = 0;
                misc[5] = 0;
            }
            // slot offsets (relative to base) of each wave's share of centre `lane`
            nqw[lane] = offq;
            nqw[64 + lane] = offq + n0;
            nqw[128 + lane] = offq + n0 + n1;
            nqw[192 + lane] = offq + n0 + n1 + n2;
        }
        __syncthreads();
        // ---- P3: dense exact stage
        {
            const uint32_t gbase = misc[4];
            const bool emit = misc[5] != 0;
            uint32_t myslot = nqw[wave * 64 + lane];   // next slot of centre `lane` for this wave
            int have = 0;
            auto drain = [&](int n) {
                // lanes 0..n-1 each resolve one (centre, candidate) pair
                if (!(CONV3P_ABLATE & 64) && lane < n) {
                    const uint32_t e = stream[lane];
                    const uint32_t ql = (e >> 6) & 63u, c = e & 63u, ct = e >> 12;
                    const CentreRec<T> cr = centres[ql];
                    const PointRec<T> v = cloud_pts[(size_t)ct * kTile + c];
                    bool out = (v.x < cr.lo[0]) | (v.x > cr.hi[0]) | (v.y < cr.lo[1]) | (v.y > cr.hi[1]) |
                               (v.z < cr.lo[2]) | (v.z > cr.hi[2]);                             // .cpp:277
                    if constexpr (WIN) { if (!out) out = outside_window(v.x, v.y, v.z, vmin, ccell + ql * 3, st); }   // .cpp:260-266
                    uint32_t fwd = kNoTap, bwd = kNoTap;
                    if (!out) {
                        const int tx = axis_tap(v.x, cr.lo[0], st.voxel, st.full[0], tapmap);
                        const int ty = axis_tap(v.y, cr.lo[1], st.voxel, st.full[1], tapmap + st.maxfull);
                        const int tz = axis_tap(v.z, cr.lo[2], st.voxel, st.full[2], tapmap + 2 * st.maxfull);
                        if ((tx | ty | tz) >= 0) {                                              // .cpp:285
                            fwd = (uint32_t)((tz * st.ext[1] + ty) * st.ext[0] + tx);           // .cpp:290
                            atomicAdd(&cnt[fwd * kCntStride + ql], 1u);
                            if (emit || want_bm) bwd = backward_tap(cr.p, v, st, tapmap);
                            if (want_bm && bwd != kNoTap) atomicOr(&bmk[ql + ((bwd >> 5) << 6)], 1u << (bwd & 31u));
                        }
                    }
                    if (emit) {
                        PairEntry pe;
                        pe.cand = (uint32_t)v.idx;
                        pe.code = pair_code(fwd, bwd, ql);
                        pairs[(size_t)gbase + stream[128 + lane]] = pe;
                    }
                }
            };
            for (int ctl = wave; ctl < ct1 - ct0; ctl += kWavesPerBlock) {
                if (tot[ctl] == 0) continue;
                uint64_t m = masks[(size_t)ctl * 64 + lane];
                const uint32_t ctbits = (uint32_t)(ct0 + ctl) << 12;
                while (true) {
                    const uint64_t act = __ballot(m != 0);
                    if (act == 0) break;
                    if (m != 0) {
                        const int P = __builtin_ctzll(m);
                        m &= m - 1;
                        const uint32_t c = (uint32_t)(P < 32 ? 31 - P : 95 - P);
                        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0));
                        stream[have + rank] = ctbits | ((uint32_t)lane << 6) | c;
                        stream[128 + have + rank] = myslot++;
                    }
                    have += __popcll(act);
                    __builtin_amdgcn_wave_barrier();
                    if (have >= 64) {
                        drain(64);
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t carry = stream[64 + lane], carry2 = stream[192 + lane];
                        __builtin_amdgcn_wave_barrier();
                        stream[lane] = carry;
                        stream[128 + lane] = carry2;
                        have -= 64;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if (have > 0) drain(have);
        }
        __syncthreads();
    }

    // populations -> global, by original index (lanes = taps, waves take every 4th centre)
    if (count != nullptr)
        for (int qq = wave; qq < kTile; qq += kWavesPerBlock) {
            const int orig = centres[qq].orig;
            if (orig < 0) continue;
            int32_t *row = count + ((size_t)b * N + orig) * st.ntap;
            for (int f = lane; f < st.ntap; f += 64) row[f] = (int32_t)cnt[f * kCntStride + qq];
        }
    // the same populations tile-major, [tile][tap][centre lane]: what the forward kernel of this tile turns into its
    // table of reciprocals with one coalesced read (no dependence on the centres' original indices)
    if (tcount != nullptr) {
        int32_t *tc = tcount + ((size_t)b * ntiles + qt) * st.ntap * kTile;
        for (int e = threadIdx.x; e < st.ntap * kTile; e += blockDim.x) tc[e] = (int32_t)cnt[(e >> 6) * kCntStride + (e & 63)];
    }
    if (want_bm && threadIdx.x < 64) {
        qbm[((size_t)b * ntiles + qt) * 64 + threadIdx.x] = bmk[threadIdx.x];
        if (st.ntap > 32) qbm_hi[((size_t)b * ntiles + qt) * 64 + threadIdx.x] = bmk[64 + threadIdx.x];
        if (st.ntap > 64) {   // planes 2 and 3 (65 .. 128 taps) follow plane 1 at the planes' common stride
            const size_t pstride = (size_t)(qbm_hi - qbm);
            qbm_hi[pstride + ((size_t)b * ntiles + qt) * 64 + threadIdx.x] = bmk[128 + threadIdx.x];
            qbm_hi[2 * pstride + ((size_t)b * ntiles + qt) * 64 + threadIdx.x] = bmk[192 + threadIdx.x];
        }
    }
    // commit: the last query tile of the cloud to finish marks the slot's lists as built from the current
    // content.  Every workgroup of the cloud passed the validity check before the ticket can reach its final
    // value, and the marks are only read by later launches, so no fence is needed.
    if (pairs != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t t = atomicAdd(&cc.ticket[b], 1u);
            if (t + 1u == (uint32_t)bm.blocks_per_cloud) {
                cc.built_version[b] = cc.version[b];
                cc.built_tag[b] = cc.tag;
                cc.ticket[b] = 0;
            }
        }
    }
}

template <typename T, bool WIN>
__global__ __launch_bounds__(256) void search_kernel(const PointRec<T> *__restrict__ pts,
                                                     const T *__restrict__ boxes, Stencil<T> st, int N,
                                                     int ntiles, int gtiles, int ngroups, BlockMap bm,
                                                     int32_t *__restrict__ count,
                                                     PairEntry *__restrict__ pairs, CacheCtl cc,
                                                     uint2 *__restrict__ segs, uint2 *__restrict__ qsegs,
                                                     const T *__restrict__ cmin, int32_t *__restrict__ tcount,
                                                     uint32_t *__restrict__ qbm, uint32_t *__restrict__ qbm_hi)
{
    search_tile<T, WIN>(pts, boxes, st, N, ntiles, gtiles, ngroups, bm, count, pairs, cc, segs, qsegs, cmin, tcount, qbm, qbm_hi);
}

// Several stencils over the same sorted points in ONE launch (blockIdx.y = stencil): the models' layers share
// `points` and differ only in stride, and one search launch is a single round of workgroups whose duration is
// set by its slowest tile; batched, the light tiles of one stencil fill in behind the heavy tiles of another.
constexpr int kMaxJobs = 8;
template <typename T> struct SearchJob {
    Stencil<T> st;
    CacheCtl cc;
    int32_t *count, *tcount;
    PairEntry *pairs;
    uint2 *segs, *qsegs;
    uint32_t *qbm, *qbm_hi;
};
template <typename T> struct SearchJobs {
    SearchJob<T> job[kMaxJobs];
};
template <typename T, bool WIN>
__global__ __launch_bounds__(256) void search_multi_kernel(const PointRec<T> *__restrict__ pts,
                                                           const T *__restrict__ boxes, int N, int ntiles,
                                                           int gtiles, int ngroups, BlockMap bm,
                                                           SearchJobs<T> jobs, const T *__restrict__ cmin)
{
    const SearchJob<T> &j = jobs.job[blockIdx.y];
    search_tile<T, WIN>(pts, boxes, j.st, N, ntiles, gtiles, ngroups, bm, j.count, j.pairs, j.cc, j.segs, j.qsegs, cmin, j.tcount,
                        j.qbm, j.qbm_hi);
}

}  // namespace conv3p
#include "conv3p_search_fused.hpp"
namespace conv3p {

// ---------------------------------------------------------------------------------
// tile_sched_kernel: launch order of the query tiles for the kernels that walk the pair lists.  A tile's cost follows
// the length of its lists, and dense regions of a cloud give tiles of 2-4x the mean (rooms, stride 1: max 18 349 pairs
// against a mean of 5 058); in Hilbert order such tiles are neighbours, land on the same few CUs and set the kernel's
// duration.  The tiles of every XCD (clouds stay on their XCD, see BlockMap) are therefore issued LONGEST FIRST:
//   sched[xcd * cap + i] = i-th tile (b * ntiles + qt) of that XCD, 0xFFFFFFFF past its last one; cap = rounds * ntiles
// so that the heavy tiles start at once, one per CU, and the light ones fill in behind them.  A pure function of the
// segment table: rebuilt after every search launch, the same for a cached and a fresh geometry.
// One workgroup per (XCD, stencil); bitonic sort of (records descending, tile id ascending) keys in LDS, up to 4096
// tiles per XCD (more: the BlockMap order is kept).  Tiles whose pair buffer overflowed search themselves: first.
constexpr int kSchedTiles = 4096;
struct SchedJob {
    const uint2 *segs;
    uint32_t *sched;
    // The slot's REGIME word: 1 when its pair lists are short on average -- at most `limit` pre-filter hits over the
    // whole batch (the clouds' allocators: `cursor`) --, else 0.  Which backward kernel is faster depends on it (crossover
    // between 27 and 41 neighbours per point); computed here, after the search, read by both
    // backward kernels, so that the choice needs neither the caller nor a host synchronisation.
    const uint32_t *cursor;
    uint32_t *regime;
    unsigned long long limit;
};
struct SchedJobs {
    SchedJob job[kMaxJobs];
};
__global__ __launch_bounds__(1024) void tile_sched_kernel(SchedJobs jobs, int B, int ntiles, int ngroups, int cap)
{
    __shared__ unsigned long long keys[kSchedTiles];
    const SchedJob &jb = jobs.job[blockIdx.y];
    const int xcd = blockIdx.x;
    if (xcd == 0 && threadIdx.x < 64 && jb.regime != nullptr) {
        // (a tile whose reservation did not fit took its request back from the allocator: such a slot's lists are long
        // whatever the sum says)
        unsigned long long sum = 0;
        bool over = false;
        for (int b = threadIdx.x; b < B; b += 64) sum += jb.cursor[b];
        for (int t = threadIdx.x; t < B * ntiles * ngroups; t += 64) over |= jb.segs[t].y == kSegOverflow;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const bool any_over = __any(over);
        if (threadIdx.x == 0) *jb.regime = (!any_over && sum <= jb.limit) ? 1u : 0u;
    }
    const int nclouds = B > xcd ? (B - xcd + 7) / 8 : 0;
    const int n = nclouds * ntiles;
    uint32_t *out = jb.sched + (size_t)xcd * cap;
    if (n > kSchedTiles) {
        for (int i = threadIdx.x; i < cap; i += blockDim.x)
            out[i] = i < n ? (uint32_t)((xcd + 8 * (i / ntiles)) * ntiles + i % ntiles) : 0xFFFFFFFFu;
        return;
    }
    int npad = 1;
    while (npad < n) npad <<= 1;
    for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        unsigned long long k = ~0ull;                        // padding sorts last
        if (i < n) {
            const uint32_t tile = (uint32_t)((xcd + 8 * (i / ntiles)) * ntiles + i % ntiles);
            uint32_t recs = 0;
            for (int g = 0; g < ngroups; ++g) {
                const uint32_t y = jb.segs[(size_t)tile * ngroups + g].y;
                recs = y == kSegOverflow || recs + y < recs ? 0xFFFFFFFEu : recs + y;
            }
            k = ((unsigned long long)(0xFFFFFFFFu - recs) << 32) | tile;
        }
        keys[i] = k;
    }
    for (int i = n + (int)threadIdx.x; i < cap; i += blockDim.x) out[i] = 0xFFFFFFFFu;
    __syncthreads();
    if (n <= 1024) {
        // rank sort: the keys are distinct (tile id in the low half), a key's rank is the number of smaller keys --
        // n broadcast reads per thread and no further barrier (the models' sizes: 128 .. 512 tiles per XCD)
        if ((int)threadIdx.x < n) {
            const unsigned long long mine = keys[threadIdx.x];
            int rank = 0;
            for (int i = 0; i < n; ++i) rank += keys[i] < mine ? 1 : 0;
            out[rank] = (uint32_t)mine;
        }
        return;
    }
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < npad; t += blockDim.x) {
                const int q = t ^ j;
                if (q > t) {
                    const unsigned long long a = keys[t], b = keys[q];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { keys[t] = b; keys[q] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = (uint32_t)keys[i];
}
// (cloud, query tile) of a workgroup by the schedule (sched == nullptr: by the BlockMap)
__device__ __forceinline__ bool block_to_tile(const BlockMap &m, const uint32_t *__restrict__ sched, int ntiles, int &cloud, int &qt)
{
    if (sched == nullptr) return block_to_cloud(m, cloud, qt);
    const uint32_t t = sched[(size_t)(blockIdx.x & 7) * ((size_t)m.rounds * ntiles) + (blockIdx.x >> 3)];
    if (t == 0xFFFFFFFFu) return false;
    cloud = (int)(t / (uint32_t)ntiles);
    qt = (int)(t - (uint32_t)cloud * (uint32_t)ntiles);
    return true;
}

// LDS layout of the forward kernel's filter copy (register path).  fp32: a tap's block is [Cin][Cout rounded up to
// even], so that a lane reads two consecutive output channels' weights with ONE aligned 8-byte access (ds_read_b64:
// 256 B / clk against 128 for 4-byte reads -- the walk's 81 weight reads per record were 80 % of the LDS's cycles and
// the LDS was what bounded the loop, profiles/r05_cfg2_pmc.txt) feeding one v_pk_fma_f32; the tap stride is a multiple
// of 2 words whose half is odd, so lanes on different taps spread over all 32 bank pairs.  fp64: [Cin][Cout], odd stride.
template <typename T> __host__ __device__ constexpr int fwd_coutp(int cout) { return sizeof(T) == 4 ? ((cout + 1) & ~1) : cout; }
template <typename T> __host__ __device__ constexpr int fwd_wstr(int cin, int cout)
{
    if (sizeof(T) != 4) return (cin * cout) | 1;
    const int w = cin * fwd_coutp<T>(cout);
    return (w / 2) % 2 == 0 ? w + 2 : w;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef CONV3P_BWD_BLOCKED
#define CONV3P_BWD_BLOCKED 1   // developer A/B: 0 = the lanes of a centre take its records interleaved (1: one run each)
#endif
// one aligned 8-byte LDS read that hipcc will not pair with its neighbour into ds_read2_b64 (which runs at the 4-byte
// reads' 128 B / clk, MI355X_MICROARCH.md, LDS table): a volatile access in the LDS address space
typedef __attribute__((address_space(3))) const volatile f32x2 lds_cv_f32x2;
__device__ __forceinline__ f32x2 lds_read_b64(const float *p) { return *(lds_cv_f32x2 *)p; }

// ---------------------------------------------------------------------------------
// forward accumulate: out[i,c] = sum over pairs of W[f,k,c] * x[j,k] / count[i,f]  (.cpp:480-494)
// One workgroup = one query tile; threads stride over the tile's pair segment (lane = pair).
// Small path (CIN/COUT compile-time): each wave owns 16 centres; its 64 lanes are dealt to them in proportion to their
// lists (share_lanes: consecutive lanes per centre, lane r of a centre's n lanes walks records r, r + n, ...); filter in
// LDS, output row in registers, the partial rows of a centre's lanes summed through LDS in ascending lane order
// (bitwise reproducible).  Generic path: thread = pair,
// global atomics into the zeroed output.
// A segment marked kSegOverflow makes the workgroup search its query tile itself.
// ---------------------------------------------------------------------------------
// What a tile's pass does at the hand-off points between the layers of a fused stack launch (conv3p_stack_fused.hpp):
// nothing, when the layer is a launch of its own.
struct NoSync {
    static constexpr bool kActive = false;
    __device__ __forceinline__ void wait() const {}     // before the first load of another tile's activations
    __device__ __forceinline__ void arrive() const {}   // after the tile's own rows are stored
};

// The pass of ONE query tile (b, qt); the whole workgroup calls it.  out2 (or nullptr): a second, dense copy of the tile's
// output rows [B][N][ld_out2] (the fused stack's hand-off buffer, see conv3p_stack_fused.hpp).
template <typename T, int CIN, int COUT, class Sync>
__device__ __forceinline__ void forward_tile(
    const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes, const int32_t *__restrict__ count,
    const PairEntry *__restrict__ pairs, const uint2 *__restrict__ segs, const uint2 *__restrict__ qsegs,
    const T *__restrict__ input, const T *__restrict__ filter, const Stencil<T> &st, int N, int ntiles, int ngroups,
    int cin_rt, int cout_rt, T *__restrict__ output,
    int act,   // act != 0 (small path only): store selu(out), the models' layer (pointcnn2_acsd.py:48-49)
    const T *__restrict__ cmin,   // per-cloud grid origin (window-mode stencils, overflow path only)
    const int32_t *__restrict__ tcount,   // populations tile-major [tile][tap][centre lane] (search_tile)
    RowLd ld, int b, int qt, T *__restrict__ out2, int ld_out2, const Sync &sync)
{
    constexpr bool kSmall = CIN > 0;
    // (fused stack launch: the same code runs once per layer in one kernel; values derived from the thread index would be
    // computed once and kept in registers across all layers -- 3 spilled registers at the 128-register cap -- unless the
    // index is opaque to the optimiser in each pass)
    uint32_t tid_ = threadIdx.x;
    if constexpr (Sync::kActive) asm volatile("" : "+v"(tid_));
    const uint32_t tid = tid_;
    const int cin = kSmall ? CIN : cin_rt;
    const int cout = kSmall ? COUT : cout_rt;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    T *w_lds = reinterpret_cast<T *>(smem + off);
    const size_t nw = (size_t)st.ntap * cin * cout;
    // LDS stride of one tap's [Cin][Cout] block: odd, so that lanes working on different taps spread over all the
    // banks (36 x 13 = 468 = 20 mod 32 would put every tap on one of 8 banks: the 36 -> 13 layer ran 2x slower)
    constexpr int COUTP = kSmall ? fwd_coutp<T>(COUT) : 1;              // row of one input channel (fp32: padded to even)
    constexpr int WSTR = kSmall ? fwd_wstr<T>(CIN, COUT) : 1;
    constexpr bool kPk = kSmall && sizeof(T) == 4;                     // fp32 register path: 8-byte weight reads, packed FMAs
    if (kSmall) off += align16((size_t)st.ntap * WSTR * sizeof(T));
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + off);   // populations of the tile's centres [tap][65] ...
    T *rcpt = reinterpret_cast<T *>(smem + off);                // ... or (dense small path) their reciprocals 1/(T)count
    off += align16((size_t)st.ntap * kCntStride * sizeof(T));
    int32_t *qorig = reinterpret_cast<int32_t *>(smem + off);
    off += 256;
    const int wave = tid >> 6, lane = tid & 63;
    float *soa = reinterpret_cast<float *>(smem + off) + wave * 192;
    off += align16((size_t)kWavesPerBlock * 192 * 4);
    T *red = reinterpret_cast<T *>(smem + off);   // [4][COUT][64], overflow path only
    int cq = wave * 16 + (lane >> 2);   // dense path: this lane's centre (re-dealt below, share_lanes)
    uint32_t sub = lane & 3u, nsub = 4u;   //   ... its index among the centre's lanes, and their number
    uint32_t *share = reinterpret_cast<uint32_t *>(soa);   // the wave's lane-sharing scratch (soa is the overflow path's)

    DEV_FWD_DECL()
    FDBG()
    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    // the tile's populations (tile-major copy): issued before the filter so that both are in flight together
    constexpr int kTcPer = 8;   // 27 taps x 64 centres = 1728 values: 7 per thread
    int32_t tcv[kTcPer];
    const bool tc_fits = kSmall && st.ntap * kTile <= kTcPer * 256;
    if (tc_fits) {
        const int32_t *tc = tcount + ((size_t)b * ntiles + qt) * st.ntap * kTile;
#pragma unroll
        for (int u = 0; u < kTcPer; ++u) {
            // unconditional load from a clamped index (a predicated load gets its own exec-mask branch and a full
            // vmcnt(0) wait from hipcc: eight memory latencies in series instead of one)
            const int e = (int)tid + 256 * u;
            tcv[u] = tc[e < st.ntap * kTile ? e : 0];
        }
    }
    if (kSmall) {
        // filter -> LDS, 8 independent loads per thread in flight (one memory latency per batch, not per element)
        for (uint32_t e0 = tid; e0 < (uint32_t)nw; e0 += 8 * 256) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = filter[e0 + u * 256 < (uint32_t)nw ? e0 + u * 256 : 0u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * 256 < (uint32_t)nw) {
                    const uint32_t e = e0 + u * 256, f = e / (CIN * COUT), kc = e - f * (CIN * COUT);
                    const uint32_t k = kc / COUT, c = kc - k * COUT;
                    w_lds[f * WSTR + k * COUTP + c] = v[u];
                    if (COUTP != COUT && c == COUT - 1) w_lds[f * WSTR + k * COUTP + COUT] = (T)0;   // the padding column
                }
        }
    }
    const PointRec<T> *cloud_pts = pts + (size_t)b * ntiles * kTile;
    const PointRec<T> me = cloud_pts[(size_t)qt * kTile + lane];
    if (wave == 0) qorig[lane] = me.idx;
    const size_t tile_id = (size_t)b * ntiles + qt;
    bool overflow = false;
    for (int g = 0; g < ngroups; ++g) overflow |= segs[tile_id * ngroups + g].y == kSegOverflow;
    // Dense small path: the pipeline over a centre's pair list (see the loop below) is STARTED here, before the barriers
    // and the table of reciprocals: the list's segment, its first records and the first neighbour row are three
    // dependent memory latencies that need nothing from LDS, so they run under the prologue instead of after it.
#ifndef CONV3P_FWD_SLOTS
#define CONV3P_FWD_SLOTS 3
#endif
    constexpr int NS = !kSmall ? 2 : sizeof(T) * CIN <= 48 ? CONV3P_FWD_SLOTS : sizeof(T) * CIN <= 144 ? 3 : 2;
    constexpr int CINR = kSmall ? CIN : 1;
    PairEntry rec[NS];
    T xs[NS][CINR];
    uint2 sg = make_uint2(0u, 0u);
    const PairEntry *pe = pairs;
    const T *in_cloud = input + (size_t)b * N * ld.in;
    // Lanes dealt to the wave's 16 centres in proportion to their lists (share_lanes, conv3p_device.hpp) when the cloud
    // was searched in one group; else four consecutive lanes per centre (a lane keeps its centre across the groups).
    const bool shared = kSmall && ngroups == 1;
    auto ld_rec = [&](uint32_t i) { return pe[i < sg.y ? i : 0u]; };
    auto ld_row = [&](int sl, uint32_t i) {
        const bool ok = i < sg.y && code_fwd(rec[sl].code) != kNoTap;
        RowLoader<T, CINR>::load(in_cloud + (size_t)(ok ? ((CONV3P_ABLATE & 1024) ? (rec[sl].cand & 63u) : rec[sl].cand) : 0u) * ld.in, xs[sl]);   // (1024: developer, every gather an L1 hit)
    };
    auto start_rows = [&]() {
#pragma unroll
        for (int sl = 0; sl < NS - 2; ++sl) ld_row(sl, sub + nsub * sl);
    };
    auto start_group = [&](int g, bool rows) {
        if (shared) {
            const LaneShare ls = share_lanes(qsegs[tile_id * 64 + wave * 16 + (lane & 15)], share);
            cq = wave * 16 + (int)ls.cl;
            sub = ls.r;
            nsub = ls.n;
            sg = ls.seg;
        } else {
            sg = qsegs[(tile_id * ngroups + g) * 64 + cq];
            if (lane < 16) share[64 + lane] = (uint32_t)(4 * lane) | (4u << 8);   // the centres' lane spans, as share_lanes leaves them
        }
        pe = pairs + sg.x;
#pragma unroll
        for (int sl = 0; sl < NS - 1; ++sl) rec[sl] = ld_rec(sub + nsub * sl);
        if (rows) start_rows();
    };
    // (fused stack launch: the neighbour rows are another tile's output of the previous layer -- everything above and the
    // records are requested first, the rows once the cloud's tiles have all arrived)
    if (kSmall && !overflow) start_group(0, !Sync::kActive);
    // (fused stack launch: the table of reciprocals needs nothing of the previous layer either -- written before the wait, the
    // one workgroup barrier after the wait publishes it: no division and no second barrier between the wait and the loop)
    const bool early_table = Sync::kActive && kSmall && !overflow && tc_fits;   // (uniform)
    if (early_table) {
#pragma unroll
        for (int u = 0; u < kTcPer; ++u) {
            const int e = (int)tid + 256 * u;
            if (e < st.ntap * kTile) rcpt[(e >> 6) * kCntStride + (e & 63)] = (T)1 / (T)tcv[u];
        }
    }
    sync.wait();
    FDBG()
    __syncthreads();
    FDBG()
    if (Sync::kActive && kSmall && !overflow) start_rows();
    if (!early_table) {
        // own populations -> LDS [tap][centre]; the dense small path keeps 1 / (T)count instead (the IEEE quotient,
        // .cpp:483: one division per (centre, tap) here rather than one per pair)
        const bool as_rcp = kSmall && !overflow;
        if (as_rcp && tc_fits) {
#pragma unroll
            for (int u = 0; u < kTcPer; ++u) {
                const int e = (int)tid + 256 * u;
                if (e < st.ntap * kTile) rcpt[(e >> 6) * kCntStride + (e & 63)] = (T)1 / (T)tcv[u];
            }
        } else {
            for (int qq = wave; qq < kTile; qq += kWavesPerBlock) {
                const int orig = qorig[qq];
                if (orig < 0) continue;
                const int32_t *row = count + ((size_t)b * N + orig) * st.ntap;
                for (int f = lane; f < st.ntap; f += 64) {
                    const int32_t cv = row[f];
                    if (as_rcp) rcpt[f * kCntStride + qq] = (T)1 / (T)cv;
                    else cnt[f * kCntStride + qq] = (uint32_t)cv;
                }
            }
        }
        __syncthreads();
    }

    FDBG()
    T *out_cloud = output + (size_t)b * N * ld.out;
    T acc[kSmall ? COUTP : 1];
    if (kSmall) {
#pragma unroll
        for (int c = 0; c < COUTP; ++c) acc[c] = (T)0;
    }
    // centre = lane `ql` (always this lane on the small path)
    auto accumulate = [&](uint32_t cand, uint32_t f, uint32_t ql, T rcp) {
        // rcp = 1 / (T)fsize (.cpp:483): from the pair record (fp32) or from the populations in LDS
        const T *xr = in_cloud + (size_t)cand * ld.in;
        if constexpr (kSmall) {
            T xs[CIN];
            RowLoader<T, CIN>::load(xr, xs);
#pragma unroll
            for (int k = 0; k < CIN; ++k) xs[k] *= rcp;                  // x / count, .cpp:492
            const T *wf = w_lds + (size_t)f * WSTR;
#pragma unroll
            for (int k = 0; k < CIN; ++k)
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] = fma_t(wf[k * COUTP + c], xs[k], acc[c]);
        } else {
            const T *wf = filter + (size_t)f * cin * cout;
            T *orow = out_cloud + (size_t)qorig[ql] * ld.out;
            for (int c = 0; c < cout; ++c) {
                T a = (T)0;
                for (int k = 0; k < cin; ++k) a = fma_t(wf[(size_t)k * cout + c], xr[k] * rcp, a);
                __hip_atomic_fetch_add(&orow[c], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };

    if (!overflow) {
        for (int g = 0; g < ngroups; ++g) {
            if constexpr (kSmall) {
                // wave w owns the centres 16w..16w+15; a centre's lanes (consecutive, share_lanes; four per centre when the
                // cloud was searched in several groups) read consecutive pair records per step
                // Software pipeline over NS NAMED slots (the loop is unrolled by NS so that a slot is a fixed
                // set of registers): the record of step k+2 and the neighbour row of step k+1 are in flight while
                // step k runs its 81 FMAs.  Rotating the slots by register copies would make every step wait for
                // the newest load (a copy reads its destination registers), i.e. no pipeline at all.  All loads are
                // unconditional from clamped addresses, so hipcc can count the ones in flight.
                // (NS = 2 for the widest rows, whose three copies would not fit the register file: the row is then
                // loaded in the step that uses it, as before)
                if (g > 0) start_group(g, true);      // (group 0 was started before the barriers)
                FDBG()
                FDBG()
                uint32_t i = sub;
                bool more = true;
                while (more) {
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        if (!__any(i < sg.y)) {
                            more = false;
                            break;
                        }
                        rec[(j + NS - 1) % NS] = ld_rec(i + nsub * (NS - 1));
                        ld_row((j + NS - 2) % NS, i + nsub * (NS - 2));
                        __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of this step's arithmetic
                        const uint32_t f = code_fwd(rec[j].code);
                        if (i < sg.y && f != kNoTap) {
                            const T rcp = rcpt[f * kCntStride + cq];
                            const T *wf = w_lds + (size_t)f * WSTR;
                            if constexpr (kPk) {
                                // two output channels per LDS access and per FMA instruction: the same products and
                                // sums, in the same order per channel, as the scalar form below
                                // (the weights of input channel k + 1 are requested before channel k's products:
                                // the volatile reads issue in source order)
                                f32x2 wq[2][COUTP / 2];
#pragma unroll
                                for (int c2 = 0; c2 < COUTP / 2; ++c2) wq[0][c2] = lds_read_b64(wf + 2 * c2);
#pragma unroll
                                for (int k = 0; k < CIN; ++k) {
                                    if (k + 1 < CIN) {
#pragma unroll
                                        for (int c2 = 0; c2 < COUTP / 2; ++c2) wq[(k + 1) & 1][c2] = lds_read_b64(wf + (k + 1) * COUTP + 2 * c2);
                                    }
                                    const float xk = xs[j][k] * rcp;              // x / count, .cpp:492
                                    const f32x2 xk2 = {xk, xk};
#pragma unroll
                                    for (int c2 = 0; c2 < COUTP / 2; ++c2) {
                                        f32x2 a = {acc[2 * c2], acc[2 * c2 + 1]};
                                        a = __builtin_elementwise_fma(wq[k & 1][c2], xk2, a);
                                        acc[2 * c2] = a.x;
                                        acc[2 * c2 + 1] = a.y;
                                    }
                                }
                            } else {
#pragma unroll
                            for (int k = 0; k < CIN; ++k) {
                                const T xk = xs[j][k] * rcp;                  // x / count, .cpp:492
#pragma unroll
                                for (int c = 0; c < COUT; ++c) acc[c] = fma_t(wf[k * COUTP + c], xk, acc[c]);
                            }
                            }
                        }
                        i += nsub;
                        DEV_FWD_STEP()
                    }
                }
                FDBG()
            } else {
                const uint2 sg = segs[tile_id * ngroups + g];
                const PairEntry *pe = pairs + sg.x;
                for (uint32_t e = tid; e < sg.y; e += blockDim.x) {
                    const PairEntry en = pe[e];
                    const uint32_t f = code_fwd(en.code);
                    if (f != kNoTap) accumulate(en.cand, f, code_q(en.code), (T)1 / (T)cnt[f * kCntStride + code_q(en.code)]);
                }
            }
        }
    } else {
        // pair buffer was full for this tile: search it here (lane = centre)
        const T *cloud_box = boxes + (size_t)b * ntiles * 6;
        Query<T> q;
        make_query(q, me, st);
        Window<T> win{};
        if (cmin != nullptr) make_window(win, me, st, cmin + (size_t)b * 3);
        for_each_neighbor(cloud_pts, cloud_box, ntiles, q, st, tapmap, soa, wave, kWavesPerBlock,
                          [&](const PointRec<T> &v, int f) {
            accumulate((uint32_t)v.idx, (uint32_t)f, (uint32_t)lane, (T)1 / (T)cnt[f * kCntStride + lane]);
        }, win, cmin != nullptr);
    }

    if constexpr (kSmall) {
        if (!overflow) {
            // the lanes of a centre (consecutive, spans in share[64 + centre]) hold partial rows: staged in the wave's part
            // of `red` and summed in ascending lane order by thread (centre, channel), which also stores the value --
            // consecutive lanes write consecutive channels of a row
            T *rw = red + (size_t)wave * COUT * 64;
#pragma unroll
            for (int c = 0; c < COUT; ++c) rw[c * 64 + lane] = acc[c];
            __builtin_amdgcn_wave_barrier();
            for (int o = lane; o < 16 * COUT; o += 64) {
                const int ci = o / COUT, ch = o - ci * COUT;
                const uint32_t span = share[64 + ci];
                const T *src = rw + ch * 64 + (span & 255u);
                const int cnt_l = (int)(span >> 8);
                T v = (T)0;
                for (int jj = 0; jj < cnt_l; ++jj) v += src[jj];
                const int orig = qorig[wave * 16 + ci];
                if (orig >= 0) {
                    const T r = act ? selu_value(v) : v;
                    out_cloud[(size_t)orig * ld.out + ch] = r;
                    if (out2 != nullptr) out2[((size_t)b * N + orig) * ld_out2 + ch] = r;
                }
            }
            DEV_FWD_PRINT(CIN, COUT, wave, lane)
        } else {
            // overflow path ran lane = centre in every wave: fixed-order sum of the per-wave partial rows
#pragma unroll
            for (int c = 0; c < COUT; ++c) red[((size_t)wave * COUT + c) * 64 + lane] = acc[c];
            __syncthreads();
            for (int e = tid; e < COUT * 64; e += blockDim.x) {
                const int c = e >> 6;   // e & 63 == lane
                T sum = red[((size_t)0 * COUT + c) * 64 + lane];
#pragma unroll
                for (int w = 1; w < kWavesPerBlock; ++w) sum += red[((size_t)w * COUT + c) * 64 + lane];
                if (me.idx >= 0) {
                    const T r = act ? selu_value(sum) : sum;
                    out_cloud[(size_t)me.idx * ld.out + c] = r;
                    if (out2 != nullptr) out2[((size_t)b * N + me.idx) * ld_out2 + c] = r;
                }
            }
        }
    }
    sync.arrive();
}

template <typename T, int CIN, int COUT>
// Cin <= 12: 4 waves per SIMD = 4 workgroups per CU, so that the 128 workgroups an XCD gets for cfg2 are resident
// in one round (the unconstrained allocation is 132 VGPRs).  Wider inputs keep their row in registers and would
// spill under that cap (36 -> 13: 50 spilled VGPRs, 4.6x slower).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 && CIN > 0 && CIN <= 12 ? 4 : 1))) void forward_kernel(
    const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes, const int32_t *__restrict__ count,
    const PairEntry *__restrict__ pairs, const uint2 *__restrict__ segs, const uint2 *__restrict__ qsegs,
    const T *__restrict__ input, const T *__restrict__ filter, Stencil<T> st, int N, int ntiles, int ngroups,
    int cin_rt, int cout_rt, BlockMap bm, T *__restrict__ output, const uint8_t *__restrict__ only_flagged,
    int act, const T *__restrict__ cmin, const int32_t *__restrict__ tcount,
    RowLd ld, const uint32_t *__restrict__ sched)   // sched: launch order of the tiles (tile_sched_kernel) or nullptr
{
    int b, qt;
    if (!block_to_tile(bm, sched, ntiles, b, qt)) return;   // uniform
    if (only_flagged != nullptr && !only_flagged[(size_t)b * ntiles + qt]) return;   // deep path did this tile
    forward_tile<T, CIN, COUT>(pts, boxes, count, pairs, segs, qsegs, input, filter, st, N, ntiles, ngroups, cin_rt, cout_rt, output,
                               act, cmin, tcount, ld, b, qt, static_cast<T *>(nullptr), 0, NoSync{});
}

// ---------------------------------------------------------------------------------
// backward accumulate.  For every stored pair (centre j = q, neighbour ii = cand):
//   f' = bwd tap (tap of j inside ii's box, no inclusion re-test, .cpp:658-677; kNoTap = hole),
//   count = population of tap f' of ii, pair skipped when 0 (.cpp:678-679),
//   g[c] = dY[ii,c] / count,  dX[j,k] += g[c] W[f',k,c],  dW[f',k,c] += g[c] X[j,k].
// Small path (one workgroup = one query tile):
//   phase A  each wave owns 16 centres; its lanes are dealt to them in proportion to their lists (share_lanes), every
//            lane walks ONE run of its centre's list:  G[(f',c)][j] += dY[ii,c] * (1/count)   (LDS [row][65]; lanes of
//            a centre that meet on one tap take turns, lower lane first -> no atomics, reproducible)
//   phase B  thread = row (f',c):  dW[f',k,c] = sum_j G[row][j] * X[j,k]  -> this workgroup's
//            partial slot (X tile broadcast from LDS)
//   phase C  lane = centre j, waves split the rows:  dX[j,k] = sum_row G[row][j] * W[row][k],
//            per-wave partial rows summed through LDS in fixed order.
//   No floating-point atomics anywhere on this path: results are bitwise reproducible.
// Generic path: lane = pair, global atomics into zeroed dX and partial slot 0.
// LDS small: tapmap | G [F*COUT][65] | qorig | { Wt [F*COUT][CIN] | X tile [64][CIN] | SoA }  (reduce aliases {})
// ---------------------------------------------------------------------------------
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void backward_kernel(
    const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes, const int32_t *__restrict__ count,
    const PairEntry *__restrict__ pairs, const uint2 *__restrict__ segs, const uint2 *__restrict__ qsegs,
    const T *__restrict__ grad_out, const T *__restrict__ input, const T *__restrict__ filter, Stencil<T> st,
    int N, int ntiles, int ngroups, int cin_rt, int cout_rt, BlockMap bm, T *__restrict__ grad_input,
    T *__restrict__ partials, const uint8_t *__restrict__ only_flagged,
    int act, const T *__restrict__ addend,   // act & 1 (small path only): `input` is a SELU output; store
                                             // (dX + addend) * selu'(input), the gradient w.r.t. that SELU's argument
    int gen_slots,                           // generic path: number of grad_filter partial slots the workgroups
                                             // spread their atomics over (slot = workgroup % gen_slots)
    const T *__restrict__ cmin,              // per-cloud grid origin (window-mode stencils, overflow path only)
    RowLd ld, const uint32_t *__restrict__ sched,   // sched: launch order of the tiles (tile_sched_kernel) or nullptr
    const uint32_t *__restrict__ regime)            // non-null: run only if the slot's lists are LONG (*regime == 0); the
                                                    // populated-rows kernel was launched for the other case
{
    if (regime != nullptr && *regime != 0u) return;   // (uniform)
    constexpr bool kSmall = CIN > 0;
    const int cin = kSmall ? CIN : cin_rt;
    const int cout = kSmall ? COUT : cout_rt;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    const size_t nw = (size_t)st.ntap * cin * cout;
    const int nrows = st.ntap * cout;
    T *G = reinterpret_cast<T *>(smem + off);         // G[row][65]
    if (kSmall) off += align16((size_t)nrows * kCntStride * sizeof(T));
    int32_t *qorig = reinterpret_cast<int32_t *>(smem + off);
    off += 256;
    T *rinv = reinterpret_cast<T *>(smem + off);      // rinv[n] = 1 / (T)n for n < 256 (the IEEE quotient): one LDS read
    if (kSmall) off += align16(256 * sizeof(T));      //   per pair instead of a division; larger populations divide
    T *red = reinterpret_cast<T *>(smem + off);       // [4][CIN][64]: ALIASES wt | xt | soa (used after them)
    T *wt = reinterpret_cast<T *>(smem + off);        // Wt[row][k], row = f*COUT + c
    if (kSmall) off += align16(nw * sizeof(T));
    T *xt = reinterpret_cast<T *>(smem + off);        // X tile [64][CIN]
    if (kSmall) off += align16((size_t)64 * cin * sizeof(T));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *soa = reinterpret_cast<float *>(smem + off) + wave * 192;
    off += align16((size_t)kWavesPerBlock * 192 * 4);
    int cq = wave * 16 + (lane >> 2);      // dense phase A: this lane's centre (re-dealt there, share_lanes), ...
    uint32_t sub = lane & 3u, maxn = 4u;   //   its index among the centre's lanes, the largest lane count of a centre

    DEV_BWD_DECL()
    BDBG()
    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    if constexpr (kSmall) {
        // filter -> LDS transposed to [row = (f,c)][k]: coalesced global reads, 8 per thread in flight
        for (uint32_t e0 = threadIdx.x; e0 < (uint32_t)nw; e0 += 8 * 256) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = e0 + u * 256 < (uint32_t)nw ? filter[e0 + u * 256] : (T)0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t e = e0 + u * 256;   // e = (f*CIN + k)*COUT + c
                if (e < (uint32_t)nw) {
                    const uint32_t fk = e / COUT, c = e - fk * COUT;
                    const uint32_t f = fk / CIN, k = fk - f * CIN;
                    wt[(f * COUT + c) * CIN + k] = v[u];
                }
            }
        }
        rinv[threadIdx.x] = (T)1 / (T)(int)threadIdx.x;   // [0] = inf, never read (populations of 0 are skipped)
        // G = 0, 16 bytes per store (G is 16-byte aligned, its length is padded to 4 by the LDS carve-up)
        {
            float4 *G4 = reinterpret_cast<float4 *>(G);
            const int n4 = (int)((nrows * kCntStride * sizeof(T) + 15) / 16);
            for (int e = threadIdx.x; e < n4; e += blockDim.x) G4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    int b, qt;
    bool live = block_to_tile(bm, sched, ntiles, b, qt);   // uniform for the workgroup
    if (live && only_flagged != nullptr && !only_flagged[(size_t)b * ntiles + qt]) live = false;   // deep path did it
    const PointRec<T> *cloud_pts = pts + (size_t)(live ? b : 0) * ntiles * kTile;
    PointRec<T> me = cloud_pts[(size_t)(live ? qt : 0) * kTile + lane];
    if (!live) me.idx = -1;
    if (wave == 0) {
        qorig[lane] = me.idx;
        if (kSmall) {
            const T *xr = input + ((size_t)(live ? b : 0) * N + (me.idx < 0 ? 0 : me.idx)) * ld.in;
            T xv[kSmall ? CIN : 1];
            RowLoader<T, (kSmall ? CIN : 1)>::load(xr, xv);   // unconditional (padding lanes read row 0), then select
#pragma unroll
            for (int k = 0; k < (kSmall ? CIN : 1); ++k) xt[lane * cin + k] = me.idx >= 0 ? xv[k] : (T)0;
        }
    }
    __syncthreads();
    BDBG()

    if (live) {
        const int32_t *cnt_cloud = count + (size_t)b * N * st.ntap;
        const T *dy_cloud = grad_out + (size_t)b * N * ld.dy;
        const T *in_cloud = input + (size_t)b * N * ld.in;
        T *dx_cloud = grad_input + (size_t)b * N * ld.dx;
        // phase A.  Small path: lane = centre `ql` == lane; wave w owns the taps f' == w (mod 4), so
        // every G element has exactly one writer and plain LDS read-modify-write is race-free.
        auto accumulate = [&](uint32_t cand, uint32_t fb, uint32_t ql, T rcp) {
            // rcp = 1 / count of tap fb of the neighbour (.cpp:678); <= 0 asks for a lookup here
            if (!(rcp > (T)0)) {
                const int cn = cnt_cloud[(size_t)cand * st.ntap + fb];
                if (cn == 0) return;                                          // .cpp:679
                rcp = (T)1 / (T)cn;
            }
            const T *dyr = dy_cloud + (size_t)cand * ld.dy;
            if constexpr (kSmall) {
                T *grow = G + ((size_t)fb * COUT) * kCntStride + ql;
#pragma unroll
                for (int c = 0; c < COUT; ++c) {
                    if (CONV3P_ABLATE & 16) { asm volatile("" :: "v"(dyr[c] * rcp)); }
                    else grow[c * kCntStride] += dyr[c] * rcp;
                }
            } else {
                const int jo = qorig[ql];
                const T *wf = filter + (size_t)fb * cin * cout;
                T *dwf = partials + (size_t)(blockIdx.x % (unsigned)gen_slots) * nw + (size_t)fb * cin * cout;
                const T *xr = in_cloud + (size_t)jo * ld.in;
                T *dxr = dx_cloud + (size_t)jo * ld.dx;
                for (int k = 0; k < cin; ++k) {
                    const T xk = xr[k];
                    T a = (T)0;
                    for (int c = 0; c < cout; ++c) {
                        const T g = dyr[c] * rcp;
                        a = fma_t(g, wf[(size_t)k * cout + c], a);                            // .cpp:692
                        __hip_atomic_fetch_add(&dwf[(size_t)k * cout + c], g * xk, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);                     // .cpp:696
                    }
                    __hip_atomic_fetch_add(&dxr[k], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };

        const size_t tile_id = (size_t)b * ntiles + qt;
        bool overflow = false;
        for (int g = 0; g < ngroups; ++g) overflow |= segs[tile_id * ngroups + g].y == kSegOverflow;
        if (!overflow) {
            for (int g = 0; g < ngroups; ++g) {
                if constexpr (kSmall) {
                    if (CONV3P_ABLATE & 1) continue;
                    // wave w owns the centres 16w..16w+15; its lanes are dealt to them in proportion to their lists
                    // (share_lanes; four consecutive lanes per centre when the cloud was searched in several groups)
                    // and every lane walks its own run of its centre's list.  Lanes of a centre that hit the same tap
                    // in one step take turns, lower lane first (fixed order) -> race-free, reproducible.
                    LaneShare ls;
                    if (ngroups == 1) {
                        ls = share_lanes(qsegs[tile_id * 64 + wave * 16 + (lane & 15)], reinterpret_cast<uint32_t *>(soa));
                        cq = wave * 16 + (int)ls.cl;
                        sub = ls.r;
                        maxn = ls.maxn;
                    } else {
                        ls = share_lanes_uniform(qsegs[(tile_id * ngroups + g) * 64 + cq]);
                    }
                    const uint2 sg = ls.seg;
                    const LaneWalk wk = lane_walk(ls, CONV3P_BWD_BLOCKED != 0);
                    const PairEntry *pe = pairs + sg.x;
                    auto live_rec = [&](const PairEntry &r, uint32_t i) {
                        return i < wk.end && code_fwd(r.code) != kNoTap && code_bwd(r.code) != kNoTap;
                    };
                    auto cnt_of = [&](const PairEntry &r, bool live) {
                        if (CONV3P_ABLATE & 2048) return 1;   // developer: timing without the gather
                        // unconditional load from a clamped address + select: a predicated load sits in its own
                        // exec-mask branch, after which hipcc waits for ALL loads in flight (vmcnt(0)) at the top of
                        // every step and the software pipeline below is worth nothing
                        // (no select on the result either -- hipcc would sink the load into the select's branch; a
                        // dead slot reads element 0 and the caller masks with `live`)
                        return cnt_cloud[live ? (size_t)r.cand * st.ntap + code_bwd(r.code) : (size_t)0];
                    };
                    // With 78 KB of LDS only two workgroups fit a CU, so the latency of the gathers is hidden by depth,
                    // not by occupancy: kDepth NAMED slots, the loop unrolled by kDepth so that a slot is a fixed set of
                    // registers; the record of step k+kDepth-1 and the dY row + population of step k+kDepth-2 are in
                    // flight while step k runs.  (Rotating slots by register copies made every step wait for the
                    // newest load -- a copy reads its destination -- so that "pipeline" drained the queue every step.)
#ifndef CONV3P_BWD_DEPTH
#define CONV3P_BWD_DEPTH 4
#endif
                    constexpr int kDepth = CONV3P_BWD_DEPTH;   // measured: 8 slots (whole lists requested up front) is 3 % slower on cfg2
                    PairEntry rec[kDepth];
                    bool lv[kDepth];
                    int cn[kDepth];
                    T val[kDepth][COUT];
                    auto ld_rec = [&](uint32_t i) { return pe[i < wk.end ? i : 0u]; };
                    auto gather = [&](int sl, uint32_t i) {
                        lv[sl] = live_rec(rec[sl], i);
                        cn[sl] = cnt_of(rec[sl], lv[sl]);
                        RowLoader<T, COUT>::load(dy_cloud + (size_t)(lv[sl] ? rec[sl].cand : 0u) * ld.dy, val[sl]);
                    };
#pragma unroll
                    for (int sl = 0; sl < kDepth - 1; ++sl) rec[sl] = ld_rec(wk.i + wk.step * sl);
                    __builtin_amdgcn_sched_barrier(0);   // the records are the oldest loads in flight on entry
#pragma unroll
                    for (int sl = 0; sl < kDepth - 2; ++sl) gather(sl, wk.i + wk.step * sl);
                    T ablate_sink = (T)0;
                    uint32_t i = wk.i;
                    bool more = true;
                    while (more) {
#pragma unroll
                        for (int j = 0; j < kDepth; ++j) {
                            if (!__any(i < wk.end)) {
                                more = false;
                                break;
                            }
                            rec[(j + kDepth - 1) % kDepth] = ld_rec(i + wk.step * (kDepth - 1));
                            gather((j + kDepth - 2) % kDepth, i + wk.step * (kDepth - 2));
                            __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of this step's arithmetic
                            // false positive, hole, or empty tap (.cpp:679) -> contributes nothing
                            bool pending = lv[j] & (cn[j] != 0);
                            const uint32_t fb = code_bwd(rec[j].code);
                            T(&v)[COUT] = val[j];
                            if (pending) {
                                const T rcpb = cn[j] < 256 ? rinv[cn[j]] : (T)1 / (T)cn[j];   // .cpp:692, :696
#pragma unroll
                                for (int c = 0; c < COUT; ++c) v[c] *= rcpb;
                            }
                            if (CONV3P_ABLATE & 256) {
                                if (pending) {
#pragma unroll
                                    for (int c = 0; c < COUT; ++c) ablate_sink += v[c];
                                }
                                pending = false;
                            }
                            if (CONV3P_ABLATE & 512) {
                                if (pending) {
                                    T *grow = G + ((size_t)fb * COUT) * kCntStride + cq;
#pragma unroll
                                    for (int c = 0; c < COUT; ++c)
                                        __hip_atomic_fetch_add(&grow[c * kCntStride], v[c], __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                                pending = false;
                            }
                            // Lanes of one centre that target the same tap take TURNS at the tap's G entries, lower lane
                            // first: turn = number of lower lanes of the centre with the same tap; a wave's LDS accesses
                            // execute in program order, so turn p adds to what turn p - 1 wrote -- race-free and in a fixed
                            // order.  (Until round 3 the lower lane absorbed the higher one's values through lane swaps
                            // first: 40 swaps and 75 selects per step.)
                            {
                                const int turn = turn_among_lower_lanes(pending ? fb : kTurnIdle, sub, (int)maxn);
                                T *grow = G + ((size_t)(pending ? fb : 0u) * COUT) * kCntStride + cq;
                                for (int p = 0; p < (int)maxn; ++p) {
                                    if (p > 0 && !__any(pending && turn >= p)) break;
                                    if (pending && turn == p) {
#pragma unroll
                                        for (int c = 0; c < COUT; ++c) grow[c * kCntStride] += v[c];
                                    }
                                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                }
                            }
                            i += wk.step;
                        }
                    }
                    if (CONV3P_ABLATE & 256) G[cq] += ablate_sink;
                } else {
                    const uint2 sg = segs[tile_id * ngroups + g];
                    const PairEntry *pe = pairs + sg.x;
                    for (uint32_t e = threadIdx.x; e < sg.y; e += blockDim.x) {
                        const PairEntry en = pe[e];
                        const uint32_t fb = code_bwd(en.code);
                        if (code_fwd(en.code) != kNoTap && fb != kNoTap) accumulate(en.cand, fb, code_q(en.code), (T)0);
                    }
                }
            }
        } else {
            // pair buffer was full for this tile: search it here.  Small path: every wave must see every
            // neighbour (it owns a tap subset), so each wave walks all candidate tiles; generic path: the
            // waves split the candidate tiles (global atomics).
            const T *cloud_box = boxes + (size_t)b * ntiles * 6;
            Query<T> q;
            make_query(q, me, st);
            Window<T> win{};
            if (cmin != nullptr) make_window(win, me, st, cmin + (size_t)b * 3);
            for_each_neighbor(cloud_pts, cloud_box, ntiles, q, st, tapmap, soa, kSmall ? 0 : wave,
                              kSmall ? 1 : kWavesPerBlock, [&](const PointRec<T> &v, int) {
                const uint32_t fb = backward_tap(q.p, v, st, tapmap);
                if (fb != kNoTap && (!kSmall || (int)(fb & (kWavesPerBlock - 1)) == wave))
                    accumulate((uint32_t)v.idx, fb, (uint32_t)lane, (T)0);
            }, win, cmin != nullptr);
        }
    }

    if constexpr (kSmall) {
        BDBG()
        __syncthreads();
        BDBG()
        // (for the wide rows of the 36 -> 13 layer: backward 0.60 -> 0.37 ms at the cfg4 size; with 9 input channels
        // the two versions take the same time, with 3 the padding to 16 columns makes the matrix version slower)
        if constexpr (sizeof(T) == 4 && CONV3P_BWD_MFMA && CIN >= 16) {
            // ---- phases B and C on the matrix cores (fp32: v_mfma_f32_16x16x4_f32 is an exact fmaf chain, so the
            // results stay deterministic; only the order of the sums differs from the vector-ALU version).
            //   B: dW[row = (f',c)][k] = sum_j G[row][j] X[j][k]      M = rows of G, N = Cin (blocks of 16), K = 64 centres
            //   C: dX[j][k]            = sum_row G[row][j] Wt[row][k]  M = 64 centres, N = Cin, K = rows of G
            // Operand maps (cdna_hip_programming.md section 3): A: lane l holds A[i = l & 15][k = l >> 4], B: lane l
            // holds B[k = l >> 4][j = l & 15]; D: register r of lane l is D[row = 4 * (l >> 4) + r][col = l & 15].
            // Rows of G past the last one are read from whatever follows G in LDS (always inside the allocation) and
            // never stored (phase B, where output rows are independent) or zeroed by selects (phase C, where they
            // are the contraction index).
            constexpr int NCB = (CIN + 15) / 16;
            const int l15 = lane & 15, l4 = lane >> 4;
            float *slot = reinterpret_cast<float *>(partials) + (size_t)blockIdx.x * nw;
            const float *Gf = reinterpret_cast<const float *>(G);
            const float *xtf = reinterpret_cast<const float *>(xt);
            const float *wtf = reinterpret_cast<const float *>(wt);
            const int nrb = (nrows + 15) / 16;
            for (int rb0 = wave; rb0 < ((CONV3P_ABLATE & 2) ? 0 : nrb); rb0 += 2 * kWavesPerBlock) {
                // two row blocks at a time (independent accumulator chains): rb0 and rb0 + 4
                const int rb1 = rb0 + kWavesPerBlock;
                const bool two = rb1 < nrb;
                f32x4 acc0[NCB], acc1[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    acc0[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc1[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const float *g0 = Gf + (size_t)(rb0 * 16 + l15) * kCntStride + l4;
                const float *g1 = Gf + (size_t)((two ? rb1 : rb0) * 16 + l15) * kCntStride + l4;
#pragma unroll 4
                for (int j0 = 0; j0 < 64; j0 += 4) {
                    const float a0 = g0[j0], a1 = g1[j0];
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        const int n = cb * 16 + l15;
                        float bv = xtf[(j0 + l4) * CIN + (n < CIN ? n : 0)];
                        bv = n < CIN ? bv : 0.0f;
                        acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc0[cb], 0, 0, 0);
                        acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc1[cb], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = cb * 16 + l15;
                        const int row0 = rb0 * 16 + 4 * l4 + r, row1 = rb1 * 16 + 4 * l4 + r;
                        if (n < CIN && row0 < nrows) {
                            const int f = row0 / COUT, c = row0 - f * COUT;
                            partial_store(&slot[((size_t)f * CIN + n) * COUT + c], acc0[cb][r]);
                        }
                        if (two && n < CIN && row1 < nrows) {
                            const int f = row1 / COUT, c = row1 - f * COUT;
                            partial_store(&slot[((size_t)f * CIN + n) * COUT + c], acc1[cb][r]);
                        }
                    }
            }
            BDBG()
            // phase C: wave w owns the centres 16w .. 16w+15 over ALL rows: no cross-wave reduction
            {
                f32x4 acc0[NCB], acc1[NCB];   // even / odd k-steps: two independent chains, added at the end
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    acc0[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc1[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const int jj = wave * 16 + l15;
                const int nr = (CONV3P_ABLATE & 4) ? 0 : nrows;
                auto step = [&](int r0, f32x4 (&acc)[NCB]) {
                    const int row = r0 + l4;
                    const bool rok = row < nr;
                    const int rc = rok ? row : 0;
                    float av = Gf[(size_t)rc * kCntStride + jj];
                    av = rok ? av : 0.0f;
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        const int n = cb * 16 + l15;
                        float bv = wtf[(size_t)rc * CIN + (n < CIN ? n : 0)];
                        bv = (rok && n < CIN) ? bv : 0.0f;
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[cb], 0, 0, 0);
                    }
                };
                for (int r0 = 0; r0 < nr; r0 += 8) {
                    step(r0, acc0);
                    step(r0 + 4, acc1);
                }
                BDBG()
                if (live) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = cb * 16 + l15;
                            const int orig = qorig[wave * 16 + 4 * l4 + r];
                            if (n < CIN && orig >= 0) {
                                float sum = acc0[cb][r] + acc1[cb][r];
                                const size_t rr = (size_t)b * N + orig;
                                if (act & 2) sum += grad_input[rr * ld.dx + n];
                                if (act & 1) sum = (addend ? sum + addend[rr * ld.add + n] : sum) * selu_slope(input[rr * ld.in + n]);
                                grad_input[rr * ld.dx + n] = sum;
                            }
                        }
                }
            }
        } else {
            // ---- phase B: dW rows.  thread = row (f,c); X tile read with wave-uniform addresses.
            T *slot = partials + (size_t)blockIdx.x * nw;
            for (int row = threadIdx.x; row < ((CONV3P_ABLATE & 2) ? 0 : nrows); row += blockDim.x) {
                T acc[CIN];
#pragma unroll
                for (int k = 0; k < CIN; ++k) acc[k] = (T)0;
                const T *grow = G + (size_t)row * kCntStride;
#pragma unroll 1
                for (int j0 = 0; j0 < 64; j0 += 8) {
                    T g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = grow[j0 + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int k = 0; k < CIN; ++k) acc[k] = fma_t(g[u], xt[(j0 + u) * CIN + k], acc[k]);
                }
                const int f = row / COUT, c = row - f * COUT;
#pragma unroll
                for (int k = 0; k < CIN; ++k) partial_store(&slot[((size_t)f * CIN + k) * COUT + c], acc[k]);
            }
            BDBG()
            // ---- phase C: dX rows.  lane = centre j, waves split the rows.
            T dx[CIN];
#pragma unroll
            for (int k = 0; k < CIN; ++k) dx[k] = (T)0;
            // (round 6) what the SELU-gradient epilogue needs of global memory -- the caller's addend and the layer's own input
            // at this thread's elements (k = wave, wave + 4, ...; centre = lane) -- is requested now, under phase C, instead of
            // as two exposed gathers at the start of the epilogue (the X tile holds the input rows, but `red` overwrites it)
            constexpr int kEpi = (CIN * 64 + 255) / 256;
            T addv[kEpi], xinv[kEpi];
#pragma unroll
            for (int it = 0; it < kEpi; ++it) addv[it] = xinv[it] = (T)0;
            if (live && (act & 1)) {
                const size_t r_ = (size_t)b * N + (me.idx >= 0 ? me.idx : 0);
#pragma unroll
                for (int it = 0; it < kEpi; ++it) {
                    const int k_ = wave + kWavesPerBlock * it;
                    xinv[it] = input[r_ * ld.in + (k_ < CIN ? k_ : 0)];
                    if (addend != nullptr) addv[it] = addend[r_ * ld.add + (k_ < CIN ? k_ : 0)];
                }
            }
            // rows are taken 4 at a time per wave so that the LDS reads of a step are independent (the loop is
            // latency-bound at 2 waves per SIMD); summation order stays fixed: ascending row within a wave
            {
                constexpr int kU = 4;
                const int nr = (CONV3P_ABLATE & 4) ? 0 : nrows;
                int row = wave;
                for (; row + (kU - 1) * kWavesPerBlock < nr; row += kU * kWavesPerBlock) {
                    T g[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) g[u] = G[(size_t)(row + u * kWavesPerBlock) * kCntStride + lane];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const T *wr = wt + (size_t)(row + u * kWavesPerBlock) * CIN;
#pragma unroll
                        for (int k = 0; k < CIN; ++k) dx[k] = fma_t(g[u], wr[k], dx[k]);
                    }
                }
                for (; row < nr; row += kWavesPerBlock) {
                    const T g = G[(size_t)row * kCntStride + lane];
                    const T *wr = wt + (size_t)row * CIN;
#pragma unroll
                    for (int k = 0; k < CIN; ++k) dx[k] = fma_t(g, wr[k], dx[k]);
                }
            }
            BDBG()
            __syncthreads();   // red aliases wt / xt: every wave is done reading them
#pragma unroll
            for (int k = 0; k < CIN; ++k) red[((size_t)wave * CIN + k) * 64 + lane] = dx[k];
            __syncthreads();
            if (live) {
#pragma unroll
                for (int it = 0; it < kEpi; ++it) {
                    const int k = wave + kWavesPerBlock * it;   // (element e = threadIdx.x + 256 it: k = e >> 6, centre = lane)
                    if (k < CIN) {
                        T sum = red[((size_t)0 * CIN + k) * 64 + lane];
#pragma unroll
                        for (int w = 1; w < kWavesPerBlock; ++w) sum += red[((size_t)w * CIN + k) * 64 + lane];
                        if (me.idx >= 0) {
                            const size_t r = (size_t)b * N + me.idx;
                            if (act & 2) sum += grad_input[r * ld.dx + k];
                            if (act & 1) sum = (addend ? sum + addv[it] : sum) * selu_slope(xinv[it]);
                            grad_input[r * ld.dx + k] = sum;
                        }
                    }
                }
            }
        }
        DEV_BWD_PRINT(CIN, COUT, wave, lane)
    }
}

// grad_filter[e] = sum over the workgroups' partial slots, in a fixed order -> run-to-run deterministic given
// deterministic partials.  A workgroup of 16 waves serves kReduceW = 16 consecutive weights: lane = (weight, one of four
// slot stripes), 64 stripes per workgroup, every stripe walks its slots 64 apart with four independent sums; the four
// stripes of a wave meet by two lane exchanges, the 16 waves through LDS in ascending order.  (Round 5: until then a
// workgroup served 64 weights -- 35 workgroups for a 9 -> 9 layer's 2 187 weights on 256 CUs, 64 dependent loads per
// thread: 10.5 us per call at the op boundary, now ~4.)
constexpr int kReduceW = 16;        // reduce_partials_kernel (one layer per launch: 137 workgroups for 2 187 weights)
constexpr int kReduceMultiW = 64;   // reduce_multi_kernel (all layers of a stack, 36 MB: bound by the read, 8.7 us; measured with
                                    // 16 weights per workgroup -- 64-byte pieces per stripe -- 13.2 us)
// ONE summation order for both: 64 stripes per weight (stripe = slot mod 64), each summed 64 slots apart into four
// interleaved accumulators, ((s0 + s1) + (s2 + s3)); the four stripes 4w .. 4w + 3 of "wave" w as (t0 + t2) + (t1 + t3); the
// 16 waves in ascending order.  W = 16: a lane group of the wave per stripe, combined by two lane exchanges; W = 64: a
// thread runs the four stripes itself.  The cached stack (multi) and the op-by-op paths (single) give the same bits.
template <typename T, int W>
__device__ __forceinline__ void reduce_slots(const T *__restrict__ partials, int nslots, size_t nw, T *__restrict__ grad_filter,
                                             unsigned block, T (*part)[W])
{
    static_assert(W == 16 || W == 64, "stripe layouts written for 16 and 64 weights per workgroup");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wl = lane & (W - 1);
    const size_t e = (size_t)block * W + wl;
    auto stripe_sum = [&](int stripe) {
        T s0 = (T)0, s1 = (T)0, s2 = (T)0, s3 = (T)0;
        int p = stripe;
        for (; p + 192 < nslots; p += 256) {
            s0 += partials[(size_t)p * nw + e];
            s1 += partials[(size_t)(p + 64) * nw + e];
            s2 += partials[(size_t)(p + 128) * nw + e];
            s3 += partials[(size_t)(p + 192) * nw + e];
        }
        for (; p < nslots; p += 64) s0 += partials[(size_t)p * nw + e];
        return (s0 + s1) + (s2 + s3);
    };
    T s = (T)0;
    if (W == 16) {
        if (e < nw) s = stripe_sum(wave * 4 + (lane >> 4));
        s += lane_xor32(s);   // t_g + t_(g ^ 2)   (a + b and b + a are the same value)
        s += lane_xor16(s);   // (t0 + t2) + (t1 + t3)
    } else if (e < nw) {
        // (the four stripes one after the other: side by side -- 16 loads per round -- measured 13.9 us against 9.8)
        const T t0 = stripe_sum(wave * 4), t1 = stripe_sum(wave * 4 + 1), t2 = stripe_sum(wave * 4 + 2), t3 = stripe_sum(wave * 4 + 3);
        s = (t0 + t2) + (t1 + t3);
    }
    if (lane < W) part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && lane < W && e < nw) {
        T t = part[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) t += part[w][lane];
        grad_filter[e] = t;
    }
}
template <typename T>
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const T *__restrict__ partials,
                                                               int nslots, size_t nw,
                                                               T *__restrict__ grad_filter)
{
    __shared__ T part[16][kReduceW];
    reduce_slots<T, kReduceW>(partials, nslots, nw, grad_filter, blockIdx.x, part);
}

// The same for several layers in ONE launch (blockIdx.y = layer): the stack-level backward leaves every layer's
// partials in its own region and reduces them all at the end, off the chain of dependent backward kernels.
constexpr int kMaxReduceJobs = 9;
template <typename T> struct ReduceJob {
    const T *partials;
    T *grad_filter;
    int nslots;
    unsigned nw;
};
template <typename T> struct ReduceJobs {
    ReduceJob<T> job[kMaxReduceJobs];
};
template <typename T>
__global__ __launch_bounds__(1024) void reduce_multi_kernel(ReduceJobs<T> jobs)
{
    __shared__ T part[16][kReduceMultiW];
    const ReduceJob<T> &j = jobs.job[blockIdx.y];
    if ((size_t)blockIdx.x * kReduceMultiW >= j.nw) return;   // uniform
    reduce_slots<T, kReduceMultiW>(j.partials, j.nslots, (size_t)j.nw, j.grad_filter, blockIdx.x, part);
}

template <typename T>
__global__ __launch_bounds__(256) void selu_kernel(const T *x, T *y, size_t n)   // y may alias x
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = selu_value(x[i]);
}
// dx = (dy [+ dy_b]) * selu'(y) over `rows` rows of `cols` values; every operand has its own row stride, so that
// the operands may be column blocks of wider buffers (dx may alias dy).
template <typename T>
__global__ __launch_bounds__(256) void selu_grad_kernel(const T *y, const T *dy, const T *dy_b, T *dx, size_t rows,
                                                        int cols, int ld_y, int ld_dy, int ld_b, int ld_dx)
{
    const size_t n = rows * (size_t)cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (size_t)cols;
        const int c = (int)(i - r * (size_t)cols);
        const T g = dy_b ? dy[r * ld_dy + c] + dy_b[r * ld_b + c] : dy[r * ld_dy + c];
        dx[r * ld_dx + c] = g * selu_slope(y[r * ld_y + c]);
    }
}
// dst[r][0..cols) = src[r][0..cols) with independent row strides (column blocks of the concat buffer <-> dense)
template <typename T>
__global__ __launch_bounds__(256) void copy_cols_kernel(const T *src, T *dst, size_t rows, int cols, int ld_s, int ld_d)
{
    const size_t n = rows * (size_t)cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (size_t)cols;
        const int c = (int)(i - r * (size_t)cols);
        dst[r * ld_d + c] = src[r * ld_s + c];
    }
}

// dst[r][0..cols) += src[r][0..cols)  (the later input-channel blocks of a wide layer, see conv3p_abi.hip wide_*)
template <typename T>
__global__ __launch_bounds__(256) void add_cols_kernel(const T *src, T *dst, size_t rows, int cols, int ld_s, int ld_d)
{
    const size_t n = rows * (size_t)cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (size_t)cols;
        const int c = (int)(i - r * (size_t)cols);
        dst[r * ld_d + c] += src[r * ld_s + c];
    }
}
// dst[f][k][c] = src[f][k][c] for f < nf, k < nk, c < nc, with independent tap / row strides on both sides: a
// [nk x nc] block of every tap of a filter <-> its packed copy
template <typename T>
__global__ __launch_bounds__(256) void copy_block_kernel(const T *src, T *dst, int nf, int nk, int nc, size_t sf_s, int sk_s,
                                                         size_t sf_d, int sk_d)
{
    const size_t n = (size_t)nf * nk * nc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / ((size_t)nk * nc), r = i - f * ((size_t)nk * nc);
        const int k = (int)(r / nc), c = (int)(r - (size_t)k * nc);
        dst[f * sf_d + (size_t)k * sk_d + c] = src[f * sf_s + (size_t)k * sk_s + c];
    }
}

}  // namespace conv3p

#include "conv3p_backward_sparse.hpp"
#include "conv3p_forward_taps.hpp"
