// conv3p_device.hpp -- gfx950 device code shared by the conv3p kernels.
//
// MI355X-first design (NOT a translation of tf_conv3p_atrous.cu, which scans all N
// candidates per point twice per op, accumulates in global memory and does one global
// float atomic per (pair, k, c)):
//
//   * a cloud is cut into TILES of 64 points = one wavefront; a prep kernel stores each
//     point as a 16-byte record {x, y, z, original index} (one coalesced dwordx4 per lane)
//     and each tile's bounding box;
//   * a wavefront owns one QUERY tile (lane = centre point).  It first culls CANDIDATE
//     tiles against the union of its 64 filter boxes (one lane per candidate tile, one
//     ballot), then for every surviving candidate tile
//        - loads the tile with one coalesced load per lane and parks it in LDS,
//        - broadcasts the 64 candidates one by one through v_readlane (SGPR operands) and
//          runs the reference's inclusive box test on all 64 centres at once, building a
//          64-bit hit mask per lane,
//        - walks the set bits: tap index with the reference's exact float arithmetic,
//          hole test, then the op-specific accumulation.
//   * tap populations live in LDS, lane-private ([tap][lane], stride 65 words);
//   * weights are staged in LDS once per workgroup; grad_filter is accumulated in an LDS
//     copy per workgroup and reduced by a second, deterministic-order kernel.
//
// Exactness: box edges are evaluated in double and rounded once (reference
// tf_conv3p_atrous.cpp:240-245), the box test is inclusive (:277), taps use IEEE
// division + truncation (:280-282; hipcc keeps `/` correctly rounded by default, this
// file is built with -ffp-contract=off), clamp + hole test + stride division (:280-288).
// Neighbour / tap decisions are therefore identical to the CPU reference; only the
// order of floating-point summation differs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace conv3p {

constexpr int kTile = 64;         // points per tile == wavefront size on gfx950
constexpr int kCntStride = 65;    // LDS row stride of the [tap][lane] tables (bank-conflict-free)
constexpr int kWavesPerBlock = 4;

// One staged point: 16 B for float, 32 B for double.
template <typename T> struct PointRec;
template <> struct __attribute__((aligned(16))) PointRec<float> {
    float x, y, z;
    int32_t idx;
};
template <> struct __attribute__((aligned(16))) PointRec<double> {
    double x, y, z;
    int32_t idx;
    int32_t pad;
};

// Filter geometry, passed by value in kernarg (wave-uniform -> SGPRs).
template <typename T> struct Stencil {
    int ext[3];      // taps along x, y, z  (fx, fy, fz)
    int step[3];     // stride along x, y, z
    int full[3];     // dilated extent (ext-1)*step+1            (.cpp:235-237)
    int ntap;        // fx*fy*fz
    int maxfull;     // max(full[a]); row length of the tap lookup table
    T voxel;
    double half[3];  // (double)full * 0.5 * (double)voxel        (.cpp:240)
};

__device__ __forceinline__ float lane_bcast(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double lane_bcast(double v, int l)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <typename T> __device__ __forceinline__ T wave_min(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

template <typename T> struct Limits;
template <> struct Limits<float> {
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
};
template <> struct Limits<double> {
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
};

// The per-lane state of a query (centre) point.
template <typename T> struct Query {
    T p[3];
    T lo[3], hi[3];     // own filter box                              (.cpp:240-245)
    T ulo[3], uhi[3];   // union of the wave's 64 boxes (wave-uniform)
    int orig;           // original index inside the cloud, -1 for padding lanes
};

template <typename T>
__device__ __forceinline__ void make_query(Query<T> &q, const PointRec<T> &me, const Stencil<T> &st)
{
    q.p[0] = me.x;
    q.p[1] = me.y;
    q.p[2] = me.z;
    q.orig = me.idx;
    const bool valid = me.idx >= 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q.lo[a] = (T)((double)q.p[a] - st.half[a]);
        q.hi[a] = (T)((double)q.p[a] + st.half[a]);
        q.ulo[a] = wave_min(valid ? q.lo[a] : Limits<T>::inf());
        q.uhi[a] = wave_max(valid ? q.hi[a] : -Limits<T>::inf());
    }
}

// tap lookup: tapmap[a*maxfull + t] = t/step[a] if t % step[a] == 0 else -1   (.cpp:285-288)
// (built once per workgroup so that the hot loop has no integer division).
__device__ __forceinline__ void build_tapmap(int16_t *tapmap, const int *full, const int *step,
                                             int maxfull)
{
    for (int e = threadIdx.x; e < 3 * maxfull; e += blockDim.x) {
        const int a = e / maxfull, t = e - a * maxfull;
        int16_t v = -1;
        if (t < full[a] && (t % step[a]) == 0) v = (int16_t)(t / step[a]);
        tapmap[e] = v;
    }
}

// Tap of coordinate v in a box with lower edge lo: IEEE divide, truncate, clamp (.cpp:280-282),
// then hole test / stride division through the lookup table.  -1 = hole.
template <typename T>
__device__ __forceinline__ int axis_tap(T v, T lo, T voxel, int full, const int16_t *map_row)
{
    int t = (int)((v - lo) / voxel);
    t = t > full - 1 ? full - 1 : t;
    if (t < 0) return -1;   // unreachable for finite data (v >= lo up to rounding truncates to 0)
    return map_row[t];
}

// Visit every candidate whose position lies inside the lane's filter box
// (inclusive test, .cpp:277).  `tile_lds` is this wave's private 64-record LDS slot.
// on_hit(const PointRec<T>&) runs with only the hit lanes active.
template <typename T, class OnHit>
__device__ __forceinline__ void for_each_box_hit(const PointRec<T> *__restrict__ cloud_pts,
                                                 const T *__restrict__ cloud_box, int ntiles,
                                                 const Query<T> &q, PointRec<T> *tile_lds,
                                                 OnHit &&on_hit)
{
    const int lane = threadIdx.x & 63;
    const bool qvalid = q.orig >= 0;
    for (int base = 0; base < ntiles; base += 64) {
        const int t = base + lane;
        bool ov = false;
        if (t < ntiles) {
            const T *bb = cloud_box + (size_t)t * 6;   // {min xyz, max xyz}
            ov = !(bb[3] < q.ulo[0] || bb[0] > q.uhi[0] || bb[4] < q.ulo[1] || bb[1] > q.uhi[1] ||
                   bb[5] < q.ulo[2] || bb[2] > q.uhi[2]);
        }
        uint64_t tiles = __ballot(ov);
        while (tiles) {
            const int ct = base + __builtin_ctzll(tiles);
            tiles &= tiles - 1;
            const PointRec<T> cand = cloud_pts[(size_t)ct * kTile + lane];
            tile_lds[lane] = cand;
            uint32_t mlo = 0, mhi = 0;
#pragma unroll
            for (int c = 0; c < 64; ++c) {
                const T vx = lane_bcast(cand.x, c);
                const T vy = lane_bcast(cand.y, c);
                const T vz = lane_bcast(cand.z, c);
                const bool out = (vx < q.lo[0]) | (vx > q.hi[0]) | (vy < q.lo[1]) |
                                 (vy > q.hi[1]) | (vz < q.lo[2]) | (vz > q.hi[2]);
                if (c < 32)
                    mlo |= out ? 0u : (1u << c);
                else
                    mhi |= out ? 0u : (1u << (c - 32));
            }
            uint64_t mask = qvalid ? (((uint64_t)mhi << 32) | mlo) : 0ull;
            __builtin_amdgcn_wave_barrier();
            while (__any(mask != 0)) {
                if (mask != 0) {
                    const int c = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const PointRec<T> v = tile_lds[c];
                    on_hit(v);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// (cloud, first query tile) of a workgroup.  Workgroup b is placed on XCD b % 8 by the
// dispatcher (observed, used for L2 locality only): clouds are dealt to XCDs round-robin
// and all tiles of a cloud run on that cloud's XCD, so points / features / counts of a
// cloud stay in one XCD's 4 MiB L2.
struct BlockMap {
    int blocks_per_cloud;   // ceil(ntiles / kWavesPerBlock)
    int clouds;             // B
    int rounds;             // ceil(B / 8)
};
__device__ __forceinline__ bool block_to_cloud(const BlockMap &m, int &cloud, int &blk_in_cloud)
{
    const int xcd = blockIdx.x & 7;
    const int r = blockIdx.x >> 3;
    const int round = r / m.blocks_per_cloud;
    blk_in_cloud = r - round * m.blocks_per_cloud;
    cloud = xcd + 8 * round;
    return cloud < m.clouds;
}

}  // namespace conv3p
