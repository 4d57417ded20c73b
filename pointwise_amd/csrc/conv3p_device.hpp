// conv3p_device.hpp -- gfx950 device code shared by the conv3p kernels.
//
// MI355X-first design (NOT a translation of tf_conv3p_atrous.cu, which scans all N
// candidates per point twice per op with the full tap arithmetic, accumulates in global
// memory and does one global float atomic per (pair, k, c)):
//
//   * a cloud is cut into TILES of 64 points = one wavefront; a prep kernel orders the
//     points along a Hilbert curve, stores each as a 16-byte record {x, y, z, original index}
//     and stores every tile's bounding box;
//   * one WORKGROUP owns one QUERY tile (lane = centre point, all waves hold the same 64
//     centres); its waves split the CANDIDATE tiles that survive a bounding-box cull
//     against the union of the 64 filter boxes;
//   * per candidate tile a wave runs a two-stage search:
//       1. PRE-FILTER (vectorised, fp32, conservative): the tile's coordinates are parked in
//          LDS as structure-of-arrays and read back with wave-uniform addresses (LDS
//          broadcast, 4 candidates per ds_read_b128), so each VALU instruction tests one
//          candidate against all 64 centres.  The filter is the atrous lattice itself: per
//          axis z = (v - lo)/(step*voxel) + shift, d = z - clamp(round(z), 0, ext-1), accept
//          iff |d| <= 1/(2*step) + eps -- the box test AND the hole test in 4 VALU per axis,
//          no integer division, no SALU.  The result is a 64-bit hit mask per lane.
//       2. EXACT (only on set bits): the reference's own arithmetic -- box edges in double
//          rounded once, inclusive comparison, IEEE division, truncation, clamp, hole test.
//          Only this stage decides; stage 1 is a strict superset (eps covers fp32 rounding).
//   * tap populations live in LDS ([tap][lane], stride 65 words), shared by the workgroup;
//   * weights are staged in LDS once per workgroup; grad_filter is accumulated in an LDS
//     copy per workgroup and reduced by a second, fixed-order kernel.
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
#include <type_traits>

namespace conv3p {

constexpr int kTile = 64;         // points per tile == wavefront size on gfx950
constexpr int kCntStride = 65;    // LDS row stride of the [tap][lane] tables (bank-conflict-free)
constexpr int kWavesPerBlock = 4;

// LDS carve helper (all offsets multiples of 16 B; one extern array per kernel)
__device__ __forceinline__ size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

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
    int reach[3];    // the reference grid's candidate window, +-n cells per axis: n = (int)((full+1)*0.5)  (.cpp:247-249)
    int window;      // != 0 when some full[a] is even: only then can the window reject a point inside the box
    T voxel;
    double half[3];  // (double)full * 0.5 * (double)voxel        (.cpp:240)
    // pre-filter constants (fp32, see scan_tile)
    float inv[3];    // 1 / (step * voxel)
    float shift[3];  // (1 - 1/step)/2 - 1/2
    float halfw[3];  // 1 / (2*step)
    float mmax[3];   // ext - 1
};

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

// A workgroup's grad_filter partial is written once and read once, by the reduction kernel: streamed past the caches'
// retention (nontemporal) when CONV3P_NT_PARTIALS, so that the 9 MB a backward launch leaves behind are on their way to
// memory before the kernel's end-of-launch write-back (developer A/B: profiles/r05_nt_partials.txt)
#ifndef CONV3P_NT_PARTIALS
#define CONV3P_NT_PARTIALS 0   // measured: no difference on the headline (0.4174 either way), --serial 0.431 against 0.429
#endif
template <typename T> __device__ __forceinline__ void partial_store(T *p, T v)
{
#if CONV3P_NT_PARTIALS
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }


// LDS accumulate without a returned value: ds_add_f32 / ds_add_f64 (fire and forget, no lgkmcnt wait for a result).
__device__ __forceinline__ void lds_add(float *p, float v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_add(double *p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Gather of a feature row with dword alignment only: rows are C*sizeof(T) bytes apart, so the compiler
// cannot prove 16-B alignment and would emit C scalar loads, each of which costs the texture-addresser a
// full pass over 64 scattered cache lines.  global_load_dwordx4 only needs dword alignment on gfx950:
// load 4 elements at a time through an under-aligned vector type.
template <typename T, int C> struct RowLoader {
    static __device__ __forceinline__ void load(const T *__restrict__ p, T (&out)[C])
    {
#pragma unroll
        for (int i = 0; i < C; ++i) out[i] = p[i];
    }
};
typedef float float4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2_a4 __attribute__((ext_vector_type(2), aligned(4)));
template <int C> struct RowLoader<float, C> {
    static __device__ __forceinline__ void load(const float *__restrict__ p, float (&out)[C])
    {
        int i = 0;
#pragma unroll
        for (; i + 4 <= C; i += 4) {
            const float4_a4 v = *reinterpret_cast<const float4_a4 *>(p + i);
            out[i] = v.x; out[i + 1] = v.y; out[i + 2] = v.z; out[i + 3] = v.w;
        }
        if (i + 2 <= C) {
            const float2_a4 v = *reinterpret_cast<const float2_a4 *>(p + i);
            out[i] = v.x; out[i + 1] = v.y;
            i += 2;
        }
        if (i < C) out[i] = p[i];
    }
};


// Value of lane (l ^ 16) / (l ^ 32) without touching LDS: gfx950's v_permlane16_swap / v_permlane32_swap
// exchange 16- / 32-lane blocks between two registers (VALU, no ds_bpermute round trip).
__device__ __forceinline__ uint32_t lane_xor16(uint32_t x)
{
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    // r[0]: odd 16-lane rows hold x of the row below; r[1]: even rows hold x of the row above
    return ((threadIdx.x >> 4) & 1) ? r[0] : r[1];
}
__device__ __forceinline__ uint32_t lane_xor32(uint32_t x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
}

// typed forms (fp64 values travel as two 32-bit halves)
__device__ __forceinline__ float lane_xor16(float v) { return __builtin_bit_cast(float, lane_xor16(__builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ float lane_xor32(float v) { return __builtin_bit_cast(float, lane_xor32(__builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ double lane_xor16(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned long long r = (unsigned long long)lane_xor16((uint32_t)b) | ((unsigned long long)lane_xor16((uint32_t)(b >> 32)) << 32);
    return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double lane_xor32(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned long long r = (unsigned long long)lane_xor32((uint32_t)b) | ((unsigned long long)lane_xor32((uint32_t)(b >> 32)) << 32);
    return __builtin_bit_cast(double, r);
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
    float add[3];       // pre-filter addend: -lo*inv + shift (per lane)
    float thr[3];       // pre-filter half width incl. rounding slack (wave-uniform)
    float cullo[3], culhi[3];   // range of pre-scaled candidate coordinates ANY lane of the wave can accept
    int orig;           // original index inside the cloud, -1 for padding lanes
};

// Window mode only (stencils with an even dilated extent): the cloud's grid origin (.cpp:163-177) and the
// centre's grid cell (.cpp:251-253).  Kept out of Query so that the common kernels carry no extra live state.
template <typename T> struct Window {
    T vmin[3];
    int cell[3];
};

// Grid cell of coordinate v (cell edge = voxel, origin vmin): IEEE divide + truncate (.cpp:192-194, :251-253).
template <typename T> __device__ __forceinline__ int grid_cell(T v, T vmin, T voxel) { return (int)((v - vmin) / voxel); }

// The reference only visits candidates in the cells centre +- reach (.cpp:260-266).  For odd dilated extents that
// window is half a cell wider than the filter box and never decides; for EVEN extents its border coincides with
// the box edge, and on voxel-aligned data rounding of the two cell indices drops candidates that pass the
// inclusive box test.  Replicated exactly: same origin, same division, same truncation.
template <typename T>
__device__ __forceinline__ bool outside_window(T vx, T vy, T vz, const T *vmin, const int *ccell, const Stencil<T> &st)
{
    const int dx = grid_cell(vx, vmin[0], st.voxel) - ccell[0];
    const int dy = grid_cell(vy, vmin[1], st.voxel) - ccell[1];
    const int dz = grid_cell(vz, vmin[2], st.voxel) - ccell[2];
    return (dx > st.reach[0]) | (dx < -st.reach[0]) | (dy > st.reach[1]) | (dy < -st.reach[1]) |
           (dz > st.reach[2]) | (dz < -st.reach[2]);
}

template <typename T>
__device__ __forceinline__ void make_window(Window<T> &w, const PointRec<T> &me, const Stencil<T> &st,
                                            const T *__restrict__ cloud_min)
{
#pragma unroll
    for (int a = 0; a < 3; ++a) w.vmin[a] = cloud_min[a];
    const bool valid = me.idx >= 0;
    w.cell[0] = valid ? grid_cell(me.x, w.vmin[0], st.voxel) : 0;
    w.cell[1] = valid ? grid_cell(me.y, w.vmin[1], st.voxel) : 0;
    w.cell[2] = valid ? grid_cell(me.z, w.vmin[2], st.voxel) : 0;
}

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
        q.add[a] = (float)((double)st.shift[a] - (double)q.lo[a] * (double)st.inv[a]);
    }
    // slack: fp32 rounding of v*inv + add at the magnitude of the coordinates involved (worst axis, so
    // that an isotropic stencil keeps one common half-width)
    float mag = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        mag = fmaxf(mag, fmaxf(fabsf((float)q.ulo[a]), fabsf((float)q.uhi[a])) * st.inv[a]);
#pragma unroll
    for (int a = 0; a < 3; ++a) q.thr[a] = st.halfw[a] + 1.0e-5f + 1.0e-6f * mag;
    // The pre-filter accepts z = s + add (s = pre-scaled candidate coordinate) when z is within thr of a
    // lattice point 0..mmax, i.e. only if s lies in [-thr - add, mmax + thr - add].  The union of that range
    // over the wave's valid lanes (plus rounding slack) culls groups of 4 staged candidates (stage_tile).
    const float slack = 1.0e-3f + 1.0e-5f * mag;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q.cullo[a] = -q.thr[a] - wave_max(valid ? q.add[a] : -Limits<float>::inf()) - slack;
        q.culhi[a] = st.mmax[a] + q.thr[a] - wave_min(valid ? q.add[a] : Limits<float>::inf()) + slack;
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
    t = t < 0 ? 0 : t;   // unreachable for finite data: v >= lo up to rounding truncates to 0 already
    return map_row[t];
}

// Exact membership + tap of candidate v for the lane's query: the reference's inclusive box
// test (.cpp:277) followed by the tap computation (.cpp:280-290).  Returns the tap or -1.
template <typename T>
__device__ __forceinline__ int exact_tap(const PointRec<T> &v, const Query<T> &q, const Stencil<T> &st,
                                         const int16_t *tapmap, const Window<T> &win, bool use_win)
{
    const bool out = (v.x < q.lo[0]) | (v.x > q.hi[0]) | (v.y < q.lo[1]) | (v.y > q.hi[1]) |
                     (v.z < q.lo[2]) | (v.z > q.hi[2]);
    if (out) return -1;
    if (use_win && outside_window(v.x, v.y, v.z, win.vmin, win.cell, st)) return -1;   // .cpp:260-266
    const int tx = axis_tap(v.x, q.lo[0], st.voxel, st.full[0], tapmap);
    const int ty = axis_tap(v.y, q.lo[1], st.voxel, st.full[1], tapmap + st.maxfull);
    const int tz = axis_tap(v.z, q.lo[2], st.voxel, st.full[2], tapmap + 2 * st.maxfull);
    if ((tx | ty | tz) < 0) return -1;                       // hole (.cpp:285)
    return (tz * st.ext[1] + ty) * st.ext[0] + tx;           // .cpp:290
}

// Stage one candidate tile for the pre-filter: lane l writes its point's fp32 coordinates, already
// multiplied by inv = 1/(step*voxel), to the wave's SoA slot (soa[0..63] = x, [64..127] = y,
// [128..191] = z), so that the per-pair arithmetic starts with a v_add instead of a v_fma.
// Returns the quad mask: bits 4k..4k+3 are set iff the bounding box of candidates 4k..4k+3 (consecutive in
// Hilbert order, so compact) meets the wave's acceptance range; scan_tile skips the other quads.
__device__ __forceinline__ float quad_swap1(float v)   // lanes {0,1,2,3} -> {1,0,3,2}
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_swap2(float v)   // lanes {0,1,2,3} -> {2,3,0,1}
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
template <typename T>
__device__ __forceinline__ uint64_t stage_tile(float *soa, const PointRec<T> &cand, const Stencil<T> &st,
                                               const Query<T> &q)
{
    const int lane = threadIdx.x & 63;
    const float s[3] = {(float)cand.x * st.inv[0], (float)cand.y * st.inv[1], (float)cand.z * st.inv[2]};
    soa[lane] = s[0];
    soa[64 + lane] = s[1];
    soa[128 + lane] = s[2];
    bool ov = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float mn = fminf(s[a], quad_swap1(s[a])), mx = fmaxf(s[a], quad_swap1(s[a]));
        mn = fminf(mn, quad_swap2(mn));
        mx = fmaxf(mx, quad_swap2(mx));
        ov &= (mx >= q.cullo[a]) & (mn <= q.culhi[a]);
    }
    return __ballot(ov);
}

// Pre-filter of the 64 staged candidates against the lane's query.  Candidate c ends up in bit
// (31 - c) of m0 for c < 32 and bit (63 - c) of m1 otherwise.
// Instruction choice follows the measured issue rates on gfx950 (tools/ubench/valu_rate.hip): plain
// VOP2 ops (v_add/v_sub) issue in ~2.7 cycles per wave, every VOP3/VOP1 op (v_fma, v_rndne, v_med3,
// v_max3, v_min3, v_cmp) in ~4.2.  z = v*inv + add is a v_add on pre-scaled coordinates.
//   TAPS3 (filters with 3 taps per axis, every layer of the reference's models):
//       |d| = min3(|z|, |z-1|, |z-2|)                 2 v_sub + 1 v_min3   (9.6 cycles/axis)
//   otherwise  d = z - med3(rndne(z), 0, ext-1)       v_rndne + v_med3 + v_sub (11 cycles/axis)
//   ISO (one half-width for all axes): hit = max3(|dx|,|dy|,|dz|) <= thr     1 v_max3
// and the mask bit is inserted with v_cmp + v_addc (hipcc's cndmask+shift+or: 7 % slower; sign bit via
// v_alignbit_b32: 10 % slower; an all-integer packed-phase filter was 1.5x cheaper per pair but its
// coarser acceptance lengthened the pair lists enough to cancel the gain).
template <typename T, bool ISO, bool TAPS3>
__device__ __forceinline__ void scan_tile_impl(const float *soa, const Query<T> &q, const Stencil<T> &st,
                                               uint64_t quads, uint32_t &m0, uint32_t &m1)
{
    m0 = 0;
    m1 = 0;
    const float4 *sx = reinterpret_cast<const float4 *>(soa);
    const float4 *sy = reinterpret_cast<const float4 *>(soa + 64);
    const float4 *sz = reinterpret_cast<const float4 *>(soa + 128);
#pragma unroll
    for (int c4 = 0; c4 < 16; ++c4) {
        if (((quads >> (4 * c4)) & 1) == 0) {   // wave-uniform: no lane can accept any of these 4 candidates
            if (c4 < 8) m0 <<= 4;
            else m1 <<= 4;
            continue;
        }
        const float4 X = sx[c4], Y = sy[c4], Z = sz[c4];   // wave-uniform address: LDS broadcast
        const float vx[4] = {X.x, X.y, X.z, X.w}, vy[4] = {Y.x, Y.y, Y.z, Y.w}, vz[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float zx = vx[u] + q.add[0], zy = vy[u] + q.add[1], zz = vz[u] + q.add[2];
            float ax, ay, az;   // distance to the nearest accepted lattice point
            if (TAPS3) {
                ax = fminf(fminf(fabsf(zx), fabsf(zx - 1.0f)), fabsf(zx - 2.0f));
                ay = fminf(fminf(fabsf(zy), fabsf(zy - 1.0f)), fabsf(zy - 2.0f));
                az = fminf(fminf(fabsf(zz), fabsf(zz - 1.0f)), fabsf(zz - 2.0f));
            } else {
                ax = fabsf(zx - __builtin_amdgcn_fmed3f(__builtin_rintf(zx), 0.0f, st.mmax[0]));
                ay = fabsf(zy - __builtin_amdgcn_fmed3f(__builtin_rintf(zy), 0.0f, st.mmax[1]));
                az = fabsf(zz - __builtin_amdgcn_fmed3f(__builtin_rintf(zz), 0.0f, st.mmax[2]));
            }
            float e, lim;
            if (ISO) {
                e = fmaxf(fmaxf(ax, ay), az);
                lim = q.thr[0];
            } else {
                e = fmaxf(fmaxf(ax - q.thr[0], ay - q.thr[1]), az - q.thr[2]);
                lim = 0.0f;
            }
            if (c4 < 8)
                asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m0) : "v"(e), "v"(lim) : "vcc");
            else
                asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m1) : "v"(e), "v"(lim) : "vcc");
        }
    }
}
template <typename T>
__device__ __forceinline__ void scan_tile(const float *soa, const Query<T> &q, const Stencil<T> &st,
                                          uint64_t quads, uint32_t &m0, uint32_t &m1)
{
    // wave-uniform choice (kernel arguments + wave-uniform slack)
    const bool iso = q.thr[0] == q.thr[1] && q.thr[1] == q.thr[2];
    const bool taps3 = st.ext[0] == 3 && st.ext[1] == 3 && st.ext[2] == 3;
    if (iso && taps3)
        scan_tile_impl<T, true, true>(soa, q, st, quads, m0, m1);
    else
        scan_tile_impl<T, false, false>(soa, q, st, quads, m0, m1);
}

// Candidate tiles whose bounding box meets the union of the wave's filter boxes: 64 tiles per
// ballot, one lane per tile.
template <typename T>
__device__ __forceinline__ uint64_t overlapping_tiles(const T *__restrict__ cloud_box, int ntiles, int base,
                                                      const Query<T> &q)
{
    const int t = base + (threadIdx.x & 63);
    bool ov = false;
    if (t < ntiles) {
        const T *bb = cloud_box + (size_t)t * 6;   // {min xyz, max xyz}
        const T b0 = bb[0], b1 = bb[1], b2 = bb[2], b3 = bb[3], b4 = bb[4], b5 = bb[5];
        ov = !((b3 < q.ulo[0]) | (b0 > q.uhi[0]) | (b4 < q.ulo[1]) | (b1 > q.uhi[1]) | (b5 < q.ulo[2]) |
               (b2 > q.uhi[2]));
    }
    return __ballot(ov);
}

// Walk the set bits of a lane's 64-bit hit mask (m0: candidates 0..31, m1: 32..63, see scan_tile).
// visit(c) runs with only the lanes that still have a candidate active.
template <class Visit>
__device__ __forceinline__ void for_each_bit(uint32_t m0, uint32_t m1, Visit &&visit)
{
    while (__any(m0 != 0)) {
        if (m0 != 0) {
            const int p = __builtin_ctz(m0);
            m0 &= m0 - 1;
            visit(31 - p);
        }
    }
    while (__any(m1 != 0)) {
        if (m1 != 0) {
            const int p = __builtin_ctz(m1);
            m1 &= m1 - 1;
            visit(63 - p);
        }
    }
}

// Search of one query tile by one wave: candidate tiles `first, first+stride, ...` of the
// surviving list (stride a power of two).  on_hit(const PointRec<T>&, int tap) is called for every EXACT neighbour.
template <typename T, class OnHit>
__device__ __forceinline__ void for_each_neighbor(const PointRec<T> *__restrict__ cloud_pts,
                                                  const T *__restrict__ cloud_box, int ntiles,
                                                  const Query<T> &q, const Stencil<T> &st,
                                                  const int16_t *tapmap, float *soa, int first, int stride,
                                                  OnHit &&on_hit, const Window<T> &win, bool use_win)   // (by reference + flag: a pointer that may be null kept the window in scratch memory)
{
    const int lane = threadIdx.x & 63;
    const bool qvalid = q.orig >= 0;
    int seen = 0;
    for (int base = 0; base < ntiles; base += 64) {
        uint64_t tiles = overlapping_tiles(cloud_box, ntiles, base, q);
        while (tiles) {
            const int ct = base + __builtin_ctzll(tiles);
            tiles &= tiles - 1;
            if ((seen++ & (stride - 1)) != first) continue;   // stride is a power of two
            const PointRec<T> *tile = cloud_pts + (size_t)ct * kTile;
            const uint64_t quads = stage_tile(soa, tile[lane], st, q);
            __builtin_amdgcn_wave_barrier();
            uint32_t m0, m1;
            scan_tile(soa, q, st, quads, m0, m1);
            __builtin_amdgcn_wave_barrier();
            if (!qvalid) m0 = m1 = 0;
            for_each_bit(m0, m1, [&](int c) {
                const PointRec<T> v = tile[c];
                const int f = exact_tap(v, q, st, tapmap, win, use_win);
                if (f >= 0) on_hit(v, f);
            });
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Pair lists.  The search kernel resolves every (centre j, neighbour ii) pair once and stores the 8-byte
//   PairEntry{ cand = original index of ii,  code = fwd_tap | bwd_tap << 12 | q << 24 }
// fwd_tap : tap of ii inside j's box                      (forward,  .cpp:280-290)
// bwd_tap : tap of j inside ii's box, kNoTap for a hole   (backward, .cpp:662-677)
// q       : lane (0..63) of the centre inside its query tile
// All pairs of a query tile are contiguous (one segment per tile), centre-major inside.  Entries
// whose exact test failed (pre-filter false positives) carry fwd_tap = kNoTap and are skipped by
// the consumers.  The normalisers are NOT stored: the forward's 1/count[j][fwd] comes from the tile's own
// population rows (an LDS table of reciprocals per workgroup), the backward's 1/count[ii][bwd] from one
// 4-byte gather of the neighbour's population next to its grad_out row (0 = empty tap, the reference's
// `count == 0` skip, .cpp:679).  No pass over the lists is needed once the populations are complete.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kNoTap = 0xFFFu;
constexpr uint32_t kSegOverflow = 0xFFFFFFFFu;
struct __attribute__((aligned(8))) PairEntry {
    uint32_t cand;
    uint32_t code;
};
__device__ __forceinline__ uint32_t pair_code(uint32_t fwd, uint32_t bwd, uint32_t q) { return fwd | (bwd << 12) | (q << 24); }
__device__ __forceinline__ uint32_t code_fwd(uint32_t c) { return c & 0xFFFu; }
__device__ __forceinline__ uint32_t code_bwd(uint32_t c) { return (c >> 12) & 0xFFFu; }
__device__ __forceinline__ uint32_t code_q(uint32_t c) { return c >> 24; }

// exclusive prefix sum of a small non-negative integer across the wave (returns total in `total`)
__device__ __forceinline__ int wave_excl_scan(int v, int &total)
{
    const int lane = threadIdx.x & 63;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(inc, o);
        if (lane >= o) inc += n;
    }
    total = __shfl(inc, 63);
    return inc - v;
}

// ---------------------------------------------------------------------------------------------------------------
// Lanes of a wave dealt to its 16 centres IN PROPORTION TO THEIR PAIR LISTS (round 5).  The list-walking kernels gave
// every centre four lanes that stepped through its list four records at a time until the longest of the wave's 16
// lists was consumed: steps = ceil(max len / 4), 59-75 % of the lane-steps on a record (profiles/r05_walk_stats.txt).
// Here centre c gets n_c = ceil(len_c / S) consecutive lanes, S the smallest step count for which the 16 centres fit
// the 64 lanes; lane r of a centre takes the records r, r + n_c, r + 2 n_c, ...: steps = S ~ ceil(sum len / 56).
// A lane keeps its centre for the whole walk, so accumulators stay in registers and the lanes of a centre (consecutive:
// [first, first + n)) are combined once, at the end, in ascending lane order -- bitwise reproducible.
//   seg      (start, length) of the list of centre (lane & 15) of this wave, the same in the four 16-lane rows
//   scratch  112 words of LDS private to the wave
// Returns the lane's centre (0 .. 15 within the wave), its index among the centre's lanes, their number (0: idle
// lane) and the centre's own segment; `spans[c]` = first lane | lanes << 8 of every centre is left in scratch[64 + c].
struct LaneShare {
    uint32_t cl, r, n;
    uint2 seg;
    uint32_t maxn;   // the largest lane count of a centre of this wave (uniform)
};
__device__ __forceinline__ uint32_t row16_sum(uint32_t v)   // sum over the 16 lanes of a row, in every lane
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);   // row_ror:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);   // row_ror:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false);   // row_ror:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false);   // row_ror:1
    return v;
}
__device__ __forceinline__ uint32_t row16_incl_scan(uint32_t v)   // inclusive prefix sum inside a 16-lane row
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1 (zeros shifted in)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    return v;
}
__device__ __forceinline__ uint32_t ceil_div_small(uint32_t a, uint32_t s, float rs)   // ceil(a / s), a + s < 2^23, rs = 1 / s
{
    const uint32_t x = a + s - 1u;
    uint32_t q = (uint32_t)((float)x * rs);   // within one of floor(x / s)
    q = q * s > x ? q - 1u : q;
    q = (q + 1u) * s <= x ? q + 1u : q;
    return q;
}
constexpr int kShareWords = 112;   // LDS words share_lanes() needs per wave: heads [64] | spans [16] | segments [16][2]
__device__ __forceinline__ LaneShare share_lanes(uint2 seg, uint32_t *scratch)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t len = seg.y == 0xFFFFFFFFu ? 0u : seg.y;   // (an overflowed tile never gets here)
    const uint32_t tot = row16_sum(len);
    // expected demand at S steps: sum len / S + half a lane per centre -> start at sum / 56; every further S is tried
    // in turn (sum / 48 always fits: sum ceil(len / S) <= sum / S + 16)
    uint32_t S = tot / 56u + 1u;
    uint32_t q;
    for (;;) {
        q = ceil_div_small(len, S, 1.0f / (float)S);
        if (__builtin_amdgcn_readfirstlane((int)row16_sum(q)) <= 64) break;
        ++S;
    }
    const uint32_t first = row16_incl_scan(q) - q;
    uint32_t qmax = q;
    qmax = max(qmax, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qmax, 0x128, 0xF, 0xF, false));   // row_ror:8
    qmax = max(qmax, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qmax, 0x124, 0xF, 0xF, false));
    qmax = max(qmax, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qmax, 0x122, 0xF, 0xF, false));
    qmax = max(qmax, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qmax, 0x121, 0xF, 0xF, false));
    scratch[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    if (lane < 16u) {
        if (q != 0u) scratch[first] = lane + 1u;
        scratch[64 + lane] = first | (q << 8);
        scratch[80 + 2 * lane] = seg.x;
        scratch[81 + 2 * lane] = seg.y;
    }
    __builtin_amdgcn_wave_barrier();
    const uint64_t heads = __ballot(scratch[lane] != 0u);
    const uint64_t below = heads & (~0ull >> (63u - lane));   // heads at or below this lane
    LaneShare ls;
    ls.maxn = (uint32_t)__builtin_amdgcn_readfirstlane((int)qmax);
    if (below == 0ull) {
        ls.cl = 0u; ls.r = 0u; ls.n = 0u;
        ls.seg = make_uint2(0u, 0u);
        return ls;
    }
    const uint32_t h = 63u - (uint32_t)__builtin_clzll(below);
    ls.cl = scratch[h] - 1u;
    const uint32_t span = scratch[64 + ls.cl];
    ls.r = lane - h;
    ls.n = span >> 8;
    ls.seg = make_uint2(scratch[80 + 2 * ls.cl], scratch[81 + 2 * ls.cl]);
    if (ls.r >= ls.n) {   // lanes past the last centre's share: idle (no records, first index 0)
        ls.r = 0u;
        ls.n = 0u;
        ls.seg.y = 0u;
    }
    return ls;
}
// A lane's walk over its share of its centre's list: records i, i + step, ... below end.
//   interleaved (forward: nothing is shared between the lanes of a centre): lane r takes r, r + n, r + 2 n, ...
//   blocked (backward: lanes of a centre that meet on a tap take turns, and neighbours in a list -- neighbours in
//   space -- mostly share their tap): lane r takes the r-th run of ceil(len / n) consecutive records
struct LaneWalk {
    uint32_t i, step, end;
};
__device__ __forceinline__ LaneWalk lane_walk(const LaneShare &ls, bool blocked)
{
    LaneWalk w;
    if (ls.n == 0u) {
        w.i = 0u; w.step = 1u; w.end = 0u;
    } else if (blocked) {
        const uint32_t per = ceil_div_small(ls.seg.y, ls.n, 1.0f / (float)ls.n);
        w.i = ls.r * per;
        w.step = 1u;
        w.end = w.i + per < ls.seg.y ? w.i + per : ls.seg.y;
        if (w.i > w.end) w.i = w.end;
    } else {
        w.i = ls.r;
        w.step = ls.n;
        w.end = ls.seg.y;
    }
    return w;
}
// value of lane U of this lane's quad (four adjacent lanes) in all four of them: v_mov_dpp quad_perm:[U,U,U,U]
template <int U> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, U * 0x55, 0xF, 0xF, true);
}
// Turn of this lane among the lanes of its centre (the r lanes just below it belong to the same centre) that want the
// same tap in this step: the number of those lower lanes whose tap equals this lane's.  `tap` = kTurnIdle for a lane
// that wants nothing; maxn = the largest lane count of a centre in this wave (uniform).
constexpr uint32_t kTurnIdle = 0xFFFFFFFFu;
__device__ __forceinline__ int turn_among_lower_lanes(uint32_t tap, uint32_t r, int maxn)
{
    int turn = 0;
    uint32_t t = tap;
    for (int d = 1; d < maxn; ++d) {
        t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFEu, (int)t, 0x138, 0xF, 0xF, false);   // wave_shr:1: lane l <- lane l - 1
        turn += ((uint32_t)d <= r && t == tap && tap != kTurnIdle) ? 1 : 0;
    }
    return turn;
}
// the same interface with four consecutive lanes per centre (clouds searched in several groups: a lane must keep its
// centre from one group's list to the next)
__device__ __forceinline__ LaneShare share_lanes_uniform(uint2 seg_of_my_centre)
{
    const uint32_t lane = threadIdx.x & 63u;
    LaneShare ls;
    ls.cl = lane >> 2;
    ls.r = lane & 3u;
    ls.n = 4u;
    ls.seg = seg_of_my_centre;
    ls.maxn = 4u;
    return ls;
}

// Per-centre exact data kept in LDS for the dense (lane = pair) stages.
template <typename T> struct CentreRec {
    T p[3];
    T lo[3];
    T hi[3];
    int32_t orig;
};

// (cloud, query tile) of a workgroup.  Workgroup b is placed on XCD b % 8 by the dispatcher
// (observed, used for L2 locality only): clouds are dealt to XCDs round-robin and all tiles of
// a cloud run on that cloud's XCD, so points / features / counts of a cloud stay in one XCD's
// 4 MiB L2.
struct BlockMap {
    int blocks_per_cloud;   // query tiles (or groups of them) per cloud
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
