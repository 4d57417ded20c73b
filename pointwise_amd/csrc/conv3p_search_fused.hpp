// conv3p_search_fused.hpp -- the neighbour search of several stencils over the same sorted points in ONE pass.
//
// The models' layers share `points` and differ only in stride (pointcnn2_acsd.py:48-66: four strides, one `points`);
// what decides a pair is tf_conv3p_atrous.cpp:232-301 (box, tap, clamp, hole).  search_tile (conv3p_kernels.hpp) tests
// every candidate of every surviving candidate tile against the 64 centres of a query tile with ~15 vector
// instructions per candidate AND stencil.  Here the per-candidate arithmetic is done ONCE PER CLOUD, by the candidate:
//
//   tile_tables_kernel   per tile and axis: bucket b of every point (1/16 voxel, counted from the tile's own minimum;
//                        coarser by powers of two for tiles wider than 6 voxels) and, for every bucket k, the 64-bit
//                        WINDOW mask W[k] = { points with bucket in [k, k + one voxel] }.  2.8 KB per tile.
//   search_fused_kernel  ONE launch for all stencils of a step (blockIdx.y = stencil), one workgroup per query tile:
//     pass 1  per surviving candidate tile: the tile's table is staged in LDS; a centre's acceptance set along an axis
//             is the union of `ext` one-voxel intervals (one per tap), i.e. `ext` table look-ups; the pre-filter hit
//             mask is the AND of the three axis masks.  ~100 vector instructions per tile pair for all 64 candidates,
//             whatever the stride -- against ~15 per candidate.  Masks stay in LDS (the first few per wave) or are
//             recomputed.
//     P2      one reservation for the query tile, centre-major slots (as search_tile)
//     pass 2  every wave streams its hits (ranked by ballot into an LDS stream, with the slot each one owns) through the
//             dense exact stage, lane = pair: the reference's arithmetic decides, exactly as in search_tile's P3 --
//             except that a pair further than a few ulps from every tap boundary needs no division and its backward tap
//             is the mirror of the forward one (fused_resolve)
//   (Several stencils per workgroup -- stage once, look up for all -- was built and measured: 263 us against 203 for
//   the cfg2 geometry; what the stencils could share is ~8 % of the instructions, what they cost is occupancy.)
//   Only the exact stage decides; the tables are a superset filter (see the slack analysis at fused_masks).
// Results: the same pairs, taps and populations as search_tile; the order of a centre's list differs (false positives
// of the two pre-filters differ), so sums differ in rounding only.
#pragma once

#include "conv3p_device.hpp"

namespace conv3p {

constexpr int kFR = 16;                        // base buckets per voxel
constexpr int kFK = 96;                        // buckets per axis in a tile's table (6 voxels at the base resolution)
constexpr int kFEntries = 116;                 // >= kFK + (kFR + 2) + 1 window entries per axis
constexpr int kFHeader = 3 * kFEntries;        // u64 index of the header: {tmin.x, tmin.y}, {tmin.z, e}, {all valid}, pad
constexpr int kFTableU64 = kFHeader + 4;       // 352 x 8 B = 2816 B = 176 x 16 B
constexpr int kFMaxExt = 8;                    // taps per axis the look-up loops take
#ifndef CONV3P_DEV_FUSED_ABLATE
#define CONV3P_DEV_FUSED_ABLATE 0   // developer timing builds (wrong results unless 16): 1 no exact stage, 2 no pass 2, 4 empty masks, 8 no epilogue, 16 no compaction (correct)
#endif
constexpr int kFStream = 192;                  // entries of a wave's pair stream (drained 128 at a time, < 64 carried over)
constexpr int kFMaxE = 12;                     // coarsest table (buckets of 2^12 / 16 voxels); beyond: everything is a candidate

__device__ __forceinline__ int floor_i32(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return (x - x) + (y - y) + (z - z) == 0.0f;   // v - v is 0 for a finite v, NaN for Inf and NaN (no min / max: they drop NaNs)
}

// ---------------------------------------------------------------------------------------------------------------
// Tables.  One wave per tile.  bucket(v) = floor((float(v) - tmin) * inv16) >> e, tmin = the tile's own minimum over
// its finite points, e the smallest shift that brings every point of the tile into [0, kFK).  Window of k:
// buckets k .. k + ws - 1, ws = (16 >> e) + 2.  Entry j of an axis holds the window of k = j - ws, so that entry 0
// (k <= -ws) and entries >= kFK + ws (k >= kFK) are empty.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void tile_tables_kernel(const PointRec<T> *__restrict__ pts, int ntiles, float inv16,
                                                          unsigned long long *__restrict__ tables,
                                                          const uint32_t *__restrict__ version,   // [B] of the clouds' contents (prep)
                                                          uint32_t *__restrict__ tab_version,     // [B] ... the tables were built from
                                                          uint32_t *__restrict__ tab_ticket,
                                                          uint32_t *__restrict__ tab_inv,         // [B] ... and the bits of the inv16 they were built with
                                                          int force)
{
    __shared__ __attribute__((aligned(16))) unsigned long long tab[kWavesPerBlock][kFTableU64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * kWavesPerBlock + wave;
    const int b = blockIdx.y;
    if (tile >= ntiles) return;   // (wave-uniform; no workgroup barrier below)
    // The tables depend on the sorted records and on the voxel size (inv16: bucket index, coarsening shift, window size):
    // a cloud whose content has not changed since they were built FOR THIS VOXEL keeps them (round 5: a framework that
    // runs the model op by op launches this kernel eight times per step for one new batch; a cache may hold stencils of
    // several voxel sizes -- the mark covers both).  It is set by the LAST tile of the cloud to finish -- by then every
    // wave of the cloud has passed this test -- and read by later launches only.
    if (!force && tab_version[b] == version[b] && tab_inv[b] == __builtin_bit_cast(uint32_t, inv16)) return;
    const size_t t = (size_t)b * ntiles + tile;
    const PointRec<T> r = pts[t * kTile + lane];
    const float v[3] = {(float)r.x, (float)r.y, (float)r.z};
    const bool valid = r.idx >= 0 && finite3(v[0], v[1], v[2]);
    float tmin[3], ext = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        tmin[a] = wave_min(valid ? v[a] : __builtin_huge_valf());
        const float tmax = wave_max(valid ? v[a] : -__builtin_huge_valf());
        ext = fmaxf(ext, (tmax - tmin[a]) * inv16);   // >= every point's pre-shift bucket coordinate (same fp32 ops, monotone)
    }
    const uint64_t allv = __ballot(valid);
    int e = 0;
    if (allv == 0) {
        tmin[0] = tmin[1] = tmin[2] = 0.0f;
    } else if (!(ext < 1.0e30f)) {
        e = kFMaxE + 1;   // (cannot happen for finite points; the search then takes every valid point as a candidate)
    } else {
        while (e <= kFMaxE && floor_i32(ext) >> e > kFK - 1) ++e;
    }
    const int es = e > 31 ? 31 : e;
    const int ws = (kFR >> es) + 2;
    int beta[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int b16 = floor_i32((v[a] - tmin[a]) * inv16) >> es;
        b16 = b16 < 0 ? 0 : (b16 > kFK - 1 ? kFK - 1 : b16);
        beta[a] = b16;
    }
    // entry j holds the points with bucket in [j - ws, j - 1]: a point of bucket beta belongs to entries beta + 1 ..
    // beta + ws -- every lane ORs its bit into that run (LDS atomics; no ballot per entry)
    unsigned long long *mytab = tab[wave];
    {
        uint4 *z = reinterpret_cast<uint4 *>(mytab);
        for (int i = lane; i < kFTableU64 / 2; i += 64) z[i] = make_uint4(0, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    if (valid) {
        const unsigned long long bit = 1ull << lane;
        for (int i = 1; i <= ws; ++i) {
#pragma unroll
            for (int a = 0; a < 3; ++a) atomicOr(&mytab[a * kFEntries + beta[a] + i], bit);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        mytab[kFHeader + 0] = (uint64_t)__builtin_bit_cast(uint32_t, tmin[0]) | ((uint64_t)__builtin_bit_cast(uint32_t, tmin[1]) << 32);
        mytab[kFHeader + 1] = (uint64_t)__builtin_bit_cast(uint32_t, tmin[2]) | ((uint64_t)(uint32_t)e << 32);
        mytab[kFHeader + 2] = allv;
        mytab[kFHeader + 3] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the LDS writes above
    const uint4 *src = reinterpret_cast<const uint4 *>(mytab);
    uint4 *dst = reinterpret_cast<uint4 *>(tables + t * kFTableU64);
    for (int i = lane; i < kFTableU64 / 2; i += 64) dst[i] = src[i];
    if (!force && lane == 0) {
        const uint32_t done = atomicAdd(&tab_ticket[b], 1u);
        if (done + 1u == (uint32_t)ntiles) {
            tab_inv[b] = __builtin_bit_cast(uint32_t, inv16);
            tab_version[b] = version[b];
            tab_ticket[b] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T> struct FusedJob {
    Stencil<T> st;
    CacheCtl cc;
    int32_t *count, *tcount;
    PairEntry *pairs;
    uint2 *segs, *qsegs;
    uint32_t *qbm, *qbm_hi;   // backward taps of every centre's list: bit f' & 31 of plane f' >> 5 (qbm_hi: filters of 33 .. 64 taps)
    // lower edge of tap k's one-voxel acceptance interval along axis a, relative to the centre, in base buckets:
    // (k * step - full / 2) * 16  (host, double -> float)
    float clo[3][kFMaxExt];
};
constexpr int kFusedMaxJobs = 8;
template <typename T> struct FusedJobs {
    FusedJob<T> job[kFusedMaxJobs];
};

// LDS carve shared by host (size) and device (offsets)
struct FusedLds {
    size_t tapmap, cnt, cen, bmk, nqw, misc, red, tab, masks, stream, total;
};
__host__ __device__ inline size_t f_a16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline FusedLds fused_lds(int ntap, int maxfull, int elem, int M)
{
    FusedLds L;
    size_t off = 0;
    L.tapmap = off; off += f_a16((size_t)3 * maxfull * 2);
    L.cnt = off; off += f_a16((size_t)ntap * kCntStride * 4);
    L.cen = off; off += (size_t)64 * 12 * elem;          // per centre: lo[3], hi[3], p[3], 3 words of padding
    L.bmk = off; off += (ntap > 64 ? 256 : ntap > 32 ? 128 : 64) * 4;   // one 64-word plane of backward-tap sets per 32 taps
    L.nqw = off; off += (size_t)kWavesPerBlock * 64 * 4;
    L.misc = off; off += 16;
    L.red = off; off += 6 * 8;
    off = f_a16(off);
    L.tab = off; off += (size_t)kWavesPerBlock * kFTableU64 * 8;
    L.masks = off; off += (size_t)kWavesPerBlock * M * 64 * 8;
    L.stream = off; off += (size_t)kWavesPerBlock * 2 * kFStream * 4;
    L.total = off;
    return L;
}

// Candidate tiles (64 per ballot) whose box meets [ulo, uhi]
template <typename T>
__device__ __forceinline__ uint64_t tiles_meeting(const T *__restrict__ cloud_box, int ntiles, int base, const T *ulo, const T *uhi)
{
    const int t = base + (threadIdx.x & 63);
    bool ov = false;
    if (t < ntiles) {
        const T *bb = cloud_box + (size_t)t * 6;
        const T b0 = bb[0], b1 = bb[1], b2 = bb[2], b3 = bb[3], b4 = bb[4], b5 = bb[5];
        ov = !((b3 < ulo[0]) | (b0 > uhi[0]) | (b4 < ulo[1]) | (b1 > uhi[1]) | (b5 < ulo[2]) | (b2 > uhi[2]));
    }
    return __ballot(ov);
}

// Pre-filter hit mask for the lane's centre against the candidate tile staged in `tab` (general filter extents).
//   g[a] = (float(p) - tmin) * inv16 (the candidate side's own fp32 operations), gs[a] = g[a] - slack + (ws << e).
// Superset argument.  A candidate accepted by the exact test for tap k of axis a satisfies, in real numbers,
//   v - p in [(k step - full/2) voxel - eta, (k step + 1 - full/2) voxel + eta],  eta = a few ulps of the coordinates.
// Its bucket coordinate b16(v) = floor(fl(fl(v - tmin) inv16)) is monotone in v and within D = ~1.5 ulp(M) inv16 +
// 1.2e-7 |r| of the real r(v) = (v - tmin) 16 / voxel; g is within the same D of r(p).  With
//   slack >= 2 D + eta 16 / voxel  (here 1e-3 + 4e-6 (|p| inv16 + |g|) >= 1e-3 + 2e-6 ((|p| + |tmin|) inv16 + |g|): ~17 ulps)
// x = g + clo - slack <= r(lower edge) - D, hence floor(x) <= b16(v): no accepted candidate lies below the window.
// Above: b16(v) <= floor(r(lower edge) + 16 + fuzz) <= floor(x) + 16 + ceil(fuzz) with fuzz <= 2.5 slack <= 2^e / 2
// (enforced: a larger slack takes every valid candidate), and ((k16 + 16 + ceil(fuzz)) >> e) - (k16 >> e) <= ws - 1
// for ws = (16 >> e) + 2.  Invalid / non-finite centres get an empty mask, non-finite candidates are in no window.
template <typename T>
__device__ __forceinline__ uint64_t fused_mask(const FusedJob<T> &job, const unsigned long long *tab, const float *gs, int e)
{
    uint64_t H = ~0ull;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        uint64_t A = 0;
        const unsigned long long *ta = tab + a * kFEntries;
        for (int k = 0; k < job.st.ext[a]; ++k) {
            int j = floor_i32(gs[a] + job.clo[a][k]) >> e;
            j = j < 0 ? 0 : (j > kFEntries - 1 ? kFEntries - 1 : j);
            A |= ta[j];
        }
        H &= A;
    }
    return H;
}

// Taps of one pair without the six divisions where they cannot change the result.
//   forward (.cpp:280-290): t_f = (int)((v - lo) / voxel), lo = fl(p - half);  backward (.cpp:662-677): t_b = (int)((p - lx) / voxel),
//   lx = fl(v - half).  In real numbers, with a = (v - p) / voxel + full / 2: the forward quotient is a + e1 and the backward
//   one full - a + e2, |e| <= E = 1.2e-7 (max(|p|, |v|) / voxel + 2 full) (rounding of lo / lx, of the two subtractions, of
//   the divisions).  If a is further than E from every integer, floor(a + e1) = floor(a) and floor(full - a + e2) =
//   full - 1 - floor(a): the backward tap is the MIRROR of the forward one (no hole either: full - 1 = (ext - 1) step), its
//   index ntap - 1 - fwd -- and q = (v - lo) * (1 / voxel), within 1.8e-7 |q| of the quotient, truncates like it.
//   A pair within 5e-7 (max(|p|, |v|) / voxel + 2 full) of a tap boundary on any axis (every pair of voxel-aligned data)
//   takes the reference's own arithmetic for both taps.
template <typename T> struct TapFast {
    static constexpr bool enabled = false;
};
template <> struct TapFast<float> {
    static constexpr bool enabled = true;
};
__device__ __forceinline__ float min3_t(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ double min3_t(double a, double b, double c) { return fmin(fmin(a, b), c); }
__device__ __forceinline__ int clamp_tap(int t, int full)
{
    t = t > full - 1 ? full - 1 : t;
    return t < 0 ? 0 : t;
}

// Per-centre exact data of the dense stage: one 48-byte (fp32) record, three 16-byte LDS reads
template <typename T> struct __attribute__((aligned(16))) CenRec {
    T lo[3], hi[3], p[3];
    T pad[3];
};

// Exact stage of one (centre ql, candidate v) pair: the reference's arithmetic (search_tile's drain), same decisions.
// `on` = the lane holds a real pair.  Returns the forward / backward taps (kNoTap: not a neighbour / hole).
template <typename T>
__device__ __forceinline__ void fused_resolve(const Stencil<T> &st, T rvoxel, const int16_t *tapmap, uint32_t *cnt, uint32_t *bmk,
                                              const CenRec<T> *cen, uint32_t ql, const PointRec<T> &v, bool on, bool want_bwd,
                                              bool want_bm, uint32_t &fwd, uint32_t &bwd)
{
    const CenRec<T> c = cen[ql];
    // inclusive box test (.cpp:277): x - lo >= 0 exactly when x >= lo (IEEE subtraction never rounds across zero)
    const T d[3] = {v.x - c.lo[0], v.y - c.lo[1], v.z - c.lo[2]};
    const T dm = min3_t(d[0], d[1], d[2]);
    const T em = min3_t(c.hi[0] - v.x, c.hi[1] - v.y, c.hi[2] - v.z);
    const bool in = on && dm >= (T)0 && em >= (T)0;
    const T vv[3] = {v.x, v.y, v.z};
    fwd = kNoTap;
    bwd = kNoTap;
    bool exact = in;                      // lanes that need the reference's own arithmetic
    if (TapFast<T>::enabled) {
        int t[3];
        bool near = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float q = (float)d[a] * (float)rvoxel;
            const float mag = fmaxf(fabsf((float)c.p[a]), fabsf((float)vv[a]));
            near |= fabsf(q - __builtin_rintf(q)) <= __builtin_fmaf(mag, 5.0e-7f * (float)rvoxel, 1.0e-6f * (float)st.full[a]);
            t[a] = (int)q;
        }
        exact = in && near;
        if (in && !near) {
            const int tx = tapmap[clamp_tap(t[0], st.full[0])];
            const int ty = tapmap[st.maxfull + clamp_tap(t[1], st.full[1])];
            const int tz = tapmap[2 * st.maxfull + clamp_tap(t[2], st.full[2])];
            if ((tx | ty | tz) >= 0) {                                                   // .cpp:285
                fwd = (uint32_t)((tz * st.ext[1] + ty) * st.ext[0] + tx);                // .cpp:290
                bwd = (uint32_t)st.ntap - 1u - fwd;                                      // the mirrored tap, see above
            }
        }
    }
    if (exact) {
        const int tx = tapmap[clamp_tap((int)(d[0] / st.voxel), st.full[0])];            // .cpp:280-282
        const int ty = tapmap[st.maxfull + clamp_tap((int)(d[1] / st.voxel), st.full[1])];
        const int tz = tapmap[2 * st.maxfull + clamp_tap((int)(d[2] / st.voxel), st.full[2])];
        if ((tx | ty | tz) >= 0) {                                                       // .cpp:285
            fwd = (uint32_t)((tz * st.ext[1] + ty) * st.ext[0] + tx);                    // .cpp:290
            if (want_bwd) {
                // tap of the centre inside the candidate's box (.cpp:662-677)
                const T lx = (T)((double)v.x - st.half[0]), ly = (T)((double)v.y - st.half[1]), lz = (T)((double)v.z - st.half[2]);
                const int bx = tapmap[clamp_tap((int)((c.p[0] - lx) / st.voxel), st.full[0])];
                const int by = tapmap[st.maxfull + clamp_tap((int)((c.p[1] - ly) / st.voxel), st.full[1])];
                const int bz = tapmap[2 * st.maxfull + clamp_tap((int)((c.p[2] - lz) / st.voxel), st.full[2])];
                if ((bx | by | bz) >= 0) bwd = (uint32_t)((bz * st.ext[1] + by) * st.ext[0] + bx);   // .cpp:672, :677
            }
        }
    }
    if (fwd != kNoTap) {
        atomicAdd(&cnt[fwd * kCntStride + ql], 1u);
        if (want_bm && bwd != kNoTap) atomicOr(&bmk[ql + ((bwd >> 5) << 6)], 1u << (bwd & 31u));
    }
}

typedef int int4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// Which stencils' lists are squeezed after pass 2 (see the compaction stage of search_fused_kernel): the dilated ones.
// A window of 16 + 2 buckets per one-voxel interval accepts ~(18/16)^3 = 1.42 x the volume: a third of a dilated
// stencil's hits are false positives, but only a ninth of an undilated one's (its three intervals per axis merge into
// one of 48 + 2 buckets: 12 % on the ModelNet-like clouds, 9 % on the rooms) -- there the pass costs more than the
// walkers' shorter lists give back.
// Undilated stencils are squeezed where the lists are long (more than 48 hits per centre on average over the tile: the
// rooms, whose stride-1 lists serve two layers of the segmentation model; cfg4 step 1.380 -> 1.358 ms) -- the pass costs
// per tile, the walkers save per record.  Decided per tile from its own hit count: the same for any batch it sits in.
template <typename T> __device__ __forceinline__ bool fused_compacts(const Stencil<T> &st, uint32_t tile_hits)
{
    return !(CONV3P_DEV_FUSED_ABLATE & 16) && (st.step[0] * st.step[1] * st.step[2] > 1 || tile_hits > 48u * 64u);
}

// EXT3: the stencil has 3 taps per axis (all layers of the reference's models).  Tap 1's acceptance interval is then
// centre -+ half a voxel whatever the stride and taps 0 / 2 sit 16 * step buckets below / above it: no per-tap
// constants, the nine look-ups of a candidate tile in flight together.  Otherwise: the general loop over FusedJob::clo.
template <typename T, bool EXT3>
__global__ __launch_bounds__(256) void search_fused_kernel(const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes,
                                                           const unsigned long long *__restrict__ tables, int N, int ntiles,
                                                           int ngroups, BlockMap bm, FusedJobs<T> jobs, int M, float inv16)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b, qt;
    if (!block_to_cloud(bm, b, qt)) return;   // uniform for the workgroup
    const FusedJob<T> &job = jobs.job[blockIdx.y];
    const bool have_pairs = job.pairs != nullptr;   // nullptr: populations only (the public neighbour-count entry point)
    if (have_pairs && slot_valid(job.cc, b)) return;   // this cloud's lists are current (uniform)
    const Stencil<T> &st = job.st;

    const FusedLds L = fused_lds(st.ntap, st.maxfull, (int)sizeof(T), M);
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem + L.tapmap);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + L.cnt);
    CenRec<T> *cen = reinterpret_cast<CenRec<T> *>(smem + L.cen);
    uint32_t *bmk = reinterpret_cast<uint32_t *>(smem + L.bmk);
    uint32_t *nqw = reinterpret_cast<uint32_t *>(smem + L.nqw);
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + L.misc);
    T *red = reinterpret_cast<T *>(smem + L.red);
    unsigned long long *tab = reinterpret_cast<unsigned long long *>(smem + L.tab) + (size_t)wave * kFTableU64;
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(smem + L.masks) + (size_t)wave * M * 64;

    const PointRec<T> *cloud_pts = pts + (size_t)b * ntiles * kTile;
    const T *cloud_box = boxes + (size_t)b * ntiles * 6;
    const unsigned long long *cloud_tab = tables + (size_t)b * ntiles * kFTableU64;
    const PointRec<T> me = cloud_pts[(size_t)qt * kTile + lane];
    const float pf[3] = {(float)me.x, (float)me.y, (float)me.z};
    const bool qvalid = me.idx >= 0 && finite3(pf[0], pf[1], pf[2]);
    const bool want_bm = job.qbm != nullptr && (st.ntap <= 32 || (st.ntap <= 128 && job.qbm_hi != nullptr));

    // ---- prologue: tap table, zeroed populations, the centres' exact data, extreme centres per axis
    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    for (int e = threadIdx.x; e < st.ntap * kCntStride; e += blockDim.x) cnt[e] = 0;
    if (wave == 0) {
        CenRec<T> r;
        const T p[3] = {me.x, me.y, me.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            r.lo[a] = (T)((double)p[a] - st.half[a]);     // .cpp:240-245
            r.hi[a] = (T)((double)p[a] + st.half[a]);
            r.p[a] = p[a];
            r.pad[a] = (T)0;
        }
        cen[lane] = r;
        bmk[lane] = 0;
        if (st.ntap > 32) bmk[64 + lane] = 0;
        if (st.ntap > 64) bmk[128 + lane] = bmk[192 + lane] = 0;
    } else {
        const int a = wave - 1;
        const T pa = a == 0 ? me.x : (a == 1 ? me.y : me.z);
        const T mn = wave_min(qvalid ? pa : Limits<T>::inf()), mx = wave_max(qvalid ? pa : -Limits<T>::inf());
        if (lane == 0) {
            red[a] = mn;
            red[3 + a] = mx;
        }
    }
    const float q0[3] = {1.0e-3f + 4.0e-6f * (fabsf(pf[0]) * inv16), 1.0e-3f + 4.0e-6f * (fabsf(pf[1]) * inv16),
                         1.0e-3f + 4.0e-6f * (fabsf(pf[2]) * inv16)};   // per-lane part of the look-up slack (see fused_mask)
    __syncthreads();
    // union of the 64 filter boxes: rounding is monotone, so it is the box of the extreme centres
    T ulo[3], uhi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ulo[a] = (T)((double)red[a] - st.half[a]);
        uhi[a] = (T)((double)red[3 + a] + st.half[a]);
    }
    const float s16[3] = {(float)(kFR * st.step[0]), (float)(kFR * st.step[1]), (float)(kFR * st.step[2])};

    // The wave's candidate tiles: tile ct belongs to wave ct % 4.  visit(ct, next ct of this wave in the block or -1)
    auto for_my_tiles = [&](auto &&visit) {
        for (int base = 0; base < ntiles; base += 64) {
            uint64_t todo = tiles_meeting(cloud_box, ntiles, base, ulo, uhi) & (0x1111111111111111ull << wave);
            while (todo) {
                const int i = __builtin_ctzll(todo);
                todo &= todo - 1;
                visit(base + i, todo ? base + __builtin_ctzll(todo) : -1);
            }
        }
    };
    // Hit mask of the lane's centre for candidate tile ct.  Stages the tile's table in the wave's LDS slot; the table of
    // the wave's next candidate tile is requested before this one is used.
    uint4 pf0 = make_uint4(0, 0, 0, 0), pf1 = pf0, pf2 = pf0;
    int pf_ct = -1;
    auto request = [&](int ct) {
        const uint4 *src = reinterpret_cast<const uint4 *>(cloud_tab + (size_t)ct * kFTableU64);
        pf0 = src[lane];
        pf1 = src[64 + lane];
        pf2 = src[128 + (lane < kFTableU64 / 2 - 128 ? lane : 0)];
        pf_ct = ct;
    };
    auto compute_mask = [&](int ct, int nxt) -> uint64_t {
        if (pf_ct != ct) request(ct);
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        __builtin_amdgcn_wave_barrier();   // (same wave: earlier readers of the slot are done in program order)
        dst[lane] = pf0;
        dst[64 + lane] = pf1;
        if (lane < kFTableU64 / 2 - 128) dst[128 + lane] = pf2;
        __builtin_amdgcn_wave_barrier();
        if (nxt >= 0) request(nxt);
        if (CONV3P_DEV_FUSED_ABLATE & 4) return 0ull;
        const uint64_t h0 = tab[kFHeader], h1 = tab[kFHeader + 1];
        const float tmin[3] = {__builtin_bit_cast(float, (uint32_t)h0), __builtin_bit_cast(float, (uint32_t)(h0 >> 32)),
                               __builtin_bit_cast(float, (uint32_t)h1)};
        const int e = (int)(uint32_t)(h1 >> 32);
        const int es = e > kFMaxE ? kFMaxE : e;
        const int ws = (kFR >> es) + 2;
        float gs[3], smax = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float g = (pf[a] - tmin[a]) * inv16;
            const float slack = __builtin_fmaf(4.0e-6f, fabsf(g), q0[a]);
            smax = fmaxf(smax, slack);
            gs[a] = (g - slack) + (float)(ws << es);
        }
        // the fuzz must stay below half a (scaled) bucket; coarser tables than 2^kFMaxE: everything is a candidate
        const bool wide = e > kFMaxE || __any(qvalid && !(smax <= 0.2f * (float)(1 << es)));
        auto entry = [&](float x) {
            int j = floor_i32(x) >> es;
            return j < 0 ? 0 : (j > kFEntries - 1 ? kFEntries - 1 : j);
        };
        uint64_t h;
        if (wide) {
            h = tab[kFHeader + 2];
        } else if (EXT3) {
            uint64_t A[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float x1 = gs[a] - (float)(kFR / 2);
                const unsigned long long *ta = tab + a * kFEntries;
                const uint64_t r0 = ta[entry(x1 - s16[a])], r1 = ta[entry(x1)], r2 = ta[entry(x1 + s16[a])];
                A[a] = r0 | r1 | r2;
            }
            h = A[0] & A[1] & A[2];
        } else {
            h = fused_mask(job, tab, gs, es);
        }
        return qvalid ? h : 0ull;
    };

    // ---- pass 1: masks and per-centre hit counts
    {
        uint32_t mine = 0;
        int ord = 0;
        for_my_tiles([&](int ct, int nxt) {
            const uint64_t m = compute_mask(ct, nxt);
            mine += (uint32_t)__popcll(m);
            if (ord < M) masks[(size_t)ord * 64 + lane] = m;
            ++ord;
        });
        nqw[wave * 64 + lane] = mine;
    }
    __syncthreads();

    // ---- P2: one reservation for the query tile, centre-major slots inside it
    if (wave == 0) {
        const uint32_t n0 = nqw[lane], n1 = nqw[64 + lane], n2 = nqw[128 + lane], n3 = nqw[192 + lane];
        const uint32_t nq = n0 + n1 + n2 + n3;
        int Ltot;
        const uint32_t offq = (uint32_t)wave_excl_scan((int)nq, Ltot);
        const uint32_t cap = job.cc.pairs_per_cloud;
        uint32_t base = 0, ok = 0;
        if (!have_pairs) {
            if (lane == 0) misc[0] = misc[1] = misc[2] = 0;
        } else {
        if (lane == 0) {
            // (see search_tile P2: a reservation that does not fit is taken back at once, the cursor cannot wrap)
            base = Ltot ? atomicAdd(&job.cc.cursor[b], (uint32_t)Ltot) : 0u;
            ok = (base <= cap && (uint32_t)Ltot <= cap - base) ? 1u : 0u;
            if (!ok) atomicSub(&job.cc.cursor[b], (uint32_t)Ltot);
            base += (uint32_t)b * cap;
            job.segs[((size_t)b * ntiles + qt) * ngroups] = ok ? make_uint2(base, (uint32_t)Ltot) : make_uint2(0u, kSegOverflow);
            for (int g = 1; g < ngroups; ++g) job.segs[((size_t)b * ntiles + qt) * ngroups + g] = make_uint2(0u, 0u);
            misc[0] = base;
            misc[1] = ok;
            misc[2] = (uint32_t)Ltot;
        }
        base = __shfl(base, 0);
        ok = __shfl(ok, 0);
        // (a tile whose reservation held gets its per-centre segments after pass 2, with the false positives squeezed out)
        if (!ok || !fused_compacts(st, (uint32_t)Ltot))
            job.qsegs[((size_t)b * ntiles + qt) * ngroups * 64 + lane] = ok ? make_uint2(base + offq, nq) : make_uint2(0u, kSegOverflow);
        for (int g = 1; g < ngroups; ++g) job.qsegs[(((size_t)b * ntiles + qt) * ngroups + g) * 64 + lane] = make_uint2(0u, 0u);
        }
        nqw[lane] = offq;
        nqw[64 + lane] = offq + n0;
        nqw[128 + lane] = offq + n0 + n1;
        nqw[192 + lane] = offq + n0 + n1 + n2;
    }
    __syncthreads();
    const bool ok = misc[1] != 0;
    const uint32_t gbase = misc[0];
    char *pbase = reinterpret_cast<char *>(job.pairs + gbase);   // this query tile's slots (uniform)
    const T rvoxel = (T)1 / st.voxel;

    // ---- pass 2: every wave turns its masks into a dense stream of (centre, candidate) pairs -- the lanes (= centres)
    //      pop one hit each per round, ranked by ballot into the wave's LDS stream together with the slot the hit owns
    //      (centre-major, running per lane) -- and resolves the stream 128 pairs at a time with all lanes busy: the
    //      reference's arithmetic decides (fused_resolve), one final PairEntry per pre-filter hit (false positives are
    //      stored as kNoTap entries).  Nothing leaves the wave in between: no barrier, no round trip through memory.
    if (!(CONV3P_DEV_FUSED_ABLATE & 2)) {
        uint32_t *sdesc = reinterpret_cast<uint32_t *>(smem + L.stream) + wave * (2 * kFStream);
        uint32_t *sslot = sdesc + kFStream;
        uint32_t off = nqw[wave * 64 + lane] * 8u;   // byte offset of the lane's next slot
        int have = 0, ord = 0;
        pf_ct = -1;
        // pairs [first, first + 64 * kU) of the stream, n of them real; kU record gathers in flight per lane
        auto drain = [&](int n) {
            if (CONV3P_DEV_FUSED_ABLATE & 1) return;
            constexpr int kU = 2;
            uint32_t d[kU], so[kU];
            PointRec<T> v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int i = u * 64 + lane;
                d[u] = sdesc[i < n ? i : 0];
                so[u] = sslot[i < n ? i : 0];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) v[u] = cloud_pts[(size_t)((d[u] >> 12) * kTile + (d[u] & 63u))];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const bool on = u * 64 + lane < n;
                const uint32_t ql = (d[u] >> 6) & 63u;
                uint32_t fwd, bwd;
                fused_resolve<T>(st, rvoxel, tapmap, cnt, bmk, cen, ql, v[u], on, true, want_bm, fwd, bwd);
                if (on && ok) {   // (pair buffer full: the consumers search this tile themselves; the populations are still owed)
                    PairEntry pe;
                    pe.cand = (uint32_t)v[u].idx;
                    pe.code = pair_code(fwd, bwd, ql);
                    *reinterpret_cast<PairEntry *>(pbase + so[u]) = pe;
                }
            }
        };
        for_my_tiles([&](int ct, int nxt) {
            uint64_t mm = ord < M ? masks[(size_t)ord * 64 + lane] : compute_mask(ct, nxt);
            ++ord;
            const uint32_t dbase = ((uint32_t)ct << 12) | ((uint32_t)lane << 6);
            while (true) {
                const uint64_t act = __ballot(mm != 0);
                if (act == 0) break;
                if (mm != 0) {
                    const uint32_t c = (uint32_t)__builtin_ctzll(mm);
                    mm &= mm - 1;
                    const int at = have + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0));
                    sdesc[at] = dbase | c;
                    sslot[at] = off;
                    off += 8u;
                }
                have += __popcll(act);
                __builtin_amdgcn_wave_barrier();
                if (have >= 128) {
                    drain(128);
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t cd = sdesc[128 + lane], cs = sslot[128 + lane];   // (have - 128 <= 63 entries carry over)
                    __builtin_amdgcn_wave_barrier();
                    sdesc[lane] = cd;
                    sslot[lane] = cs;
                    have -= 128;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        });
        if (have > 0) drain(have);
    }
    // (the compaction below lets wave w read and rewrite records the OTHER waves stored to global memory in pass 2: their
    // stores must have left the waves before the barrier -- a workgroup-scope release; all four waves share the CU's one
    // vector L1, the default non-tgsplit mode, so nothing more is needed for them to be read back)
    __threadfence_block();
    __syncthreads();

    // ---- compaction (round 5).  The tables are a superset filter: on the models' dilated stencils a third of the
    //      pre-filter hits are false positives (window of 18 buckets where 16 decide, on three axes), stored above as
    //      kNoTap records that every list-walking kernel then steps over -- a third of the forward's and backward's steps.
    //      Here every centre's list is squeezed to its live records (forward tap present), order kept, in place: wave w
    //      takes the contiguous run of the centres 16w .. 16w + 15, lane = record, 64 at a time; a live record's place
    //      in its centre's list is the number of live records of the same centre before it (one ballot; the centre's
    //      first lane in the chunk from its slot offset; a count carried over when a list straddles two chunks).  A
    //      record only ever moves DOWN inside its own centre's slots, so nothing unread is overwritten; the slots past
    //      the live records are rewritten as kNoTap records (tile-level readers of `segs` see no duplicates).  The live
    //      count of a centre is the sum of its populations, already in LDS.
    if (have_pairs && ok && fused_compacts(st, misc[2])) {
        uint32_t *livec = nqw + 64;   // [64]; entry c is written and read by the wave that owns centre c only
        {
            const int c = wave * 16 + (lane & 15);
            uint32_t sacc = 0;
            for (int f = lane >> 4; f < st.ntap; f += 4) sacc += cnt[f * kCntStride + c];
            sacc += lane_xor16(sacc);
            sacc += lane_xor32(sacc);
            if (lane < 16) livec[c] = sacc;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t r0 = nqw[wave * 16], r1 = wave == kWavesPerBlock - 1 ? misc[2] : nqw[wave * 16 + 16];
        PairEntry *tp = job.pairs + gbase;
        const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        uint32_t carry = 0;
        // kCB chunks of 64 records per batch, the next batch requested before this one is worked on: the run of a wave
        // (~530 records on the cfg2 clouds) costs two or three memory round trips instead of one per chunk (first
        // version, one chunk ahead: +22 us on the 98-us search of the cfg2 step)
        constexpr int kCB = 4;
        PairEntry cur[kCB], nxt[kCB];
        auto request = [&](PairEntry (&dst)[kCB], uint32_t pos) {
#pragma unroll
            for (int u = 0; u < kCB; ++u) {
                const uint32_t i = pos + (uint32_t)(u * 64 + lane);
                dst[u] = tp[i < r1 ? i : r0];
            }
        };
        if (r0 < r1) request(nxt, r0);   // (uniform)
        for (uint32_t pos = r0; pos < r1; pos += 64 * kCB) {
#pragma unroll
            for (int u = 0; u < kCB; ++u) cur[u] = nxt[u];
            if (pos + 64 * kCB < r1) request(nxt, pos + 64 * kCB);   // (its slots lie above everything this batch stores to)
#pragma unroll
            for (int u = 0; u < kCB; ++u) {
                const uint32_t cpos = pos + (uint32_t)(u * 64);
                if (cpos >= r1) break;   // (uniform)
                const uint32_t i = cpos + (uint32_t)lane;
                const bool valid = i < r1;
                const uint32_t c = code_q(cur[u].code) & 63u;
                const bool live = valid && code_fwd(cur[u].code) != kNoTap;
                const uint32_t oc = nqw[c], lv = livec[c];
                const uint64_t Lm = __ballot(live);
                const uint32_t sc = oc > cpos ? oc - cpos : 0u;   // first lane of centre c in this chunk (<= lane)
                const uint64_t before = Lm & lt & ~((1ull << (sc & 63u)) - 1ull);
                const uint32_t rank = (uint32_t)__popcll(before) + (oc < cpos ? carry : 0u);
                if (live && oc + rank != i) tp[oc + rank] = cur[u];
                if (valid && i - oc >= lv) {
                    PairEntry dead;
                    dead.cand = cur[u].cand;
                    dead.code = pair_code(kNoTap, kNoTap, c);
                    tp[i] = dead;
                }
                carry = (uint32_t)__builtin_amdgcn_readlane((int)(rank + (live ? 1u : 0u)), 63);
            }
        }
        if (lane < 16) {
            const int c = wave * 16 + lane;
            job.qsegs[((size_t)b * ntiles + qt) * ngroups * 64 + c] = make_uint2(gbase + nqw[c], livec[c]);
        }
    }

    // ---- epilogue: populations in both layouts, backward-tap sets, slot commit.  Thread (wave w, lane q) serves
    //      centre q (its record is still in the thread's registers).
    if (!(CONV3P_DEV_FUSED_ABLATE & 8)) {
        if (job.count != nullptr && me.idx >= 0) {
            // row of centre q by ORIGINAL index, four taps (16 bytes) per store; the waves take every fourth piece
            int32_t *row = job.count + ((size_t)b * N + me.idx) * st.ntap;
            const int pieces = st.ntap >> 2;
            for (int pc = wave; pc < pieces; pc += kWavesPerBlock) {
                const uint32_t *src = cnt + (size_t)(pc * 4) * kCntStride + lane;
                int4_a4 w;
                w.x = (int)src[0];
                w.y = (int)src[kCntStride];
                w.z = (int)src[2 * kCntStride];
                w.w = (int)src[3 * kCntStride];
                *reinterpret_cast<int4_a4 *>(row + pc * 4) = w;
            }
            if (wave == (pieces & 3))
                for (int f = pieces * 4; f < st.ntap; ++f) row[f] = (int32_t)cnt[f * kCntStride + lane];
        }
        if (job.tcount != nullptr) {
            // the same populations tile-major, [tap][centre lane]: the forward kernel's table of reciprocals
            int32_t *tc = job.tcount + ((size_t)b * ntiles + qt) * st.ntap * kTile;
            for (int f = wave; f < st.ntap; f += kWavesPerBlock) tc[f * kTile + lane] = (int32_t)cnt[f * kCntStride + lane];
        }
        if (want_bm && wave == 0) {
            job.qbm[((size_t)b * ntiles + qt) * 64 + lane] = bmk[lane];
            if (st.ntap > 32) job.qbm_hi[((size_t)b * ntiles + qt) * 64 + lane] = bmk[64 + lane];
            if (st.ntap > 64) {   // planes 2 and 3 (65 .. 128 taps) follow plane 1 at the planes' common stride
                const size_t pstride = (size_t)(job.qbm_hi - job.qbm);
                job.qbm_hi[pstride + ((size_t)b * ntiles + qt) * 64 + lane] = bmk[128 + lane];
                job.qbm_hi[2 * pstride + ((size_t)b * ntiles + qt) * 64 + lane] = bmk[192 + lane];
            }
        }
    }
    // commit: the last query tile of the cloud to finish marks the slot's lists as built from the current content
    __syncthreads();
    if (threadIdx.x == 0 && have_pairs) {
        const CacheCtl &cc = job.cc;
        const uint32_t t = atomicAdd(&cc.ticket[b], 1u);
        if (t + 1u == (uint32_t)bm.blocks_per_cloud) {
            cc.built_version[b] = cc.version[b];
            cc.built_tag[b] = cc.tag;
            cc.ticket[b] = 0;
        }
    }
}

}  // namespace conv3p
