// conv3p_deep.hpp -- deep-channel path (e.g. 128 -> 256, BASELINE config 5) on the matrix cores.
//
// For wide layers the per-pair form (Cin*Cout FMAs per neighbour pair) is wasteful; the op factorises as
//     out[i,:]  = sum_f  M_f[i,:]  . W[f]          M_f[i,:] = sum_{j in tap f of i} x[j,:] / count[i,f]
//     dX[j,:]   = sum_f' G_f'[j,:] . W[f']^T       G_f'[j,:] = sum_{ii: bwd tap f'} dY[ii,:] / count[ii,f']
//     dW[f']    = sum_pairs(f')  x[j,:]^T . (dY[ii,:] / count[ii,f'])
// (same sums as tf_conv3p_atrous.cpp:480-494 and :682-698, re-associated; inside the fp32 tolerance).
// The contractions with the filter run on v_mfma_f32_32x32x2_f32 (exact fp32, an fmaf chain per output,
// 64 FLOP/clk/SIMD); the per-centre reductions M_f / G_f' are segmented sums on the vector ALUs:
//   * a tile's records are put in tap-major order once (deep_order_kernel, a stable counting sort);
//   * per tap, M_f is a branch-free segmented sum over the tap's run of records (deep_gemm_kernel stage 1);
//   * M_f . W[f] follows from LDS, the W[f] operand streamed from L2 sixteen k-rows ahead;
//   * dW[f'] = sum over tiles of X_tile^T . G_f'[tile]: the grad_input kernel leaves every G_f' tile ([64][Cout], the
//     per-centre reduction it computes anyway) in a scratch buffer, and the grad_filter kernel is a plain
//     [Cin x 64] . [64 x Cout] product per (tile, tap) -- 2*27*Cin*Cout flops per point, the dense-equivalent count
//     (the earlier pair-indexed form issued 3.6x that).
// Taps with no neighbour in the tile are skipped.
// Channel counts: the kernels are instantiated for KDIM, NDIM in {32, 64, 128} (+ the 128 -> 256 layer) and take the
// REAL row lengths at run time: rows are read / written with their real length and stride, columns past the real
// count are zero in LDS, and the filter operand is a zero-padded copy [F][KDIM][NDIM] (pad_filter_kernel).  Any
// fp32 layer with up to 128 channels on either side therefore runs here, deterministically.
//
// Fragment layouts of mfma_f32_32x32x2f32 (cdna_hip_programming.md section 3):
//   A operand: lane l holds A[i = l & 31][k = l >> 5];  B operand: lane l holds B[k = l >> 5][j = l & 31];
//   C/D: register r of lane l is C[row = (r & 3) + 8*(r >> 2) + 4*(l >> 5)][col = l & 31].
#pragma once

#include "conv3p_dev.hpp"
#include "conv3p_device.hpp"
#include <type_traits>

#ifndef DEEP_GEMM_WAVES
#define DEEP_GEMM_WAVES 2
#endif
#ifndef DEEP_SKIP_STORES
#define DEEP_SKIP_STORES 1
#endif
#ifndef DEEP_DW_ROWS
#define DEEP_DW_ROWS 32   // packed rows of a tile that deep_dw_kernel holds in LDS at a time (64: all)
#endif
#ifndef DEEP_DW_WAVES
#define DEEP_DW_WAVES (DEEP_DW_ROWS == 64 ? 2 : 3)
#endif
#ifndef DEEP_KU
#define DEEP_KU 4   // records per pipeline step of deep_gemm_kernel's stage 1
#endif
#ifndef DEEP_KD
#define DEEP_KD 4   // ... and pipeline slots
#endif

namespace conv3p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 4 consecutive floats of a row of `len` values starting at column `col` (zeros past the end); rows are only
// dword-aligned in general (global_load_dwordx4 needs no more on gfx950)
__device__ __forceinline__ float4 load_row4(const float *__restrict__ row, int col, int len)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col + 4 <= len) {
        const float4_a4 t = *reinterpret_cast<const float4_a4 *>(row + col);
        v = make_float4(t.x, t.y, t.z, t.w);
    } else if (col < len) {
        v.x = row[col];
        if (col + 1 < len) v.y = row[col + 1];
        if (col + 2 < len) v.z = row[col + 2];
    }
    return v;
}

constexpr int kDeepBlk = 32;      // records per staged block (= 16 MFMA k-steps)

// ---------------------------------------------------------------------------------------------
// deep_order_kernel: tap-major order of every query tile's records (stable counting sort by tap: taps ascending,
// inside a tap the search's record order -> deterministic).  One workgroup per tile.
//   tap_off[tile][f] .. tap_off[tile][f+1] : slots of tap f (relative to the tile's segment)
//   tap_meta[segment start + slot]         : {neighbour's original index | centre lane << 24, 1 / population} of the
//                                            record in that slot -- everything the GEMM kernels need of a record, so
//                                            they fetch it with ONE load (no order -> record -> population chain)
// BWD: taps = backward taps, population = that of the neighbour's tap (one 4-byte gather); holes and records whose
// population is 0 (.cpp:679) dropped.  Else forward taps, population of the centre's tap; false positives dropped.
// Tiles whose pair reservation overflowed get an empty order and tile_flag = 1 (the generic kernel takes them).
// ---------------------------------------------------------------------------------------------
constexpr int kOrderR = 16;   // records per thread and round of deep_order_kernel (4096 per round)

// Rank of this lane's record among the records of the SAME tap in its wave-chunk (the 64 records the wave holds in
// one register), the chunk's number of records of that tap, and whether this lane is the first of them.  The lanes
// with the same key are found bit by bit (ntap <= 64: six ballots) instead of one ballot per tap.
__device__ __forceinline__ uint32_t chunk_rank(uint32_t key, uint32_t &cnt, bool &leader)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t eq = __ballot(key != 0xFFFFFFFFu);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const bool bit = (key >> b) & 1u;
        const uint64_t m = __ballot(bit);
        eq &= bit ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(eq & ((1ull << lane) - 1ull));
    cnt = (uint32_t)__popcll(eq);
    leader = key != 0xFFFFFFFFu && rank == 0;
    return rank;
}

template <bool BWD>
__global__ __launch_bounds__(256) void deep_order_kernel(const PointRec<float> *__restrict__ pts,
                                                         const int32_t *__restrict__ count,
                                                         const PairEntry *__restrict__ pairs,
                                                         const uint2 *__restrict__ segs, int N, int ntiles, int ntap,
                                                         int ngroups,                        // segments per tile (search groups)
                                                         const uint2 *__restrict__ qsegs,    // per (tile, group, centre) sub-lists (ngroups > 1 only)
                                                         uint2 *__restrict__ dsegs,          // out [tile]: {first slot of the tile in tap_meta, records} or kSegOverflow
                                                         uint32_t *__restrict__ meta_cursor, // [clouds] slot allocator of tap_meta (ngroups > 1 only; zeroed by the host)
                                                         uint32_t pairs_per_cloud,
                                                         uint2 *__restrict__ tap_meta,
                                                         uint32_t *__restrict__ tap_off,
                                                         uint8_t *__restrict__ tile_flag,
                                                         uint32_t *__restrict__ tap_total,   // [ntap] += (may be null)
                                                         unsigned long long *__restrict__ pop_mask,   // [tile] (may be null)
                                                         uint4 *__restrict__ tap_split,   // [tile][ntap] quarter points of every run
                                                         unsigned long long *__restrict__ tap_cmask)   // [tile][ntap] centres with records of the tap (may be null)
{
    constexpr int R = kOrderR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *tot = reinterpret_cast<uint32_t *>(smem);            // [ntap]        records per tap (whole tile)
    uint32_t *base = tot + ntap;                                    // [ntap]        next free slot of each tap
    uint32_t *wcnt = base + ntap;                                   // [ntap][4 * R] per wave-chunk counts of a round
    int32_t *qorig = reinterpret_cast<int32_t *>(wcnt + 4 * R * ntap);   // [64] original indices of the tile's centres
    uint32_t *hist = reinterpret_cast<uint32_t *>(qorig + 64);     // [ntap][64] records per (tap, centre)
    const size_t tile = blockIdx.x;
    const int32_t *cnt_cloud = count + (tile / (size_t)ntiles) * (size_t)N * ntap;
    if (threadIdx.x < 64) qorig[threadIdx.x] = pts[tile * kTile + threadIdx.x].idx;
    // The tile's records.  One search group (N <= 8192): its single segment of `pairs`, centre-major.  Several groups
    // (one segment per 128 candidate tiles, each centre-major on its own): read as ONE centre-major list -- centre 0's
    // sub-lists of all groups, then centre 1's, ... -- through a table of the 64 x ngroups sub-lists (first virtual
    // index, address); the tap-major order written below is then centre-major inside every tap like the single-group
    // one, which the quarter points and deep_gemm's segmented sums rely on.  Their slots in tap_meta come from a
    // per-cloud allocator (the groups' own segments are not adjacent); the placement does not enter any result.
    uint32_t *vstart = hist + (size_t)ntap * 64;                    // [64 * ngroups + 1]   (ngroups > 1)
    uint32_t *vaddr = vstart + 64 * ngroups + 1;                    // [64 * ngroups]
    uint32_t *wsum = vaddr + 64 * ngroups;                          // [4] + [1] slot base + [1] overflow
    uint32_t *toff = tap_off + tile * (size_t)(ntap + 1);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint2 tseg;
    const int nsub = 64 * ngroups;
    if (ngroups == 1) {
        tseg = segs[tile];
    } else {
        bool ovf = false;
        for (int g = 0; g < ngroups; ++g) ovf |= segs[tile * ngroups + g].y == kSegOverflow;
        const int per = (nsub + 255) / 256;                         // consecutive sub-lists per thread
        uint32_t local = 0;
        for (int e = 0; e < per; ++e) {
            const int sidx = (int)tid * per + e;                    // sub-list (centre q, group g) = (sidx / ngroups, sidx % ngroups)
            if (sidx < nsub && !ovf) {
                const uint2 qs = qsegs[((size_t)tile * ngroups + (sidx % ngroups)) * 64 + (sidx / ngroups)];
                vaddr[sidx] = qs.x;
                vstart[sidx + 1] = qs.y;                            // (lengths for now)
                local += qs.y;
            }
        }
        int wtot;
        const int excl = wave_excl_scan((int)local, wtot);
        if (lane == 63) wsum[wave] = (uint32_t)wtot;
        __syncthreads();
        uint32_t before = (uint32_t)excl;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();                                            // (wsum is reused below)
        for (int e = 0; e < per; ++e) {
            const int sidx = (int)tid * per + e;
            if (sidx < nsub && !ovf) {
                const uint32_t len = vstart[sidx + 1];
                vstart[sidx + 1] = before + len;                    // inclusive end = next start (own entries only: no race)
                before += len;
            }
        }
        if (tid == 0) {
            vstart[0] = 0;
            wsum[4] = ovf ? 0u : (uint32_t)(tile / (size_t)ntiles) * pairs_per_cloud + atomicAdd(&meta_cursor[tile / (size_t)ntiles], total);
        }
        __syncthreads();
        tseg = make_uint2(wsum[4], ovf ? kSegOverflow : total);
    }
    if (tid == 0) dsegs[tile] = tseg;
    if (tseg.y == kSegOverflow) {
        for (uint32_t f = tid; f <= (uint32_t)ntap; f += 256) toff[f] = 0;
        if (tid == 0) {
            tile_flag[tile] = 1;
            if (pop_mask != nullptr) pop_mask[tile] = 0ull;
        }
        if (tap_cmask != nullptr)
            for (uint32_t f = tid; f < (uint32_t)ntap; f += 256) tap_cmask[tile * (size_t)ntap + f] = 0ull;
        return;
    }
    if (tid == 0) tile_flag[tile] = 0;
    const PairEntry *seg = pairs + tseg.x;                          // (single group)
    const uint32_t n = tseg.y;
    // record i of the tile's centre-major list
    auto record = [&](uint32_t i) -> PairEntry {
        if (ngroups == 1) return seg[i];
        int lo = 0, hi = nsub;                                      // last sub-list whose first index is <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (vstart[mid] <= i) lo = mid; else hi = mid;
        }
        return pairs[vaddr[lo] + (i - vstart[lo])];
    };
    auto zero_wcnt = [&]() {
        for (uint32_t e = tid; e < (uint32_t)(4 * R * ntap); e += 256) wcnt[e] = 0;
    };
    for (uint32_t f = tid; f < (uint32_t)ntap; f += 256) tot[f] = 0;
    for (uint32_t e = tid; e < (uint32_t)ntap * 64u; e += 256) hist[e] = 0;
    zero_wcnt();
    __syncthreads();                                                // qorig, tot, wcnt, hist
    // One round = 256 x R records, R per thread (record s0 + u * 256 + tid: every load of a step is coalesced), all
    // R record loads issued together, then all R population gathers: two memory latencies per round.  Every load is
    // unconditional from a clamped index (see deep_gemm_kernel) and nothing is selected on the gathered value itself.
    //   key[u]  tap of the record (forward / backward), 0xFFFFFFFF = dropped: past the end, false positive, hole,
    //           or (BWD) empty tap of the neighbour (.cpp:679)
    //   rec[u]  {neighbour | centre lane << 24, 1 / population}
    auto load_round = [&](uint32_t s0, uint32_t (&key)[R], uint2 (&rec)[R]) {
        PairEntry en[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const uint32_t i = s0 + (uint32_t)u * 256u + tid;
            en[u] = record(i < n ? i : 0u);
        }
        uint32_t pop[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const uint32_t fw = code_fwd(en[u].code), bw = code_bwd(en[u].code);
            const bool ok = s0 + (uint32_t)u * 256u + tid < n && fw != kNoTap && (!BWD || bw != kNoTap);
            key[u] = ok ? (BWD ? bw : fw) : 0xFFFFFFFFu;
            const uint32_t row = BWD ? en[u].cand : (uint32_t)qorig[code_q(en[u].code) & 63u];
            pop[u] = (uint32_t)cnt_cloud[ok ? (size_t)row * ntap + key[u] : (size_t)0];
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (BWD) key[u] = ((key[u] != 0xFFFFFFFFu) & (pop[u] != 0u)) ? key[u] : 0xFFFFFFFFu;   // .cpp:679
            // {neighbour | centre lane << 24, bits of 1 / population}: the IEEE quotient, as the register path
            // computes it (a dropped record's population may be 0: its reciprocal is never used)
            rec[u] = make_uint2(en[u].cand | (code_q(en[u].code) << 24), __builtin_bit_cast(uint32_t, 1.0f / (float)pop[u]));
        }
    };
    // ranks inside the wave-chunks (chunk u * 4 + wave = 64 consecutive records) + the chunks' per-tap counts
    auto rank_round = [&](const uint32_t (&key)[R], uint32_t (&rank)[R], bool totals) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            uint32_t c;
            bool leader;
            rank[u] = chunk_rank(key[u], c, leader);
            if (leader) {
                wcnt[key[u] * (4u * R) + (uint32_t)u * 4u + wave] = c;
                if (totals) atomicAdd(&tot[key[u]], c);             // one add per (chunk, tap): distinct addresses
            }
        }
    };
    auto hist_round = [&](const uint32_t (&key)[R], const uint2 (&rec)[R]) {
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (key[u] != 0xFFFFFFFFu) atomicAdd(&hist[key[u] * 64u + (rec[u].x >> 24)], 1u);
    };
    uint32_t key[R], rank[R];
    uint2 rec[R];
    const bool single = n <= 256u * R;                              // the whole tile in one round: records stay in registers
    if (single) {
        load_round(0, key, rec);
        rank_round(key, rank, true);
        hist_round(key, rec);
    } else {
        for (uint32_t s0 = 0; s0 < n; s0 += 256u * R) {             // totals only (wcnt is rebuilt per round below)
            load_round(s0, key, rec);
#pragma unroll
            for (int u = 0; u < R; ++u) {
                uint32_t c;
                bool leader;
                chunk_rank(key[u], c, leader);
                if (leader) atomicAdd(&tot[key[u]], c);
            }
            hist_round(key, rec);
        }
    }
    __syncthreads();
    if (tap_total != nullptr && tid < (uint32_t)ntap && tot[tid] != 0) atomicAdd(&tap_total[tid], tot[tid]);
    if (tid < 64) {                                                 // exclusive scan of the totals (ntap <= 64)
        int total;
        const int off = wave_excl_scan(tid < (uint32_t)ntap ? (int)tot[tid] : 0, total);
        if (tid < (uint32_t)ntap) {
            toff[tid] = (uint32_t)off;
            base[tid] = (uint32_t)off;
        }
        if (tid == 0) toff[ntap] = (uint32_t)total;
        const uint64_t populated = __ballot(tid < (uint32_t)ntap && tot[tid < (uint32_t)ntap ? tid : 0u] != 0u);
        if (tid == 0 && pop_mask != nullptr) pop_mask[tile] = populated;
    }
    __syncthreads();
    // Quarter points of every tap's run, moved forward to the next change of centre (a run is centre-major, centres in
    // lane order): deep_gemm_kernel's stage 1 walks the four parts of a run with four thread groups, and a centre
    // must belong to exactly one of them.  split[f] = {a1, a2, a3, length}, positions relative to the run's start.
    // the centres that have records of tap f: the rows of G_f' that are not zero (deep_gemm stores only those for the
    // grad_filter kernel, which then contracts over them instead of all 64)
    if (tap_cmask != nullptr)
        for (uint32_t f = wave; f < (uint32_t)ntap; f += 4) {
            const uint64_t m = __ballot(hist[f * 64u + lane] != 0u);
            if (lane == 0) tap_cmask[tile * (size_t)ntap + f] = m;
        }
    if (tap_split != nullptr)
        for (uint32_t f = wave; f < (uint32_t)ntap; f += 4) {
            int len;
            const int excl = wave_excl_scan((int)hist[f * 64u + lane], len);   // records of the centres before `lane`
            uint32_t a[3];
#pragma unroll
            for (int r = 1; r <= 3; ++r) {
                const uint32_t goal = (uint32_t)len * (uint32_t)r / 4u;
                uint32_t best = (uint32_t)excl >= goal ? (uint32_t)excl : (uint32_t)len;   // boundaries at or after the goal
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const uint32_t other = __shfl_xor(best, o);
                    best = other < best ? other : best;
                }
                a[r - 1] = best;
            }
            if (lane == 0) tap_split[tile * (size_t)ntap + f] = make_uint4(a[0], a[1], a[2], (uint32_t)len);
        }
    uint2 *meta = tap_meta + tseg.x;
    for (uint32_t s0 = 0; s0 < n; s0 += 256u * R) {
        if (!single) {
            zero_wcnt();
            __syncthreads();
            load_round(s0, key, rec);
            rank_round(key, rank, false);
            __syncthreads();
        }
        // exclusive offsets over the 4 * R = 64 wave-chunks of every tap: one wave-wide scan per tap (lane = chunk)
        for (uint32_t f = wave; f < (uint32_t)ntap; f += 4) {
            int total;
            const int c = (int)wcnt[f * (4u * R) + lane];
            const int off = wave_excl_scan(c, total);
            const uint32_t b0 = base[f];
            wcnt[f * (4u * R) + lane] = b0 + (uint32_t)off;
            if (lane == 0) base[f] = b0 + (uint32_t)total;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (key[u] != 0xFFFFFFFFu) meta[wcnt[key[u] * (4u * R) + (uint32_t)u * 4u + wave] + rank[u]] = rec[u];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// deep_sched_kernel: launch order of the query tiles for deep_gemm_kernel.  A tile's cost grows with its number
// of records (boundary tiles of a cloud hold a fraction of an interior tile's), and the kernel runs ~4 rounds of
// long workgroups, so the tiles of every XCD (clouds stay on their XCD, see BlockMap) are issued longest first:
// sched[xcd * cap + i] = i-th tile (b * ntiles + qt) of that XCD, 0xFFFFFFFF past its last one.
// One workgroup per XCD; bitonic sort of (records descending, tile id ascending) keys in LDS, up to 4096 tiles per
// XCD (more: the BlockMap order is kept).
// ---------------------------------------------------------------------------------------------
constexpr int kSchedMax = 4096;
__global__ __launch_bounds__(1024) void deep_sched_kernel(const uint2 *__restrict__ segs, int B, int ntiles, int cap,
                                                          uint32_t *__restrict__ sched)
{
    __shared__ unsigned long long keys[kSchedMax];
    const int xcd = blockIdx.x;
    const int nclouds = B > xcd ? (B - xcd + 7) / 8 : 0;
    const int n = nclouds * ntiles;
    uint32_t *out = sched + (size_t)xcd * cap;
    if (n > kSchedMax) {
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
            const uint32_t recs = segs[tile].y == kSegOverflow ? 0u : segs[tile].y;
            k = ((unsigned long long)(0xFFFFFFFFu - recs) << 32) | tile;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < npad; t += blockDim.x) {
                const int p = t ^ j;
                if (p > t) {
                    const unsigned long long a = keys[t], b = keys[p];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { keys[t] = b; keys[p] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < cap; i += blockDim.x) out[i] = i < n ? (uint32_t)keys[i] : 0xFFFFFFFFu;
}

__device__ __forceinline__ int qorig_early(const PointRec<float> *__restrict__ pts, size_t tile_id, int q)
{
    return pts[tile_id * kTile + q].idx;
}

// ---------------------------------------------------------------------------------------------
// deep_gemm_kernel: out[centre, 0..NDIM) = sum_f M_f[centre, 0..KDIM) . Bm[f][KDIM][NDIM]
//   BWD = false : forward.    src = input    (rows of KDIM = Cin),  order/weights of the forward taps, Bm = filter
//   BWD = true  : grad_input. src = grad_out (rows of KDIM = Cout), backward taps, Bm = filter^T ([F][Cout][Cin])
// One workgroup (4 waves) per query tile; per tap:
//   stage 1  M_f[centre] = segmented sum of the tap's neighbour rows / count on the vector ALUs.  The 256 threads
//            form TG = 256 / KDIM groups of KDIM threads (thread = column); every group takes ONE tap of the current
//            super-step and walks that tap's run on its own (rows and record metadata straight from global memory,
//            software-pipelined), so the super-step's TG taps proceed concurrently without any barrier or shared
//            staging between them -> LDS M [TG][64][KDIM+1]
//   stage 2  out += M_f . Bm[f] for the super-step's taps: the 2 x NDIM/32 output blocks dealt to the waves,
//            accumulators in registers across all taps, B operand streamed from L2
// The populated taps are taken longest run first (taps running together have similar lengths).
// LDS: M [TG][64][KDIM+4] | taps [64] | qorig [64] | scrap [256][4] | rowc [TG][64] bytes
// ---------------------------------------------------------------------------------------------
template <int KDIM, int NDIM, bool BWD, bool WIDE>   // WIDE: source rows have at least 4 floats (one 16-byte load per lane)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DEEP_GEMM_WAVES))) void deep_gemm_kernel(const PointRec<float> *__restrict__ pts,
                                                        const PairEntry *__restrict__ pairs,
                                                        const uint2 *__restrict__ segs,
                                                        const float *__restrict__ src,
                                                        const float *__restrict__ Bm, int N, int ntiles, int ntap,
                                                        const uint32_t *__restrict__ sched, int sched_cap,
                                                        float *__restrict__ out,
                                                        const uint2 *__restrict__ tap_meta,
                                                        const uint32_t *__restrict__ tap_off,
                                                        uint8_t *__restrict__ tile_flag,
                                                        int kreal, int nreal,   // real row lengths of src / out
                                                        float *__restrict__ gbuf,   // BWD: [tile][tap][64][KDIM] <- M_f
                                                        const float *__restrict__ xin,   // BWD: the layer's input
                                                        const uint4 *__restrict__ tap_split,   // [tile][tap] quarter points
                                                        const unsigned long long *__restrict__ tap_cmask)   // BWD: [tile][tap] centres with records
{
    constexpr int LDA = KDIM + 4;                         // rows 16-byte aligned (stage 1 stores 4 columns at once)
    constexpr int TG = KDIM < 256 ? 256 / KDIM : 1;        // taps per super-step = thread groups of stage 1
    // The 2 x NDIM/32 output blocks (32 x 32) are dealt to the 4 waves so that every index below is a compile-time
    // constant (accumulators stay in registers, no predicated MFMAs).  With four or more column blocks a wave owns
    // BOTH row blocks of its column blocks w, w+4, ...: every B-operand value it fetches from L2 feeds two MFMAs and
    // no two waves fetch the same value (the filter block is re-read for every (tile, tap): 2*64*N flops per
    // K*N*4 bytes, the stage's L2 traffic).  With two column blocks: wave = (row block w & 1, column block w >> 1);
    // with one: waves 0 and 1 take a row block each.
    constexpr int CBK = KDIM / 32;                        // column blocks of M_f
    constexpr int CBN = NDIM / 32;                        // column blocks of the output
    constexpr bool kBoth = CBN >= 4;
    constexpr int RBW = kBoth ? 2 : 1;                    // row blocks per wave
    constexpr int CPW = kBoth ? CBN / 4 : 1;              // column blocks per wave
    constexpr int CSTEP = kBoth ? 4 : 2;                  // distance of a wave's column blocks
    constexpr int OPW = RBW * CPW;                        // accumulator blocks per wave
    constexpr int KG = 8;                                 // MFMA k-steps per B-operand prefetch group
    static_assert(KDIM % 32 == 0 && NDIM % 32 == 0 && KDIM <= 256 && (CBK == 1 || CBK % 2 == 0) && (CBN == 1 || CBN % 2 == 0),
                  "deep path: channel counts are 32 or multiples of 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *A = reinterpret_cast<float *>(smem);           // M_f of the super-step's taps [TG][64][LDA]
    size_t off = align16((size_t)TG * 64 * LDA * 4);
    uint32_t *taps = reinterpret_cast<uint32_t *>(smem + off);   // [64] populated taps, longest run first; [64] = how many
    int32_t *qorig = reinterpret_cast<int32_t *>(taps + 68);
    float *scrap = reinterpret_cast<float *>(qorig + 64);   // [256][4] write-only (stage 1's predicated-off stores)
    uint8_t *rowc = reinterpret_cast<uint8_t *>(scrap + 1024);   // [TG][64] BWD: centre of every packed row of the G block

    // workgroup -> tile: XCD (blockIdx.x & 7, as in BlockMap) and position in that XCD's longest-first order
    const uint32_t tile_sched = sched[(size_t)(blockIdx.x & 7) * sched_cap + (blockIdx.x >> 3)];
    if (tile_sched == 0xFFFFFFFFu) return;
    const int b = (int)(tile_sched / (uint32_t)ntiles), qt = (int)(tile_sched % (uint32_t)ntiles);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rb = kBoth ? 0 : (wave & 1), cb0 = kBoth ? wave : (wave >> 1);   // first row / column block of this wave
    const bool o_on = CBN >= 2 || wave < 2;
    const uint32_t myrow = (uint32_t)(rb * 32 + (lane & 31));
    const size_t tile_id = (size_t)b * ntiles + qt;
    const uint2 tseg = segs[tile_id];
    if (wave == 0) qorig[lane] = pts[tile_id * kTile + lane].idx;
    float *out_cloud = out + (size_t)b * N * nreal;
    if (tseg.y == kSegOverflow) {
        // the cloud's pair region overflowed: leave zero rows; the generic kernel is launched afterwards for the
        // tiles deep_order_kernel flagged (it can search the tile itself)
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * NDIM; e += 256) {
            const int orig = qorig[e / NDIM];
            if (orig >= 0 && (e % NDIM) < nreal) out_cloud[(size_t)orig * nreal + (e % NDIM)] = 0.0f;
        }
        return;
    }
    const uint2 *meta = tap_meta + tseg.x;
    const uint32_t *toff = tap_off + tile_id * (size_t)(ntap + 1);
    const float *src_cloud = src + (size_t)b * N * kreal;

    f32x16 acc[OPW];
#pragma unroll
    for (int i = 0; i < OPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // Tiles that meet a non-finite value are handed to the exact generic kernel (tile_flag), so that NaN / Inf
    // reach exactly the outputs they reach in the reference whatever the association of the sums.
    float badsum = 0.0f;
    // grad_input pass: the grad_filter kernel will multiply ALL 64 input rows of this tile with G_f' (zero rows for
    // centres without that tap), which is exact only for finite inputs -- a non-finite input row sends the tile to
    // the generic kernel as well
    if (BWD && xin != nullptr) {
        const float *xc = xin + (size_t)b * N * nreal;
        for (int e = threadIdx.x; e < 64 * nreal; e += 256) {
            const int orig = qorig_early(pts, tile_id, e / nreal);
            if (orig >= 0) {
                const float v = xc[(size_t)orig * nreal + (e % nreal)];
                badsum += v - v;
            }
        }
    }
    DEV_GEMM_DECL()
    // populated taps, longest run first (ties: lower tap): position = number of taps ahead (wave 0, lane = tap)
    if (wave == 0) {
        const uint32_t len = lane < ntap ? toff[lane + 1] - toff[lane] : 0u;
        uint32_t pos = 0;
        for (int g = 0; g < ntap; ++g) {
            const uint32_t lg = __shfl(len, g);
            pos += (lg > len || (lg == len && g < lane)) ? 1u : 0u;
        }
        if (len != 0) taps[pos] = (uint32_t)lane;
        const uint64_t ne = __ballot(len != 0);
        if (lane == 0) taps[64] = (uint32_t)__popcll(ne);
    }
    __syncthreads();
    const int nne = (int)taps[64];
    // stage 1 thread layout: LPR = KDIM / 4 lanes per record (thread = 4 consecutive columns, ONE 16-byte load per
    // record and lane), 256 / LPR = 4 * TG groups: the four groups 4g .. 4g+3 walk the four parts of tap slot g's run
    constexpr int LPR = KDIM / 4;
    const int tl = (int)threadIdx.x % LPR, grp = (int)threadIdx.x / LPR;
    const int gslot = grp >> 2, part = grp & 3;
    const int col0 = 4 * tl;
    // columns [col0, col0 + 4) of a row of kreal floats with one unconditional 16-byte load: the load starts at
    // min(col0, kreal - 4) and the values are shifted into place (`shl` = 0 for full chunks; >= 4: all padding)
    constexpr bool wide = WIDE;
    const int lstart = wide ? (col0 < kreal - 4 ? col0 : kreal - 4) : 0;
    const int shl = col0 - lstart;
    float *Ag = A + gslot * 64 * LDA + col0;
    float *dummy = scrap + 4 * threadIdx.x;                  // where the stores of steps past the run's end go
    for (int t0 = 0; t0 < nne; t0 += TG) {
        // ---- stage 1: M_f[centre] = sum over the centre's records of tap f of row[neighbour] / count.
        // A tap's records are centre-major (the search's order, kept by the stable sort of deep_order_kernel), so
        // this is a segmented sum: the thread walks its part of the run, adds the neighbour's values to running sums
        // that restart at every change of centre, and stores them to M_f[centre] after EVERY record -- the last
        // store of a centre is its total (the parts are cut at changes of centre: deep_order_kernel's tap_split).
        // No branch per record: selects, four fmas, one 16-byte LDS store.  Exact for any values.
        // What bounds this stage is the NUMBER of vector-memory instructions (a wave-level gather costs ~30-40 cycles
        // of the CU's address unit whatever its width, tools/ubench/gather_rate.hip): one 16-byte load per lane
        // fetches a record's row with KDIM / 4 lanes, i.e. 256 / KDIM records per wave instruction -- the
        // column-per-thread layout of the first version spent 4x as many.
        // Software pipeline over kD NAMED slots of kU records (loop unrolled by kD, all loads unconditional from
        // clamped indices, wave-uniform control flow: see the register path's backward kernel): the metadata of
        // step s+3 and the rows of step s+2 are in flight while step s is consumed.
        __syncthreads();                                     // previous super-step's stage 2 has read A
        for (int e = threadIdx.x; e < TG * 64 * LDA / 4; e += 256)
            reinterpret_cast<float4 *>(A)[e] = make_float4(0.f, 0.f, 0.f, 0.f);   // centres without the tap stay zero
        const int ti = t0 + gslot;
        const uint32_t f1 = ti < nne ? taps[ti] : 0u;
        const uint4 sp = tap_split[tile_id * (size_t)ntap + f1];
        __syncthreads();
        {
            const uint32_t rbeg = part == 0 ? 0u : part == 1 ? sp.x : part == 2 ? sp.y : sp.z;
            const uint32_t rend = part == 0 ? sp.x : part == 1 ? sp.y : part == 2 ? sp.z : sp.w;
            const uint32_t len = ti < nne ? rend - rbeg : 0u;
            // (an empty part still issues clamped loads of "its" record 0: point it at the tap's first record, which
            // exists -- the slot after the run's end may never have been written)
            // Addresses are 32-bit offsets from wave-uniform bases (the tile's records, the cloud's rows: the host
            // checks N * kreal * 4 < 2^32): one multiply-add per load instead of a 64-bit product.
            const uint32_t mbase = ti < nne ? toff[f1] + (len != 0 ? rbeg : 0u) : 0u;
            const uint32_t lenm1 = len != 0 ? len - 1u : 0u;
            const uint32_t rowb = (uint32_t)kreal * 4u, lbyte = (uint32_t)lstart * 4u;
            const char *srcb = reinterpret_cast<const char *>(src_cloud);
            constexpr int kU = DEEP_KU, kD = DEEP_KD;
            uint2 m[kD][kU];
            float4 v[kD][kU];
            auto ld_meta = [&](int sl, uint32_t p0) {
#pragma unroll
                for (int u = 0; u < kU; ++u) m[sl][u] = meta[mbase + (p0 + u < lenm1 ? p0 + u : lenm1)];   // (clamped: v_min)
            };
            auto ld_rows = [&](int sl) {
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const uint32_t boff = __umul24(m[sl][u].x, rowb);   // (low 24 bits of .x = the neighbour; rowb <= 1024)
                    if constexpr (wide) {
                        const float4_a4 t = *reinterpret_cast<const float4_a4 *>(srcb + (boff + lbyte));
                        v[sl][u] = make_float4(t.x, t.y, t.z, t.w);
                    } else {                                 // rows of 1..3 floats: clamped scalar loads
                        const float *row = reinterpret_cast<const float *>(srcb + boff);
                        v[sl][u] = make_float4(row[0], row[kreal > 1 ? 1 : 0], row[kreal > 2 ? 2 : 0], 0.f);
                    }
                }
            };
            float sum[4] = {0.f, 0.f, 0.f, 0.f};
            uint32_t prev = 64;                              // centre of the running sums (64 = none)
            // One record.  What bounds this stage is the NUMBER of vector instructions a wave issues per record (two
            // waves per SIMD: ~9 cycles per instruction; by removal the stage takes the same time whether its rows come
            // from L1 or from the memory-side cache, and twice as long for 1-KiB rows -- one record per wave instruction
            // -- as for 512-byte rows -- two): 45 in round 3, ~22 now --
            //   FULL (kreal == KDIM, the common case): the loaded 16 bytes ARE the lane's four columns, no shifting;
            //   PRED = false (every lane's step lies inside its run): no select on "inside the run";
            //   the running sum restarts through its addend (same ? sum : 0), one fma per column;
            //   non-finite values are no longer looked for here: they reach the accumulators (a non-finite row makes its
            //   centre's M row non-finite, 1 / population being finite and non-zero; Inf x 0 = NaN in the products),
            //   where the epilogue finds them -- once per tile instead of once per record.
            auto rec = [&](auto full_c, auto pred_c, int j, int u, uint32_t p0) {
                constexpr bool kFull = decltype(full_c)::value, kPred = decltype(pred_c)::value;
                const bool in = kPred ? p0 + u < len : true;
                const uint32_t q = m[j][u].x >> 24;
                const float w = __builtin_bit_cast(float, m[j][u].y);
                const float4 r = v[j][u];
                float x[4];
                if constexpr (kFull) {
                    x[0] = r.x; x[1] = r.y; x[2] = r.z; x[3] = r.w;
                } else if constexpr (wide) {
                    const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        x[c] = 0.0f;
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            if (d >= c) x[c] = (shl == d - c) ? rr[d] : x[c];
                    }
                } else {
                    x[0] = col0 + 0 < kreal ? r.x : 0.0f;
                    x[1] = col0 + 1 < kreal ? r.y : 0.0f;
                    x[2] = col0 + 2 < kreal ? r.z : 0.0f;
                    x[3] = 0.0f;
                }
                const bool same = q == prev;
#pragma unroll
                for (int c = 0; c < 4; ++c) {    // (scalars: a select between float4 objects goes through scratch)
                    const float t = __builtin_fmaf(x[c], w, same ? sum[c] : 0.0f);
                    sum[c] = kPred ? (in ? t : sum[c]) : t;
                }
                prev = kPred ? (in ? q : prev) : q;
                if constexpr (kPred || !(DEEP_SKIP_STORES && KDIM == 256)) {   // (256 columns: one record per wave, the test is uniform)
                    float *dst = kPred ? (in ? Ag + q * LDA : dummy) : Ag + q * LDA;
                    *reinterpret_cast<float4 *>(dst) = make_float4(sum[0], sum[1], sum[2], sum[3]);
                } else {
                    // inside the run: only a centre's LAST record needs its store (the next record, already loaded,
                    // names another centre, or the run ends) -- a 16-byte LDS store occupies the CU's store path for
                    // 13 cycles per wave (MI355X_MICROARCH.md), one record in five is the last of its centre
                    const uint32_t qn = (u + 1 < kU ? m[j][(u + 1) % kU].x : m[(j + 1) % kD][0].x) >> 24;
                    if (qn != q || p0 + u + 1 >= len)
                        *reinterpret_cast<float4 *>(Ag + q * LDA) = make_float4(sum[0], sum[1], sum[2], sum[3]);
                }
            };
            // (control flow is kept WAVE-UNIFORM -- a wave holds several groups with different run lengths, the
            // shorter ones idle on clamped loads and masked stores: with a per-lane loop exit hipcc's wait counts
            // degenerate to draining the queue at the top of every unrolled iteration)
            auto walk = [&](auto full_c) {
                ld_meta(0, 0);
                ld_meta(1, kU);
                ld_meta(2, 2 * kU);
                __builtin_amdgcn_sched_barrier(0);
                ld_rows(0);
                ld_rows(1);
                uint32_t p0 = 0;
                bool more = true;
                while (more) {
#pragma unroll
                    for (int j = 0; j < kD; ++j) {
                        if (!__any(p0 < len)) {
                            more = false;
                            break;
                        }
                        ld_meta((j + 3) % kD, p0 + 3 * kU);
                        ld_rows((j + 2) % kD);
                        __builtin_amdgcn_sched_barrier(0);
                        if (__all(p0 + kU <= len)) {
#pragma unroll
                            for (int u = 0; u < kU; ++u) rec(full_c, std::false_type{}, j, u, p0);
                        } else {
#pragma unroll
                            for (int u = 0; u < kU; ++u) rec(full_c, std::true_type{}, j, u, p0);
                        }
                        p0 += kU;
                    }
                }
            };
            if (__any(len != 0) && !(CONV3P_ABLATE & 131072)) {   // (developer bit: no walk)
                if (kreal == KDIM) walk(std::true_type{});
                else walk(std::false_type{});
            }
        }
        if (BWD && gbuf != nullptr) {
            // packed row -> centre of the G blocks stored below (row r of tap f' = its r-th centre with records): one
            // table per tap, written here so that the barrier that completes M publishes it too (round 3: every thread
            // searched the mask for each of its 16-byte pieces, ~50 instructions per piece, 315 us of the kernel)
            const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
            for (int g = wave; g < TG && t0 + g < nne; g += 4) {
                const unsigned long long cm = tap_cmask[tile_id * (size_t)ntap + taps[t0 + g]];
                if ((cm >> lane) & 1ull) rowc[g * 64 + __popcll(cm & lt)] = (uint8_t)lane;
            }
        }
        __syncthreads();                                     // M of the super-step's taps complete in LDS
        GDBG(0)
        const int ntp = nne - t0 < TG ? nne - t0 : TG;       // taps of this super-step
        // grad_input pass: every M_f (= G_f' of this tile) also goes to gbuf, the grad_filter kernel's B operand
        // (only the rows of the centres that have the tap -- the others are zero --, packed in centre order: row r of
        // the stored block is the r-th set bit of tap_cmask; on the SceneNN-shaped rooms 42 % of the rows)
        if (BWD && gbuf != nullptr && !(CONV3P_ABLATE & 262144)) {   // (developer bit: no G store)
            for (int g = 0; g < ntp; ++g) {
                const float *Af = A + g * 64 * LDA;
                const uint32_t fg = taps[t0 + g];
                const unsigned long long cm = tap_cmask[tile_id * (size_t)ntap + fg];
                const int nrow = __popcll(cm);
                float *gt = gbuf + (tile_id * (size_t)ntap + fg) * 64 * KDIM;
                for (int e = threadIdx.x; e < nrow * (KDIM / 4); e += 256) {
                    const int rr = e / (KDIM / 4), c4 = e % (KDIM / 4);
                    const int pos = rowc[g * 64 + rr];       // centre of packed row rr
                    const float *ar = Af + pos * LDA + 4 * c4;
                    *reinterpret_cast<float4 *>(gt + (size_t)rr * KDIM + 4 * c4) = make_float4(ar[0], ar[1], ar[2], ar[3]);
                }
            }
        }
        // ---- stage 2: out += M_f . Bm[f] for the super-step's taps, as ONE flat sequence of k-groups (KG MFMA
        // k-steps each) over all of them: the B operand of group i+2 is requested before group i's MFMAs are issued
        // (three NAMED buffers, loop unrolled by three; loads unconditional from a clamped group index) -- with one
        // group of look-ahead and a register copy at the end, every group waited a full L2 latency for its operand.
        if (o_on) {
            constexpr int GPT = KDIM / (2 * KG);             // k-groups per tap
            const int ng = ntp * GPT;
            const float *Bl = Bm + (lane >> 5) * NDIM + cb0 * 32 + (lane & 31);
            const float *Al = A + myrow * LDA + (lane >> 5);
            auto load_g = [&](int gi, float (&bv)[KG][CPW]) {
                const int gc = gi < ng ? gi : ng - 1;
#ifdef DEEP_DEV_B_L1   // developer timing probe (wrong results): every B-operand group is the SAME 2 KG rows -- L1 hits instead of L2 traffic
                const float *Bf = Bl + (size_t)(gc & 0) * NDIM;
#else
                const float *Bf = Bl + (size_t)taps[t0 + gc / GPT] * KDIM * NDIM + (size_t)(gc % GPT) * (2 * KG) * NDIM;
#endif
#pragma unroll
                for (int s2 = 0; s2 < KG; ++s2)
#pragma unroll
                    for (int j = 0; j < CPW; ++j) bv[s2][j] = Bf[(size_t)(2 * s2) * NDIM + j * (CSTEP * 32)];
            };
            auto mma_g = [&](int gi, const float (&bv)[KG][CPW]) {
                const float *arow = Al + (gi / GPT) * 64 * LDA + (gi % GPT) * (2 * KG);
#pragma unroll
                for (int s2 = 0; s2 < KG; ++s2) {
                    float a[RBW];
#pragma unroll
                    for (int i = 0; i < RBW; ++i) a[i] = arow[i * 32 * LDA + 2 * s2];
#pragma unroll
                    for (int i = 0; i < RBW; ++i)
#pragma unroll
                        for (int j = 0; j < CPW; ++j)
                            acc[i * CPW + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[s2][j], acc[i * CPW + j], 0, 0, 0);
                }
            };
            float b0[KG][CPW], b1[KG][CPW], b2[KG][CPW];
            load_g(0, b0);
            load_g(1, b1);
            for (int gi = 0; gi < ((CONV3P_ABLATE & 65536) ? 0 : ng); gi += 3) {
                load_g(gi + 2, b2);
                __builtin_amdgcn_sched_barrier(0);
                mma_g(gi, b0);
                if (gi + 1 >= ng) break;
                load_g(gi + 3, b0);
                __builtin_amdgcn_sched_barrier(0);
                mma_g(gi + 1, b1);
                if (gi + 2 >= ng) break;
                load_g(gi + 4, b1);
                __builtin_amdgcn_sched_barrier(0);
                mma_g(gi + 2, b2);
            }
        }
        GDBG(4)
    }

    GDBG(5)
    DEV_GEMM_PRINT(KDIM, NDIM, BWD, wave, lane, nne)
    // ---- epilogue: C fragments -> out rows (by original index)
    // non-finite rows of `src` (or a sum that overflowed) have made some accumulator non-finite: x - x is 0 only for
    // finite x
    if (o_on) {
#pragma unroll
        for (int i = 0; i < OPW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) badsum += acc[i][r] - acc[i][r];
    }
    const bool bad = __syncthreads_or(!(badsum == 0.0f)) != 0;
    // bad: zero rows now, exact accumulation by the generic kernel.  The flag is rewritten by EVERY pass over the tile:
    // the record order (and with it this array) outlives the call when the geometry is reused -- a later call with
    // finite data, or the next channel block of a wide layer, must not find the mark of an earlier one (the generic
    // kernel would add its result to rows this kernel has filled).  (Overflowed tiles returned above: their mark, set
    // by deep_order_kernel, stays.)
    if (threadIdx.x == 0) tile_flag[tile_id] = bad ? 1 : 0;
    if (o_on) {
#pragma unroll
        for (int i = 0; i < RBW; ++i)
#pragma unroll
            for (int j = 0; j < CPW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (rb + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int orig = qorig[row];
                    const int col = (cb0 + CSTEP * j) * 32 + (lane & 31);
                    if (orig >= 0 && col < nreal) out_cloud[(size_t)orig * nreal + col] = bad ? 0.0f : acc[i * CPW + j][r];
                }
    }
}

// ---------------------------------------------------------------------------------------------
// deep_plan_kernel: work items of deep_dw_kernel.  The cost of (tile, tap) is proportional to its number of
// pairs: on surface-like clouds the central tap holds ~3x the average tap's and boundary tiles a fraction of an
// interior tile's, so neither equal tile ranges nor equal tap shares balance the workgroups.  Tap f gets
// n_f ~ (its share of all pairs) x `target` items (exactly `target` in total: a whole number of rounds of the
// resident workgroups), and its tiles are cut into n_f ranges of equal WORK (32-record blocks + a fixed cost per
// populated tile, from a prefix sum over tap_off).  Items are issued in order of their first tile, so that the
// items running together read the same rows (L2); item j of tap f writes partial slot ibeg_f + j.
//   items[i] = {tap, first tile, end tile, slot};  tap_rng[f] = {ibeg_f, n_f};  *nitems = number of items
// One workgroup of 1024 threads.  More than kPlanTiles tiles: equal tile ranges.
// ---------------------------------------------------------------------------------------------
constexpr int kPlanMax = 2048;
constexpr int kPlanTiles = 8192;
__global__ __launch_bounds__(1024) void deep_plan_kernel(const uint32_t *__restrict__ tap_total,
                                                         const unsigned long long *__restrict__ pop_mask, int ntap, int tiles,
                                                         int target, uint4 *__restrict__ items,
                                                         uint2 *__restrict__ tap_rng, uint32_t *__restrict__ nitems)
{
    __shared__ unsigned long long keys[kPlanMax];
    __shared__ uint32_t t1s[kPlanMax];
    __shared__ uint32_t nf[64], ibeg[65];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < 64) {
        // n_f = floor share, at least 1, at most one item per tile; the items left over go to the taps with the
        // largest remainders (ties: lower tap), so that the total is EXACTLY `target` whenever that is possible.
        // lane = tap (ntap <= 64).
        const bool on = tid < (uint32_t)ntap;
        const uint32_t mytot = on ? tap_total[tid] : 0u;
        unsigned long long sum = mytot;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const unsigned long long share = sum ? (unsigned long long)mytot * (unsigned)target : 0ull;
        uint32_t n = sum ? (uint32_t)(share / sum) : 1u;
        uint32_t rem = sum ? (uint32_t)(share % sum * 1024ull / sum) : 0u;
        if (n < 1u) { n = 1u; rem = 0; }
        if (n > (uint32_t)tiles) { n = (uint32_t)tiles; rem = 0; }
        if (n < 1u) n = 1u;
        if (!on) { n = 0; rem = 0; }
        int used_i;
        (void)wave_excl_scan((int)n, used_i);
        const uint32_t left = (uint32_t)used_i < (uint32_t)target ? (uint32_t)target - (uint32_t)used_i : 0u;
        const bool elig = on && n < (uint32_t)tiles && rem > 0;
        uint32_t ahead = 0;                                   // eligible taps that are served before this one
        for (int g = 0; g < ntap; ++g) {
            const uint32_t rg = __shfl(rem, g);
            const bool eg = __shfl((int)elig, g) != 0;
            ahead += (eg && (rg > rem || (rg == rem && (uint32_t)g < tid))) ? 1u : 0u;
        }
        if (elig && ahead < left) n += 1;
        int run_total;
        const uint32_t run = (uint32_t)wave_excl_scan((int)n, run_total);
        if (on) {
            nf[tid] = n;
            ibeg[tid] = run;
            tap_rng[tid] = make_uint2(run, n);
        }
        if (tid == 0) {
            ibeg[ntap] = (uint32_t)run_total;
            *nitems = (uint32_t)run_total;
        }
    }
    __syncthreads();
    const uint32_t total = ibeg[ntap];                       // <= target + ntap <= kPlanMax (host checks)
    const uint32_t T = (uint32_t)tiles;
    // Every WAVE takes taps wave, wave + 16, ... on its own (no workgroup barrier inside): lane l walks the contiguous
    // run of `per` tiles [l * per, (l + 1) * per) twice -- once to count the tiles that have the tap (then an exclusive
    // scan over the lanes), once to write the items that start in its run.  (Round 3: all 1 024 threads on one tap
    // after the other, a scan through LDS and a barrier per tap: 86 us on the critical path of the in-line cfg5 step.)
    extern __shared__ unsigned long long pm[];             // [T] the tiles' populated-tap masks (T <= kPlanTiles), read twice per tap
    if (T <= (uint32_t)kPlanTiles) {
        for (uint32_t t = tid; t < T; t += 1024) pm[t] = pop_mask[t];
        __syncthreads();
    }
    for (int f = (int)wave; f < ntap; f += 16) {
        const uint32_t n = nf[f], i0 = ibeg[f];
        if (T > (uint32_t)kPlanTiles) {                      // too many tiles for the walk below: equal tile ranges
            for (uint32_t j = lane; j < n; j += 64) {
                const uint32_t t0 = (uint32_t)((unsigned long long)j * T / n);
                keys[i0 + j] = ((unsigned long long)t0 << 32) | ((unsigned long long)f << 16) | j;
                t1s[i0 + j] = (uint32_t)((unsigned long long)(j + 1) * T / n);
            }
            continue;
        }
        // work of a tile for this tap = 1 if it is populated (one [Cin x 64].[64 x Cout] product).  Item j of the
        // tap starts at b_j = the first tile whose inclusive work prefix exceeds floor(wall * j / n) (b_0 = 0), i.e.
        // the populated tile with exclusive prefix e starts exactly the items j in
        // [ceil(e * n / wall), ceil((e + 1) * n / wall)).
        // Lane l takes the tiles l, l + 64, ...: a row of 64 consecutive masks per LDS read (a contiguous run per lane
        // put all 64 lanes on one bank: 49 of the kernel's 68 us), the prefix of a tile from the rows' ballots.
        const uint64_t lt_lane = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        uint32_t wall = 0;
        for (uint32_t t0 = 0; t0 < T; t0 += 64) {
            const uint32_t t = t0 + lane;
            wall += (uint32_t)__popcll(__ballot(t < T && ((pm[t < T ? t : 0u] >> f) & 1ull) != 0));
        }
        if (lane == 0) {
            keys[i0] = ((unsigned long long)f << 16);        // item 0 starts at the first tile
            t1s[i0 + n - 1] = T;                              // the last item ends at the last tile
        }
        if (wall == 0) continue;                             // (uniform; a tap no tile has: n items of nothing)
        uint32_t base = 0;
        for (uint32_t t0 = 0; t0 < T; t0 += 64) {
            const uint32_t t = t0 + lane;
            const bool has = t < T && ((pm[t < T ? t : 0u] >> f) & 1ull) != 0;
            const uint64_t m = __ballot(has);
            if (has) {
                // (e < tiles <= 8192, n <= 2048: 32-bit products)
                const uint32_t e = base + (uint32_t)__popcll(m & lt_lane);
                const uint32_t ja = (e * n + wall - 1u) / wall;
                uint32_t jb = ((e + 1u) * n + wall - 1u) / wall;
                if (jb > n) jb = n;
                for (uint32_t j = ja > 0 ? ja : 1u; j < jb; ++j) {
                    keys[i0 + j] = ((unsigned long long)t << 32) | ((unsigned long long)f << 16) | j;
                    t1s[i0 + j - 1] = t;
                }
            }
            base += (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    // issue order: by first tile (then tap, then j); keys are distinct, so an item's position is the number of smaller
    // keys (every thread reads the same LDS words: broadcasts) -- a bitonic sort spent ~70 barriers here
    for (uint32_t i = tid; i < total; i += 1024) {
        const unsigned long long k = keys[i];
        // position = number of smaller keys.  The keys of one tap ascend with j (first tiles do), so that number is a
        // sum of ntap binary searches (~6 steps each) instead of a pass over all `total` keys (1 024 reads per item:
        // 35 of the kernel's 78 us)
        uint32_t pos = 0;
        for (int g = 0; g < ntap; ++g) {
            const uint32_t base = ibeg[g];
            uint32_t lo = 0, hi = nf[g];
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (keys[base + mid] < k) lo = mid + 1;
                else hi = mid;
            }
            pos += lo;
        }
        const uint32_t f = (uint32_t)(k >> 16) & 0xFFFFu, j = (uint32_t)k & 0xFFFFu;
        const uint32_t slot = ibeg[f] + j;
        items[pos] = make_uint4(f, (uint32_t)(k >> 32), t1s[slot], slot);
    }
}

// grad_filter[f] = sum of tap f's item partials (ascending j: fixed order) + the generic kernel's contribution
// (partial slots are [CINP][COUTP] of the padded instantiation; grad_filter and `extra` have the real layout)
__global__ __launch_bounds__(256) void deep_reduce_kernel(const float *__restrict__ partials,
                                                          const uint2 *__restrict__ tap_rng,
                                                          const float *__restrict__ extra, int per_tap, int coutp,
                                                          int cin, int cout, float *__restrict__ grad_filter)
{
    const int f = blockIdx.y;
    const int er = blockIdx.x * 256 + threadIdx.x;       // element of the real [cin][cout] block
    if (er >= cin * cout) return;
    const int e = (er / cout) * coutp + (er % cout);      // ... and of the padded one
    const uint2 rg = tap_rng[f];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    uint32_t j = 0;
    for (; j + 4 <= rg.y; j += 4) {
        s0 += partials[(size_t)(rg.x + j) * per_tap + e];
        s1 += partials[(size_t)(rg.x + j + 1) * per_tap + e];
        s2 += partials[(size_t)(rg.x + j + 2) * per_tap + e];
        s3 += partials[(size_t)(rg.x + j + 3) * per_tap + e];
    }
    for (; j < rg.y; ++j) s0 += partials[(size_t)(rg.x + j) * per_tap + e];
    grad_filter[(size_t)f * cin * cout + er] = ((s0 + s1) + (s2 + s3)) + extra[(size_t)f * cin * cout + er];
}

// ---------------------------------------------------------------------------------------------
// deep_dw_kernel: grad_filter partials.  Workgroup = (work item = backward tap f and a range of query tiles,
// half of the output columns):
//   dW[f][0..CIN)[cols] = sum over the item's tiles of  X_tile^T [CIN x 64] . G_f[tile] [64 x cols]
// G_f[tile] (row j = sum over the pairs of centre j with backward tap f of dY[neighbour] / count) was stored by the
// grad_input pass of deep_gemm_kernel; rows of X by the centres' original indices.  32 MFMA k-steps per tile, the
// waves form a WM x WN grid over the CIN/32 x cols/32 output blocks, accumulators stay in registers over the whole
// item; one partial per item ([CIN][COUT], each half writes its columns), summed by deep_reduce_kernel in fixed order.
// LDS: X tile [64][CIN+1] | G tile [64][cols+1] | qorig [64]   (<= 66 KB: two workgroups per CU)
// ---------------------------------------------------------------------------------------------
// column parts of a work item (blockIdx.y): halves; quarters for 256 input channels, where a wave's 8 row blocks x 1
// column block of accumulators (128 registers) plus the operands of four k-steps did not fit three waves per SIMD (35
// spilled registers, 144 B of scratch per lane until round 5)
template <int CIN, int COUT> constexpr int deep_dw_parts() { return CIN >= 256 && COUT >= 128 ? 4 : COUT >= 64 ? 2 : 1; }
template <int CIN, int COUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DEEP_DW_WAVES))) void deep_dw_kernel(
    const PointRec<float> *__restrict__ pts, const uint32_t *__restrict__ tap_off, const float *__restrict__ gbuf,
    const float *__restrict__ input, int N, int ntiles, int ntap, const uint8_t *__restrict__ tile_flag,
    const uint4 *__restrict__ items, const uint32_t *__restrict__ nitems, float *__restrict__ partials,
    int cin,   // real input channels (<= CIN); the partial slots are [CIN][COUT]
    const unsigned long long *__restrict__ tap_cmask)   // [tile][tap] centres with records: the packed rows of gbuf
{
    constexpr int NH = deep_dw_parts<CIN, COUT>();          // column parts (blockIdx.y)
    constexpr int CH = COUT / NH;
    constexpr int LDX = CIN + 1, LDG = CH + 1;
    constexpr int MB = CIN / 32, NB = CH / 32;
    constexpr int WN = NB >= 4 ? 4 : NB, WM = 4 / WN;      // wave grid
    constexpr int PM = (MB + WM - 1) / WM, PN = NB / WN;   // blocks per wave along each axis
    static_assert(CIN % 32 == 0 && CH % 32 == 0 && NB % WN == 0, "deep path: channel counts are 32 or multiples of 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kRows = DEEP_DW_ROWS;                    // packed rows per pass
    float *X = reinterpret_cast<float *>(smem);            // [kRows][LDX]
    float *G = X + kRows * LDX;                            // [kRows][LDG]
    int32_t *qorig = reinterpret_cast<int32_t *>(G + kRows * LDG);
    if (blockIdx.x >= *nitems) return;
    const uint4 item = items[blockIdx.x];                  // {tap, first tile, end tile, partial slot}
    const int f = (int)item.x, c0 = (int)blockIdx.y * CH;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const bool w_on = wm * PM < MB;
    f32x16 acc[PM][PN];
#pragma unroll
    for (int i = 0; i < PM; ++i)
#pragma unroll
        for (int j = 0; j < PN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // A tile's packed rows (the centres that have the tap: 27 of 64 on average) go through LDS kRows at a time: with
    // room for all 64 the two tiles took 66 KiB (two workgroups per CU, 50 % matrix-pipe busy: a workgroup's loads are
    // exposed, only the CU's other workgroup covers them); 32 rows take 33 KiB, three workgroups per CU (the
    // registers' limit), and most tiles still need one pass.
    for (size_t tile = item.y; tile < item.z; ++tile) {
        const uint32_t *toff = tap_off + tile * (size_t)(ntap + 1);
        if (toff[f] == toff[f + 1] || tile_flag[tile]) continue;   // uniform: nothing with this tap / generic kernel's tile
        const int b = (int)(tile / ntiles);
        // the tile's centres that have tap f: only their rows of G_f' were stored (packed, centre order) and only they
        // enter the contraction; K is padded to a multiple of 8 with zero rows of G
        const unsigned long long cm = tap_cmask[tile * (size_t)ntap + f];
        const int nrow = __popcll(cm), npad = (nrow + 7) & ~7;
        __syncthreads();                                    // previous tile's X / G consumed (qorig too)
        if (wave == 0) {
            const int orig = pts[tile * kTile + lane].idx;
            if ((cm >> lane) & 1ull) qorig[__popcll(cm & ((1ull << lane) - 1ull))] = orig;   // packed row -> original index
        }
        for (int r0 = 0; r0 < npad; r0 += kRows) {
            const int cr = npad - r0 < kRows ? npad - r0 : kRows;   // rows of this pass (a multiple of 8)
            if (r0 > 0) __syncthreads();                    // previous pass consumed
            // G rows r0 .. r0 + cr of the stored block half (16-byte loads, all of a thread's in flight together)
            {
                const float *gt = gbuf + (tile * (size_t)ntap + f) * 64 * COUT + c0;
                constexpr int GPT = (kRows * (CH / 4)) / 256;   // float4 per thread
                float4 gv[GPT];
#pragma unroll
                for (int u = 0; u < GPT; ++u) {
                    const int e = (int)threadIdx.x + 256 * u;
                    const int rr = r0 + e / (CH / 4);
                    gv[u] = *reinterpret_cast<const float4 *>(gt + (size_t)(rr < nrow ? rr : 0) * COUT + 4 * (e % (CH / 4)));
                }
#pragma unroll
                for (int u = 0; u < GPT; ++u) {
                    const int e = (int)threadIdx.x + 256 * u;
                    const int rl = e / (CH / 4);
                    if (rl < cr) {
                        float *gr = G + rl * LDG + 4 * (e % (CH / 4));
                        const bool on = r0 + rl < nrow;
                        gr[0] = on ? gv[u].x : 0.f; gr[1] = on ? gv[u].y : 0.f; gr[2] = on ? gv[u].z : 0.f; gr[3] = on ? gv[u].w : 0.f;
                    }
                }
            }
            if (r0 == 0) __syncthreads();                   // qorig visible
            {
                constexpr int XPT = (kRows * (CIN / 4)) / 256;   // float4 per thread (CIN >= 32)
                float4 xv[XPT];
#pragma unroll
                for (int u = 0; u < XPT; ++u) {
                    const int e = (int)threadIdx.x + 256 * u;
                    const int rr = r0 + e / (CIN / 4);
                    const int orig = rr < nrow ? qorig[rr] : -1;
                    xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (orig >= 0) xv[u] = load_row4(input + ((size_t)b * N + orig) * cin, 4 * (e % (CIN / 4)), cin);
                }
#pragma unroll
                for (int u = 0; u < XPT; ++u) {
                    const int e = (int)threadIdx.x + 256 * u;
                    if (e / (CIN / 4) < cr) {
                        float *xr = X + (e / (CIN / 4)) * LDX + 4 * (e % (CIN / 4));
                        xr[0] = xv[u].x; xr[1] = xv[u].y; xr[2] = xv[u].z; xr[3] = xv[u].w;
                    }
                }
            }
            __syncthreads();
            if (w_on) {
                // A[i = k][kk = row] = X[row][k], B[kk = row][j = c] = G[row][c]: cr / 2 k-steps, 4 at a time
                const float *xa = X + half * LDX + wm * PM * 32 + (lane & 31);
                const float *gb = G + half * LDG + wn * PN * 32 + (lane & 31);
                for (int s4 = 0; s4 < cr / 2; s4 += 4) {
                    float a[4][PM], bv[4][PN];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int i = 0; i < PM; ++i) a[t][i] = xa[2 * (s4 + t) * LDX + i * 32];
#pragma unroll
                        for (int j = 0; j < PN; ++j) bv[t][j] = gb[2 * (s4 + t) * LDG + j * 32];
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < PM; ++i)
#pragma unroll
                            for (int j = 0; j < PN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][i], bv[t][j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    // partial slot of this item: [k][c]
    float *slot = partials + (size_t)item.w * CIN * COUT;
    if (w_on) {
#pragma unroll
        for (int i = 0; i < PM; ++i)
#pragma unroll
            for (int j = 0; j < PN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = (wm * PM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (k < CIN) slot[(size_t)k * COUT + c0 + (wn * PN + j) * 32 + (lane & 31)] = acc[i][j][r];
                }
    }
}

// filter [F][cin][cout] -> zero-padded B operand of deep_gemm_kernel:
//   transpose == 0: [F][kp][np] with rows k < cin, columns n < cout             (forward:   out = M_f . W[f])
//   transpose != 0: [F][kp][np] with rows k < cout, columns n < cin = W[f]^T    (grad_input: dX = G_f . W[f]^T)
__global__ __launch_bounds__(256) void pad_filter_kernel(const float *__restrict__ w, int ntap, int cin, int cout,
                                                         int kp, int np, int transpose, float *__restrict__ wp)
{
    const size_t n = (size_t)ntap * kp * np;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t f = e / ((size_t)kp * np), r = e % ((size_t)kp * np);
        const int k = (int)(r / np), c = (int)(r % np);
        float v = 0.0f;
        if (!transpose) { if (k < cin && c < cout) v = w[(f * cin + k) * cout + c]; }
        else if (k < cout && c < cin) v = w[(f * cin + c) * cout + k];
        wp[e] = v;
    }
}

}  // namespace conv3p
