// conv3p_backward_sparse.hpp -- Conv3pGrad accumulate (tf_conv3p_atrous.cpp:608-716) for the models' narrow layers, with the
// per-tile G matrix kept only where it is populated.
//
// backward_kernel (conv3p_kernels.hpp) builds G[(f', c)][j] = sum over the pairs (j, ii) with backward tap f' of
// dY[ii, c] / count dense: 27 taps x Cout rows x 64 centres = 63 KiB for 9 output channels, which allows two workgroups
// per CU -- 1 024 query tiles then run in two resident rounds and never share a CU with the next batch's search
// (35 KiB per workgroup) running beside them.  A centre only has pairs in the taps its neighbourhood populates:
// 9 of 27 at stride 1, 3-5 at strides 2-4.  Here G holds one row of Cout values per POPULATED
// (centre, tap) only:
//     slot(j, f') = tapbase[f'] + |{ centres j' < j of the tile with tap f' }|        (tap-major, centres ascending)
// from the per-centre sets of backward taps the search leaves in `qbm` (search_tile).  With 9 -> 9 channels: ~390 / 260 /
// 190 slots at strides 2 / 3 / 4 instead of 1 728, 37 KiB of LDS in all: FOUR workgroups per CU, one resident round.
//   prologue  tapmask[f'] (64-bit set of centres) by ballots over the centres' masks; taps are dealt to ROUNDS whose
//             slots fit the LDS capacity (one round for the models' strides >= 2; dense stride-1 tiles take two)
//   phase A   the centre-major walk of backward_kernel (4 sub-lanes per centre, records 3 / rows 2 steps ahead in
//             named register slots, equal-tap sub-lanes merged by lane swaps), RMW into G[slot][c]
//   phase B   thread = (f', c): dW[f', k, c] = sum over the tap's slots of G[slot][c] * X[j][k]   (slots of a tap are
//             consecutive, its centres the set bits of tapmask[f'])  -> this workgroup's partial
//   phase C   lane = centre, waves split the taps: dX[j, k] += sum_c G[slot(j, f')][c] * W[f', k, c]; accumulated in
//             registers across rounds, per-wave partial rows summed through LDS in fixed order
// No floating-point atomics: bitwise reproducible.  A tile whose pair segment overflowed searches itself (as in
// backward_kernel); its slots come from the same masks (the search computes them for such tiles too).
#pragma once

#ifndef CONV3P_SP_DEPTH_NARROW
#define CONV3P_SP_DEPTH_NARROW 3
#endif
#ifndef CONV3P_SP_DEPTH_WIDE
#define CONV3P_SP_DEPTH_WIDE 4   // phase A gather depth of the >= 16-input layers (2 waves per SIMD: 256 registers)
#endif
#ifndef CONV3P_SP_BLOCKED
#define CONV3P_SP_BLOCKED 1   // developer A/B: 0 = the lanes of a centre take its records interleaved (1: one run each)
#endif
#ifndef CONV3P_SP_CHSPLIT
#define CONV3P_SP_CHSPLIT 1   // developer A/B: 0 = a centre's sub-lanes take different records (merged by lane swaps)
#endif
#ifndef CONV3P_SP_FUSE_BC
#define CONV3P_SP_FUSE_BC 0   // developer A/B: 1 = phases B and C of a one-round tile interleaved tap by tap (9 -> 9 at the cfg2 size: 45.1 us against 45.8, but 2 spilled registers at the 128-register cap: not shipped)
#endif

namespace conv3p {

struct __attribute__((aligned(16))) TapInfo {
    uint32_t mask_lo, mask_hi;   // centres of the tile that have this backward tap
    uint32_t base;               // first row of G of the tap inside its round
    uint32_t gbase;              // first entry of the tap in the slot -> centre map (all rounds)
};

// LDS of the sparse kernel apart from G [cap][Cout]:
// tapmap | tapinfo[32] | rounds | qorig | slot -> centre map u8[64 * 32] | rinv | xt | soa     (red aliases G)
// (tap tables for 32 taps, or 64 for filters of 33 .. 64 taps: the kernel's BIG instantiation)
template <typename T> __host__ __device__ inline size_t sparse_fixed_lds(int maxfull, int ntap, int cin, int cout)
{
    const size_t maxt = ntap > 64 ? 128 : ntap > 32 ? 64 : 32;
    return (((size_t)3 * maxfull * 2 + 15) & ~(size_t)15) + maxt * sizeof(TapInfo) + (maxt + 32) * 4 + 256 + 64 * maxt +
           ((256 * sizeof(T) + 15) & ~(size_t)15) + (((size_t)64 * cin * sizeof(T) + 15) & ~(size_t)15) + (size_t)kWavesPerBlock * 192 * 4;
}
// G must also hold the final cross-wave sum [4][Cin][64]
template <typename T> __host__ __device__ inline size_t sparse_min_rows(int cin, int cout)
{
    return ((size_t)kWavesPerBlock * cin * 64 + cout - 1) / cout;
}

// The pass of ONE query tile; the whole workgroup calls it.  live: the workgroup has a tile (else it only zeroes its
// grad_filter partial); slot_idx: index of the workgroup's partial; sync: the hand-off points of a fused stack launch
// (conv3p_stack_fused.hpp) or NoSync.
template <typename T, int CIN, int COUT, int BIG, class Sync>   // BIG: 1 = filters of 33 .. 64 taps (64-bit tap sets), 2 = 65 .. 128 taps (128-bit); narrow layers only
__device__ __forceinline__ void backward_sparse_tile(
    const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes, const int32_t *__restrict__ count,
    const PairEntry *__restrict__ pairs, const uint2 *__restrict__ segs, const uint2 *__restrict__ qsegs,
    const uint32_t *__restrict__ qbm, const uint32_t *__restrict__ qbm_hi, const T *__restrict__ grad_out, const T *__restrict__ input,
    const T *__restrict__ filter, const Stencil<T> &st, int N, int ntiles, int ngroups, T *__restrict__ grad_input,
    T *__restrict__ partials, int act, const T *__restrict__ addend, const T *__restrict__ cmin, RowLd ld,
    int cap,   // G rows (slots) the LDS allocation holds; >= 64, so every tap fits a round of its own
    bool live, int b, int qt, unsigned slot_idx, const Sync &sync)
{
    // (fused stack launch: the same code once per layer in one kernel -- thread-index-derived values must not stay live across
    // the layers, see forward_tile)
    uint32_t tid_ = threadIdx.x;
    if constexpr (Sync::kActive) asm volatile("" : "+v"(tid_));
    const uint32_t tid = tid_;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    constexpr int kMaxT = BIG == 2 ? 128 : BIG ? 64 : 32;         // taps the tables hold
    using BM = std::conditional_t<BIG == 2, unsigned __int128, std::conditional_t<BIG == 1, uint64_t, uint32_t>>;   // set of backward taps of a centre
    static_assert(!BIG || CIN < 16, "64-bit tap sets: narrow layers only (phase C of the wide ones shuffles 32-bit sets)");
    TapInfo *tapinfo = reinterpret_cast<TapInfo *>(smem + off);
    off += kMaxT * sizeof(TapInfo);
    uint32_t *rinfo = reinterpret_cast<uint32_t *>(smem + off);   // [0] rounds, [1 + r] first tap of round r, ... (<= taps + 2 entries)
    off += (kMaxT + 32) * 4;
    int32_t *qorig = reinterpret_cast<int32_t *>(smem + off);
    off += 256;
    uint8_t *sj = reinterpret_cast<uint8_t *>(smem + off);        // centre of every (tap, slot), taps ascending
    off += 64 * kMaxT;
    T *rinv = reinterpret_cast<T *>(smem + off);                  // rinv[n] = 1 / (T)n for n < 256
    off += align16(256 * sizeof(T));
    const size_t nw = (size_t)st.ntap * CIN * COUT;
    T *xt = reinterpret_cast<T *>(smem + off);                    // X tile [64][CIN]
    off += align16((size_t)64 * CIN * sizeof(T));
    const int wave = tid >> 6, lane = tid & 63;
    float *soa = reinterpret_cast<float *>(smem + off) + wave * 192;
    off += align16((size_t)kWavesPerBlock * 192 * 4);
    T *G = reinterpret_cast<T *>(smem + off);                     // [cap][COUT]
    T *red = G;                                                   // [4][CIN][64]: ALIASES G (used after the last round)
    // narrow layers: sub-lanes of a centre 16 lanes apart (their merge swaps 16- / 32-lane blocks); wide layers: ADJACENT
    // lanes, so that their pieces of one dY row and their common record form one access for the texture addresser
    // narrow layers: the lanes of the wave are dealt to its 16 centres in proportion to their lists (share_lanes; four
    // consecutive lanes per centre when the cloud was searched in several groups) and every lane walks its own run of
    // the list; wide layers: four ADJACENT lanes walk a centre's list together and split the output channels, so that
    // their pieces of one dY row and their common record form one access for the texture addresser
    constexpr bool kChSplit = CONV3P_SP_CHSPLIT && CIN >= 16;
#ifndef CONV3P_SP_HALVES
#define CONV3P_SP_HALVES 1   // developer A/B: 0 = tiles over the capacity take rounds by taps (every list walked once per round)
#endif
    constexpr bool kHalves = CONV3P_SP_HALVES && kChSplit && sizeof(T) == 4 && CIN >= 16;   // (needs phase C on the matrix cores)
    int cq = wave * 16 + (lane >> 2);
    uint32_t sub = lane & 3u, maxn = 4u;
    uint32_t *share = reinterpret_cast<uint32_t *>(soa);   // the wave's lane-sharing scratch (soa is the overflow path's)

    DEV_SP_DECL()
    SDBG()
    build_tapmap(tapmap, st.full, st.step, st.maxfull);
    rinv[tid] = (T)1 / (T)(int)tid;

    const PointRec<T> *cloud_pts = pts + (size_t)(live ? b : 0) * ntiles * kTile;
    PointRec<T> me = cloud_pts[(size_t)(live ? qt : 0) * kTile + lane];
    if (!live) me.idx = -1;
    const size_t tile_id = (size_t)(live ? b : 0) * ntiles + (live ? qt : 0);
    // loads that need nothing from LDS go out now, under the prologue: the tile's tap sets, its segment records (is
    // the tile's pair list complete?), this lane's centre segment and the first records of its list
    BM bm_raw = qbm[tile_id * 64 + lane];
    if constexpr (BIG >= 1) bm_raw |= (BM)qbm_hi[tile_id * 64 + lane] << 32;
    if constexpr (BIG == 2) {   // planes 2 and 3 follow plane 1 at the planes' common stride (qbm_hi - qbm)
        const size_t pstride = (size_t)(qbm_hi - qbm);
        bm_raw |= (BM)qbm_hi[pstride + tile_id * 64 + lane] << 64;
        bm_raw |= (BM)qbm_hi[2 * pstride + tile_id * 64 + lane] << 96;
    }
    bool overflow = false;
    for (int g = 0; g < ngroups; ++g) overflow |= segs[tile_id * ngroups + g].y == kSegOverflow;
    LaneShare ls0;
    if (!kChSplit && ngroups == 1) {
        ls0 = share_lanes(qsegs[tile_id * 64 + wave * 16 + (lane & 15)], share);
        cq = wave * 16 + (int)ls0.cl;
        sub = ls0.r;
        maxn = ls0.maxn;
    } else {
        ls0 = share_lanes_uniform(qsegs[(tile_id * ngroups) * 64 + cq]);
    }
    const uint2 sg0 = ls0.seg;
    const LaneWalk wk0 = lane_walk(ls0, CONV3P_SP_BLOCKED != 0);
    constexpr int kDepth = CIN < 16 ? CONV3P_SP_DEPTH_NARROW : CONV3P_SP_DEPTH_WIDE;   // records in flight per lane in phase A
    PairEntry rec0[kDepth - 1];
#pragma unroll
    for (int sl = 0; sl < kDepth - 1; ++sl) {
        const uint32_t i0 = kChSplit ? (uint32_t)sl : wk0.i + wk0.step * (uint32_t)sl;
        const uint32_t e0 = kChSplit ? sg0.y : wk0.end;
        rec0[sl] = pairs[sg0.x + (i0 < e0 && sg0.y != kSegOverflow ? i0 : 0u)];
    }
    if (wave == 0) {
        qorig[lane] = me.idx;
        const T *xr = input + ((size_t)(live ? b : 0) * N + (me.idx < 0 ? 0 : me.idx)) * ld.in;
        T xv[CIN];
        RowLoader<T, CIN>::load(xr, xv);
#pragma unroll
        for (int k = 0; k < CIN; ++k) xt[lane * CIN + k] = me.idx >= 0 ? xv[k] : (T)0;
    }
    // ---- which (centre, tap) rows exist, and in which round each tap is handled.  Every wave runs the same scalar
    // bookkeeping over all taps (ballots + counts) and publishes the taps f == wave (mod 4): no serial section.
    const BM mybm_all = live && me.idx >= 0 ? bm_raw : (BM)0;     // lane = centre in every wave
    BM mybm = mybm_all;                                            // ... restricted to the centres of the running half (below)
    auto build_rows = [&](BM bmv) {
        uint32_t base = 0, gbase = 0, nrounds = 1;
        for (int f = 0; f < st.ntap; ++f) {   // (ntap <= kMaxT: host)
            const bool has = (uint32_t)(bmv >> f) & 1u;
            const uint64_t m = __ballot(has);
            const uint32_t n = (uint32_t)__popcll(m);
            if (base + n > (uint32_t)cap) {
                if (tid == 0) rinfo[1 + nrounds] = (uint32_t)f;
                ++nrounds;
                base = 0;
            }
            if ((f & (kWavesPerBlock - 1)) == wave) {
                if (lane == 0) {
                    TapInfo ti;
                    ti.mask_lo = (uint32_t)m;
                    ti.mask_hi = (uint32_t)(m >> 32);
                    ti.base = base;
                    ti.gbase = gbase;
                    tapinfo[f] = ti;
                }
                if (has) sj[gbase + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = (uint8_t)lane;
            }
            base += n;
            gbase += (n + 3u) & ~3u;   // (a tap's entries start 4-byte aligned: phase B reads four centres per LDS load)
        }
        if (tid == 0) {
            rinfo[0] = nrounds;
            rinfo[1] = 0;
            rinfo[1 + nrounds] = (uint32_t)st.ntap;
        }
    };
    build_rows(mybm);
    __syncthreads();
    if (!live) {   // (uniform) a workgroup past the last cloud: its grad_filter partial is summed like the others
        T *z = partials + (size_t)slot_idx * nw;
        for (uint32_t e = tid; e < (uint32_t)nw; e += blockDim.x) z[e] = (T)0;
        return;
    }

    const int32_t *cnt_cloud = count + (size_t)b * N * st.ntap;
    const T *dy_cloud = grad_out + (size_t)b * N * ld.dy;
    const uint64_t lt_cq = cq == 0 ? 0ull : (~0ull >> (64 - cq));
    const int orig_lane = me.idx;   // (the rest of `me` is only needed by the overflow path, which reloads it)
    const int nrounds = (int)rinfo[0];
    // row of G for (centre of this lane, backward tap fb); fb must belong to the running round
    auto slot_of = [&](uint32_t fb, uint64_t lower) {
        const TapInfo ti = tapinfo[fb];
        const uint64_t m = ((uint64_t)ti.mask_hi << 32) | ti.mask_lo;
        return ti.base + (uint32_t)__popcll(m & lower);
    };
    auto zero_G = [&](int t1) {
        // G = 0 for the round's slots
        {
            const TapInfo last = tapinfo[t1 - 1];
            const int used = (int)last.base + __popc(last.mask_lo) + __popc(last.mask_hi);
            float4 *G4 = reinterpret_cast<float4 *>(G);
            const int n4 = (int)(((size_t)used * COUT * sizeof(T) + 15) / 16);
            for (int e = tid; e < n4; e += blockDim.x) G4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // ---- phase A of the wide layers (>= 16 inputs; 36 -> 13 of the scene-segmentation stack).  Four ADJACENT lanes work on
    // one record and split the output channels (13: 4 + 4 + 4 + 1), so every lane owns its entries of G: nothing to merge
    // across the lanes of a record.  The walk is bound by the CU's address units (a wave-level gather costs them 33-45
    // cycles whatever its width, profiles/HISTORY.md), so records, populations and rows of G are fetched for FOUR steps at
    // a time: sub-lane s of a centre loads record 4 k + s, its population and its row of G (one instruction each per
    // four steps instead of one per step) and the four lanes pass them round by quad broadcasts (v_mov_dpp quad_perm);
    // only the pieces of the dY rows are still one load per step -- 6 memory instructions per 4 steps instead of 12.
    // Groups are pipelined over two named slots: the records of group k + 2 and the gathers of group k + 1 are in
    // flight while group k is accumulated.
    //   h < 0:  the whole tile, 16 centres per wave (tiles whose populated rows fit G; `t0 .. t1` = every tap)
    //   h >= 0: the centres 32 h .. 32 h + 31, eight per wave, of a tile whose rows exceed the capacity.  Rounds by taps
    //           walk every list once per round; split by CENTRES, a list is walked once: lanes 32-63 take the same
    //           centres' ODD records while lanes 0-31 take the even ones.  Two records of one centre that meet on a row
    //           of G in one step take turns, the even record first (a wave's LDS accesses execute in program order):
    //           race-free, in a fixed order.
    auto phase_A_wide = [&](int h, int t0, int t1) {
        constexpr int CPL = (COUT + 3) / 4;                       // channels per sub-lane
        const bool two = h >= 0;                                  // (uniform)
        const uint32_t nstr = two ? 2u : 1u;
        const uint32_t strm = two ? (uint32_t)lane >> 5 : 0u;
        const int cqw = two ? 32 * h + 8 * wave + ((lane & 31) >> 2) : wave * 16 + (lane >> 2);
        const uint32_t subw = (uint32_t)lane & 3u;
        const uint64_t lt_w = cqw == 0 ? 0ull : (~0ull >> (64 - cqw));
        const int c0 = (int)subw * CPL;
        const int nc = COUT - c0 < 0 ? 0 : (COUT - c0 < CPL ? COUT - c0 : CPL);
        constexpr int kShiftLast = 4 * CPL - COUT;                // the last piece is read from the row's end backwards
        const int start = c0 < COUT - CPL ? c0 : COUT - CPL;
        for (int g = 0; g < ngroups; ++g) {
            const uint2 sg = qsegs[(tile_id * ngroups + g) * 64 + cqw];
            const PairEntry *pe = pairs + sg.x;
            PairEntry rec[2];
            uint32_t fbv[2], row[2];
            bool lv[2];
            int cn[2];
            T val[2][4][CPL];
            auto own = [&](uint32_t k) { return strm + nstr * (4u * k + subw); };   // this lane's record of group k
            auto ld_rec = [&](int sl, uint32_t k) {
                const uint32_t i = own(k);
                rec[sl] = pe[i < sg.y ? i : 0u];
            };
            auto gather = [&](int sl, uint32_t k) {
                const uint32_t fb = code_bwd(rec[sl].code);
                const bool l = own(k) < sg.y && code_fwd(rec[sl].code) != kNoTap && fb != kNoTap && (int)fb >= t0 && (int)fb < t1;
                lv[sl] = l;
                fbv[sl] = l ? fb : kTurnIdle;
                row[sl] = slot_of(l ? fb : (uint32_t)t0, lt_w);
                cn[sl] = cnt_cloud[(l && !(CONV3P_SP_ABLATE & 2048)) ? (size_t)rec[sl].cand * st.ntap + fb : (size_t)0];
                const uint32_t cand = (l && !(CONV3P_SP_ABLATE & 2048)) ? rec[sl].cand : 0u;   // (2048: developer timing, every dY piece an L1 hit)
                RowLoader<T, CPL>::load(dy_cloud + (size_t)quad_bcast<0>(cand) * ld.dy + start, val[sl][0]);
                RowLoader<T, CPL>::load(dy_cloud + (size_t)quad_bcast<1>(cand) * ld.dy + start, val[sl][1]);
                RowLoader<T, CPL>::load(dy_cloud + (size_t)quad_bcast<2>(cand) * ld.dy + start, val[sl][2]);
                RowLoader<T, CPL>::load(dy_cloud + (size_t)quad_bcast<3>(cand) * ld.dy + start, val[sl][3]);
            };
            auto accumulate = [&](int sl) {
                const bool pend_own = lv[sl] & (cn[sl] != 0);                                   // .cpp:679
                const uint32_t fb_own = pend_own ? fbv[sl] : kTurnIdle;
                const T rcp_own = cn[sl] < 256 ? rinv[cn[sl] > 0 ? cn[sl] : 0] : (T)1 / (T)cn[sl];   // .cpp:692, :696
                // the group's four records of this centre: row of G (a value no other step has where the step is idle) and
                // this lane's piece of dY / count
                uint32_t rw[4];
                bool pd[4];
                T x[4][CPL];
                auto prep = [&](auto uc) {
                    constexpr int U = decltype(uc)::value;
                    pd[U] = quad_bcast<U>(fb_own) != kTurnIdle;
                    rw[U] = pd[U] ? quad_bcast<U>(row[sl]) : 0xFFFFFFF0u + (uint32_t)U;
                    const T rcpb = __builtin_bit_cast(T, quad_bcast<U>(__builtin_bit_cast(uint32_t, rcp_own)));
#pragma unroll
                    for (int c = 0; c < CPL; ++c) x[U][c] = ((subw == 3u && c + kShiftLast < CPL) ? val[sl][U][(c + kShiftLast) % CPL] : val[sl][U][c]) * rcpb;
                };
                prep(std::integral_constant<int, 0>{});
                prep(std::integral_constant<int, 1>{});
                prep(std::integral_constant<int, 2>{});
                prep(std::integral_constant<int, 3>{});
                // ONE LDS round trip for the four steps (it was one per step, each waiting for the one before: 100 of the
                // heaviest tile's 153 us, profiles/r05_wide_phaseA.txt): the four rows are read together, the sums are formed
                // in record order in registers -- a step whose row an earlier step of the group has already updated takes
                // that step's value, not the stale one read -- and written back in order (the last write of a row holds
                // its total).  Same additions in the same order as the step-by-step form.
                auto rmw = [&]() {
                    if (CONV3P_SP_ABLATE & 1024) {          // developer timing: no read-modify-write at all
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int c = 0; c < CPL; ++c) asm volatile("" :: "v"(x[u][c]));
                        return;
                    }
                    T g[4][CPL];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const T *gr = G + (size_t)(pd[u] ? rw[u] : 0u) * COUT + c0;   // (idle step: some valid row, value unused)
#pragma unroll
                        for (int c = 0; c < CPL; ++c) g[u][c] = gr[c < nc ? c : 0];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int v = 0; v < u; ++v) {       // (ascending: the latest earlier step on the same row wins)
                            const bool same = rw[v] == rw[u];
#pragma unroll
                            for (int c = 0; c < CPL; ++c) g[u][c] = same ? g[v][c] : g[u][c];
                        }
#pragma unroll
                        for (int c = 0; c < CPL; ++c) g[u][c] += x[u][c];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (pd[u]) {
                            T *gw = G + (size_t)rw[u] * COUT + c0;
#pragma unroll
                            for (int c = 0; c < CPL; ++c)
                                if (c < nc) gw[c] = g[u][c];
                        }
                };
                if (!two) {
                    rmw();
                } else {
                    // the even records' group first, then the odd records': the two may meet on a row
                    if (strm == 0u) rmw();
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (strm == 1u) rmw();
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            };
            ld_rec(0, 0u);
            ld_rec(1, 1u);
            __builtin_amdgcn_sched_barrier(0);
            gather(0, 0u);
            uint32_t k = 0;
            bool more = true;
            while (more) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!__any(strm + nstr * 4u * k < sg.y)) {
                        more = false;
                        break;
                    }
                    gather(j ^ 1, k + 1u);        // (its records were requested an iteration ago)
                    ld_rec(j, k + 2u);            // (group k's records are no longer needed: its gathers are out)
                    __builtin_amdgcn_sched_barrier(0);
                    accumulate(j);
                    k += 1u;
                }
            }
        }
    };
    auto phase_A = [&](int t0, int t1) {
        // ---- phase A
        if (CONV3P_SP_ABLATE & 1) {
        } else if (!overflow && kChSplit) {
            phase_A_wide(-1, t0, t1);
        } else if (!overflow) {
            for (int g = 0; g < ngroups; ++g) {
                LaneShare lsg = ls0;
                if (g > 0) lsg.seg = qsegs[(tile_id * ngroups + g) * 64 + cq];
                const uint2 sg = lsg.seg;
                const LaneWalk wk = g == 0 ? wk0 : lane_walk(lsg, CONV3P_SP_BLOCKED != 0);
                const PairEntry *pe = pairs + sg.x;
                auto live_rec = [&](const PairEntry &rc, uint32_t i) {
                    const uint32_t fb = code_bwd(rc.code);
                    return i < wk.end && code_fwd(rc.code) != kNoTap && fb != kNoTap && (int)fb >= t0 && (int)fb < t1;
                };
                PairEntry rec[kDepth];
                bool lv[kDepth];
                int cn[kDepth];
                uint32_t row[kDepth];   // G row of the record's (centre, tap): looked up when the record arrives
                T val[kDepth][COUT];
                auto ld_rec = [&](uint32_t i) { return pe[i < wk.end ? i : 0u]; };
                auto gather = [&](int sl, uint32_t i) {
                    lv[sl] = live_rec(rec[sl], i);
                    row[sl] = slot_of(lv[sl] ? code_bwd(rec[sl].code) : (uint32_t)t0, lt_cq);
                    cn[sl] = cnt_cloud[lv[sl] && !(CONV3P_SP_ABLATE & 32) ? (size_t)rec[sl].cand * st.ntap + code_bwd(rec[sl].code) : (size_t)0];
                    if (CONV3P_SP_ABLATE & 64) {   // developer: one 16-byte load instead of the whole row (timing only)
                        T four[4];
                        RowLoader<T, 4>::load(dy_cloud + (size_t)(lv[sl] ? rec[sl].cand : 0u) * ld.dy, four);
#pragma unroll
                        for (int c = 0; c < COUT; ++c) val[sl][c] = four[c & 3];
                    } else
                    RowLoader<T, COUT>::load(dy_cloud + (size_t)(lv[sl] && !(CONV3P_SP_ABLATE & 32) ? rec[sl].cand : 0u) * ld.dy, val[sl]);
                };
#pragma unroll
                for (int sl = 0; sl < kDepth - 1; ++sl) rec[sl] = g == 0 ? rec0[sl] : ld_rec(wk.i + wk.step * sl);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sl = 0; sl < kDepth - 2; ++sl) gather(sl, wk.i + wk.step * sl);
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
                        __builtin_amdgcn_sched_barrier(0);
                        const bool pending = lv[j] & (cn[j] != 0);                             // .cpp:679
                        const uint32_t fb = code_bwd(rec[j].code);
                        T(&v)[COUT] = val[j];
                        if (pending) {
                            const T rcpb = cn[j] < 256 ? rinv[cn[j]] : (T)1 / (T)cn[j];        // .cpp:692, :696
#pragma unroll
                            for (int c = 0; c < COUT; ++c) v[c] *= rcpb;
                        }
                        // Lanes of one centre that meet on a tap take TURNS at its row of G, lower lane first: turn = number
                        // of lower lanes of the centre with the same tap; a wave's LDS accesses execute in program order, so
                        // turn p adds to what turn p - 1 wrote -- race-free, in a fixed order.
                        {
                            const int turn = turn_among_lower_lanes(pending ? fb : kTurnIdle, sub, (int)maxn);
                            T *grow = G + (size_t)row[j] * COUT;
                            if (CONV3P_SP_ABLATE & 256) {          // developer timing: no read-modify-write at all
                                if (pending) {
#pragma unroll
                                    for (int c = 0; c < COUT; ++c) asm volatile("" :: "v"(v[c]));
                                }
                            } else if (CONV3P_SP_ABLATE & 512) {   // developer timing: one (racy) round whatever the taps
                                if (pending) {
#pragma unroll
                                    for (int c = 0; c < COUT; ++c) grow[c] += v[c];
                                }
                            } else
                            for (int p = 0; p < (int)maxn; ++p) {
                                if (p > 0 && !__any(pending && turn >= p)) break;
                                if (pending && turn == p) {
#pragma unroll
                                    for (int c = 0; c < COUT; ++c) grow[c] += v[c];
                                }
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            }
                        }
                        i += wk.step;
                    }
                }
            }
        } else {
            // pair buffer was full for this tile: search it here, lane = centre.  Every wave walks all candidate
            // tiles and keeps the taps f' == wave (mod 4): one writer per row of G.
            const T *cloud_box = boxes + (size_t)b * ntiles * 6;
            const PointRec<T> mine = cloud_pts[(size_t)qt * kTile + lane];
            const uint64_t lt_lane = lane == 0 ? 0ull : (~0ull >> (64 - lane));
            Query<T> q;
            make_query(q, mine, st);
            Window<T> win{};
            if (cmin != nullptr) make_window(win, mine, st, cmin + (size_t)b * 3);
            for_each_neighbor(cloud_pts, cloud_box, ntiles, q, st, tapmap, soa, 0, 1, [&](const PointRec<T> &v, int) {
                const uint32_t fb = backward_tap(q.p, v, st, tapmap);
                if (fb == kNoTap || (int)(fb & (kWavesPerBlock - 1)) != wave || (int)fb < t0 || (int)fb >= t1) return;
                const int cn = cnt_cloud[(size_t)v.idx * st.ntap + fb];
                if (cn == 0) return;                                                           // .cpp:679
                const T rcp = (T)1 / (T)cn;
                const T *dyr = dy_cloud + (size_t)v.idx * ld.dy;
                T *grow = G + (size_t)slot_of(fb, lt_lane) * COUT;
#pragma unroll
                for (int c = 0; c < COUT; ++c) grow[c] += dyr[c] * rcp;
            }, win, cmin != nullptr);
        }
    };
    auto phase_B = [&](int t0, int t1, bool accumulate = false) {
        T *slot_out = partials + (size_t)slot_idx * nw;
        // ---- phase B: dW[f'][k][c] = sum over the tap's slots of X[j(slot)][k] * G[slot][c].  fp32: on the matrix cores
        // (v_mfma_f32_16x16x4_f32, an exact fmaf chain: deterministic), wave w takes the round's taps f' == w (mod 4);
        // K = the tap's slots, four per step: A[i][kk] = X[j][i], B[kk][n] = G[slot][n], D[k][c] in registers over the tap.
        if (CONV3P_SP_ABLATE & 2) {
        } else if constexpr (sizeof(T) == 4) {
            const int l15 = lane & 15, l4 = lane >> 4;
            const float *Gf = reinterpret_cast<const float *>(G);
            const float *xtf = reinterpret_cast<const float *>(xt);
            float *so = reinterpret_cast<float *>(slot_out);
            for (int f = t0 + ((wave - t0) & (kWavesPerBlock - 1)); f < t1; f += kWavesPerBlock) {
                const TapInfo ti = tapinfo[f];
                const int n = __popc(ti.mask_lo) + __popc(ti.mask_hi);
                constexpr int NKB = (CIN + 15) / 16;   // blocks of 16 input channels (M of the product)
                f32x4 acc0[NKB], acc1[NKB];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    acc0[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc1[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // (second half of a tile split by centres: this lane wrote the first half's sums itself; they are requested now
                // and arrive under the tap's products)
                float old[NKB][4];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int k = kb * 16 + 4 * l4 + rr;
                        old[kb][rr] = accumulate ? so[((size_t)f * CIN + (k < CIN ? k : 0)) * COUT + (l15 < COUT ? l15 : 0)] : 0.0f;
                    }
                for (int s0 = 0; s0 < n; s0 += 16) {   // 16 slots per iteration: lane group l4 takes slots s0 + 4 l4 .. + 3
                    const int sb = s0 + 4 * l4;
                    const uint32_t cj4 = *reinterpret_cast<const uint32_t *>(sj + ti.gbase + (sb < n ? sb : 0));   // four centres
                    float av[4][NKB], bv[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        ok[u] = sb + u < n;
                        const int sl = ti.base + (ok[u] ? sb + u : 0);
                        const float *xr = xtf + ((cj4 >> (8 * u)) & 0xFFu) * CIN;
#pragma unroll
                        for (int kb = 0; kb < NKB; ++kb) av[u][kb] = xr[kb * 16 + l15 < CIN ? kb * 16 + l15 : 0];
                        bv[u] = Gf[(size_t)sl * COUT + (l15 < COUT ? l15 : 0)];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float bb = (ok[u] && l15 < COUT) ? bv[u] : 0.0f;
#pragma unroll
                        for (int kb = 0; kb < NKB; ++kb) {
                            const float a = (ok[u] && kb * 16 + l15 < CIN) ? av[u][kb] : 0.0f;
                            if (u & 1) acc1[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc1[kb], 0, 0, 0);
                            else acc0[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc0[kb], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int k = kb * 16 + 4 * l4 + rr;
                        if (k < CIN && l15 < COUT) {
                            const float sum = acc0[kb][rr] + acc1[kb][rr];
                            partial_store(&so[((size_t)f * CIN + k) * COUT + l15], accumulate ? old[kb][rr] + sum : sum);
                        }
                    }
            }
        } else {
            for (int row = t0 * COUT + (int)tid; row < t1 * COUT; row += blockDim.x) {
                const int f = row / COUT, c = row - f * COUT;
                const TapInfo ti = tapinfo[f];
                uint64_t m = ((uint64_t)ti.mask_hi << 32) | ti.mask_lo;
                const T *gp = G + (size_t)ti.base * COUT + c;
                T acc[CIN];
#pragma unroll
                for (int k = 0; k < CIN; ++k) acc[k] = (T)0;
                while (m != 0) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    const T g = *gp;
                    gp += COUT;
#pragma unroll
                    for (int k = 0; k < CIN; ++k) acc[k] = fma_t(g, xt[j * CIN + k], acc[k]);
                }
#pragma unroll
                for (int k = 0; k < CIN; ++k) partial_store(&slot_out[((size_t)f * CIN + k) * COUT + c], acc[k]);
            }
        }
    };
    auto phase_C = [&](int t0, int t1, T (&dx)[CIN]) {
        const uint64_t lt_lane = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        // ---- phase C: lane = centre, wave w takes the round's taps f' == w (mod 4)
        {
            for (int f = t0 + ((wave - t0) & (kWavesPerBlock - 1)); f < ((CONV3P_SP_ABLATE & 4) ? t0 : t1); f += kWavesPerBlock) {
                const bool has = (uint32_t)(mybm >> f) & 1u;
                if (!__any(has)) continue;
                const uint32_t s = has ? slot_of((uint32_t)f, lt_lane) : 0u;
                T g[COUT];
#pragma unroll
                for (int c = 0; c < COUT; ++c) g[c] = has ? G[(size_t)s * COUT + c] : (T)0;
                // W[f'][k][c] straight from the caller's filter: f' is wave-uniform, so these are scalar loads (the filter,
                // 8.7 KB for 9 -> 9, stays in the scalar cache) and the products take the weight as a scalar operand --
                // no copy of the filter in LDS, whose 8.7 KB hold 240 more rows of G
                const T *wf = filter + (size_t)f * CIN * COUT;
#pragma unroll
                for (int k = 0; k < CIN; ++k)
#pragma unroll
                    for (int c = 0; c < COUT; ++c) dx[k] = fma_t(g[c], wf[k * COUT + c], dx[k]);
            }
        }
    };
    // ---- phase C of the wide layers (fp32, >= 16 inputs) on the matrix cores.  The vector-ALU form above fetches its
    // 36 x 13 weights per tap through the scalar path in batches of 16 and waits for each: 36 us of a 36 -> 13 tile's
    // 130 on the rooms, 121 of the heaviest tile's 460 (profiles/r05_phase_trace_36_13.txt) -- the tile the whole launch
    // waits for.  Here: dX[centre][k] += sum_c G[slot(centre, f')][c] . W[f'][k][c] as v_mfma_f32_16x16x4_f32 with
    // M = centres (four blocks of 16), N = input channels (blocks of 16), K = output channels four at a time;
    // A[i][kk] = G[slot][4 cs + kk] (0 where the centre has no row for the tap), B[kk][n] = W[f'][16 kb + n][4 cs + kk]
    // from a [Cin][4] slice of the tap's filter block staged in the wave's scratch (one 16-byte load per input channel).
    // A wave takes the taps f' == wave (mod 4) for all 64 centres, as above; accumulators stay in registers across the
    // taps and the rounds.  An exact fmaf chain per output: deterministic.
#ifndef CONV3P_SP_MFMA_C_NARROW
#define CONV3P_SP_MFMA_C_NARROW 0   // developer A/B: 1 = the 9-input layers take it too (measured: 46.3 against 46.1 us, no gain)
#endif
    constexpr bool kMfmaC = sizeof(T) == 4 && (CIN >= 16 || (CONV3P_SP_MFMA_C_NARROW && CIN == 9));
    constexpr int NKC = (CIN + 15) / 16;
    // (the narrow layers, at 128 registers, keep the accumulators only where a tile takes ONE round -- every tile of the
    // models' dilated strides: across rounds they would sit in registers through phase A)
    constexpr bool kMfmaCRounds = sizeof(T) == 4 && CIN >= 16;
    auto zero_acc = [&](f32x4 (&acc)[4][NKC]) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int kb = 0; kb < NKC; ++kb) acc[mb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // D fragment -> the waves' partial rows: register r of lane l is D[centre = 16 mb + 4 (l >> 4) + r][k = 16 kb + (l & 15)]
    auto red_from_acc = [&](const f32x4 (&acc)[4][NKC]) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int kb = 0; kb < NKC; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * kb + (lane & 15);
                    if (k < CIN) red[((size_t)wave * CIN + k) * 64 + 16 * mb + 4 * (lane >> 4) + r] = acc[mb][kb][r];
                }
    };
    auto phase_C_mfma = [&](int t0, int t1, f32x4 (&accC)[4][NKC], int half = -1) {   // half >= 0: only its centres have rows
        if constexpr (kMfmaC) {
            const int l15 = lane & 15, l4 = lane >> 4;
            // tap sets and lower-centre masks of the four centres this lane feeds into the A operand (computed here, not
            // kept across phase A: the narrow layers run at 128 registers)
            uint32_t bmc[4];
            uint64_t ltc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int ci = 16 * mb + l15;
                bmc[mb] = (uint32_t)__shfl((int)mybm, ci);
                ltc[mb] = ci == 0 ? 0ull : (~0ull >> (64 - ci));
            }
            const float *Gf = reinterpret_cast<const float *>(G);
            const float *ff = reinterpret_cast<const float *>(filter);
            constexpr int NCS = (COUT + 3) / 4;
            // B operand of (tap, channel slice cs, input block kb): B[kk = l4][n = l15] = W[f'][16 kb + l15][4 cs + l4], read
            // straight from the caller's filter (50 KB for 36 -> 13: cache resident) into registers, the NEXT tap's twelve
            // values requested before this tap's products (two named slots).  (Round 5, first form: a [Cin][4] slice staged
            // in the wave's scratch per (tap, cs) -- one exposed memory latency each, 46 us of the heaviest tile's 260.)
            float bw[2][NCS][NKC];
            auto ld_w = [&](int sl, int f) {
#pragma unroll
                for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
                    for (int kb = 0; kb < NKC; ++kb) {
                        const int k = 16 * kb + l15, c = 4 * cs + l4;
                        const bool ok = k < CIN && c < COUT;
                        bw[sl][cs][kb] = ff[((size_t)f * CIN + (ok ? k : 0)) * COUT + (ok ? c : 0)];
                    }
            };
            const int tend = (CONV3P_SP_ABLATE & 4) ? t0 : t1;
            int f = t0 + ((wave - t0) & (kWavesPerBlock - 1));
            if (f < tend) ld_w(0, f);
            bool more = f < tend;
            while (more) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (f >= tend) {
                        more = false;
                        break;
                    }
                    if (f + kWavesPerBlock < tend) ld_w(j ^ 1, f + kWavesPerBlock);   // (uniform)
                    const TapInfo ti = tapinfo[f];
                    const uint64_t m = ((uint64_t)ti.mask_hi << 32) | ti.mask_lo;
                    if (m != 0ull) {                               // (uniform)
                        uint32_t slot[4];
                        bool has[4];
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) {
                            has[mb] = (bmc[mb] >> f) & 1u;
                            slot[mb] = has[mb] ? ti.base + (uint32_t)__popcll(m & ltc[mb]) : 0u;
                        }
#pragma unroll
                        for (int cs = 0; cs < NCS; ++cs) {
                            const int c = 4 * cs + l4;
                            float bv[NKC];
#pragma unroll
                            for (int kb = 0; kb < NKC; ++kb) bv[kb] = (16 * kb + l15 < CIN && c < COUT) ? bw[j][cs][kb] : 0.0f;
#pragma unroll
                            for (int mb = 0; mb < 4; ++mb) {
                                if (half >= 0 && (mb >> 1) != half) continue;   // (uniform)
                                float a = Gf[(size_t)slot[mb] * COUT + (c < COUT ? c : 0)];
                                a = (has[mb] && c < COUT) ? a : 0.0f;
#pragma unroll
                                for (int kb = 0; kb < NKC; ++kb) accC[mb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[kb], accC[mb][kb], 0, 0, 0);
                            }
                        }
                    }
                    f += kWavesPerBlock;
                }
            }
        }
    };
    T dx[CIN];
    bool red_written = false;   // (uniform)
    // The SELU-gradient epilogue needs, per output element, the caller's addend (global, by original index) and the slope at
    // the layer's own input -- which is the X tile already in LDS.  The addend values of this thread's elements (k = wave,
    // wave + 4, ...; centre = lane) are requested before phase C of a one-round tile, so that the epilogue does not start
    // with two exposed gathers (round 6: reduce + store 3.9 us per wave, max 10-14, on every tile's path).
    constexpr int kEpi = (CIN * 64 + 255) / 256;
    constexpr bool kEarlyAddend = (CIN == 9 && COUT == 9) || CIN >= 16;   // (9 -> 3 at the 128-register cap: two spills)
    T addv[kEpi];
#pragma unroll
    for (int it = 0; it < kEpi; ++it) addv[it] = (T)0;
    const T *addend_row = addend + ((size_t)b * N + (orig_lane >= 0 ? orig_lane : 0)) * ld.add;   // (only dereferenced if addend)
#define CONV3P_SP_LOAD_ADDEND()                                                                     \
    if ((act & 1) && addend != nullptr) {                                                           \
        _Pragma("unroll") for (int it = 0; it < kEpi; ++it) {                                       \
            const int k_ = wave + kWavesPerBlock * it;                                              \
            addv[it] = addend_row[k_ < CIN ? k_ : 0];                                               \
        }                                                                                           \
    }
    SDBG()
    sync.wait();   // (fused stack launch: phase A gathers other tiles' grad rows of the layer above; every path below has a
                   // workgroup barrier between here and its first gather)
    if (nrounds == 1) {
        // the common case (every tile of the models' strides >= 2): nothing but phase A's own state is live across it
        zero_G(st.ntap);
        __syncthreads();
        SDBG()
        phase_A(0, st.ntap);
        SDBG()
        __syncthreads();
        SDBG()
#pragma unroll
        for (int k = 0; k < CIN; ++k) dx[k] = (T)0;
#if CONV3P_SP_FUSE_BC
        if constexpr (!kMfmaC) {
            // phases B and C tap by tap (both give wave w the taps f' == w (mod 4)): C's weights arrive through the scalar
            // path while B's products of the same tap are in the matrix pipe
            for (int f = wave; f < st.ntap; f += kWavesPerBlock) {
                phase_B(f, f + 1);
                phase_C(f, f + 1, dx);
            }
            SDBG()
        } else
#endif
        phase_B(0, st.ntap);
        SDBG()
        if constexpr (kEarlyAddend) { CONV3P_SP_LOAD_ADDEND() }
        if constexpr (kMfmaC) {
            f32x4 acc1[4][NKC];
            zero_acc(acc1);
            phase_C_mfma(0, st.ntap, acc1);
            SDBG()
            __syncthreads();   // red aliases G: every wave is done with phase C
            red_from_acc(acc1);
            red_written = true;
        } else {
#if !CONV3P_SP_FUSE_BC
            phase_C(0, st.ntap, dx);
#endif
            SDBG()
        }
    } else if (kHalves && !overflow) {
        // (uniform) more rows than G holds: the two halves of the centres one after the other, each with rows of its own
        if constexpr (kHalves) {
            f32x4 accR[4][NKC];
            zero_acc(accR);
            DEV_SP_ROUNDS_DECL()
            for (int h = 0; h < 2; ++h) {
                __syncthreads();   // the row bookkeeping and G of the previous half (of the whole tile) are no longer read
                mybm = (lane >> 5) == h ? mybm_all : (BM)0;
                build_rows(mybm);
                __syncthreads();
                const int nr = (int)rinfo[0];
                for (int r = 0; r < nr; ++r) {
                    const int t0 = (int)rinfo[1 + r], t1 = (int)rinfo[2 + r];
                    if (r > 0) __syncthreads();
                    zero_G(t1);
                    __syncthreads();
                    DEV_SP_ACC(rs_)
                    phase_A_wide(h, t0, t1);
                    DEV_SP_ACC(ra_)
                    __syncthreads();
                    DEV_SP_ACC(rs_)
                    phase_B(t0, t1, h > 0);
                    DEV_SP_ACC(rb_)
                    phase_C_mfma(t0, t1, accR, h);
                    DEV_SP_ACC(rc_)
                    DEV_SP_ROUND()
                }
            }
            DEV_SP_ROUNDS_PRINT(CIN, COUT, wave, lane, nrs_)
            __syncthreads();   // red aliases G: every wave is done with its last phase C
            red_from_acc(accR);
            red_written = true;
        }
    } else {
#pragma unroll
        for (int k = 0; k < CIN; ++k) dx[k] = (T)0;
        f32x4 accR[kMfmaCRounds ? 4 : 1][kMfmaCRounds ? NKC : 1];
        if constexpr (kMfmaCRounds) zero_acc(accR);
        DEV_SP_ROUNDS_DECL()
        for (int r = 0; r < nrounds; ++r) {
            const int t0 = (int)rinfo[1 + r], t1 = (int)rinfo[2 + r];
            if (r > 0) __syncthreads();   // the previous round's phases B / C are done with G
            zero_G(t1);
            __syncthreads();
            DEV_SP_ACC(rs_)
            phase_A(t0, t1);
            DEV_SP_ACC(ra_)
            __syncthreads();
            DEV_SP_ACC(rs_)
            phase_B(t0, t1);
            DEV_SP_ACC(rb_)
            if constexpr (kMfmaCRounds) phase_C_mfma(t0, t1, accR);
            else phase_C(t0, t1, dx);
            DEV_SP_ACC(rc_)
        }
        if constexpr (kMfmaCRounds) {
            __syncthreads();   // red aliases G: every wave is done with its last phase C
            red_from_acc(accR);
            red_written = true;
        }
        DEV_SP_ROUNDS_PRINT(CIN, COUT, wave, lane, nrounds)
    }
    // ---- grad_input rows: fixed-order sum of the four waves' partial rows
    if (!red_written) {
        __syncthreads();   // red aliases G: every wave is done with its last phase C
#pragma unroll
        for (int k = 0; k < CIN; ++k) red[((size_t)wave * CIN + k) * 64 + lane] = dx[k];
    }
    if constexpr (kEarlyAddend) {
        if (nrounds != 1) { CONV3P_SP_LOAD_ADDEND() }   // (tiles of several rounds: here)
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kEpi; ++it) {
            const int k = wave + kWavesPerBlock * it;   // (element e = threadIdx.x + 256 it: k = e >> 6, centre = lane)
            if (k < CIN) {
                T sum = red[((size_t)0 * CIN + k) * 64 + lane];
#pragma unroll
                for (int w = 1; w < kWavesPerBlock; ++w) sum += red[((size_t)w * CIN + k) * 64 + lane];
                if (orig_lane >= 0) {
                    const size_t rr = (size_t)b * N + orig_lane;
                    if (act & 2) sum += grad_input[rr * ld.dx + k];
                    // (xt[lane][k] IS input[rr][k]: the X tile loaded in the prologue)
                    if (act & 1) sum = (addend ? sum + addv[it] : sum) * selu_slope(xt[lane * CIN + k]);
                    grad_input[rr * ld.dx + k] = sum;
                }
            }
        }
    } else {   // (the other shapes, some of them at the 128-register cap: as before)
        __syncthreads();
        for (int e = tid; e < CIN * 64; e += blockDim.x) {
            const int k = e >> 6;   // e & 63 == lane
            T sum = red[((size_t)0 * CIN + k) * 64 + lane];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) sum += red[((size_t)w * CIN + k) * 64 + lane];
            if (orig_lane >= 0) {
                const size_t rr = (size_t)b * N + orig_lane;
                if (act & 2) sum += grad_input[rr * ld.dx + k];
                if (act & 1) sum = (addend ? sum + addend[rr * ld.add + k] : sum) * selu_slope(input[rr * ld.in + k]);
                grad_input[rr * ld.dx + k] = sum;
            }
        }
    }
    DEV_SP_PRINT(CIN, COUT, wave, lane, nrounds == 1)
    sync.arrive();
}

template <typename T, int CIN, int COUT, int BIG = 0>
// (four waves per SIMD -- four workgroups per CU -- for the models' 3- and 9-input layers; 6 and 12 inputs, SceneNN's first
// layer, which only gets here when dilated, would spill 2 / 16 registers under that cap; so would the 64-bit tap sets of BIG)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((!BIG && (CIN == 3 || CIN == 9)) ? 4 : 2))) void backward_sparse_kernel(
    const PointRec<T> *__restrict__ pts, const T *__restrict__ boxes, const int32_t *__restrict__ count,
    const PairEntry *__restrict__ pairs, const uint2 *__restrict__ segs, const uint2 *__restrict__ qsegs,
    const uint32_t *__restrict__ qbm, const uint32_t *__restrict__ qbm_hi, const T *__restrict__ grad_out, const T *__restrict__ input,
    const T *__restrict__ filter, Stencil<T> st, int N, int ntiles, int ngroups, BlockMap bm, T *__restrict__ grad_input,
    T *__restrict__ partials, int act, const T *__restrict__ addend, const T *__restrict__ cmin, RowLd ld, int cap,
    const uint32_t *__restrict__ sched,   // launch order of the tiles (tile_sched_kernel) or nullptr
    const uint32_t *__restrict__ regime)  // non-null: run only if the slot's lists are SHORT (*regime == 1, tile_sched_kernel);
                                          // the host then launches backward_kernel too, which runs in the other case
{
    if (regime != nullptr && *regime != 1u) return;   // (uniform)
    int b, qt;
    const bool live = block_to_tile(bm, sched, ntiles, b, qt);   // uniform for the workgroup
    backward_sparse_tile<T, CIN, COUT, BIG>(pts, boxes, count, pairs, segs, qsegs, qbm, qbm_hi, grad_out, input, filter, st, N, ntiles, ngroups,
                                            grad_input, partials, act, addend, cmin, ld, cap, live, b, qt, blockIdx.x, NoSync{});   // (BIG: 0 / 1 / 2)
}

}  // namespace conv3p
