// conv3p_deep.hpp -- deep-channel path (e.g. 128 -> 256, BASELINE config 5) on the matrix cores.
//
// For wide layers the per-pair form (Cin*Cout FMAs per neighbour pair) is wasteful; the op factorises as
//     out[i,:]  = sum_f  M_f[i,:]  . W[f]          M_f[i,:] = sum_{j in tap f of i} x[j,:] / count[i,f]
//     dX[j,:]   = sum_f' G_f'[j,:] . W[f']^T       G_f'[j,:] = sum_{ii: bwd tap f'} dY[ii,:] / count[ii,f']
//     dW[f']    = sum_j  X[j,:]^T . G_f'[j,:]
// (same sums as tf_conv3p_atrous.cpp:480-494 and :682-698, re-associated; inside the fp32 tolerance).
// M_f / G_f' are built per query tile and per tap in LDS from the pair lists (gather-reduce, lanes = channels,
// coalesced 4*K-byte rows), and every product is a [64 x K].[K x N] GEMM on v_mfma_f32_32x32x2_f32 -- exact
// fp32 (an fmaf chain), 64 FLOP/clk/SIMD.  Taps with no neighbour in the tile are skipped.
//
// Fragment layouts of mfma_f32_32x32x2f32 (cdna_hip_programming.md section 3):
//   A operand: lane l holds A[i = l & 31][k = l >> 5];  B operand: lane l holds B[k = l >> 5][j = l & 31];
//   C/D: register r of lane l is C[row = (r & 3) + 8*(r >> 2) + 4*(l >> 5)][col = l & 31].
#pragma once

#include "conv3p_device.hpp"

namespace conv3p {

typedef float f32x16 __attribute__((ext_vector_type(16)));


// Per-centre stable bucketing of a tile's pair records by tap (lane = centre, wave 0 only).
//   start[q*(ntap+1) + f] .. start[q*(ntap+1) + f+1]  : positions (relative to the centre's bucket base) of tap f
//   order[qbase[q] + p]                               : record index relative to the CENTRE's list start
//                                                       (u16: a centre has < 65536 neighbours); `order` lives in
//                                                       global scratch, one u16 per pair slot of the tile
// Returns (through total[f]) the number of records of each tap in the whole tile.
template <bool BWD>
__device__ __forceinline__ void bucket_by_tap(const PairEntry *__restrict__ seg, const uint2 *__restrict__ qseg_tile,
                                              uint32_t seg_start, int ntap, uint16_t *start,
                                              uint16_t *__restrict__ order, uint32_t *qbase, uint32_t *total)
{
    const int lane = threadIdx.x & 63;
    if ((threadIdx.x >> 6) == 0) {
        const uint2 sg = qseg_tile[lane];
        const uint32_t n = sg.y, rel = sg.x - seg_start;
        uint16_t *st = start + lane * (ntap + 1);
        for (int f = 0; f <= ntap; ++f) st[f] = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const PairEntry en = seg[rel + i];
            const uint32_t f = BWD ? code_bwd(en.code) : code_fwd(en.code);
            const bool ok = BWD ? (en.rcp_bwd > 0.0f) : (code_fwd(en.code) != kNoTap);
            if (ok) st[f + 1] += 1;
        }
        uint32_t run = 0;
        for (int f = 0; f < ntap; ++f) {     // counts -> exclusive starts (st[f+1] held the count of tap f)
            const uint32_t c = st[f + 1];
            st[f] = (uint16_t)run;
            run += c;
        }
        st[ntap] = (uint16_t)run;
        int tot;
        const uint32_t base = (uint32_t)wave_excl_scan((int)run, tot);
        qbase[lane] = base;
        // second pass: stable scatter (st[f] is advanced and restored afterwards)
        for (uint32_t i = 0; i < n; ++i) {
            const PairEntry en = seg[rel + i];
            const uint32_t f = BWD ? code_bwd(en.code) : code_fwd(en.code);
            const bool ok = BWD ? (en.rcp_bwd > 0.0f) : (code_fwd(en.code) != kNoTap);
            if (ok) {
                order[base + st[f]] = (uint16_t)i;
                st[f] += 1;
            }
        }
        // st[f] now holds the END of tap f == start of tap f+1: shift back to starts
        uint32_t prev = 0;
        for (int f = 0; f < ntap; ++f) {
            const uint32_t end = st[f];
            st[f] = (uint16_t)prev;
            prev = end;
        }
        // per-tap totals over the tile
        for (int f = 0; f < ntap; ++f) {
            uint32_t c = (uint32_t)(st[f + 1] - st[f]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            if (lane == 0) total[f] = c;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// deep_gemm_kernel: out[centre, 0..NDIM) = sum_f A_f[centre, 0..KDIM) . Bm[f][KDIM][NDIM]
//   BWD = false : forward.   src = input  (rows of KDIM = Cin),  taps = fwd, weights rcp_fwd, Bm = filter
//   BWD = true  : grad_input. src = grad_out (rows of KDIM = Cout), taps = bwd, weights rcp_bwd, Bm = filter^T
//                 ([F][Cout][Cin]); additionally publishes the tile's tap-major record order for deep_dw_kernel.
// One workgroup (4 waves) per query tile.  The 2 x NDIM/32 output blocks of 32x32 are dealt to the waves
// round-robin; accumulators stay in registers across all taps.
// LDS: A_f [64][KDIM+1] | start [64][ntap+1] u16 | qbase [64] | qrel [64] | total [ntap] | qorig [64]
// ---------------------------------------------------------------------------------------------
template <int KDIM, int NDIM, bool BWD>
__global__ __launch_bounds__(256) void deep_gemm_kernel(const PointRec<float> *__restrict__ pts,
                                                        const PairEntry *__restrict__ pairs,
                                                        const uint2 *__restrict__ segs,
                                                        const uint2 *__restrict__ qsegs,
                                                        const float *__restrict__ src,
                                                        const float *__restrict__ Bm, int N, int ntiles, int ntap,
                                                        BlockMap bm, float *__restrict__ out,
                                                        uint16_t *__restrict__ bucket_order,
                                                        uint32_t *__restrict__ tap_order,
                                                        uint32_t *__restrict__ tap_off,
                                                        uint8_t *__restrict__ tile_flag)
{
    constexpr int LDA = KDIM + 1;
    constexpr int NBLK = 2 * (NDIM / 32);                 // 32x32 output blocks of the tile
    constexpr int PER_WAVE = (NBLK + 3) / 4;
    constexpr int KPL = (KDIM + 63) / 64;                 // channels per lane in the gather stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *A = reinterpret_cast<float *>(smem);
    size_t off = align16((size_t)64 * LDA * 4);
    uint16_t *start = reinterpret_cast<uint16_t *>(smem + off);
    off += align16((size_t)64 * (ntap + 1) * 2);
    uint32_t *qbase = reinterpret_cast<uint32_t *>(smem + off);
    off += 256;
    uint32_t *qrel = reinterpret_cast<uint32_t *>(smem + off);   // start of each centre's list inside the segment
    off += 256;
    uint32_t *total = reinterpret_cast<uint32_t *>(smem + off);
    off += align16((size_t)ntap * 4);
    int32_t *qorig = reinterpret_cast<int32_t *>(smem + off);

    int b, qt;
    if (!block_to_cloud(bm, b, qt)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t tile_id = (size_t)b * ntiles + qt;
    const uint2 tseg = segs[tile_id];                      // ngroups == 1 on this path (host checks)
    if (wave == 0) qorig[lane] = pts[tile_id * kTile + lane].idx;
    if (tseg.y == kSegOverflow) {
        // the cloud's pair region overflowed: leave zero rows and flag the tile; the generic kernel is launched
        // afterwards for flagged tiles only (it can search the tile itself)
        if (threadIdx.x == 0) tile_flag[tile_id] = 1;
        __syncthreads();
        float *oc = out + (size_t)b * N * NDIM;
        for (int e = threadIdx.x; e < 64 * NDIM; e += 256) {
            const int orig = qorig[e / NDIM];
            if (orig >= 0) oc[(size_t)orig * NDIM + (e % NDIM)] = 0.0f;
        }
        if (BWD && threadIdx.x == 0) {
            uint32_t *toff = tap_off + tile_id * (ntap + 1);
            for (int f = 0; f <= ntap; ++f) toff[f] = 0;    // deep_dw_kernel skips the tile
        }
        return;
    }
    const PairEntry *seg = pairs + tseg.x;
    uint16_t *order = bucket_order + tseg.x;               // this tile's share of the global scratch
    if (threadIdx.x == 0) tile_flag[tile_id] = 0;
    if (wave == 0) qrel[lane] = qsegs[tile_id * 64 + lane].x - tseg.x;
    bucket_by_tap<BWD>(seg, qsegs + tile_id * 64, tseg.x, ntap, start, order, qbase, total);
    __syncthreads();

    if (BWD) {
        // tap-major order of the tile's records (tap ascending, centre ascending, list order) for deep_dw_kernel:
        // tap_order[slot] = record index relative to the tile's segment
        uint32_t *toff = tap_off + tile_id * (ntap + 1);
        if (wave == 0) {
            uint32_t run = 0;
            for (int f = 0; f < ntap; ++f) {
                const uint32_t c = (uint32_t)(start[lane * (ntap + 1) + f + 1] - start[lane * (ntap + 1) + f]);
                int tot;
                const uint32_t pos = run + (uint32_t)wave_excl_scan((int)c, tot);
                const uint32_t from = qbase[lane] + start[lane * (ntap + 1) + f];
                for (uint32_t p = 0; p < c; ++p) tap_order[tseg.x + pos + p] = qrel[lane] + order[from + p];
                if (lane == 0) toff[f] = run;
                run += (uint32_t)tot;
            }
            if (lane == 0) toff[ntap] = run;
        }
    }

    f32x16 acc[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const float *src_cloud = src + (size_t)b * N * KDIM;
    for (int f = 0; f < ntap; ++f) {
        if (total[f] == 0) continue;                        // block-uniform
        // ---- gather-reduce: A[q][:] = sum of the tap-f neighbour rows of centre q, normalised
        for (int q = wave; q < 64; q += 4) {
            const uint32_t r0 = start[q * (ntap + 1) + f], r1 = start[q * (ntap + 1) + f + 1];
            float av[KPL];
#pragma unroll
            for (int u = 0; u < KPL; ++u) av[u] = 0.0f;
            for (uint32_t p0 = r0; p0 < r1; p0 += 64) {
                // lanes fetch (neighbour, weight) of up to 64 records in parallel: one latency, not one per record
                uint32_t mcand = 0;
                float mw = 0.0f;
                if (p0 + lane < r1) {
                    const PairEntry en = seg[qrel[q] + order[qbase[q] + p0 + lane]];
                    mcand = en.cand;
                    mw = BWD ? en.rcp_bwd : en.rcp_fwd;
                }
                const int nrec = (int)min(64u, r1 - p0);
                for (int r = 0; r < nrec; r += 4) {
                    // four independent row loads in flight
                    float rv[4][KPL], wv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int rr = r + t < nrec ? r + t : r;
                        const uint32_t cand = __shfl(mcand, rr);
                        wv[t] = r + t < nrec ? __shfl(mw, rr) : 0.0f;
                        const float *row = src_cloud + (size_t)cand * KDIM;
#pragma unroll
                        for (int u = 0; u < KPL; ++u) rv[t][u] = lane + 64 * u < KDIM ? row[lane + 64 * u] : 0.0f;
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int u = 0; u < KPL; ++u) av[u] = __builtin_fmaf(rv[t][u], wv[t], av[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < KPL; ++u)
                if (lane + 64 * u < KDIM) A[q * LDA + lane + 64 * u] = av[u];
        }
        __syncthreads();
        // ---- [64 x KDIM] . [KDIM x NDIM] on the matrix cores
        const float *Bf = Bm + (size_t)f * KDIM * NDIM;
        for (int k0 = 0; k0 < KDIM; k0 += 2) {
            const int kk = k0 + (lane >> 5);
            const float a0 = A[(lane & 31) * LDA + kk];
            const float a1 = A[(32 + (lane & 31)) * LDA + kk];
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) {
                const int blk = wave + 4 * i;                // block id = rb * (NDIM/32) + cb
                if (blk < NBLK) {
                    const int rb = blk / (NDIM / 32), cb = blk % (NDIM / 32);
                    const float bv = Bf[(size_t)kk * NDIM + cb * 32 + (lane & 31)];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb ? a1 : a0, bv, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: C fragment -> out rows (by original index)
    float *out_cloud = out + (size_t)b * N * NDIM;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int blk = wave + 4 * i;
        if (blk < NBLK) {
            const int rb = blk / (NDIM / 32), cb = blk % (NDIM / 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int orig = qorig[row];
                if (orig >= 0) out_cloud[(size_t)orig * NDIM + cb * 32 + (lane & 31)] = acc[i][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// deep_dw_kernel: grad_filter partials.  Workgroup = (tap f, 64-column slice of Cout, chunk of query tiles):
//   dW[f][0..CIN)[n0..n0+64) = sum over the chunk's tiles of  X_tile^T [CIN x 64] . G_f [64 x 64]
// G_f (this tap, this column slice) is gather-reduced per tile from the tap-major record order written by
// deep_gemm_kernel<.., BWD=true>.  Wave w owns the row blocks w, w+4, ... of CIN; accumulators live in
// registers across the whole chunk; one partial per chunk, summed by reduce_partials_kernel.
// LDS: X tile [64][CIN+1] | G [64][65]
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void deep_dw_kernel(const PointRec<float> *__restrict__ pts,
                                                      const PairEntry *__restrict__ pairs,
                                                      const uint2 *__restrict__ segs,
                                                      const uint32_t *__restrict__ tap_order,
                                                      const uint32_t *__restrict__ tap_off,
                                                      const float *__restrict__ grad_out,
                                                      const float *__restrict__ input, int B, int N, int ntiles,
                                                      int ntap, int nchunks, float *__restrict__ partials)
{
    constexpr int LDX = CIN + 1;
    constexpr int RB = CIN / 32;                           // row blocks of CIN
    constexpr int PER_WAVE = (RB + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *X = reinterpret_cast<float *>(smem);
    float *G = reinterpret_cast<float *>(smem + align16((size_t)64 * LDX * 4));
    __shared__ int32_t qorig[64];
    __shared__ uint32_t mcand[256], mq[256];
    __shared__ float mrcp[256];

    const int f = blockIdx.x, slice = blockIdx.y, chunk = blockIdx.z;
    const int n0 = slice * 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[PER_WAVE][2];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.0f;

    const size_t total_tiles = (size_t)B * ntiles;
    const size_t per = (total_tiles + nchunks - 1) / nchunks;
    const size_t t0 = per * chunk, t1 = (t0 + per < total_tiles) ? t0 + per : total_tiles;
    for (size_t tile = t0; tile < t1; ++tile) {
        const uint32_t *toff = tap_off + tile * (ntap + 1);
        const uint32_t e0 = toff[f], e1 = toff[f + 1];
        if (e0 == e1) continue;                             // uniform: no neighbour with this tap in the tile
        const int b = (int)(tile / ntiles);
        const uint2 tseg = segs[tile];
        const PairEntry *seg = pairs + tseg.x;
        const uint32_t *ord = tap_order + tseg.x;
        if (wave == 0) qorig[lane] = pts[tile * kTile + lane].idx;
        for (int e = threadIdx.x; e < 64 * 65; e += 256) G[e] = 0.0f;
        __syncthreads();
        // X tile (rows by original index; padding centres -> 0)
        for (int q = wave; q < 64; q += 4) {
            const int orig = qorig[q];
            const float *xr = input + ((size_t)b * N + (orig < 0 ? 0 : orig)) * CIN;
            for (int k = lane; k < CIN; k += 64) X[q * LDX + k] = orig >= 0 ? xr[k] : 0.0f;
        }
        // G[q][0..64) += dY[cand][n0..n0+64) / count   (records of a centre are consecutive; wave = q & 3 owns it)
        const float *dy_cloud = grad_out + (size_t)b * N * COUT;
        for (uint32_t eb = e0; eb < e1; eb += 256) {
            // 256 threads fetch the metadata of up to 256 records in parallel, then every wave walks them
            __syncthreads();
            if (eb + threadIdx.x < e1) {
                const PairEntry en = seg[ord[eb + threadIdx.x]];
                mcand[threadIdx.x] = en.cand;
                mq[threadIdx.x] = code_q(en.code);
                mrcp[threadIdx.x] = en.rcp_bwd;
            }
            __syncthreads();
            const int nrec = (int)min(256u, e1 - eb);
            for (int r = 0; r < nrec; ++r) {
                const uint32_t q = mq[r];
                if ((int)(q & 3) == wave)
                    G[q * 65 + lane] += dy_cloud[(size_t)mcand[r] * COUT + n0 + lane] * mrcp[r];
            }
        }
        __syncthreads();
        // X^T [CIN x 64] . G [64 x 64]
        for (int k0 = 0; k0 < 64; k0 += 2) {
            const int kk = k0 + (lane >> 5);
            const float b0 = G[kk * 65 + (lane & 31)], b1 = G[kk * 65 + 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) {
                const int rb = wave + 4 * i;
                if (rb < RB) {
                    const float a = X[kk * LDX + rb * 32 + (lane & 31)];
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[i][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // partial slot of this chunk: layout of grad_filter, [(f*CIN + k)*COUT + c]
    float *slot = partials + (size_t)chunk * ntap * CIN * COUT + (size_t)f * CIN * COUT;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int rb = wave + 4 * i;
        if (rb < RB)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    slot[(size_t)k * COUT + n0 + c * 32 + (lane & 31)] = acc[i][c][r];
                }
    }
}

// filter [F][Cin][Cout] -> [F][Cout][Cin]
__global__ __launch_bounds__(256) void transpose_filter_kernel(const float *__restrict__ w, int ntap, int cin,
                                                               int cout, float *__restrict__ wt)
{
    const size_t n = (size_t)ntap * cin * cout;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t f = e / ((size_t)cin * cout), r = e % ((size_t)cin * cout);
        const size_t c = r / cin, k = r % cin;              // e indexes wt[f][c][k]
        wt[e] = w[(f * cin + k) * cout + c];
    }
}

}  // namespace conv3p
