// conv3p_forward_wide.hpp -- Conv3p accumulate (tf_conv3p_atrous.cpp:453-504) for the layers of 16..48 input channels
// and <= 16 output channels (the segmentation model's 36 -> 13 head, pointcnn_scene_seg_acsd.py:57), fp32.
//
// forward_kernel (conv3p_kernels.hpp) multiplies every pair's input row with the [Cin][Cout] block of ITS tap: 468
// weight reads from LDS per pair for 36 -> 13, with a different block in every lane -- LDS bandwidth bounds it
// (0.43 ms for the 5.2 M pairs of the cfg4 rooms).  The sum factors per (centre, tap):
//     out[i, :] = sum_f ( (1 / count[i, f]) * sum_{j in tap f of i} x[j, :] ) . W[f]            (.cpp:483-492)
// so the weights are needed once per (centre, tap) instead of once per pair, and the product over the 16 centres of a
// wave is a [16 x Cin] x [Cin x Cout] matrix product for the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA
// chains, deterministic).  What it takes is a centre's records grouped by tap; the lists are centre-major in search
// order, so every wave first bucket-sorts the lists of its 16 centres by forward tap in LDS:
//   pass 1   lane (centre, q) counts its records r == q (mod 4) per tap in a byte of cur[centre][tap] (own byte: no
//            two lanes ever write one address)
//   prefix   first position of every (tap, q) run inside the centre's sorted array, first position of every tap
//   pass 2   the records again (L1 / L2 hits): neighbour index -> sorted[centre][position++]
//   product  per tap: lane (centre, q) sums the input-row chunks k = 4 (q + 4 t) .. + 3 of the tap's neighbours (the four
//            lanes of a centre read 64 contiguous bytes of a row), scales by 1 / count, and the wave issues
//            4 * ceil(Cin / 16) MFMAs with B = W[f] (12 dword loads per lane, L1 / L2 resident)
// Lists longer than kWideQuota records are taken in batches of kWideQuota (positions fit a byte; the sums are linear).
// No barriers after the prologue: a wave's LDS accesses execute in program order.  The sum order differs from the
// reference's pair order (by tap, then by sub-lane run, then list order): fixed, so results are bitwise reproducible;
// parity with the oracle is within the fp32 tolerance of the tests, as for every other kernel here.
// A tile whose pair segment overflowed searches itself, as in forward_kernel (lane = centre, VALU products).
#pragma once

namespace conv3p {

#ifndef CONV3P_FW_ABLATE
#define CONV3P_FW_ABLATE 0   // developer ablation switch; 0 in every shipped build
#endif
#ifndef CONV3P_FW_QUOTA
#define CONV3P_FW_QUOTA 128
#endif
#ifndef CONV3P_FW_PRED
#define CONV3P_FW_PRED 0
#endif
#ifndef CONV3P_FW_U
#define CONV3P_FW_U 4
#endif
#ifndef CONV3P_FW_WAVES
#define CONV3P_FW_WAVES 2
#endif
constexpr int kWideQuota = CONV3P_FW_QUOTA;
#if CONV3P_FW_ABLATE & 32   // developer instrumentation build: shader-clock ticks per stage, printed by a sample of waves
#define FWT(acc_) { __builtin_amdgcn_s_waitcnt(0); const long long now_ = (long long)__builtin_amdgcn_s_memtime(); acc_ += now_ - tlast; tlast = now_; }
#else
#define FWT(acc_)
#endif   // records of one centre sorted per batch

// LDS: tapmap | sorted u32 [64][quota] (the overflow path's cross-wave sum aliases it) | cur u32 [64][32] |
//      tapstart u8 [64][32] | 1 / count [ntap][64] | qorig | SoA staging of the overflow path
__host__ __device__ inline size_t forward_wide_lds(int maxfull, int ntap)
{
    return (((size_t)3 * maxfull * 2 + 15) & ~(size_t)15) + (size_t)64 * kWideQuota * 4 + 64 * 32 * 4 + 64 * 32 +
           (((size_t)ntap * 64 * 4 + 15) & ~(size_t)15) + 256 + (size_t)kWavesPerBlock * 192 * 4;
}

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CONV3P_FW_WAVES))) void forward_wide_kernel(
    const PointRec<float> *__restrict__ pts, const float *__restrict__ boxes, const int32_t *__restrict__ count,
    const PairEntry *__restrict__ pairs, const uint2 *__restrict__ segs, const uint2 *__restrict__ qsegs,
    const float *__restrict__ input, const float *__restrict__ filter, Stencil<float> st, int N, int ntiles, int ngroups,
    BlockMap bm, float *__restrict__ output, int act, const float *__restrict__ cmin,
    const int32_t *__restrict__ tcount, RowLd ld)
{
    static_assert(CIN % 4 == 0 && CIN >= 16 && CIN <= 48 && COUT >= 1 && COUT <= 16, "shape outside the wide forward");
    constexpr int KT = (CIN + 15) / 16;   // 4-channel chunks per lane
    constexpr int Q = kWideQuota;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    uint32_t *sorted = reinterpret_cast<uint32_t *>(smem + off);
    float *red = reinterpret_cast<float *>(smem + off);   // [4][COUT][64], overflow path only
    off += (size_t)64 * Q * 4;
    uint32_t *cur = reinterpret_cast<uint32_t *>(smem + off);
    off += 64 * 32 * 4;
    uint8_t *tapstart = reinterpret_cast<uint8_t *>(smem + off);
    off += 64 * 32;
    float *rcpt = reinterpret_cast<float *>(smem + off);   // [tap][64]
    off += align16((size_t)st.ntap * 64 * 4);
    int32_t *qorig = reinterpret_cast<int32_t *>(smem + off);
    off += 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *soa = reinterpret_cast<float *>(smem + off) + wave * 192;
    const int l15 = lane & 15, l4 = lane >> 4;
    // Sorting and gathering: the four lanes of a centre are ADJACENT (centre lane >> 2, chunk lane & 3), so that their
    // 16-byte pieces of one input row / their consecutive records form one 64- / 32-byte access for the texture
    // addresser (lanes 16 apart are processed as 64 separate accesses: measured 4x the gather time).  The matrix
    // cores want row m of A in lanes m, m + 16, m + 32, m + 48: a fixed lane permutation (ds_bpermute) per operand.
    const int gc = lane >> 2, gq = lane & 3;
    const int cq = wave * 16 + gc;   // centre of this lane; its input chunks: k = 4 (gq + 4 t) .. + 3

    int b, qt;
    if (!block_to_cloud(bm, b, qt)) return;   // uniform
    const size_t tile_id = (size_t)b * ntiles + qt;
    const PointRec<float> *cloud_pts = pts + (size_t)b * ntiles * kTile;
    const PointRec<float> me = cloud_pts[(size_t)qt * kTile + lane];
    if (wave == 0) qorig[lane] = me.idx;
    bool overflow = false;
    for (int g = 0; g < ngroups; ++g) overflow |= segs[tile_id * ngroups + g].y == kSegOverflow;
    const uint2 sg0 = qsegs[(tile_id * ngroups) * 64 + cq];
    {
        // 1 / (T)count of the tile's (tap, centre) populations (.cpp:483: the IEEE quotient 1 / count, once per population)
        const int32_t *tc = tcount + tile_id * st.ntap * kTile;
        const int ne = st.ntap * kTile;
        for (int e0 = threadIdx.x; e0 < ne; e0 += 4 * 256) {
            int32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = tc[e0 + 256 * u < ne ? e0 + 256 * u : 0];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + 256 * u < ne) rcpt[e0 + 256 * u] = 1.0f / (float)v[u];
        }
    }
    if (overflow) build_tapmap(tapmap, st.full, st.step, st.maxfull);   // (uniform)
    __syncthreads();

    const float *in_cloud = input + (size_t)b * N * ld.in;
    float *out_cloud = output + (size_t)b * N * ld.out;

    if (overflow) {
        // pair buffer was full for this tile: search it here, lane = centre, the waves split the candidate tiles
        const float *cloud_box = boxes + (size_t)b * ntiles * 6;
        const int32_t *cnt_row = count + ((size_t)b * N + (me.idx < 0 ? 0 : me.idx)) * st.ntap;
        Query<float> q;
        make_query(q, me, st);
        Window<float> win;
        if (cmin != nullptr) make_window(win, me, st, cmin + (size_t)b * 3);
        float acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = 0.0f;
        for_each_neighbor(cloud_pts, cloud_box, ntiles, q, st, tapmap, soa, wave, kWavesPerBlock,
                          [&](const PointRec<float> &v, int f) {
            const float rcp = 1.0f / (float)cnt_row[f];
            const float *xr = in_cloud + (size_t)v.idx * ld.in;
            const float *wf = filter + (size_t)f * CIN * COUT;
            for (int k = 0; k < CIN; ++k) {
                const float xk = xr[k] * rcp;                                  // x / count, .cpp:492
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] = fma_t(wf[k * COUT + c], xk, acc[c]);
            }
        }, cmin != nullptr ? &win : nullptr);
#pragma unroll
        for (int c = 0; c < COUT; ++c) red[((size_t)wave * COUT + c) * 64 + lane] = acc[c];
        __syncthreads();
        for (int e = threadIdx.x; e < COUT * 64; e += blockDim.x) {
            const int c = e >> 6;   // e & 63 == lane
            float sum = red[((size_t)0 * COUT + c) * 64 + lane];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) sum += red[((size_t)w * COUT + c) * 64 + lane];
            if (me.idx >= 0) out_cloud[(size_t)me.idx * ld.out + c] = act ? selu_value(sum) : sum;
        }
        return;
    }

    uint32_t *my_sorted = sorted + (size_t)cq * Q;
    uint32_t *my_cur = cur + cq * 32;
    uint8_t *my_cur8 = reinterpret_cast<uint8_t *>(my_cur) + gq;   // + 4 f: this lane's byte of tap f
    uint8_t *my_ts = tapstart + cq * 32;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // D[centre 4 l4 + rr][channel l15], two chains
    bool chunk_ok[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) chunk_ok[t] = 4 * (gq + 4 * t) < CIN;

#if CONV3P_FW_ABLATE & 32
    long long tlast = (long long)__builtin_amdgcn_s_memtime(), t_pro = 0, t_sort = 0, t_w = 0, t_g = 0, t_m = 0, t_epi = 0;
    const long long tstart = tlast;
    int n_batch = 0, n_tap = 0, n_grp = 0;
#endif
    for (int g = 0; g < ((CONV3P_FW_ABLATE & 8) ? 0 : ngroups); ++g) {
        const uint2 sg = g == 0 ? sg0 : qsegs[(tile_id * ngroups + g) * 64 + cq];
        const PairEntry *pe = pairs + sg.x;
        for (uint32_t r0 = 0; __any(r0 < sg.y); r0 += Q) {
            const uint32_t r1 = sg.y < r0 + Q ? (sg.y < r0 ? r0 : sg.y) : r0 + Q;   // this batch: records [r0, r1) of the centre
            // ---- bucket sort of the batch by forward tap
            FWT(t_pro)
            reinterpret_cast<uint4 *>(my_cur)[2 * gq] = make_uint4(0u, 0u, 0u, 0u);
            reinterpret_cast<uint4 *>(my_cur)[2 * gq + 1] = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_fence();
            for (uint32_t base = r0 + gq; __any(base < r1); base += 32) {
                uint32_t code[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) code[u] = pe[base + 4 * u < r1 ? base + 4 * u : 0u].code;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t f = code_fwd(code[u]);
                    if (base + 4 * u < r1 && f != kNoTap) my_cur8[4 * f] = (uint8_t)(my_cur8[4 * f] + 1);
                }
            }
            wave_lds_fence();
            {
                uint32_t run = 0;
                for (int f = 0; f < st.ntap; ++f) {   // (ntap <= 31: host)
                    const uint32_t w = my_cur[f];
                    const uint32_t b0 = w & 0xFFu, b1 = (w >> 8) & 0xFFu, b2 = (w >> 16) & 0xFFu, b3 = w >> 24;
                    const uint32_t s1 = run + b0, s2 = s1 + b1, s3 = s2 + b2;
                    if (gq == 0) {
                        my_cur[f] = run | (s1 << 8) | (s2 << 16) | (s3 << 24);
                        my_ts[f] = (uint8_t)run;
                    }
                    run = s3 + b3;
                }
                if (gq == 0) my_ts[st.ntap] = (uint8_t)run;
            }
            wave_lds_fence();
            for (uint32_t base = r0 + gq; !(CONV3P_FW_ABLATE & 4) && __any(base < r1); base += 32) {
                PairEntry rc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) rc[u] = pe[base + 4 * u < r1 ? base + 4 * u : 0u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t f = code_fwd(rc[u].code);
                    if (base + 4 * u < r1 && f != kNoTap) {
                        const uint32_t p = my_cur8[4 * f];
                        my_cur8[4 * f] = (uint8_t)(p + 1);
                        my_sorted[p] = rc[u].cand;
                    }
                }
            }
            wave_lds_fence();
            FWT(t_sort)
#if CONV3P_FW_ABLATE & 32
            ++n_batch;
#endif
            // ---- per tap: sum of the neighbours' rows (this lane's chunks), then the product with W[f]
            for (int f = 0; f < ((CONV3P_FW_ABLATE & 2) ? 0 : st.ntap); ++f) {
                const int s = my_ts[f], nf = (int)my_ts[f + 1] - s;
                if (!__any(nf > 0)) continue;
                float wreg[KT][4];
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = 4 * (l4 + 4 * t) + u;
                        const bool ok = k < CIN && l15 < COUT;
                        const float wv = filter[ok ? ((size_t)f * CIN + k) * COUT + l15 : (size_t)0];
                        wreg[t][u] = ok ? wv : 0.0f;
                    }
                FWT(t_w)
#if CONV3P_FW_ABLATE & 32
                ++n_tap;
#endif
                f32x4 S[KT];
#pragma unroll
                for (int t = 0; t < KT; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                constexpr int U = CONV3P_FW_U;
                for (int rb = 0; !(CONV3P_FW_ABLATE & 1) && __any(rb < nf); rb += U) {
#if CONV3P_FW_ABLATE & 32
                    ++n_grp;
#endif
                    uint32_t cand[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) cand[u] = my_sorted[rb + u < nf ? s + rb + u : 0];
                    if (CONV3P_FW_ABLATE & 16)
#pragma unroll
                        for (int u = 0; u < U; ++u) cand[u] &= 63u;
                    float4_a4 xs[U][KT];
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int t = 0; t < KT; ++t) {
                            const bool ok = rb + u < nf && chunk_ok[t];
#if CONV3P_FW_PRED
                            xs[u][t] = float4_a4{0.f, 0.f, 0.f, 0.f};
                            if (ok) xs[u][t] = *reinterpret_cast<const float4_a4 *>(in_cloud + (size_t)cand[u] * ld.in + 4 * (gq + 4 * t));
#else
                            xs[u][t] = *reinterpret_cast<const float4_a4 *>(
                                in_cloud + (size_t)(ok ? cand[u] : 0u) * ld.in + (ok ? 4 * (gq + 4 * t) : 0));
#endif
                        }
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int t = 0; t < KT; ++t) {
                            const bool ok = rb + u < nf && chunk_ok[t];
                            S[t][0] += ok ? xs[u][t].x : 0.0f;
                            S[t][1] += ok ? xs[u][t].y : 0.0f;
                            S[t][2] += ok ? xs[u][t].z : 0.0f;
                            S[t][3] += ok ? xs[u][t].w : 0.0f;
                        }
                }
                FWT(t_g)
                const float rcp = nf > 0 ? rcpt[f * 64 + cq] : 0.0f;
                const int src4 = 4 * (4 * l15 + l4);   // lane that gathered centre l15's chunk l4 (byte address for bpermute)
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float mine = nf > 0 ? S[t][u] * rcp : 0.0f;
                        const float a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src4, __builtin_bit_cast(int, mine)));
                        acc[(t * 4 + u) & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[t][u], acc[(t * 4 + u) & 1], 0, 0, 0);
                    }
                FWT(t_m)
            }
            FWT(t_m)
            wave_lds_fence();   // the next batch rewrites the wave's part of `sorted`
        }
    }
#if CONV3P_FW_ABLATE & 32
    FWT(t_epi)
    if (lane == 0 && (blockIdx.x % 97) == 5)
        printf("fww wg %d wave %d: total %lld pro %lld sort %lld wload %lld gather %lld mfma %lld | batches %d taps %d groups %d n0 %u\n", (int)blockIdx.x, wave,
               tlast - tstart, t_pro, t_sort, t_w, t_g, t_m, n_batch, n_tap, n_grp, sg0.y);
#endif
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int orig = qorig[wave * 16 + 4 * l4 + rr];
        if (orig >= 0 && l15 < COUT) {
            const float v = acc[0][rr] + acc[1][rr];
            out_cloud[(size_t)orig * ld.out + l15] = act ? selu_value(v) : v;
        }
    }
}

}  // namespace conv3p
