// conv3p_forward_taps.hpp -- Conv3p accumulate (tf_conv3p_atrous.cpp:453-504) for layers with many more input than
// output channels (the segmentation model's 36 -> 13 head, pointcnn_scene_seg_acsd.py:57), fp32: transform, then gather.
//
// The filter block a pair multiplies its neighbour's row with depends on the pair's tap only, not on the centre:
//     out[i, :] = sum_{pairs (i, j)} (x[j, :] . W[f(i, j)]) / count[i, f(i, j)]                      (.cpp:483-492)
// so  Z[j, f, :] = x[j, :] . W[f]  for EVERY point and tap is one dense [B N x Cin] x [Cin x 27 * 16] product for the
// matrix cores (tap_transform_kernel), after which a pair costs a 64-byte gather of Z[j, f, 0:16] and Cout FMAs with
// 1 / count (tap_gather_kernel) -- instead of its neighbour's 144-byte row and 468 FMAs whose weights every lane reads
// from its own tap's block in LDS (forward_kernel: LDS-bandwidth bound, 0.43 ms on the cfg4 rooms).
// Cost: Z is B N * ntap * 64 bytes of scratch (113 MB at the cfg4 size; written once, every 64-byte piece read by the
// pairs that name it -- from the 256 MB memory-side cache at that size).  Worth it when Cin is well above Cout; the
// models' 9 -> 9 layers keep forward_kernel (their X rows stay in L2, Z would not).
// Sum order: per pair the dot product over Cin first (an exact FMA chain on the matrix cores), then the pairs in list
// order -- fixed, bitwise reproducible; within the tests' fp32 tolerance of the reference's order.
#pragma once

namespace conv3p {

constexpr int kZRow = 16;   // floats per (point, tap) piece of Z: Cout padded to one 64-byte line piece

// ---------------------------------------------------------------------------------------------
// Z[p, f, 0:16] = X[p, :] . W[f][:, 0:Cout] (columns >= Cout zero).  One wave = 32 points (two 16-column blocks of the
// MFMA's N), all taps; v_mfma_f32_16x16x4_f32 with A = W[f] (M = output channel) and B = X^T (N = point), so that a
// lane ends up with four CONSECUTIVE channels of one point: one 16-byte store per (block, tap).
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void tap_transform_kernel(const float *__restrict__ input, const float *__restrict__ filter,
                                                            float *__restrict__ z, size_t rows, int ntap, int ld_in)
{
    static_assert(CIN % 4 == 0 && COUT <= 16, "shape outside the tap transform");
    constexpr int KS = CIN / 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const size_t p0 = ((size_t)blockIdx.x * kWavesPerBlock + wave) * 32;
    if (p0 >= rows) return;
    float xb[2][KS];   // B[kk = l4][n = l15] of step s: X[p0 + 16 mb + l15][4 s + l4]
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const size_t p = p0 + 16 * mb + l15;
        const float *xr = input + (p < rows ? p : rows - 1) * (size_t)ld_in;
#pragma unroll
        for (int s = 0; s < KS; ++s) xb[mb][s] = xr[4 * s + l4];
    }
    for (int f = 0; f < ntap; ++f) {
        float wa[KS];   // A[m = l15][kk = l4] of step s: W[f][4 s + l4][l15]
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float wv = filter[((size_t)f * CIN + 4 * s + l4) * COUT + (l15 < COUT ? l15 : 0)];
            wa[s] = l15 < COUT ? wv : 0.0f;
        }
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], xb[mb][s], acc[mb], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {   // D[m = 4 l4 + rr][n = l15]: channels 4 l4 .. + 3 of point p0 + 16 mb + l15
            const size_t p = p0 + 16 * mb + l15;
            if (p < rows) *reinterpret_cast<f32x4 *>(z + (p * ntap + f) * kZRow + 4 * l4) = acc[mb];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// out[i, c] = sum over the stored pairs (i, j) of Z[j, f, c] / count[i, f].  Same walk as forward_kernel's register
// path: wave w owns the centres 16 w .. + 15, the 4 lanes {c, c + 16, c + 32, c + 48} share centre c and take 4
// consecutive records per step; record 2 steps / Z piece 1 step ahead in named register slots.
template <int COUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void tap_gather_kernel(
    const PointRec<float> *__restrict__ pts, const float *__restrict__ boxes, const int32_t *__restrict__ count,
    const PairEntry *__restrict__ pairs, const uint2 *__restrict__ segs, const uint2 *__restrict__ qsegs,
    const float *__restrict__ z, Stencil<float> st, int N, int ntiles, int ngroups, BlockMap bm,
    float *__restrict__ output, int act, const float *__restrict__ cmin, const int32_t *__restrict__ tcount, RowLd ld,
    const uint32_t *__restrict__ sched)   // launch order of the tiles (tile_sched_kernel) or nullptr
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *tapmap = reinterpret_cast<int16_t *>(smem);
    size_t off = align16((size_t)3 * st.maxfull * 2);
    float *rcpt = reinterpret_cast<float *>(smem + off);   // 1 / count [tap][65]
    off += align16((size_t)st.ntap * kCntStride * 4);
    int32_t *qorig = reinterpret_cast<int32_t *>(smem + off);
    off += 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *soa = reinterpret_cast<float *>(smem + off) + wave * 192;
    off += align16((size_t)kWavesPerBlock * 192 * 4);
    float *red = reinterpret_cast<float *>(smem + off);   // [4][COUT][64], overflow path only
    const int cq = wave * 16 + (lane & 15), sub = lane >> 4;

    int b, qt;
    if (!block_to_tile(bm, sched, ntiles, b, qt)) return;   // uniform
    const size_t tile_id = (size_t)b * ntiles + qt;
    const PointRec<float> *cloud_pts = pts + (size_t)b * ntiles * kTile;
    const PointRec<float> me = cloud_pts[(size_t)qt * kTile + lane];
    if (wave == 0) qorig[lane] = me.idx;
    bool overflow = false;
    for (int g = 0; g < ngroups; ++g) overflow |= segs[tile_id * ngroups + g].y == kSegOverflow;
    const float *z_cloud = z + (size_t)b * N * st.ntap * kZRow;

    constexpr int NS = 4;
    PairEntry rec[NS];
    f32x4 zs[NS][4];
    uint2 sg = make_uint2(0u, 0u);
    const PairEntry *pe = pairs;
    auto ld_rec = [&](uint32_t i) { return pe[i < sg.y ? i : 0u]; };
    auto ld_row = [&](int sl, uint32_t i) {
        const uint32_t f = code_fwd(rec[sl].code);
        const bool ok = i < sg.y && f != kNoTap;
        const f32x4 *zr = reinterpret_cast<const f32x4 *>(z_cloud + ((size_t)(ok ? rec[sl].cand : 0u) * st.ntap + (ok ? f : 0u)) * kZRow);
#pragma unroll
        for (int v = 0; v < (COUT + 3) / 4; ++v) zs[sl][v] = zr[v];
    };
    auto start_group = [&](int g) {
        sg = qsegs[(tile_id * ngroups + g) * 64 + cq];
        pe = pairs + sg.x;
#pragma unroll
        for (int sl = 0; sl < NS - 1; ++sl) rec[sl] = ld_rec(sub + 4 * sl);
#pragma unroll
        for (int sl = 0; sl < NS - 2; ++sl) ld_row(sl, sub + 4 * sl);
    };
    if (!overflow) start_group(0);   // (the first loads need nothing from LDS: they run under the prologue)
    {
        const int32_t *tc = tcount + tile_id * st.ntap * kTile;
        const int ne = st.ntap * kTile;
        for (int e0 = threadIdx.x; e0 < ne; e0 += 4 * 256) {
            int32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = tc[e0 + 256 * u < ne ? e0 + 256 * u : 0];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 256 * u;
                if (e < ne) rcpt[(e >> 6) * kCntStride + (e & 63)] = 1.0f / (float)v[u];   // .cpp:483
            }
        }
    }
    if (overflow) build_tapmap(tapmap, st.full, st.step, st.maxfull);   // (uniform)
    __syncthreads();

    float *out_cloud = output + (size_t)b * N * ld.out;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.0f;

    if (!overflow) {
        for (int g = 0; g < ngroups; ++g) {
            if (g > 0) start_group(g);
            uint32_t i = sub;
            bool more = true;
            while (more) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    if (!__any(i < sg.y)) {
                        more = false;
                        break;
                    }
                    rec[(j + NS - 1) % NS] = ld_rec(i + 4 * (NS - 1));
                    ld_row((j + NS - 2) % NS, i + 4 * (NS - 2));
                    __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of this step's arithmetic
                    const uint32_t f = code_fwd(rec[j].code);
                    if (i < sg.y && f != kNoTap) {
                        const float rcp = rcpt[f * kCntStride + cq];
#pragma unroll
                        for (int c = 0; c < COUT; ++c) acc[c] = fma_t(zs[j][c >> 2][c & 3], rcp, acc[c]);
                    }
                    i += 4;
                }
            }
        }
        const int orig = qorig[cq];
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            float v = acc[c];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (sub == 0 && orig >= 0) out_cloud[(size_t)orig * ld.out + c] = act ? selu_value(v) : v;
        }
    } else {
        // pair buffer was full for this tile: search it here (lane = centre, the waves split the candidate tiles)
        const float *cloud_box = boxes + (size_t)b * ntiles * 6;
        const int32_t *cnt_row = count + ((size_t)b * N + (me.idx < 0 ? 0 : me.idx)) * st.ntap;
        Query<float> q;
        make_query(q, me, st);
        Window<float> win;
        if (cmin != nullptr) make_window(win, me, st, cmin + (size_t)b * 3);
        for_each_neighbor(cloud_pts, cloud_box, ntiles, q, st, tapmap, soa, wave, kWavesPerBlock,
                          [&](const PointRec<float> &v, int f) {
            const float rcp = 1.0f / (float)cnt_row[f];
            const float *zr = z_cloud + ((size_t)v.idx * st.ntap + f) * kZRow;
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[c] = fma_t(zr[c], rcp, acc[c]);
        }, win, cmin != nullptr);
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
    }
}

}  // namespace conv3p
