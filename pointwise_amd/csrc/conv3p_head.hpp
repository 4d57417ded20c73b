// conv3p_head.hpp -- the dense head of the classification model on the matrix cores.
//
//   feat (B, N, 36) -> view (B, N*36) -> fc1 = selu(view . W1 + b1)   W1: (N*36) x 512   (pointcnn2_acsd.py:69-71)
// For N = 2048 that is a 73 728 x 512 fp32 matrix -- 151 MB, against 29 KB for all four conv3p filters together -- and
// a batch of 32 rows per GPU: every pass over it is bound by HBM (one full read of W1 forward, one read and one
// write of its size backward), not by arithmetic.  The kernels therefore stream W1 exactly once per pass with
// 16-byte loads, many in flight, and do the (exact fp32) products on v_mfma_f32_32x32x2_f32 so that the vector ALU
// stays free for address generation:
//
//   fc_forward_kernel   y_part[chunk] = x[:, chunk] . W[chunk, :]      split over K in `chunks` workgroups
//   fc_finish_kernel    y = act(sum_chunks y_part + b)                 fixed order: bitwise reproducible
//   fc_dx_kernel        dx = dz . W^T      (dz = dy * act'(y), computed once by fc_dz_kernel)
//   fc_dw_kernel        dW = x^T . dz      one pass that writes every element of dW once
//
// Layout of the 32x32x2 fp32 MFMA (cdna_hip_programming.md section 3): A operand lane l = A[i = l & 31][k = l >> 5],
// B operand lane l = B[k = l >> 5][j = l & 31], C/D register r of lane l = C[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
// Rows of x (the batch) are the M dimension: up to 32 per block (blockIdx.y covers larger batches).
#pragma once

#include "conv3p_deep.hpp"

namespace conv3p {

constexpr int kFcCols = 512;      // columns one workgroup covers (4 waves x 4 interleaved 32-column blocks)
constexpr int kFcDepth = 16;      // W loads (1 KiB per wave each) issued per group, one group ahead of the MFMAs

// ---------------------------------------------------------------------------------------------
// forward partials.  grid = (chunks, ceil(M / 32), ceil(N / 512)); workgroup = 4 waves.
// Wave w owns columns [512 z + 128 w, +128): its four accumulators take the columns = 4 j + t (j = lane & 31,
// t = 0..3), so that one float4 load per lane -- W[k][4 j .. 4 j + 3], a full 512 B row segment per half-wave --
// feeds four MFMAs.  The x chunk sits in LDS ([32][kc + 1], read along the rows with stride kc + 1: conflict-free).
// part[((chunk * gridDim.y + y) * 32 + m) * N + n]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void fc_forward_kernel(
    const float *__restrict__ x, const float *__restrict__ W, int M, int K, int N, int kc, float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);          // [32][ldx]: the chunk, zero past its end
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k0 = blockIdx.x * kc, klen = min(kc, K - k0);
    const int m0 = blockIdx.y * 32;
    const int steps = (klen + 1) / 2;
    const int psteps = (steps + kFcDepth - 1) / kFcDepth * kFcDepth;    // whole groups: straight-line inner loop
    const int ldx = 2 * psteps + 1;                        // odd: rows land on different banks
    for (int m = wave; m < 32; m += 4) {                   // a wave per row: coalesced along k
        const float *xr = x + (size_t)(m0 + m) * K + k0;
        for (int k = lane; k < 2 * psteps; k += 64) xs[m * ldx + k] = (m0 + m < M && k < klen) ? xr[k] : 0.0f;
    }
    __syncthreads();
    const int col = blockIdx.z * kFcCols + wave * 128 + 4 * (lane & 31);
    const bool col_ok = col + 4 <= N;                      // N % 4 == 0 (host checks)
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int half = lane >> 5;
    const float *xa = xs + (lane & 31) * ldx + half;
    // row k0 + 2 s + half of W, clamped to the chunk's last row: every load is unconditional (so that the compiler
    // can count them: kFcDepth loads stay in flight across the MFMAs), rows past the end meet x == 0 in LDS and are
    // zeroed here as well (0 * Inf)
    auto loadw = [&](int s) -> float4 {
        int k = 2 * s + half;
        k = k < klen ? k : klen - 1;
        return *reinterpret_cast<const float4 *>(W + (size_t)(k0 + k) * N + (col_ok ? col : 0));
    };
    // Double-buffered groups: the NEXT group's kFcDepth loads are issued first, then the current group's MFMAs run
    // (kFcDepth x 4 x 64 cycles, about one HBM latency) -- sched_barrier keeps hipcc from sinking the loads behind
    // the MFMAs, where it would wait for them at once.
    float4 cur[kFcDepth], nxt[kFcDepth];
#pragma unroll
    for (int u = 0; u < kFcDepth; ++u) cur[u] = loadw(u);
    for (int s0 = 0; s0 < psteps; s0 += kFcDepth) {
#pragma unroll
        for (int u = 0; u < kFcDepth; ++u) nxt[u] = loadw(s0 + kFcDepth + u);
        float a[kFcDepth];
#pragma unroll
        for (int u = 0; u < kFcDepth; ++u) a[u] = xa[2 * (s0 + u)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kFcDepth; ++u) {
            float4 w = cur[u];
            if (2 * (s0 + u) + half >= klen) w = make_float4(0.f, 0.f, 0.f, 0.f);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], w.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], w.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], w.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], w.w, acc[3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kFcDepth; ++u) cur[u] = nxt[u];
    }
    if (!col_ok) return;
    float *pp = part + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * 32) * N + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        *reinterpret_cast<float4 *>(pp + (size_t)m * N) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    }
}

// y[m][n] = act(sum over chunks of part[chunk][m][n] + b[n]); act: 0 = identity, 1 = SELU.
// Workgroup = 64 consecutive outputs x 16 waves; wave w adds the chunks w, w + 16, ... (four running sums), the 16
// per-wave sums are combined in wave order: a fixed association, bitwise reproducible, and 16x the parallelism of one
// thread per output (the partials are 16 MB for the model's fc1).
__global__ __launch_bounds__(1024) void fc_finish_kernel(const float *__restrict__ part, const float *__restrict__ b,
                                                         int M, int N, int chunks, int mblocks, int act,
                                                         float *__restrict__ y)
{
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const bool ok = i < (size_t)M * N;
    const int m = ok ? (int)(i / N) : 0, n = ok ? (int)(i - (size_t)m * N) : 0;
    const size_t base = ((size_t)(m >> 5) * 32 + (m & 31)) * N + n, stride = (size_t)mblocks * 32 * N;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = wave;
    for (; c + 48 < chunks; c += 64) {
        s0 += part[base + (size_t)c * stride];
        s1 += part[base + (size_t)(c + 16) * stride];
        s2 += part[base + (size_t)(c + 32) * stride];
        s3 += part[base + (size_t)(c + 48) * stride];
    }
    for (; c < chunks; c += 16) s0 += part[base + (size_t)c * stride];
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && ok) {
        float v = red[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) v += red[w][lane];
        v += b ? b[n] : 0.0f;
        y[i] = act ? selu_value(v) : v;
    }
}

// dz = dy * act'(y) (act = SELU: through the OUTPUT y, as TensorFlow's SeluGrad) and db[n] = sum_m dz[m][n]
// (one workgroup per 256 columns; rows added in ascending order)
__global__ __launch_bounds__(256) void fc_dz_kernel(const float *__restrict__ y, const float *__restrict__ dy, int M,
                                                    int N, int act, float *__restrict__ dz, float *__restrict__ db)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.0f;
    for (int m = 0; m < M; ++m) {
        const size_t i = (size_t)m * N + n;
        const float g = act ? dy[i] * selu_slope(y[i]) : dy[i];
        dz[i] = g;
        s += g;
    }
    if (db) db[n] = s;
}

// ---------------------------------------------------------------------------------------------
// dx[m][k] = sum_n dz[m][n] * W[k][n].  grid = (ceil(K / (32 nw)), ceil(M / 32)), nw = blockDim.x / 64 waves; wave w owns
// the 32 rows k = 32 nw x + 32 w .. +32 of W and computes the transposed block dx^T[k][m]:  A[i = k][kk = n] = W[k][n] comes
// straight from global (lane = one row; 16 bytes per lane and step, each 128-byte line of a row is consumed over 8
// consecutive steps out of L1, so HBM sees every byte of W once), B[kk = n][j = m] = dz[m][n] from LDS
// ([32][N + 4], 16-byte reads).  The 32x32 result goes through LDS once more (the dz tile's space, after a barrier) to be
// stored along k (coalesced).  N <= 1024, N % 8 == 0.
// Workgroup size (round 4): the host picks nw so that the whole grid is ONE resident round (66 KiB of LDS: two
// workgroups per CU, 512 slots; the model's fc1: nw = 5, 461 workgroups -- with four waves and a transpose buffer of
// its own it was 576 workgroups of 83 KiB, one per CU: three rounds, the last a quarter full; fc_dx 95 -> 55 us with kD = 8, fc_dw 78 -> 60 us).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void fc_dx_kernel(
    const float *__restrict__ dz, const float *__restrict__ W, int M, int K, int N, float *__restrict__ dx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 8 loads in flight per lane, twice: ~130 VGPRs, three waves per SIMD, so that TWO five-wave workgroups share a CU
    // (with 16 the kernel took 216 VGPRs: two waves per SIMD = one workgroup per CU, i.e. two rounds again)
    constexpr int kD = 8;
    const int steps = N / 8;                                            // 8 columns per step (4 per half-wave)
    const int psteps = (steps + kD - 1) / kD * kD;
    const int ldz = 8 * psteps + 4;
    float *zs = reinterpret_cast<float *>(smem);                       // [32][ldz], zero past column N
    float *tr = zs;                                                    // [nw waves][32][33] transpose tiles, AFTER the products
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, nw = (int)blockDim.x >> 6;
    const int m0 = blockIdx.y * 32;
    for (int m = wave; m < 32; m += nw)
        for (int n = lane; n < 8 * psteps; n += 64) zs[m * ldz + n] = (m0 + m < M && n < N) ? dz[(size_t)(m0 + m) * N + n] : 0.0f;
    __syncthreads();
    const int kb = ((int)blockIdx.x * nw + wave) * 32;
    const int krow = min(kb + (lane & 31), K - 1);                      // clamped: rows past K are computed, not stored
    const float *wrow = W + (size_t)krow * N + 4 * half;                // columns 8 s + 4 half .. + 3
    const float *zrow = zs + (lane & 31) * ldz + 4 * half;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    auto loadw = [&](int s) -> float4 {
        s = s < steps ? s : steps - 1;                                  // unconditional loads (see fc_forward_kernel)
        return *reinterpret_cast<const float4 *>(wrow + 8 * s);
    };
    float4 cur[kD], nxt[kD];
#pragma unroll
    for (int u = 0; u < kD; ++u) cur[u] = loadw(u);
    for (int s0 = 0; s0 < psteps; s0 += kD) {
#pragma unroll
        for (int u = 0; u < kD; ++u) nxt[u] = loadw(s0 + kD + u);      // next group first (see fc_forward_kernel)
        float4 z[kD];
#pragma unroll
        for (int u = 0; u < kD; ++u) z[u] = *reinterpret_cast<const float4 *>(zrow + 8 * (s0 + u));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            float4 w = cur[u];
            if (s0 + u >= steps) w = make_float4(0.f, 0.f, 0.f, 0.f);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, z[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, z[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, z[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, z[u].w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kD; ++u) cur[u] = nxt[u];
    }
    // acc holds dx^T[k = kb + row(r)][m = lane & 31]: transpose through LDS, store rows of dx along k
    __syncthreads();                                                    // every wave is done with the dz tile
    float *t = tr + wave * 32 * 33;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + (lane & 31)] = acc[r];
    __builtin_amdgcn_wave_barrier();
    for (int mm = half; mm < 32; mm += 2) {
        const int k = kb + (lane & 31);
        if (m0 + mm < M && k < K) dx[(size_t)(m0 + mm) * K + k] = t[(lane & 31) * 33 + mm];
    }
}

// ---------------------------------------------------------------------------------------------
// dW[k][n] = sum_m x[m][k] * dz[m][n].  grid = ceil(K / (32 nw)), nw = blockDim.x / 64 waves chosen by the host for one
// resident round (see fc_dx_kernel); wave w owns rows k = 32 nw x + 32 w .. +32: its x
// operand (A[i = k][kk = m], one coalesced 128-byte read per batch row) is loaded ONCE into 16 registers and reused
// for every 32-column block of dz (B from LDS [Mpad][N + 1]).  Every element of dW is written exactly once,
// 128 contiguous bytes per row and store.  Batches larger than 32 rows add further k-steps (M <= 128).
// ---------------------------------------------------------------------------------------------
template <int STEPS>   // MFMA k-steps = batch rows / 2 (16 for the models' 32 clouds per GPU; 32, 64 for larger batches)
__global__ __launch_bounds__(512) void fc_dw_kernel(const float *__restrict__ x, const float *__restrict__ dz, int M,
                                                    int K, int N, float *__restrict__ dW)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ldz = N + 1;
    float *zs = reinterpret_cast<float *>(smem);                       // [2 STEPS][N + 1], zero rows past M
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, nw = (int)blockDim.x >> 6;
    for (int m = wave; m < 2 * STEPS; m += nw)
        for (int n = lane; n < N; n += 64) zs[m * ldz + n] = m < M ? dz[(size_t)m * N + n] : 0.0f;
    __syncthreads();
    const int kb = ((int)blockIdx.x * nw + wave) * 32;
    const int k = min(kb + (lane & 31), K - 1);
    float a[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int m = min(2 * s + half, M - 1);
        a[s] = x[(size_t)m * K + k];                                    // rows past M meet dz == 0
    }
    for (int nb = 0; nb < N; nb += 32) {
        const int n = min(nb + (lane & 31), N - 1);
        const float *zb = zs + half * ldz + n;
        float bv[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) bv[s] = zb[2 * s * ldz];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv[s], acc, 0, 0, 0);
        if (nb + (lane & 31) < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kb + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (kk < K) dW[(size_t)kk * N + nb + (lane & 31)] = acc[r];
            }
        }
    }
}

}  // namespace conv3p
