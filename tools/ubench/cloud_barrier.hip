// micro-probe (developer tool, round 6): what a PER-CLOUD barrier between the layers of a fused stack kernel costs.
// Geometry of the cfg2 stack: 1024 workgroups of 256 threads, 4 per CU (37 KiB of LDS each), workgroup b on XCD b % 8,
// cloud = xcd + 8 * (b / 8 / 32), 32 tiles per cloud -> the 32 workgroups of a cloud share one XCD's L2.
// Every "layer" a workgroup (optionally) gathers rows of the previous layer's columns written by OTHER tiles of its cloud
// from a (B, N, 36) buffer (checked word for word against what must be there: catches stale L1 / L2 lines), writes its own
// 64 rows x 9 columns, and meets the other 31 tiles of its cloud at a barrier.
// Variants (struct Cfg): counter per cloud or for the grid; arrive / poll with agent-scope atomics (sc1) or with atomics that
// stay in the XCD's own L2; release / acquire fences at agent scope or just "stores drained" (s_waitcnt vmcnt(0)) with or
// without an L1 invalidate (buffer_inv sc1 / sc0); every layer's rows in a buffer of its own (no line a workgroup read
// earlier in the launch can hold them) or in column blocks of one (B, N, 36) buffer (the stale-L1-line hazard).
//   hipcc --offload-arch=gfx950 -O3 -o cloud_barrier cloud_barrier.hip && ./cloud_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kB = 32, kTiles = 32, kN = kTiles * 64, kLd = 36, kCols = 9;

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float expect(int cloud, int row, int layer, int col, uint32_t salt)
{
    return (float)(mix((uint32_t)(((cloud * kN + row) * 8 + layer) * 16 + col) ^ salt) & 0xFFFFF);
}

constexpr int kMaxSpins = 20000;
// returning atomic add executed in the XCD's own L2 (no sc1; sc0 = return the old value).  Inline asm: hipcc turns an atomic
// add of 0 at workgroup scope into a plain load, which may hit the CU's L1 for ever.
__device__ __forceinline__ uint32_t l2_atomic_add(uint32_t *p, uint32_t v)
{
    uint32_t old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
    return old;
}
struct Cfg {
    int grid_wide;   // 1: one counter for the whole grid (else one per cloud)
    int wg_atomics;  // 1: arrive / poll with WORKGROUP-scope atomics (no sc1: executed in the XCD's own L2); 0: agent scope
    int poll_rmw;    // 1: poll with a returning atomic add of 0; 0: with an atomic load
    int fence;       // 0: s_waitcnt vmcnt(0) only; 1: release / acquire fences at agent scope
    int inv;         // 0: none; 1: buffer_inv sc1; 2: buffer_inv sc0
    int separate;    // 1: every layer writes a buffer of its own (dense [B][N][9]); 0: column blocks of one (B, N, 36) buffer
};

__global__ __launch_bounds__(256) void probe(float *buf, uint32_t *counters, int layers, int work_iters, int gather, uint32_t salt,
                                             unsigned long long *stats, uint32_t *placement, Cfg cfg)
{
    extern __shared__ char smem[];
    const int b = blockIdx.x, xcd = b & 7, r = b >> 3;
    const int cloud = xcd + 8 * (r / kTiles), tile = r % kTiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *cnt = counters + (cfg.grid_wide ? 0 : cloud * 32);   // one 128-byte line per cloud
    const uint32_t waiters = cfg.grid_wide ? gridDim.x : kTiles;
    if (threadIdx.x == 0 && placement) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
        const uint32_t hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        placement[2 * b] = xcc;
        placement[2 * b + 1] = hwid;
    }
    unsigned long long bad = 0, waited = 0;
    for (int l = 0; l < layers; ++l) {
        // where layer l's rows live: column block l & 3 of the (B, N, 36) buffer, or a dense buffer of its own
        const size_t lbytes = (size_t)kB * kN * kCols;
        const int ldw = cfg.separate ? kCols : kLd;
        float *wr = cfg.separate ? buf + (size_t)l * lbytes + (size_t)cloud * kN * kCols : buf + (size_t)cloud * kN * kLd + (l & 3) * kCols;
        const float *rd = cfg.separate ? buf + (size_t)(l - 1) * lbytes + (size_t)cloud * kN * kCols : buf + (size_t)cloud * kN * kLd + ((l + 3) & 3) * kCols;
        float dummy = (float)lane;
        const int it = work_iters ? (int)(mix((uint32_t)(b * 131 + l)) % (uint32_t)work_iters) : 0;
        for (int i = 0; i < it; ++i) dummy = __builtin_fmaf(dummy, 1.0000001f, 0.5f);
        if (gather && l > 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = (int)(mix((uint32_t)(b * 977 + l * 31 + threadIdx.x * 7 + u)) % (uint32_t)kN);
                const float *p = rd + (size_t)row * ldw;
#pragma unroll
                for (int c = 0; c < kCols; ++c) bad += p[c] != expect(cloud, row, l - 1, c, salt) ? 1 : 0;
            }
        }
        for (int c = wave; c < kCols; c += 4) wr[(size_t)(tile * 64 + lane) * ldw + c] = expect(cloud, tile * 64 + lane, l, c, salt) + (dummy < 0.0f ? 1.0f : 0.0f);
        // ---- barrier
        const long long t0 = wall_clock64();
        if (cfg.fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t target = waiters * (uint32_t)(l + 1);
            int spins = 0;   // EVERY spin is bounded: a variant that cannot see the counter move must not hang the GPU
            if (cfg.wg_atomics) {
                l2_atomic_add(cnt, 1u);
                if (cfg.poll_rmw) { while (l2_atomic_add(cnt, 0u) + 0u < target && ++spins < kMaxSpins) __builtin_amdgcn_s_sleep(1); }
                else { while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < kMaxSpins) __builtin_amdgcn_s_sleep(1); }
            } else {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cfg.poll_rmw) { while (__hip_atomic_fetch_add(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < kMaxSpins) __builtin_amdgcn_s_sleep(1); }
                else { while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < kMaxSpins) __builtin_amdgcn_s_sleep(1); }
            }
            if (spins >= kMaxSpins) atomicAdd(&stats[2], 1ull);
        }
        __syncthreads();
        if (cfg.fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        else if (cfg.inv == 1) asm volatile("buffer_inv sc1" ::: "memory");
        else if (cfg.inv == 2) asm volatile("buffer_inv sc0" ::: "memory");
        waited += (unsigned long long)(wall_clock64() - t0);
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if (lane == 0 && bad) atomicAdd(&stats[0], bad);
    if (threadIdx.x == 0) atomicAdd(&stats[1], waited);
}

int run(const char *name, Cfg cfg, int layers, int work_iters, int gather, float *buf, uint32_t *counters, unsigned long long *stats,
        uint32_t *placement)
{
    const int lds = 37 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    unsigned long long h[3] = {0, 0, 0};
    unsigned long long bad_total = 0, gave_up = 0;
    for (int rep = 0; rep < 12; ++rep) {
        CK(hipMemsetAsync(counters, 0, 4 * 32 * kB, 0));
        CK(hipMemsetAsync(stats, 0, 24, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe, dim3(kB * kTiles), dim3(256), lds, 0, buf, counters, layers, work_iters, gather, 0x1234567u * (rep + 1) + 77u * cfg.inv, stats,
                           rep == 0 ? placement : nullptr, cfg);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        CK(hipMemcpy(h, stats, 24, hipMemcpyDeviceToHost));
        bad_total += h[0];
        gave_up += h[2];
        if (rep >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    printf("%-58s layers %3d work<=%5d gather %d: kernel %8.1f us, %6.2f us per layer; mean wait of a workgroup %5.2f us per barrier; wrong words %llu%s\n", name, layers,
           work_iters, gather, med * 1e3, med * 1e3 / layers, (double)h[1] * 0.01 / (kB * kTiles) / layers, bad_total, gave_up ? "  [SPIN BOUND HIT: counter not seen]" : "");
    fflush(stdout);
    return 0;
}

int main()
{
    const int L = 48;
    float *buf; uint32_t *counters, *placement; unsigned long long *stats;
    CK(hipMalloc(&buf, (size_t)L * kB * kN * kLd * 4));
    CK(hipMemset(buf, 0, (size_t)L * kB * kN * kLd * 4));
    CK(hipMalloc(&counters, 4 * 32 * kB));
    CK(hipMalloc(&stats, 24));
    CK(hipMalloc(&placement, 8 * kB * kTiles));
    //                       grid wg  rmw fence inv sep
    const Cfg light_sc1   = {0,   0,  0,  0,    1,  0};
    if (run("warm-up", light_sc1, 8, 0, 0, buf, counters, stats, placement)) return 1;
    {
        std::vector<uint32_t> pl(2 * kB * kTiles);
        CK(hipMemcpy(pl.data(), placement, 8 * kB * kTiles, hipMemcpyDeviceToHost));
        int mism = 0; int per_xcc[16] = {0};
        for (int b = 0; b < kB * kTiles; ++b) { mism += (pl[2 * b] & 15u) != (uint32_t)(b & 7); per_xcc[pl[2 * b] & 15u]++; }
        printf("placement: %d of %d workgroups NOT on XCC blockIdx %% 8; per XCC:", mism, kB * kTiles);
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
        printf("\n");
    }
    struct Named { const char *name; Cfg c; };
    const Named all[] = {
        {"cloud, agent atomics, load poll, inv sc1", {0, 0, 0, 0, 1, 0}},
        {"cloud, agent atomics, load poll, no inv", {0, 0, 0, 0, 0, 0}},
        {"cloud, agent atomics, load poll, no inv, own buffers", {0, 0, 0, 0, 0, 1}},
        {"cloud, agent atomics, rmw poll, no inv, own buffers", {0, 0, 1, 0, 0, 1}},
        {"cloud, L2-local atomics, rmw poll, no inv", {0, 1, 1, 0, 0, 0}},
        {"cloud, L2-local atomics, rmw poll, no inv, own buffers", {0, 1, 1, 0, 0, 1}},
        {"cloud, L2-local atomics, load(sc1) poll, no inv, own bufs", {0, 1, 0, 0, 0, 1}},
        {"cloud, L2-local atomics, rmw poll, inv sc0", {0, 1, 1, 0, 2, 0}},
        {"cloud, L2-local atomics, rmw poll, inv sc1", {0, 1, 1, 0, 1, 0}},
        {"cloud, agent atomics, full agent fences", {0, 0, 0, 1, 0, 0}},
        {"grid, agent atomics, full agent fences", {1, 0, 0, 1, 0, 0}},
    };
    for (int pass = 0; pass < 2; ++pass) {
        for (const Named &n : all) if (run(n.name, n.c, L, 0, 0, buf, counters, stats, nullptr)) return 1;
        for (const Named &n : all) if (run(n.name, n.c, L, 0, 1, buf, counters, stats, nullptr)) return 1;
        for (const Named &n : all) if (run(n.name, n.c, L, 2000, 1, buf, counters, stats, nullptr)) return 1;
    }
    return 0;
}
