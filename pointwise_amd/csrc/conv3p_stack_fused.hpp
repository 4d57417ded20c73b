// conv3p_stack_fused.hpp -- the models' narrow layer stack (pointcnn2_acsd.py:48-67: four dependent conv3p layers over ONE
// `points`) as ONE launch per pass instead of one per layer (round 6).
//
// Why: a narrow layer's kernel is a workgroup's chain of short phases -- forward_kernel<9,9> = prologue 6.5 us + loop 6.3 +
// epilogue 2.4 of a 19.5-us launch, backward_sparse<9,9> = prologue 5.3 + A 13.6 + B 7.1 + C 4.6 + store 2.4 of 46
// (profiles/r06_phase_trace.txt) -- and the ten dependent launches of a step pay a grid-wide boundary each.  What a
// layer needs from the layer before it are the rows of ITS OWN CLOUD only.  Here a workgroup keeps its (cloud, tile) for
// the whole pass; between two layers the 32 (rooms: 64) tiles of a cloud meet at a PER-CLOUD barrier, and everything of the
// next layer that does not depend on activations (filter -> LDS, reciprocal populations, lane sharing, the first pair
// records; tap sets and the rows bookkeeping in the backward) is done BEFORE the wait.
//
// The barrier (measured first: tools/ubench/cloud_barrier.hip, profiles/r06_cloud_barrier.txt).  All tiles of a cloud run
// on one XCD (workgroup b -> XCD b % 8, BlockMap), so the XCD's L2 is their coherence point:
//   arrive  s_waitcnt vmcnt(0) (the tile's stores have reached L2) -> workgroup barrier -> one returning-free atomic add
//           executed in that L2 (global_atomic_add without sc1)
//   wait    one lane polls with a returning L2 atomic add of 0 (inline asm: hipcc folds an atomic add of 0 into a plain
//           load, which may hit the CU's L1 for ever); bounded -- a wait that gives up sets the error word
//   = 2.9 us per layer with nothing else in the kernel.  Agent-scope fences (buffer_wbl2 / buffer_inv sc1) cost 140 us per
//   layer at 1024 workgroups, an L1 invalidate alone (buffer_inv sc1) 30 us, agent-scope polls (sc1 loads) 9.6 us.
//   NO cache invalidate is needed because of how the hand-off rows are laid out: every layer writes the rows the next
//   layer gathers into a dense buffer OF ITS OWN that no workgroup reads before the barrier, so no L1 can hold a line of
//   it from before it was written (the probe: 0 wrong words with own buffers; 1.3e5 stale words when the layers share the
//   (B, N, 36) concat rows, whose lines hold the previous layer's columns next to the current one's).  The forward therefore
//   stores each activation twice: its column block of `concat` (the API) and the dense hand-off copy.
// Residency: a waiting workgroup needs the rest of its cloud resident or scheduled; the host launches the fused kernels only
// when the whole grid fits the device at once (occupancy query; 1024 workgroups at 4 per CU for cfg2 / cfg4) and a placement
// census (workgroups of one residue mod 8 share an XCC) has passed on this device; every launch re-checks it per cloud
// (publish_placement / check_placement; a violation sets the cache's error word).  Kernels of OTHER streams (the next batch's search) only delay the dispatch of some tiles: they end
// without waiting for anything.
// Results: the tile passes are the per-layer kernels' own code (forward_tile / backward_sparse_tile): bit-identical.
#pragma once

#include "conv3p_kernels.hpp"
#include "conv3p_backward_sparse.hpp"

namespace conv3p {

constexpr int kStackMaxFused = 8;              // layers one fused launch takes (CONV3P_STACK_MAX_LAYERS)
constexpr int kSyncLineWords = 32;             // one 128-byte line per cloud and pass kind
constexpr int kSyncMaxSpins = 1 << 20;         // ~0.6 us per poll: gives up after more than half a second
#ifndef CONV3P_SYNC_SLEEP
#define CONV3P_SYNC_SLEEP 16                   // s_sleep argument between two polls (64 clocks each).  16: with 64 waiters per cloud (cfg4) polls every 64 clocks
                                               // queue up on the counter's line in front of the arrivals (fused forward 1.447 -> 1.386 ms per cfg4 step; cfg2: no effect)
#endif

// returning atomic add executed in the XCD's own L2 (sc0 = return the old value; no sc1)
__device__ __forceinline__ uint32_t l2_atomic_add_ret(uint32_t *p, uint32_t v)
{
    uint32_t old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
    return old;
}
__device__ __forceinline__ void l2_atomic_add(uint32_t *p, uint32_t v)
{
    asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }   // HW_REG_XCC_ID[3:0]

// An error of a fused launch (a wait gave up, a cloud's tiles on different XCCs): the bit goes to word 2 of the cloud's line
// (conv3p_cache_fused_status) AND to a word of host-mapped memory whose address the host left in words 4-5 of every line --
// the host looks at that word before the next fused launch on the cache and fails it loudly (CONV3P_ERR_LAUNCH).
__device__ __forceinline__ void report_error(uint32_t *line, uint32_t bit)
{
    __hip_atomic_fetch_or(line + 2, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t *host_word = reinterpret_cast<uint32_t *>(((unsigned long long)line[5] << 32) | line[4]);
    if (host_word != nullptr) __hip_atomic_fetch_or(host_word, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct CloudSync {
    static constexpr bool kActive = true;
    uint32_t *cnt;       // the cloud's line: [0] arrival counter (monotonic; the host tracks its value between launches),
                         // [1] placement word, [2] error bits (1: a wait gave up, 2: the cloud's tiles did not share an XCC)
    uint32_t target;     // the counter's value once every tile of the cloud has stored the rows this layer gathers
    bool do_wait, do_arrive;
#if CONV3P_SP_ABLATE & 4096   // developer stamps: [0] ticks spun in wait() (thread 0), [1] store drain, [2] barrier of arrive()
    long long *dbg = nullptr;
#endif
    __device__ __forceinline__ void wait() const
    {
#if CONV3P_SP_ABLATE & 4096
        const long long w0_ = wall_clock64();
#endif
        if (do_wait && threadIdx.x == 0) {
            int spins = 0;
            while ((int32_t)(l2_atomic_add_ret(cnt, 0u) - target) < 0) {
                if (++spins >= kSyncMaxSpins) {
                    report_error(cnt, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(CONV3P_SYNC_SLEEP);
            }
        }
#if CONV3P_SP_ABLATE & 4096
        if (dbg) dbg[0] += wall_clock64() - w0_;
#endif
    }
    __device__ __forceinline__ void arrive() const
    {
        if (do_arrive) {
#if CONV3P_SP_ABLATE & 4096
            const long long a0_ = wall_clock64();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have been acknowledged by L2
#if CONV3P_SP_ABLATE & 4096
            const long long a1_ = wall_clock64();
#endif
            __syncthreads();                                   // ... every wave's; and nobody reads this layer's LDS any more
#if CONV3P_SP_ABLATE & 4096
            if (dbg) { dbg[1] += a1_ - a0_; dbg[2] += wall_clock64() - a1_; }
#endif
            if (threadIdx.x == 0) l2_atomic_add(cnt, 1u);
        }
    }
};

template <typename T> struct StackFwdLayer {
    Stencil<T> st;
    const int32_t *count, *tcount;
    const PairEntry *pairs;
    const uint2 *segs, *qsegs;
    const T *input, *filter;
    T *output, *out2;          // column block of the concat / dense hand-off copy (or nullptr: last layer)
    RowLd ld;
    int ld_out2;
};
template <typename T> struct StackFwdArgs {
    const PointRec<T> *pts;
    const T *boxes, *cmin;
    int N, ntiles, nl;
    BlockMap bm;
    uint32_t *sync;            // [clouds][kSyncLineWords]
    uint32_t base;             // the counters' value before this launch
    StackFwdLayer<T> layer[kStackMaxFused];
};

// Placement check of one launch.  What the per-cloud barrier relies on is that the tiles of a cloud (workgroups b of one
// residue mod 8) share an XCC -- which XCC that is varies from launch to launch (the dispatcher's round-robin does not
// restart at XCC 0: seen on the first run of this file, which compared with the census' own table and trapped).  Tile 0 of
// the cloud publishes {launch tag, its XCC} in word 1 of the cloud's line before it first arrives; every other tile
// compares after its first wait (one more L2 atomic of one lane, once per launch) and sets bit 1 of the line's error word on a
// mismatch (conv3p_cache_fused_status).
__device__ __forceinline__ void publish_placement(uint32_t *line, uint32_t tag, int qt)
{
    if (qt == 0 && threadIdx.x == 0) {
        const uint32_t v = (tag << 4) | xcc_id();
        asm volatile("global_atomic_swap %0, %1, off" : : "v"(line + 1), "v"(v) : "memory");
    }
}
__device__ __forceinline__ void check_placement(uint32_t *line, uint32_t tag, int qt)
{
    if (qt != 0 && threadIdx.x == 0 && l2_atomic_add_ret(line + 1, 0u) != ((tag << 4) | xcc_id())) report_error(line, 2u);
}

// hidden layers 0 .. nl-1 of the stack; layer 0 has CIN0 inputs, the others H
template <typename T, int CIN0, int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void stack_forward_kernel(StackFwdArgs<T> a)
{
    int b, qt;
    if (!block_to_cloud(a.bm, b, qt)) return;   // (uniform) a workgroup past the last cloud
    uint32_t *cnt = a.sync + (size_t)b * kSyncLineWords;
    publish_placement(cnt, a.base, qt);
    {
        const StackFwdLayer<T> &L = a.layer[0];
        const CloudSync sy{cnt, a.base, false, a.nl > 1};
        forward_tile<T, CIN0, H>(a.pts, a.boxes, L.count, L.pairs, L.segs, L.qsegs, L.input, L.filter, L.st, a.N, a.ntiles, 1, CIN0, H,
                                 L.output, 1, static_cast<const T *>(nullptr), L.tcount, L.ld, b, qt, L.out2, L.ld_out2, sy);
    }
    for (int l = 1; l < a.nl; ++l) {
        const StackFwdLayer<T> &L = a.layer[l];
        const CloudSync sy{cnt, a.base + (uint32_t)(a.ntiles * l), true, l + 1 < a.nl};
        forward_tile<T, H, H>(a.pts, a.boxes, L.count, L.pairs, L.segs, L.qsegs, L.input, L.filter, L.st, a.N, a.ntiles, 1, H, H,
                              L.output, 1, static_cast<const T *>(nullptr), L.tcount, L.ld, b, qt, L.out2, L.ld_out2, sy);
        if (l == 1) check_placement(cnt, a.base, qt);
    }
}

template <typename T> struct StackBwdLayer {
    Stencil<T> st;
    const int32_t *count;
    const PairEntry *pairs;
    const uint2 *segs, *qsegs;
    const uint32_t *qbm;
    const T *grad_out, *input, *filter, *addend;
    T *grad_input, *partials;
    RowLd ld;
    int cap;
};
template <typename T> struct StackBwdArgs {
    const PointRec<T> *pts;
    const T *boxes, *cmin;
    int N, ntiles, nl;
    BlockMap bm;
    uint32_t *sync;
    uint32_t base;
    // the gradient that enters the deepest layer: g_top = ext_top * selu'(act_top), dense [B][N][H]
    const T *top_act, *top_ext;
    T *top_g;
    int ld_act, ld_ext;
    size_t nw;                 // weights per layer (taps * H * H): a workgroup's partial
    StackBwdLayer<T> layer[kStackMaxFused];   // in execution order: deepest hidden layer first
};

// the dilated H -> H layers of the stack, deepest first (populated-rows backward); the first hidden layer (undilated,
// dense G: 63 KiB of LDS) stays a launch of its own
template <typename T, int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void stack_backward_kernel(StackBwdArgs<T> a)
{
    int b, qt;
    const bool live = block_to_cloud(a.bm, b, qt);   // (uniform)
    if (!live) {   // a workgroup past the last cloud: its grad_filter partials are summed like the others
        for (int l = 0; l < a.nl; ++l) {
            T *z = a.layer[l].partials + (size_t)blockIdx.x * a.nw;
            for (uint32_t e = threadIdx.x; e < (uint32_t)a.nw; e += blockDim.x) z[e] = (T)0;
        }
        return;
    }
    uint32_t *cnt = a.sync + (size_t)b * kSyncLineWords;
    publish_placement(cnt, a.base, qt);
    {
        // the tile's own rows of the gradient that enters the deepest layer (what selu_grad_kernel computes)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int orig = a.pts[((size_t)b * a.ntiles + qt) * kTile + lane].idx;
        if (orig >= 0) {
            const size_t r = (size_t)b * a.N + orig;
            for (int c = wave; c < H; c += kWavesPerBlock) a.top_g[r * H + c] = a.top_ext[r * a.ld_ext + c] * selu_slope(a.top_act[r * a.ld_act + c]);
        }
        const CloudSync s0{cnt, a.base, false, true};
        s0.arrive();
    }
#if CONV3P_SP_ABLATE & 4096
    long long lt_[9];
    long long dbg_[3] = {0, 0, 0};
    __builtin_amdgcn_s_waitcnt(0);
    const long long ltop_ = wall_clock64();
#endif
    for (int l = 0; l < a.nl; ++l) {
        const StackBwdLayer<T> &L = a.layer[l];
#if CONV3P_SP_ABLATE & 4096
        const CloudSync sy{cnt, a.base + (uint32_t)(a.ntiles * (l + 1)), true, l + 1 < a.nl, dbg_};
#else
        const CloudSync sy{cnt, a.base + (uint32_t)(a.ntiles * (l + 1)), true, l + 1 < a.nl};
#endif
#if CONV3P_SP_ABLATE & 4096   // developer stamps: the whole tile pass of every layer, wait and arrive included
        __builtin_amdgcn_s_waitcnt(0);
        lt_[l < 7 ? l : 7] = wall_clock64();
#endif
        backward_sparse_tile<T, H, H, 0>(a.pts, a.boxes, L.count, L.pairs, L.segs, L.qsegs, L.qbm, static_cast<const uint32_t *>(nullptr),
                                             L.grad_out, L.input, L.filter, L.st, a.N, a.ntiles, 1, L.grad_input, L.partials, 1, L.addend,
                                             static_cast<const T *>(nullptr), L.ld, L.cap, true, b, qt, blockIdx.x, sy);
        if (l == 0) check_placement(cnt, a.base, qt);
    }
#if CONV3P_SP_ABLATE & 4096
    __builtin_amdgcn_s_waitcnt(0);
    lt_[a.nl < 8 ? a.nl : 8] = wall_clock64();
    if ((threadIdx.x & 63) == 0 && (blockIdx.x % 53) == 7)   // (10-ns ticks: top = the SELU-gradient rows + first arrive)
        printf("fusedbwd wg %d wave %d: top %lld  layer0 %lld  layer1 %lld  layer2 %lld  spin %lld  drain %lld  arrivebar %lld\n", (int)blockIdx.x,
               (int)(threadIdx.x >> 6), lt_[0] - ltop_, lt_[1] - lt_[0], lt_[2] - lt_[1], lt_[3] - lt_[2], dbg_[0], dbg_[1], dbg_[2]);
#endif
}

// placement census (host, once per device): XCC id of every workgroup of a grid shaped like the fused launches
__global__ __launch_bounds__(256) void xcc_census_kernel(uint32_t *out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

}  // namespace conv3p
