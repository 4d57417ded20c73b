// conv3p_abi.hip -- host side of libconv3p_hip.so: the C ABI declared in include/conv3p.h.
//
// Replaces the host glue of the reference's GPU op (tf_conv3p_atrous.cu:541-642, :659-775):
// no blocking D2H copies (stride / voxel arrive by value), no per-call temp allocation
// (caller-provided workspace / cache), no default-stream launches (everything on `stream`),
// status codes instead of OP_REQUIRES / exit().
//
// Pipeline of one op call (all on the caller's stream, no host synchronisation):
//   prep_sort_kernel   hash the clouds; re-sort + re-box only clouds whose content changed
//   search_kernel      per-tap populations + centre-major pair lists   (skipped per cloud when the slot is
//                      current; its last workgroup per cloud commits the slot)
//   forward_kernel / backward_kernel + reduce_partials_kernel          the accumulation (register path), or
//   deep_order / deep_sched / deep_plan + deep_gemm / deep_dw / deep_reduce   (matrix-core path, conv3p_deep.hpp), or
//   the generic forms of forward_kernel / backward_kernel (any channel counts, fp64)
// The stateless entry points run the same kernels on the caller's scratch with force = 1.
#include "../../include/conv3p.h"
#include "conv3p_kernels.hpp"
#include "conv3p_stack_fused.hpp"
#include "conv3p_prestep.hpp"
#include "conv3p_head.hpp"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

using namespace conv3p;

namespace {

constexpr size_t kAlign = 256;
inline size_t up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// ----------------------------------------------------------------------------- profiling
enum Kind { K_PREP = 0, K_SEARCH, K_FORWARD, K_BACKWARD, K_REDUCE, K_SELU, K_SELU_GRAD, K_MEMSET,
            K_DEEP_GEMM, K_DEEP_DW, K_TRANSPOSE, K_DEEP_ORDER, K_FC_FWD, K_FC_DX, K_FC_DW, K_NKINDS };
const char *const kKindName[K_NKINDS] = {"prep_kernel", "search_kernel", "forward_kernel",
                                         "backward_kernel", "reduce_partials_kernel", "selu_kernel",
                                         "selu_grad_kernel", "memset", "deep_gemm_kernel", "deep_dw_kernel",
                                         "transpose_filter_kernel", "deep_order_kernel", "fc_forward_kernel",
                                         "fc_dx_kernel", "fc_dw_kernel"};
struct Prof {
    std::mutex mu;
    bool on = false;
    struct Rec { int kind; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
} g_prof;

struct Scope {
    hipStream_t s;
    hipEvent_t b = nullptr;
    bool on;
    Scope(int kind, hipStream_t stream) : s(stream)
    {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        on = g_prof.on;
        if (!on) return;
        hipEvent_t a = g_prof.get();
        b = g_prof.get();
        (void)hipEventRecord(a, s);
        g_prof.recs.push_back({kind, a, b});
    }
    ~Scope()
    {
        if (on) (void)hipEventRecord(b, s);
    }
};

// ----------------------------------------------------------------------------- validation
struct Dims {
    int B, N, Cin, Cout, fz, fy, fx;
    int ntap, ntiles;
};

int check(Dims &d, const int32_t *stride, double voxel, bool need_channels)
{
    // mirrors the reference's OP_REQUIRES checks as far as a flat C signature can
    // (ranks and matching batch sizes are the host mirror's job, .cpp:410-443)
    if (d.B < 0 || d.N < 0) return CONV3P_ERR_INVALID_ARGUMENT;
    if (d.fz <= 0 || d.fy <= 0 || d.fx <= 0) return CONV3P_ERR_INVALID_ARGUMENT;
    if (need_channels && (d.Cin < 0 || d.Cout < 0)) return CONV3P_ERR_INVALID_ARGUMENT;
    if (!stride || stride[0] <= 0 || stride[1] <= 0 || stride[2] <= 0) return CONV3P_ERR_INVALID_ARGUMENT;
    if (!(voxel > 0.0)) return CONV3P_ERR_INVALID_ARGUMENT;   // also rejects NaN
    const long long ntap = (long long)d.fz * d.fy * d.fx;
    if (ntap >= (long long)kNoTap) return CONV3P_ERR_UNSUPPORTED;   // taps are 12-bit fields of the pair code
    d.ntap = (int)ntap;
    d.ntiles = (d.N + kTile - 1) / kTile;
    const int ext[3] = {d.fx, d.fy, d.fz};
    for (int a = 0; a < 3; ++a) {
        const long long full = (long long)(ext[a] - 1) * stride[a] + 1;
        if (full > 4096) return CONV3P_ERR_UNSUPPORTED;
    }
    return CONV3P_OK;
}

template <typename T> Stencil<T> make_stencil(const Dims &d, const int32_t *stride, T voxel)
{
    Stencil<T> st;
    st.ext[0] = d.fx; st.ext[1] = d.fy; st.ext[2] = d.fz;
    st.maxfull = 1;
    for (int a = 0; a < 3; ++a) {
        st.step[a] = stride[a];
        st.full[a] = (st.ext[a] - 1) * st.step[a] + 1;
        st.half[a] = ((double)st.full[a] * 0.5) * (double)voxel;   // .cpp:240, evaluated in double
        if (st.full[a] > st.maxfull) st.maxfull = st.full[a];
        st.inv[a] = (float)(1.0 / ((double)st.step[a] * (double)voxel));
        st.shift[a] = (float)((1.0 - 1.0 / (double)st.step[a]) * 0.5 - 0.5);
        st.halfw[a] = (float)(0.5 / (double)st.step[a]);
        st.mmax[a] = (float)(st.ext[a] - 1);
        st.reach[a] = (int)((st.full[a] + 1) * 0.5);              // .cpp:247-249
    }
    st.window = ((st.full[0] & 1) == 0 || (st.full[1] & 1) == 0 || (st.full[2] & 1) == 0) ? 1 : 0;
    st.ntap = d.ntap;
    st.voxel = voxel;
    return st;
}

// 64-bit tag of a stencil: what a cache slot was built for
unsigned long long stencil_tag(const Dims &d, const int32_t *stride, double voxel, int elem)
{
    unsigned long long h = 0xcbf29ce484222325ull;
    auto mix = [&](unsigned long long v) {
        h ^= v;
        h *= 0x100000001b3ull;
        h ^= h >> 29;
    };
    unsigned long long vb;
    std::memcpy(&vb, &voxel, 8);
    mix((unsigned long long)d.fz); mix((unsigned long long)d.fy); mix((unsigned long long)d.fx);
    mix((unsigned long long)stride[0]); mix((unsigned long long)stride[1]); mix((unsigned long long)stride[2]);
    mix(vb); mix((unsigned long long)elem); mix((unsigned long long)d.B); mix((unsigned long long)d.N);
    return h | 1ull;
}

BlockMap make_blockmap(const Dims &d)
{
    BlockMap m;
    m.blocks_per_cloud = d.ntiles;   // one workgroup per query tile
    m.clouds = d.B;
    m.rounds = (d.B + 7) / 8;
    return m;
}
inline unsigned grid_of(const BlockMap &m) { return 8u * (unsigned)m.rounds * (unsigned)m.blocks_per_cloud; }
// (Tried and dropped: two consecutive query tiles per backward workgroup -- half the grad_filter partials, one resident
// round of 512 workgroups -- 0.247 -> 0.306 ms/step: the kernel is latency-bound per workgroup and wants as many tiles
// in flight as fit; and handling the taps in 2-3 ranges to shrink G and fit a third workgroup per CU -- 0.247 -> 0.385.)

// ----------------------------------------------------------------------------- buffer layout
// One layout serves both the per-call workspace (1 slot, rebuilt every call) and the persistent
// neighbour cache (several slots).  Everything a slot holds depends only on (points, stencil).
constexpr int kGroupTiles = 128;      // candidate tiles per search group (64 KiB of hit masks in LDS)
constexpr int kFusedMaxPoints = 16384;   // clouds the prep kernel sorts: only their tiles are compact enough for the window tables
constexpr int kDefaultPairsPerPoint = 256;   // 16-B records; room-like clouds reach ~170 neighbours/point

template <typename T> struct Layout {
    // per cloud, independent of the stencil
    unsigned long long *hash;
    uint32_t *version;
    PointRec<T> *pts;
    T *boxes;
    T *cmin;   // [B][3] origin of the reference's uniform grid (stencils with an even dilated extent only)
    unsigned long long *ftab;   // [B][ntiles][kFTableU64] per-tile window tables of the fused search (sorted clouds only)
    uint32_t *tab_version, *tab_ticket;   // [B] version of the cloud its tables were built from / tiles done (tile_tables_kernel)
    uint32_t *tab_inv;                    // [B] bits of the inv16 (= 16 / voxel) the tables were built with
    // per slot
    struct Slot {
        uint32_t *built_version, *cursor, *ticket;
        unsigned long long *built_tag;
        int32_t *count, *tcount;   // populations: original-index order (B,N,F) / tile-major [tile][F][64]
        uint2 *segs, *qsegs;
        uint32_t *qbm;             // [B][ntiles][64] backward taps each centre's list holds (bit f' of taps 0 .. 31)
        uint32_t *qbm_hi;          // ... taps 32 .. 63 (caches / workspaces for filters of more than 32 taps; else nullptr)
        uint32_t *sched;           // [8][ceil(B / 8) * ntiles] launch order of the tiles per XCD (tile_sched_kernel)
        uint32_t *regime;          // [1] 1: short pair lists on average (tile_sched_kernel), see SchedJob
        PairEntry *pairs;
    };
    std::vector<Slot> slot;
    uint32_t *cursor_all;   // [nslots][2][B]: pair allocator and completion ticket of every slot
    int nclouds;
    uint32_t pairs_per_cloud;
    int gtiles, ngroups;
    // per-call scratch
    T *partials;
    size_t bytes;
};

// ntap_max: taps the slots must be able to hold; scratch_bytes: bytes of per-call scratch
template <typename T>
Layout<T> carve(int B, int N, int ntiles, int ntap_max, int nslots, int pairs_per_point, size_t scratch_bytes,
                void *base)
{
    Layout<T> L;
    size_t off = 0;
    char *p = static_cast<char *>(base);
    auto take = [&](size_t n) { char *r = p ? p + off : nullptr; off += up(n); return r; };
    // hit masks take 512 B per candidate tile of a group; leave room for the [taps][65] populations
    int gmax = kGroupTiles;
    {
        const size_t fixed = (size_t)ntap_max * kCntStride * 4 + 16 * 1024;
        const size_t room = fixed < 150 * 1024 ? 150 * 1024 - fixed : 0;
        const int fit = (int)(room / 516);
        if (fit < gmax) gmax = fit < 4 ? 4 : fit;
    }
    L.gtiles = ntiles < gmax ? (ntiles > 0 ? ntiles : 1) : gmax;
    L.ngroups = ntiles > 0 ? (ntiles + L.gtiles - 1) / L.gtiles : 1;
    L.hash = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * (size_t)B));
    L.version = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t) * (size_t)B));
    L.pts = reinterpret_cast<PointRec<T> *>(take(sizeof(PointRec<T>) * (size_t)B * ntiles * kTile));
    L.boxes = reinterpret_cast<T *>(take(sizeof(T) * (size_t)B * ntiles * 6));
    L.cmin = reinterpret_cast<T *>(take(sizeof(T) * (size_t)B * 3));
    L.ftab = N <= kFusedMaxPoints
                 ? reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * (size_t)B * ntiles * kFTableU64))
                 : nullptr;
    L.tab_version = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t) * (size_t)B));
    L.tab_ticket = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t) * (size_t)B));
    L.tab_inv = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t) * (size_t)B));
    size_t ppc = (size_t)N * (size_t)pairs_per_point;
    if ((size_t)B * ppc > 0xFFFFFFF0ull) ppc = B ? 0xFFFFFFF0ull / (size_t)B : 0;
    if (ppc > 0x7FFFFFFFull) ppc = 0x7FFFFFFFull;   // headroom for the allocator's transient overshoot (search_tile P2)
    L.pairs_per_cloud = (uint32_t)ppc;
    L.slot.resize(nslots);
    L.nclouds = B;
    L.cursor_all = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t) * (size_t)B * nslots * 2));
    for (int s = 0; s < nslots; ++s) {
        auto &S = L.slot[s];
        S.built_version = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t) * (size_t)B));
        S.cursor = L.cursor_all ? L.cursor_all + (size_t)(2 * s) * B : nullptr;
        S.ticket = L.cursor_all ? L.cursor_all + (size_t)(2 * s + 1) * B : nullptr;
        S.built_tag = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * (size_t)B));
        S.count = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)B * N * ntap_max));
        S.tcount = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)B * ntiles * kTile * ntap_max));
        S.segs = reinterpret_cast<uint2 *>(take(sizeof(uint2) * (size_t)B * ntiles * L.ngroups));
        S.qsegs = reinterpret_cast<uint2 *>(take(sizeof(uint2) * (size_t)B * ntiles * L.ngroups * 64));
        S.qbm = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t) * (size_t)B * ntiles * 64 * (ntap_max > 64 ? 4 : ntap_max > 32 ? 2 : 1)));   // one plane per 32 taps (up to 128)
        S.qbm_hi = ntap_max > 32 && S.qbm != nullptr ? S.qbm + (size_t)B * ntiles * 64 : nullptr;
        S.sched = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t) * 8 * (size_t)((B + 7) / 8) * ntiles));
        S.regime = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t)));
        S.pairs = reinterpret_cast<PairEntry *>(take(sizeof(PairEntry) * (size_t)B * ppc));
    }
    L.partials = reinterpret_cast<T *>(take(scratch_bytes));
    L.bytes = off;
    return L;
}

// ----------------------------------------------------------------------------- dispatch table
// (Cin, Cout) pairs with register-resident rows: the layers of the reference's two models
// (pointcnn2_acsd.py:48-66: Cin->9, 9->9; pointcnn_scene_seg_acsd.py:51-57: +36->num_class,
// 13 classes for S3DIS) plus a few neighbours.  Everything else takes the generic path.
#define CONV3P_SMALL_SHAPES(X) X(3, 9) X(6, 9) X(9, 9) X(12, 9) X(36, 13) X(3, 3) X(9, 3)

inline bool small_shape(int elem, int cin, int cout)
{
#ifdef CONV3P_DEV_SKIP_SMALL   // developer build only (-DCONV3P_DEV_SKIP_SMALL): time the other paths on the models' shapes
    return false;
#endif
    (void)elem;   // fp32 and fp64 (the kernels are templates on T; shapes whose LDS does not fit fall back at launch)
#define X(ci, co) if (cin == ci && cout == co) return true;
    CONV3P_SMALL_SHAPES(X)
#undef X
    return false;
}

inline bool deep_shape(int elem, int cin, int cout);
struct DeepScratch;
size_t deep_scratch_bytes(const Dims &d, size_t pair_slots);
size_t deep_forward_bytes(const Dims &d, size_t pair_slots);

// Generic backward: every pair adds Cin*Cout products to grad_filter with global atomics.  One shared copy
// serialises them all on the same few addresses (cfg2-sized 5->7: 67 ms); the workgroups therefore spread over
// `slots` partial copies (workgroup % slots), at most one per workgroup and at most 64 MiB in total, summed by
// reduce_partials_kernel.
inline int generic_slots(const Dims &d, int elem)
{
    const size_t nw = (size_t)d.ntap * d.Cin * d.Cout;
    const size_t grid = (size_t)grid_of(make_blockmap(d));
    size_t s = nw ? ((size_t)64 << 20) / (nw * (size_t)elem) : 1;
    if (s > grid) s = grid;
    return s < 1 ? 1 : (int)s;
}

// fp32 layers with more than 256 channels on a side (round 4): column blocks of at most 256 x 256 channels on the
// matrix-core kernels -- the op is linear in the input-channel blocks and independent over the output-channel blocks
// -- instead of the global-atomics kernels.  Scratch: the matrix-core path's own for a 256 x 256 block, then packed
// copies of the block's rows (input / grad_out / result: 3 x [B N][256]) and of its filter block and grad_filter block.
constexpr int kWideBlk = 256;
inline bool wide_shape(int elem, int cin, int cout)
{
    return elem == 4 && cin >= 1 && cout >= 1 && (cin > kWideBlk || cout > kWideBlk);
}
inline size_t wide_scratch_bytes(const Dims &d, size_t pair_slots, bool forward_only = false)
{
    Dims db = d;
    db.Cin = kWideBlk;    // (blocks are padded to 128 or 256 channels: sized for the largest)
    db.Cout = kWideBlk;
    const size_t rows = (size_t)d.B * d.N;
    return up(forward_only ? deep_forward_bytes(db, pair_slots) : deep_scratch_bytes(db, pair_slots)) + 3 * up(rows * kWideBlk * 4) +
           2 * up((size_t)d.ntap * kWideBlk * kWideBlk * 4);
}

// fp64 layers outside the register-path shapes (round 4): blocks of 16 input x 8 output channels, zero-padded, on the
// register-path kernels <double, 16, 8> (their G matrix, 27 x 8 rows of 65 doubles = 112 KB, and transposed filter fit LDS
// in double) -- the op is linear in the input-channel blocks and independent over the output-channel blocks -- instead
// of the global-atomics kernels.  Block sizes measured in round 6 (profiles/r06_f64_blocks.txt; 32 -> 64 at the cfg2
// size): 16 x 4 1.36 / 3.81 ms, 16 x 8 0.91 / 3.41, 16 x 16 forward 1.21, 32 x 16 forward 0.90, 32 x 4 backward 3.77.
// Scratch: the block kernel's grad_filter partials, packed rows (input [B N][16], grad_out / result [B N][8] and
// [B N][16]) and the packed filter block and its grad_filter block.
#ifndef CONV3P_F64_FWD_KI
#define CONV3P_F64_FWD_KI 16
#define CONV3P_F64_FWD_CO 8
#endif
#ifndef CONV3P_F64_BWD_KI
#define CONV3P_F64_BWD_KI 16
#define CONV3P_F64_BWD_CO 8
#endif
constexpr int kF64FwdKi = CONV3P_F64_FWD_KI, kF64FwdCo = CONV3P_F64_FWD_CO;   // block of the forward pass
constexpr int kF64BwdKi = CONV3P_F64_BWD_KI, kF64BwdCo = CONV3P_F64_BWD_CO;   // block of the backward pass
constexpr int kF64Ki = kF64FwdKi > kF64BwdKi ? kF64FwdKi : kF64BwdKi;         // (scratch: sized for the larger of the two)
constexpr int kF64Co = kF64FwdCo > kF64BwdCo ? kF64FwdCo : kF64BwdCo;
constexpr int kF64Row = kF64Ki > kF64Co ? kF64Ki : kF64Co;
inline bool f64_blocked_shape(int elem, int cin, int cout) { return elem == 8 && cin >= 1 && cout >= 1 && !small_shape(elem, cin, cout); }
inline size_t f64_blocked_bytes(const Dims &d)
{
    const size_t rows = (size_t)d.B * d.N, nwb = (size_t)d.ntap * kF64Ki * kF64Co;
    return up((size_t)grid_of(make_blockmap(d)) * nwb * 8) + 3 * up(rows * kF64Row * 8) + 2 * up(nwb * 8);
}

// ppp: pair slots per point of the buffer the call runs in (a cache may be configured with fewer than the default)
size_t backward_scratch_bytes(const Dims &d, int elem, int ppp = kDefaultPairsPerPoint)
{
    const size_t nw = (size_t)d.ntap * d.Cin * d.Cout;
    size_t deep = 0;
    if (wide_shape(elem, d.Cin, d.Cout)) deep = wide_scratch_bytes(d, (size_t)d.B * d.N * (size_t)ppp);
    if (f64_blocked_shape(elem, d.Cin, d.Cout)) deep = f64_blocked_bytes(d);
    if (!small_shape(elem, d.Cin, d.Cout) && deep_shape(elem, d.Cin, d.Cout))
        deep = deep_scratch_bytes(d, (size_t)d.B * d.N * (size_t)ppp);
    // register path: one partial per workgroup; generic path (also the fallback of the other two): generic_slots
    const size_t slots = small_shape(elem, d.Cin, d.Cout) ? (size_t)grid_of(make_blockmap(d)) : (size_t)generic_slots(d, elem);
    const size_t plain = nw * slots * (size_t)elem;
    return deep > plain ? deep : plain;
}

// Shapes whose forward runs as transform + gather (conv3p_forward_taps.hpp): fp32 register-path shapes with at least
// 16 inputs and at most 16 outputs.  Their scratch: Z [B N][ntap][16] floats.
inline bool tap_forward_shape(int elem, int cin, int cout)
{
#ifdef CONV3P_DEV_NO_TAP_FORWARD   // developer A/B build
    return false;
#endif
    return elem == 4 && small_shape(elem, cin, cout) && cin >= 16 && cin % 4 == 0 && cout <= 16;
}
inline size_t tap_forward_bytes(const Dims &d) { return (size_t)d.B * d.N * (size_t)d.ntap * kZRow * 4; }

// mandatory_only: what the forward cannot run without.  The transform + gather forward's Z array is optional (without
// it the layer runs on forward_kernel, same results): a persistent cache sized for narrower layers must not be refused
// for it (Call::tap_scratch_ok decides per call).
size_t forward_scratch_bytes(const Dims &d, int elem, int ppp = kDefaultPairsPerPoint, bool mandatory_only = false)
{
    if (tap_forward_shape(elem, d.Cin, d.Cout)) return mandatory_only ? 0 : tap_forward_bytes(d);
    if (!small_shape(elem, d.Cin, d.Cout) && deep_shape(elem, d.Cin, d.Cout))
        return deep_forward_bytes(d, (size_t)d.B * d.N * (size_t)ppp);
    if (wide_shape(elem, d.Cin, d.Cout)) return mandatory_only ? 0 : wide_scratch_bytes(d, (size_t)d.B * d.N * (size_t)ppp, true);
    if (f64_blocked_shape(elem, d.Cin, d.Cout)) return mandatory_only ? 0 : f64_blocked_bytes(d);
    return 0;
}

bool cache_cfg_ok(const conv3p_cache_config *cfg);

int hip_ok()
{
    return hipGetLastError() == hipSuccess ? CONV3P_OK : CONV3P_ERR_LAUNCH;
}

template <typename T> size_t lds_common(const Stencil<T> &st) { return (3 * (size_t)st.maxfull * 2 + 15) & ~(size_t)15; }
inline size_t a16(size_t x) { return (x + 15) & ~(size_t)15; }
constexpr size_t kMaxLds = 160 * 1024;

#define TRY(expr) do { int rc_ = (expr); if (rc_ != CONV3P_OK) return rc_; } while (0)

// Everything one call needs.
template <typename T> struct Call {
    Dims d;
    Stencil<T> st;
    Layout<T> L;
    int slot;
    CacheCtl cc;
    hipStream_t s;
    bool deep_scratch_ok = false;  // the scratch region can hold the deep path's side arrays
    bool deep_fwd_ok = false;      // ... what its FORWARD needs of them (no G tiles, no partials)
    bool wide_fwd_ok = false;      // ... the forward of the blocked path (packs placed after the forward-only deep part)
    bool tap_scratch_ok = false;   // ... the transform + gather forward's Z array
    bool wide_scratch_ok = false;  // ... the blocked path of layers with more than 256 channels
    bool f64_scratch_ok = false;   // ... the blocked path of fp64 layers outside the register-path shapes
    bool order_ok[2] = {false, false};   // the deep path's forward / backward record order of this geometry is in the scratch
    void *cache_key = nullptr;     // persistent cache the call runs in (host bookkeeping of the record orders)
    uint64_t gen = 0;
    bool evicted_hinted = false;   // slot re-assigned to a new stencil while prep is skipped: reset its allocators
    bool skip_prep = false;     // caller promised unchanged points
    bool skip_search = false;   // ... and this slot's lists were already enqueued for them
    // conv3p_layer_*: SELU fused into the op (pointcnn2_acsd.py:48-49).  forward: output = selu(conv);
    // backward: grad_input = (dX + addend) * selu'(input)
    bool act = false;
    bool sparse_hint = false;      // CONV3P_CACHE_SPARSE_NEIGHBOURHOODS: short pair lists expected (see conv3p.h)
    bool dense_hint = false;       // CONV3P_CACHE_DENSE_NEIGHBOURHOODS: long ones (neither: decided on the device)
    bool accum = false;            // backward: add to the grad_input already there (column-split passes)
    const T *addend = nullptr;
    RowLd ld{0, 0, 0, 0, 0};       // row strides of the feature tensors; filled with the dense values by set_ld()
    bool strided = false;          // some tensor is a column block of a wider buffer (register-path shapes only)
};

template <typename T> void set_ld(Call<T> &c, const RowLd *ld, int Cin, int Cout)
{
    c.ld = RowLd{Cin, Cout, Cout, Cin, Cin};
    if (ld) {
        if (ld->in > 0) c.ld.in = ld->in;
        if (ld->out > 0) c.ld.out = ld->out;
        if (ld->dy > 0) c.ld.dy = ld->dy;
        if (ld->dx > 0) c.ld.dx = ld->dx;
        if (ld->add > 0) c.ld.add = ld->add;
    }
    c.strided = c.ld.in != Cin || c.ld.out != Cout || c.ld.dy != Cout || c.ld.dx != Cin || c.ld.add != Cin;
}

template <typename T> CacheCtl make_ctl(const Layout<T> &L, int slot, unsigned long long tag, uint32_t epoch, int force)
{
    CacheCtl cc;
    cc.hash = L.hash;
    cc.version = L.version;
    cc.built_version = L.slot[slot].built_version;
    cc.built_tag = L.slot[slot].built_tag;
    cc.cursor = L.slot[slot].cursor;
    cc.ticket = L.slot[slot].ticket;
    cc.cursor_all = L.cursor_all;
    cc.tab_version = L.tab_version;
    cc.tab_ticket = L.tab_ticket;
    cc.nslots = (int)L.slot.size();
    cc.nclouds = L.nclouds;
    cc.tag = tag;
    cc.epoch = epoch;
    cc.pairs_per_cloud = L.pairs_per_cloud;
    cc.force = force;
    return cc;
}

template <typename T> int run_prep(const T *points, const Call<T> &c)
{
    if (c.skip_prep) {
        if (c.evicted_hinted)   // prep_sort_kernel would have done this for an un-hinted call
            return hipMemsetAsync(c.L.slot[c.slot].cursor, 0, sizeof(uint32_t) * 2 * (size_t)c.d.B, c.s) == hipSuccess
                       ? CONV3P_OK : CONV3P_ERR_LAUNCH;
        return CONV3P_OK;
    }
    const Dims &d = c.d;
    Scope sc(K_PREP, c.s);
    if (d.N <= 16384 && d.N > kTile) {
        int npad = 128;
        while (npad < d.N) npad <<= 1;
        const int threads = npad / 2 < 1024 ? (npad / 2 < 64 ? 64 : npad / 2) : 1024;
        const size_t lds = (size_t)npad * 8 + 256;
        auto launch = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(d.B), dim3(threads), lds, c.s, points, d.N, d.ntiles, npad, c.L.pts, c.L.boxes, c.cc);
        };
        switch (npad / threads) {   // keys per thread
        case 2: launch(prep_sort_kernel<T, 2>); break;
        case 4: launch(prep_sort_kernel<T, 4>); break;
        case 8: launch(prep_sort_kernel<T, 8>); break;
        default: launch(prep_sort_kernel<T, 16>); break;
        }
        return hip_ok();
    }
    dim3 grid((d.ntiles + kWavesPerBlock - 1) / kWavesPerBlock, d.B);
    hipLaunchKernelGGL(prep_kernel<T>, grid, dim3(256), 0, c.s, points, d.N, d.ntiles, c.L.pts, c.L.boxes, c.cc);
    return hip_ok();
}

template <typename T> size_t search_lds_bytes(const Stencil<T> &st, int gtiles)
{
    // (+ 1024: search_tile's static array of the centres' backward-tap sets -- it counts against the 160 KiB like the dynamic part)
    return lds_common(st) + a16((size_t)st.ntap * kCntStride * 4) + a16(sizeof(CentreRec<T>) * 64) + 64 * 3 * 4 +
           a16((size_t)gtiles * 64 * 8) + a16((size_t)gtiles * 4) + 32 + kWavesPerBlock * 64 * 4 +
           a16((size_t)kWavesPerBlock * 192 * 4) + a16((size_t)kWavesPerBlock * 256 * 4) + 1024;
}

// grid origin of every cloud, for stencils whose candidate window can decide (even dilated extents)
template <typename T> int run_cloud_min(const T *points, const Call<T> &c)
{
    if (!c.st.window) return CONV3P_OK;
    hipLaunchKernelGGL(cloud_min_kernel<T>, dim3((unsigned)c.d.B), dim3(256), 0, c.s, points, c.d.N, c.L.cmin);
    return hip_ok();
}

// search.  count: where the populations go.
// launch order of a slot's tiles for the list-walking kernels
template <typename SlotT> inline const uint32_t *sched_of(const SlotT &S)
{
#ifdef CONV3P_DEV_NO_SCHED   // developer A/B build: BlockMap order
    return nullptr;
#else
    return S.sched;
#endif
}

// mean pre-filter hits per point below which a slot's lists count as SHORT (populated-rows backward for the dilated narrow
// layers): Conv3pStack.tune()'s threshold of 32 neighbours per point plus the pre-filter's ~10 % of false positives
// The regime word only matters for DILATED narrow layers (undilated ones never take the populated-rows kernel).  Of a
// dilated stencil's hits the window tables of the fused search let through ~1.5 per neighbour (a third are false
// positives, see fused_compacts), the tile-pair pre-filter of search_tile ~1.1: the same 32 neighbours per point are 48
// hits there and 35 here.
constexpr unsigned long long kShortListsPerPoint = 35, kShortListsPerPointFused = 48;
template <typename T> bool fused_ok(const Call<T> &c);
template <typename T> SchedJob make_sched_job(const Call<T> &c, int slot, bool fused)
{
    const auto &S = c.L.slot[slot];
    return SchedJob{S.segs, S.sched, S.cursor, S.regime,
                    (fused ? kShortListsPerPointFused : kShortListsPerPoint) * (unsigned long long)c.d.B * (unsigned long long)c.d.N};
}

// ----------------------------------------------------------------------------- fused search (conv3p_search_fused.hpp)
#ifndef CONV3P_DEV_FUSED_M
#define CONV3P_DEV_FUSED_M 6     // candidate tiles per wave whose hit masks stay in LDS between the passes
#endif
template <typename T> bool fused_ok(const Call<T> &c)
{
#ifdef CONV3P_DEV_NO_FUSED_SEARCH   // developer A/B build: the tile-pair pre-filter of search_tile for everything
    return false;
#endif
    const Stencil<T> &st = c.st;
    if (c.L.ftab == nullptr || st.window) return false;
    for (int a = 0; a < 3; ++a)
        if (st.ext[a] > kFMaxExt) return false;
    return true;
}
template <typename T> FusedJob<T> make_fused_job(const Call<T> &c)
{
    const auto &S = c.L.slot[c.slot];
    FusedJob<T> j;
    j.st = c.st;
    j.cc = c.cc;
    j.count = S.count;
    j.tcount = S.tcount;
    j.pairs = S.pairs;
    j.segs = S.segs;
    j.qsegs = S.qsegs;
    j.qbm = S.qbm;
    j.qbm_hi = S.qbm_hi;
    for (int a = 0; a < 3; ++a)
        for (int k = 0; k < kFMaxExt; ++k)
            j.clo[a][k] = (float)(((double)k * c.st.step[a] - (double)c.st.full[a] * 0.5) * (double)kFR);
    return j;
}
// tables of every tile + ONE search launch for `njobs` stencils over the same points + the launch order of the tiles
// hit masks of a wave's first candidate tiles kept in LDS between the passes: as many as fit 40 KiB (0: not even one fits
// the LDS at all)
inline int fused_mask_depth(int elem, int ntiles, int ntap_max, int maxfull_max)
{
    const int mmax = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    int M = std::min(mmax, (int)CONV3P_DEV_FUSED_M);
    // small clouds (<= 32 tiles: a wave meets two or three candidate tiles): ONE stored mask -- 31 KiB instead of 41, a
    // fifth workgroup per CU; recomputing the others in pass 2 costs less than the occupancy gives (cfg2 search alone
    // 111 -> 106 us, under the backward 141 -> 123; on the rooms' 64 tiles per cloud the step does not move)
    if (ntiles <= 32) M = 1;
    if (M < 1) M = 1;
    while (M > 1 && fused_lds(ntap_max, maxfull_max, elem, M).total > 40 * 1024) --M;   // large filters: fewer stored masks
    return fused_lds(ntap_max, maxfull_max, elem, M).total > kMaxLds ? 0 : M;
}
template <typename T> int launch_fused(const Call<T> &c, const FusedJobs<T> &jobs, const SchedJobs &sjobs, int njobs, bool schedule = true)
{
    const Dims &d = c.d;
    const BlockMap bm = make_blockmap(d);
    int ntap_max = 1, maxfull_max = 1;
    for (int k = 0; k < njobs; ++k) {
        ntap_max = std::max(ntap_max, jobs.job[k].st.ntap);
        maxfull_max = std::max(maxfull_max, jobs.job[k].st.maxfull);
    }
    const int M = fused_mask_depth((int)sizeof(T), d.ntiles, ntap_max, maxfull_max);
    if (M == 0) return CONV3P_ERR_UNSUPPORTED;   // (callers test fused_fits() first and take the tile-pair search instead)
#ifdef CONV3P_DEV_FUSED_LDS_KB   // developer: pad the LDS request (fewer workgroups per CU) to see how the kernel scales with occupancy
    const size_t lds = std::max(fused_lds(ntap_max, maxfull_max, (int)sizeof(T), M).total, (size_t)CONV3P_DEV_FUSED_LDS_KB * 1024);
#else
    const size_t lds = fused_lds(ntap_max, maxfull_max, (int)sizeof(T), M).total;
#endif
    if (lds > kMaxLds) return CONV3P_ERR_UNSUPPORTED;
    const float inv16 = (float)((double)kFR / (double)c.st.voxel);
    Scope sc(K_SEARCH, c.s);
    hipLaunchKernelGGL(tile_tables_kernel<T>, dim3((d.ntiles + kWavesPerBlock - 1) / kWavesPerBlock, d.B), dim3(256), 0, c.s,
                       c.L.pts, d.ntiles, inv16, c.L.ftab, c.cc.version, c.L.tab_version, c.L.tab_ticket, c.L.tab_inv, c.cc.force);
    bool ext3 = true;
    for (int k = 0; k < njobs; ++k)
        for (int a = 0; a < 3; ++a) ext3 &= jobs.job[k].st.ext[a] == 3;
    auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid_of(bm), njobs), dim3(256), lds, c.s, c.L.pts, c.L.boxes, c.L.ftab, d.N, d.ntiles,
                           c.L.ngroups, bm, jobs, M, inv16);
    };
    if (ext3) launch(search_fused_kernel<T, true>);
    else launch(search_fused_kernel<T, false>);
#ifndef CONV3P_DEV_NO_SCHED   // developer A/B build: no launch order at all (BlockMap order; needs a hint: no regime word either)
    if (schedule)
        hipLaunchKernelGGL(tile_sched_kernel, dim3(8, njobs), dim3(1024), 0, c.s, sjobs, d.B, d.ntiles, c.L.ngroups, bm.rounds * d.ntiles);
#endif
    return hip_ok();
}

// the other stencils a persistent cache holds, as jobs of the same launch (defined with the cache's host record below)
template <typename T> int add_companions(const Call<T> &c, FusedJobs<T> &fj, SchedJobs &sj, int n);
template <typename T> void note_companions_built(const Call<T> &c, const FusedJobs<T> &fj, int n);

template <typename T> int run_search(const Call<T> &c, int32_t *count, bool with_pairs)
{
    if (c.skip_search && with_pairs) return CONV3P_OK;
    const Dims &d = c.d;
    const Stencil<T> &st = c.st;
    const auto &S = c.L.slot[c.slot];
    if (fused_ok(c) && fused_mask_depth((int)sizeof(T), d.ntiles, st.ntap, st.maxfull) > 0 && (!with_pairs || count == S.count)) {
        FusedJobs<T> fj;
        SchedJobs sj;
        fj.job[0] = make_fused_job(c);
        sj.job[0] = make_sched_job(c, c.slot, true);
        if (!with_pairs) {   // populations only (conv3p_neighbor_count_*): into the caller's tensor, nothing else kept
            FusedJob<T> &j = fj.job[0];
            j.count = count;
            j.tcount = nullptr;
            j.pairs = nullptr;
            j.segs = nullptr;
            j.qsegs = nullptr;
            j.qbm = nullptr;
            j.qbm_hi = nullptr;
            return launch_fused<T>(c, fj, sj, 1, /*schedule=*/false);
        }
        const int nj = add_companions<T>(c, fj, sj, 1);
        TRY(launch_fused<T>(c, fj, sj, nj));
        note_companions_built<T>(c, fj, nj);
        return CONV3P_OK;
    }
    const size_t lds = search_lds_bytes(st, c.L.gtiles);
    if (lds > kMaxLds) return CONV3P_ERR_UNSUPPORTED;   // very large filters (> ~340 taps): populations alone exceed LDS
    const BlockMap bm = make_blockmap(d);
    {
        Scope sc(K_SEARCH, c.s);
        auto launch = [&](auto kern, const T *cmin) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(grid_of(bm)), dim3(256), lds, c.s, c.L.pts, c.L.boxes, st, d.N, d.ntiles,
                               c.L.gtiles, c.L.ngroups, bm, count, with_pairs ? S.pairs : nullptr, c.cc, S.segs, S.qsegs,
                               cmin, with_pairs ? S.tcount : nullptr, with_pairs ? S.qbm : nullptr, with_pairs ? S.qbm_hi : nullptr);
        };
        if (st.window) launch(search_kernel<T, true>, c.L.cmin);
        else launch(search_kernel<T, false>, static_cast<const T *>(nullptr));
        if (with_pairs) {
            SchedJobs sj;
            sj.job[0] = make_sched_job(c, c.slot, false);
            hipLaunchKernelGGL(tile_sched_kernel, dim3(8, 1), dim3(1024), 0, c.s, sj, d.B, d.ntiles, c.L.ngroups, bm.rounds * d.ntiles);
        }
    }
    return hip_ok();
}

#ifdef CONV3P_DEV_WALK_STATS
// developer instrumentation build only (-DCONV3P_DEV_WALK_STATS, tools/walk_stats.py): how well the list-walking kernels'
// lane mapping uses a wave.  Per query tile, from the per-centre segment table: records, steps of the shipped mapping
// (wave = 16 centres x 4 sub-lanes: ceil(longest list / 4)), steps with the lanes of a wave dealt to its centres in
// proportion to their lists, the same over the whole workgroup, and the bound ceil(records / 64).
__global__ void walk_stats_kernel(const uint2 *qsegs, const PairEntry *pairs, int ntiles_all, unsigned long long *out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles_all) return;
    uint32_t len[64];
    unsigned long long recs = 0, live = 0;
    for (int c = 0; c < 64; ++c) {
        const uint2 sg = qsegs[(size_t)t * 64 + c];
        len[c] = sg.y == kSegOverflow ? 0u : sg.y;
        recs += len[c];
        for (uint32_t i = 0; i < len[c]; ++i) live += code_fwd(pairs[sg.x + i].code) != kNoTap ? 1u : 0u;
    }
    unsigned long long cur = 0, pslw = 0, ideal = 0, curmax = 0, pslmax = 0, idealmax = 0;
    for (int w = 0; w < 4; ++w) {
        uint32_t mx = 0, sum = 0;
        for (int c = 0; c < 16; ++c) { mx = max(mx, len[16 * w + c]); sum += len[16 * w + c]; }
        const unsigned long long a = (mx + 3) / 4;
        uint32_t S = max(1u, (sum + 63) / 64);
        for (;; ++S) {
            uint32_t need = 0;
            for (int c = 0; c < 16; ++c) need += (len[16 * w + c] + S - 1) / S;
            if (need <= 64) break;
        }
        const unsigned long long idl = (sum + 63) / 64;
        cur += a; pslw += S; ideal += idl;
        curmax = max(curmax, a); pslmax = max(pslmax, (unsigned long long)S); idealmax = max(idealmax, idl);
    }
    uint32_t St = max(1u, (uint32_t)((recs + 255) / 256));
    for (;; ++St) {
        uint32_t need = 0;
        for (int c = 0; c < 64; ++c) need += (len[c] + St - 1) / St;
        if (need <= 256) break;
    }
    atomicAdd(&out[0], recs); atomicAdd(&out[1], live); atomicAdd(&out[2], cur); atomicAdd(&out[3], pslw);
    atomicAdd(&out[4], 4ull * St); atomicAdd(&out[5], ideal);
    atomicMax(&out[6], curmax); atomicMax(&out[7], pslmax); atomicMax(&out[8], (unsigned long long)St); atomicMax(&out[9], idealmax);
}
template <typename T> void dev_walk_stats(const Call<T> &c, const char *what, int ci, int co)
{
    const auto &S = c.L.slot[c.slot];
    if (c.L.ngroups != 1) return;
    unsigned long long *d = nullptr, h[10];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return;
    (void)hipMemsetAsync(d, 0, sizeof(h), c.s);
    const int nt = c.d.B * c.d.ntiles;
    hipLaunchKernelGGL(walk_stats_kernel, dim3((nt + 63) / 64), dim3(64), 0, c.s, S.qsegs, S.pairs, nt, d);
    (void)hipStreamSynchronize(c.s);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    const double r = (double)h[0];
    fprintf(stderr, "walk_stats %s<%d,%d> stride %d B=%d N=%d: records/point %.2f (true pairs %.2f) | lane use: shipped %.3f (true-pair %.3f), "
            "per-wave proportional %.3f, per-workgroup proportional %.3f, bound %.3f | steps of the slowest wave: shipped %llu, "
            "per-wave prop. %llu, per-wg prop. %llu, bound %llu | mean steps per wave: %.2f / %.2f / %.2f / %.2f\n",
            what, ci, co, c.st.step[0], c.d.B, c.d.N, r / ((double)c.d.B * c.d.N), (double)h[1] / ((double)c.d.B * c.d.N),
            r / (64.0 * h[2]), (double)h[1] / (64.0 * h[2]), r / (64.0 * h[3]), r / (64.0 * h[4]), r / (64.0 * h[5]),
            h[6], h[7], h[8], h[9], h[2] / (4.0 * nt), h[3] / (4.0 * nt), h[4] / (4.0 * nt), h[5] / (4.0 * nt));
}
#endif

template <typename T, int CI, int CO>
int launch_forward(const Call<T> &c, const T *input, const T *filter, T *output, const uint8_t *only_flagged = nullptr)
{
    const Dims &d = c.d;
    const Stencil<T> &st = c.st;
    const auto &S = c.L.slot[c.slot];
    if constexpr (sizeof(T) == 4 && CI >= 16 && CI % 4 == 0 && CO <= 16) {
        // transform + gather (conv3p_forward_taps.hpp): Z = X . W[f] for every point and tap, then 64 bytes per pair
        if (only_flagged == nullptr && c.tap_scratch_ok) {
            float *z = c.L.partials;
            const size_t rows = (size_t)d.B * d.N;
            const BlockMap bm = make_blockmap(d);
            const size_t glds = lds_common(st) + a16((size_t)st.ntap * kCntStride * 4) + 256 +
                                a16((size_t)kWavesPerBlock * 192 * 4) + a16((size_t)kWavesPerBlock * CO * 64 * 4);
#ifdef CONV3P_DEV_WALK_STATS
            dev_walk_stats(c, "tap_gather_kernel", CI, CO);
#endif
            Scope sc(K_FORWARD, c.s);
            hipLaunchKernelGGL((tap_transform_kernel<CI, CO>), dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, c.s, input,
                               filter, z, rows, st.ntap, c.ld.in);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(tap_gather_kernel<CO>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds);
            hipLaunchKernelGGL((tap_gather_kernel<CO>), dim3(grid_of(bm)), dim3(256), glds, c.s, c.L.pts, c.L.boxes, S.count,
                               S.pairs, S.segs, S.qsegs, z, st, d.N, d.ntiles, c.L.ngroups, bm, output, c.act ? 1 : 0,
                               st.window ? c.L.cmin : nullptr, S.tcount, c.ld, sched_of(S));
            return hip_ok();
        }
    }
    const size_t lds = lds_common(st) + (CI > 0 ? a16((size_t)st.ntap * fwd_wstr<T>(CI, CO) * sizeof(T)) : 0) +
                       a16((size_t)st.ntap * kCntStride * sizeof(T)) +
                       256 + a16((size_t)kWavesPerBlock * 192 * 4) +
                       (CI > 0 ? a16((size_t)kWavesPerBlock * CO * 64 * sizeof(T)) : 0);
    if (lds > kMaxLds) return CONV3P_ERR_UNSUPPORTED;
    const BlockMap bm = make_blockmap(d);
#ifdef CONV3P_DEV_WALK_STATS
    if (CI > 0) dev_walk_stats(c, "forward_kernel", CI, CO);
#endif
    Scope sc(K_FORWARD, c.s);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(forward_kernel<T, CI, CO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((forward_kernel<T, CI, CO>), dim3(grid_of(bm)), dim3(256), lds, c.s, c.L.pts, c.L.boxes,
                       S.count, S.pairs, S.segs, S.qsegs, input, filter, st, d.N, d.ntiles, c.L.ngroups, d.Cin, d.Cout,
                       bm, output, only_flagged, (CI > 0 && c.act) ? 1 : 0, st.window ? c.L.cmin : nullptr, S.tcount, c.ld,
                       sched_of(S));
    return hip_ok();
}

// Rows of the populated-rows G matrix (conv3p_backward_sparse.hpp) that fit LDS next to the kernel's other arrays with
// four (else three, else two) workgroups per CU; 0: use backward_kernel's dense G.
// LDS of backward_kernel's dense G for a register-path shape (what launch_backward asks for)
template <typename T> size_t dense_backward_lds(const Stencil<T> &st, int cin, int cout)
{
    const size_t nw = (size_t)st.ntap * cin * cout;
    size_t tail = a16(nw * sizeof(T)) + a16((size_t)64 * cin * sizeof(T)) + a16((size_t)kWavesPerBlock * 192 * 4);
    const size_t red = a16((size_t)kWavesPerBlock * cin * 64 * sizeof(T));
    if (tail < red) tail = red;
    return lds_common(st) + a16((size_t)st.ntap * cout * kCntStride * sizeof(T)) + a16(256 * sizeof(T)) + 256 + tail;
}
template <typename T> int sparse_cap(const Stencil<T> &st, int cin, int cout, size_t &lds)
{
    // (phase B holds the output channels in one 16-wide block; 33 .. 64 taps: the narrow layers' 64-bit tap sets, 65 .. 128:
    // 128-bit ones -- round 6: a 5 x 5 x 5 filter stays on the deterministic kernels)
    if (st.ntap > 128 || (st.ntap > 32 && cin >= 16) || cin < 1 || cout < 1 || cout > 16) return 0;
    // Undilated stencils populate a third of a centre's taps and more (adjacent cells: 9 of 27 on the ModelNet-shaped
    // clouds, 650-850 rows per tile): their tiles would take two rounds, each walking the pair lists again
    // (measured: 3 -> 9 stride 1 at the cfg2 size 63.8 us against 49.5 us dense).  Dilated ones: 210-450 rows.
    // (Unless the dense G does not fit LDS at all -- filters of many taps: then the rounds are the only register path.)
#ifndef CONV3P_DEV_SPARSE_UNDILATED   // developer A/B build: the populated-rows kernel for undilated narrow layers too
    if (cin < 16 && st.step[0] * st.step[1] * st.step[2] == 1 && dense_backward_lds<T>(st, cin, cout) <= kMaxLds) return 0;
#endif
    const size_t fixed = sparse_fixed_lds<T>(st.maxfull, st.ntap, cin, cout) + 64;
    const size_t budgets[3] = {40960, 54608, 81920};
    for (size_t bud : budgets) {
        if (cin >= 16 && bud < 81920) continue;   // wide rows: 2 waves per SIMD for the registers (dX rows of Cin values)
        if (bud <= fixed) continue;
        long long cap = (long long)((bud - fixed) / ((size_t)cout * sizeof(T)));
        if (cap > 64LL * st.ntap) cap = 64LL * st.ntap;
        if (cap < (long long)sparse_min_rows<T>(cin, cout)) continue;
        if (cap >= 256 || cap == 64LL * st.ntap) {
            lds = fixed + (size_t)cap * cout * sizeof(T);
            return (int)cap;
        }
    }
    return 0;
}

template <typename T, int CI, int CO>
int launch_backward(const Call<T> &c, const T *grad_out, const T *input, const T *filter, T *grad_input,
                    T *partials = nullptr, const uint8_t *only_flagged = nullptr, int gen_slots = 1)
{
    const Dims &d = c.d;
    const Stencil<T> &st = c.st;
    const auto &S = c.L.slot[c.slot];
    const size_t nw = (size_t)st.ntap * d.Cin * d.Cout;
    const uint32_t *regime = nullptr;   // non-null: the populated-rows kernel was launched for the short-lists case
#ifndef CONV3P_DEV_DENSE_BACKWARD   // developer A/B build: always the dense-G kernel
    if constexpr (CI > 0 && CO <= 16 && sizeof(T) == 4) {
        size_t slds = 0;
        // narrow layers: on the caller's hint (short pair lists); layers of >= 16 inputs: always -- their dense G
        // plus the transposed filter take 151 KiB of LDS (one workgroup per CU), the populated rows fit two
        // layers of >= 16 inputs: always (their dense G plus the transposed filter take 151 KiB of LDS, one workgroup per
        // CU; the populated rows fit two).  Narrow dilated layers: when the pair lists are short -- on the caller's hint
        // (CONV3P_CACHE_SPARSE_NEIGHBOURHOODS) this kernel alone; without it BOTH kernels are launched and the slot's
        // regime word (tile_sched_kernel, from the lists just built) lets exactly one of them run: the choice needs
        // neither the caller nor a host synchronisation, at the price of one empty launch (~5 us).
        // (a filter whose dense G does not fit LDS -- more than ~60 taps at 9 output channels -- has only this kernel: no
        // regime word, no hint can send it to the dense one)
        const bool dense_fits = dense_backward_lds<T>(st, CI, CO) <= kMaxLds;
        const int cap = only_flagged == nullptr && !(CI < 16 && c.dense_hint && dense_fits) ? sparse_cap<T>(st, CI, CO, slds) : 0;
        const bool by_regime = CI < 16 && !c.sparse_hint && dense_fits;
        if (cap > 0) {
            const BlockMap bm = make_blockmap(d);
            Scope sc(K_BACKWARD, c.s);
            auto go = [&](auto kern) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds);
                hipLaunchKernelGGL(kern, dim3(grid_of(bm)), dim3(256), slds, c.s, c.L.pts, c.L.boxes,
                                   S.count, S.pairs, S.segs, S.qsegs, S.qbm, S.qbm_hi, grad_out, input, filter, st, d.N, d.ntiles, c.L.ngroups,
                                   bm, grad_input, partials ? partials : c.L.partials, (c.act ? 1 : 0) | (c.accum ? 2 : 0), c.addend,
                                   st.window ? c.L.cmin : nullptr, c.ld, cap, sched_of(S),
                                   by_regime ? S.regime : static_cast<const uint32_t *>(nullptr));
            };
            if constexpr (CI < 16) {
                if (st.ntap > 64) go(backward_sparse_kernel<T, CI, CO, 2>);        // 128-bit tap sets
                else if (st.ntap > 32) go(backward_sparse_kernel<T, CI, CO, 1>);   // 64-bit tap sets
                else go(backward_sparse_kernel<T, CI, CO, 0>);
            } else {
                go(backward_sparse_kernel<T, CI, CO, 0>);
            }
            if (!by_regime) return hip_ok();
            regime = S.regime;
        }
    }
#endif
    // reduce buffer [4][CI][64] aliases { Wt | X tile | SoA }
    size_t tail = (CI > 0 ? a16(nw * sizeof(T)) + a16((size_t)64 * CI * sizeof(T)) : 0) + a16((size_t)kWavesPerBlock * 192 * 4);
    const size_t red = CI > 0 ? a16((size_t)kWavesPerBlock * CI * 64 * sizeof(T)) : 0;
    if (tail < red) tail = red;
    const size_t lds = lds_common(st) + (CI > 0 ? a16((size_t)st.ntap * CO * kCntStride * sizeof(T)) + a16(256 * sizeof(T)) : 0) + 256 + tail;
    if (lds > kMaxLds) return CONV3P_ERR_UNSUPPORTED;
    const BlockMap bm = make_blockmap(d);
    Scope sc(K_BACKWARD, c.s);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(backward_kernel<T, CI, CO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((backward_kernel<T, CI, CO>), dim3(grid_of(bm)), dim3(256), lds, c.s, c.L.pts, c.L.boxes,
                       S.count, S.pairs, S.segs, S.qsegs, grad_out, input, filter, st, d.N, d.ntiles, c.L.ngroups,
                       d.Cin, d.Cout, bm, grad_input, partials ? partials : c.L.partials, only_flagged,
                       ((CI > 0 && c.act) ? 1 : 0) | ((CI > 0 && c.accum) ? 2 : 0), c.addend, gen_slots,
                       st.window ? c.L.cmin : nullptr, c.ld, sched_of(S), regime);
    return hip_ok();
}

int zero_async(void *p, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return CONV3P_OK;
    Scope sc(K_MEMSET, s);
    return hipMemsetAsync(p, 0, bytes, s) == hipSuccess ? CONV3P_OK : CONV3P_ERR_LAUNCH;
}

// ----------------------------------------------------------------------------- deep-channel path
// Channel shapes served by the matrix-core kernels of conv3p_deep.hpp (fp32 only): every layer with up to 128
// channels on either side, padded to the instantiated sizes {32, 64, 128}, plus the 128 -> 256 layer of BASELINE
// config 5.  (Cin, Cout) below are the PADDED sizes.
#define CONV3P_DEEP_SHAPES(X) X(128, 256) X(256, 256) X(256, 128) X(32, 32) X(32, 64) X(32, 128) X(64, 32) X(64, 64) X(64, 128) X(128, 32) X(128, 64) X(128, 128)

inline int deep_pad(int c) { return c <= 32 ? 32 : c <= 64 ? 64 : c <= 128 ? 128 : c <= 256 ? 256 : 0; }
// padded (Cin, Cout) of a layer the deep path takes, or false
inline bool deep_class(int elem, int cin, int cout, int &cip, int &cop)
{
    if (elem != 4 || cin < 1 || cout < 1) return false;
    cip = deep_pad(cin);
    cop = deep_pad(cout);
    if (cip == 0 || cop == 0) return false;
    // (padded to the instantiated sizes; 129 .. 256 channels on either side: the 128 -> 256, 256 -> 256 and 256 -> 128 classes)
#define X(ci, co) if (cip == ci && cop == co) return true;
    CONV3P_DEEP_SHAPES(X)
#undef X
    return false;
}
inline bool deep_shape(int elem, int cin, int cout)
{
    int a, b;
    return deep_class(elem, cin, cout, a, b);
}

constexpr int kDwItems = 1024;        // target number of deep_dw_kernel work items (2 rounds at 2 per CU)
struct DeepScratch {   // carved from the per-call scratch region
    float *wt;             // zero-padded (and, for grad_input, transposed) filter [F][Kpad][Npad]
    uint2 *tap_meta;       // per pair slot, tap-major inside a tile: {neighbour, centre lane | population << 8}
    uint32_t *tap_off;     // [tiles][F+1]
    uint8_t *tile_flag;    // [tiles]
    unsigned long long *pop_mask;   // [tiles] bit f: the tile has records of backward tap f (deep_order -> deep_plan)
    uint4 *tap_split;      // [tiles][F] quarter points of every (tile, tap) run (deep_order -> deep_gemm stage 1)
    unsigned long long *tap_cmask;   // [tiles][F] centres with records of the backward tap: the rows of G_f' that exist
    uint32_t *sched;       // [8][sched_cap] launch order of the tiles per XCD (deep_sched_kernel)
    int sched_cap;
    uint2 *dsegs;          // [tiles] {first slot in tap_meta, records} of every tile (deep_order; all search groups together)
    uint32_t *meta_cursor; // [B] slot allocator of tap_meta for tiles searched in several groups (N > 8192)
    uint32_t *tap_total;   // [64] pairs per backward tap, then [1] number of work items  (deep_plan_kernel)
    uint2 *tap_rng;        // [64] partial slots of each tap
    uint4 *items;          // [kDwItems + 64] deep_dw_kernel work items
    float *partials;       // [kDwItems + 64][Cin*Cout] item partials, then [F*Cin*Cout] the generic kernel's share
    float *gbuf;           // [tiles][F][64][Coutpad]: G_f' of every (tile, backward tap), grad_input -> grad_filter pass
    size_t bytes;
};

// The record orders (deep_order_kernel) come FIRST and twice -- forward taps and backward taps -- at offsets that do not
// depend on the channel counts: a geometry prefetch (CONV3P_CACHE_PREPARE_DEEP_ORDERS) builds both ahead of the
// layer's calls, which then find them in place.  `which`: 0 = the forward's set, 1 = the backward's.
DeepScratch carve_deep(const Dims &d, size_t pair_slots, void *base, int which, size_t *order_bytes = nullptr, size_t *forward_bytes = nullptr)
{
    DeepScratch s{};
    size_t off = 0;
    char *p = static_cast<char *>(base);
    auto take = [&](size_t n) { char *r = p ? p + off : nullptr; off += up(n); return r; };
    const size_t nw = (size_t)d.ntap * d.Cin * d.Cout;
    const size_t cip = (size_t)deep_pad(d.Cin), cop = (size_t)deep_pad(d.Cout);
    s.sched_cap = ((d.B + 7) / 8) * d.ntiles;
    for (int w = 0; w < 2; ++w) {
        uint2 *tap_meta = reinterpret_cast<uint2 *>(take(pair_slots * 8));
        uint32_t *tap_off = reinterpret_cast<uint32_t *>(take((size_t)d.B * d.ntiles * (d.ntap + 1) * 4));
        uint8_t *tile_flag = reinterpret_cast<uint8_t *>(take((size_t)d.B * d.ntiles));
        uint4 *tap_split = reinterpret_cast<uint4 *>(take((size_t)d.B * d.ntiles * d.ntap * 16));
        uint32_t *sched = reinterpret_cast<uint32_t *>(take((size_t)8 * s.sched_cap * 4));
        uint2 *dsegs = reinterpret_cast<uint2 *>(take((size_t)d.B * d.ntiles * 8));
        uint32_t *meta_cursor = reinterpret_cast<uint32_t *>(take((size_t)d.B * 4));
        if (w == which) {
            s.tap_meta = tap_meta; s.tap_off = tap_off; s.tile_flag = tile_flag; s.tap_split = tap_split;
            s.sched = sched; s.dsegs = dsegs; s.meta_cursor = meta_cursor;
        }
    }
    // backward only
    s.pop_mask = reinterpret_cast<unsigned long long *>(take((size_t)d.B * d.ntiles * 8));
    s.tap_cmask = reinterpret_cast<unsigned long long *>(take((size_t)d.B * d.ntiles * d.ntap * 8));
    s.tap_total = reinterpret_cast<uint32_t *>(take(65 * 4));
    s.tap_rng = reinterpret_cast<uint2 *>(take(64 * 8));
    s.items = reinterpret_cast<uint4 *>(take((size_t)(kDwItems + 64) * 16));
    if (order_bytes) *order_bytes = off;
    s.wt = reinterpret_cast<float *>(take((size_t)d.ntap * cip * cop * 4));
    if (forward_bytes) *forward_bytes = off;   // everything below is the backward's (work-item partials, the G tiles)
    s.partials = reinterpret_cast<float *>(take((size_t)(kDwItems + 64) * cip * cop * 4 + nw * 4));
    s.gbuf = reinterpret_cast<float *>(take((size_t)d.B * d.ntiles * d.ntap * 64 * cop * 4));
    s.bytes = off;
    return s;
}

size_t deep_scratch_bytes(const Dims &d, size_t pair_slots) { return carve_deep(d, pair_slots, nullptr, 0).bytes; }
// what a FORWARD call of the matrix-core path touches: the record orders and the padded filter -- not the backward's G tiles
// (B x tiles x taps x 64 x Cout floats: 3.6 GB for the cfg5 shard) nor its partials
size_t deep_forward_bytes(const Dims &d, size_t pair_slots)
{
    size_t b = 0;
    (void)carve_deep(d, pair_slots, nullptr, 0, nullptr, &b);
    return b;
}
// the part of it the record orders take (the same for every channel shape)
size_t deep_order_bytes(const Dims &d, size_t pair_slots)
{
    size_t b = 0;
    (void)carve_deep(d, pair_slots, nullptr, 0, &b);
    return b;
}

void note_deep_order(const Call<float> &c, int which);

template <bool BWD> int launch_deep_order(const Call<float> &c, const DeepScratch &ds)
{
    const Dims &d = c.d;
    if (d.ntap > 64) return CONV3P_ERR_UNSUPPORTED;
    if (c.order_ok[BWD ? 1 : 0]) return CONV3P_OK;   // built for this geometry by an earlier call, scratch untouched since
    const auto &S = c.L.slot[c.slot];
    const int ng = c.L.ngroups;
    const size_t lds = (size_t)(2 + 4 * kOrderR + 64) * d.ntap * 4 + 256 + (size_t)(2 * 64 * ng + 1 + 8) * 4;
    if (lds > kMaxLds) return CONV3P_ERR_UNSUPPORTED;
    Scope sc(K_DEEP_ORDER, c.s);
    if (BWD) TRY(zero_async(ds.tap_total, 65 * 4, c.s));
    if (ng > 1) TRY(zero_async(ds.meta_cursor, (size_t)d.B * 4, c.s));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(deep_order_kernel<BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(deep_order_kernel<BWD>, dim3((unsigned)(d.B * d.ntiles)), dim3(256), lds, c.s, c.L.pts, S.count,
                       S.pairs, S.segs, d.N, d.ntiles, d.ntap, ng, S.qsegs, ds.dsegs, ds.meta_cursor, c.L.pairs_per_cloud,
                       ds.tap_meta, ds.tap_off, ds.tile_flag,
                       BWD ? ds.tap_total : nullptr, BWD ? ds.pop_mask : nullptr, ds.tap_split, BWD ? ds.tap_cmask : nullptr);
    hipLaunchKernelGGL(deep_sched_kernel, dim3(8), dim3(1024), 0, c.s, ds.dsegs, d.B, d.ntiles, ds.sched_cap, ds.sched);
    if (BWD)
        hipLaunchKernelGGL(deep_plan_kernel, dim3(1), dim3(1024), (size_t)(d.B * d.ntiles <= kPlanTiles ? d.B * d.ntiles : 0) * 8, c.s,
                           ds.tap_total, ds.pop_mask, d.ntap, d.B * d.ntiles, kDwItems,
                           ds.items, ds.tap_rng, ds.tap_total + 64);
    note_deep_order(c, BWD ? 1 : 0);
    return hip_ok();
}

template <int KD, int ND, bool BWD>
int launch_deep_gemm(const Call<float> &c, const float *src, const float *Bm, float *out, const DeepScratch &ds,
                     int kreal, int nreal, float *gbuf = nullptr, const float *xin = nullptr)
{
    const Dims &d = c.d;
    const auto &S = c.L.slot[c.slot];
    const size_t lds = a16((size_t)(KD < 256 ? 256 / KD : 1) * 64 * (KD + 4) * 4) + 68 * 4 + 256 + 4096 + 512;
    if (lds > kMaxLds) return CONV3P_ERR_UNSUPPORTED;
    if ((size_t)d.N * (size_t)kreal * 4 > 0xFFFFFFFFull) return CONV3P_ERR_UNSUPPORTED;   // (stage 1 addresses rows by 32-bit offsets)
    const BlockMap bm = make_blockmap(d);
    Scope sc(K_DEEP_GEMM, c.s);
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid_of(bm)), dim3(256), lds, c.s, c.L.pts, S.pairs, ds.dsegs, src, Bm, d.N, d.ntiles,
                           d.ntap, ds.sched, ds.sched_cap, out, ds.tap_meta, ds.tap_off, ds.tile_flag, kreal, nreal, gbuf, xin,
                           ds.tap_split, ds.tap_cmask);
    };
    // rows shorter than 4 floats (only possible in the 32-column class) take the variant with scalar row loads
    if constexpr (KD == 32) {
        if (kreal < 4) {
            go(deep_gemm_kernel<KD, ND, BWD, false>);
            return hip_ok();
        }
    }
    go(deep_gemm_kernel<KD, ND, BWD, true>);
    return hip_ok();
}

int launch_pad_filter(const Call<float> &c, const float *filter, int kp, int np, int transpose, float *wp)
{
    const size_t n = (size_t)c.d.ntap * kp * np;
    Scope sc(K_TRANSPOSE, c.s);
    hipLaunchKernelGGL(pad_filter_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0,
                       c.s, filter, c.d.ntap, c.d.Cin, c.d.Cout, kp, np, transpose, wp);
    return hip_ok();
}

// CI, CO: the padded instantiation; c.d.Cin / c.d.Cout: the layer's real channel counts
template <int CI, int CO>
int deep_forward(const Call<float> &c, const float *input, const float *filter, float *output)
{
    const Dims &d = c.d;
    const DeepScratch ds = carve_deep(d, (size_t)d.B * c.L.pairs_per_cloud, c.L.partials, 0);
    for (int w = 0; w < 2; ++w)
        if (c.order_ok[w]) note_deep_order(c, w);   // this path leaves both orders in place
    TRY(launch_deep_order<false>(c, ds));
    const float *Bm = filter;
    if (d.Cin != CI || d.Cout != CO) {
        TRY(launch_pad_filter(c, filter, CI, CO, 0, ds.wt));
        Bm = ds.wt;
    }
    TRY((launch_deep_gemm<CI, CO, false>(c, input, Bm, output, ds, d.Cin, d.Cout)));
    // tiles the deep kernels could not take (pair buffer overflow, non-finite rows): generic kernel, flagged tiles
    return launch_forward<float, 0, 0>(c, input, filter, output, ds.tile_flag);
}

template <int CI, int CO>
int deep_backward(const Call<float> &c, const float *grad_out, const float *input, const float *filter,
                  float *grad_input, float *grad_filter)
{
    const Dims &d = c.d;
    const size_t nw = (size_t)d.ntap * d.Cin * d.Cout;
    const DeepScratch ds = carve_deep(d, (size_t)d.B * c.L.pairs_per_cloud, c.L.partials, 1);
    for (int w = 0; w < 2; ++w)
        if (c.order_ok[w]) note_deep_order(c, w);   // this path leaves both orders in place
    TRY(launch_deep_order<true>(c, ds));
    TRY(launch_pad_filter(c, filter, CO, CI, 1, ds.wt));
    // dX = sum_f' G_f' . W[f']^T  (K = Cout, N = Cin)
    TRY((launch_deep_gemm<CO, CI, true>(c, grad_out, ds.wt, grad_input, ds, d.Cout, d.Cin, ds.gbuf, input)));
    {
        constexpr int NH = deep_dw_parts<CI, CO>();
        const size_t lds = (size_t)DEEP_DW_ROWS * (CI + 1) * 4 + (size_t)DEEP_DW_ROWS * (CO / NH + 1) * 4 + 256;
        if (lds > kMaxLds) return CONV3P_ERR_UNSUPPORTED;
        Scope sc(K_DEEP_DW, c.s);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(deep_dw_kernel<CI, CO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((deep_dw_kernel<CI, CO>), dim3(kDwItems + 64, NH), dim3(256), lds, c.s, c.L.pts, ds.tap_off,
                           ds.gbuf, input, d.N, d.ntiles, d.ntap, ds.tile_flag, ds.items, ds.tap_total + 64, ds.partials,
                           d.Cin, ds.tap_cmask);
    }
    TRY(hip_ok());
    // flagged tiles: generic kernel adds into the zeroed rows / into its own grad_filter-shaped buffer
    float *extra = ds.partials + (size_t)(kDwItems + 64) * CI * CO;
    TRY(zero_async(extra, nw * 4, c.s));
    TRY((launch_backward<float, 0, 0>(c, grad_out, input, filter, grad_input, extra, ds.tile_flag)));
    {
        Scope sc(K_REDUCE, c.s);
        hipLaunchKernelGGL(deep_reduce_kernel, dim3((unsigned)((d.Cin * d.Cout + 255) / 256), (unsigned)d.ntap), dim3(256), 0,
                           c.s, ds.partials, ds.tap_rng, extra, CI * CO, CO, d.Cin, d.Cout, grad_filter);
    }
    return hip_ok();
}

// ----------------------------------------------------------------------------- layers of more than 256 channels
// A block of kw <= 256 channels is packed into rows of 128 or 256 floats, zero-filled past kw: every block then IS one of
// the instantiated shapes {128, 256}^2 with full rows (the kernels' fast path), whatever the layer's channel counts.
inline int wide_pad(int n) { return n <= 128 ? 128 : 256; }
struct WideScratch {
    float *xp, *yp, *zp;   // [B N][256] packed rows: block input, block grad_out (backward), block result
    float *wp, *dwp;       // [ntap][256][256] packed filter block, its grad_filter block
};
inline WideScratch carve_wide(const Call<float> &c, bool forward_only = false)
{
    const Dims &d = c.d;
    Dims db = d;
    db.Cin = kWideBlk;
    db.Cout = kWideBlk;
    const size_t rows = (size_t)d.B * d.N;
    // (a forward-only workspace ends after the forward's part of the block scratch: the packs follow that; a cache sized
    // for the backward keeps ONE placement for both passes)
    const size_t slots = (size_t)d.B * c.L.pairs_per_cloud;
    char *p = reinterpret_cast<char *>(c.L.partials) + up(forward_only && !c.wide_scratch_ok ? deep_forward_bytes(db, slots) : deep_scratch_bytes(db, slots));
    WideScratch w{};
    w.xp = reinterpret_cast<float *>(p); p += up(rows * kWideBlk * 4);
    w.yp = reinterpret_cast<float *>(p); p += up(rows * kWideBlk * 4);
    w.zp = reinterpret_cast<float *>(p); p += up(rows * kWideBlk * 4);
    w.wp = reinterpret_cast<float *>(p); p += up((size_t)d.ntap * kWideBlk * kWideBlk * 4);
    w.dwp = reinterpret_cast<float *>(p);
    return w;
}
inline unsigned grid_1d(size_t n) { return (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }
// dst [rows][ldp] = src[:, 0 .. cols) zero-filled to ldp columns
inline int wide_pack_rows(const Call<float> &c, const float *src, int ld_s, float *dst, size_t rows, int cols, int ldp)
{
    if (cols < ldp) TRY(zero_async(dst, rows * (size_t)ldp * 4, c.s));
    hipLaunchKernelGGL(copy_cols_kernel<float>, dim3(grid_1d(rows * cols)), dim3(256), 0, c.s, src, dst, rows, cols, ld_s, ldp);
    return CONV3P_OK;
}
// wp [ntap][kwp][cwp] = filter[:, k0 .. k0 + kw, c0 .. c0 + cw) zero-filled
inline int wide_pack_filter(const Call<float> &c, const float *filter, int k0, int kw, int c0, int cw, int kwp, int cwp, float *wp)
{
    const Dims &d = c.d;
    if (kw < kwp || cw < cwp) TRY(zero_async(wp, (size_t)d.ntap * kwp * cwp * 4, c.s));
    hipLaunchKernelGGL(copy_block_kernel<float>, dim3(grid_1d((size_t)d.ntap * kw * cw)), dim3(256), 0, c.s,
                       filter + (size_t)k0 * d.Cout + c0, wp, d.ntap, kw, cw, (size_t)d.Cin * d.Cout, d.Cout, (size_t)kwp * cwp, cwp);
    return CONV3P_OK;
}

// out[:, c0 .. c0 + cw) = sum over the input-channel blocks of conv(input[:, k0 .. k0 + kw), filter[:, k-block, c-block]):
// every block runs on the matrix-core kernels from packed copies (the same geometry and record order for all of them),
// the blocks' results are added in ascending k0 -- deterministic.
int wide_forward(const Call<float> &c, const float *input, const float *filter, float *output)
{
    const Dims &d = c.d;
    const size_t rows = (size_t)d.B * d.N;
    const WideScratch w = carve_wide(c, /*forward_only=*/true);
    bool have_order = c.order_ok[0];
    for (int c0 = 0; c0 < d.Cout; c0 += kWideBlk) {
        const int cw = d.Cout - c0 < kWideBlk ? d.Cout - c0 : kWideBlk, cwp = wide_pad(cw);
        for (int k0 = 0; k0 < d.Cin; k0 += kWideBlk) {
            const int kw = d.Cin - k0 < kWideBlk ? d.Cin - k0 : kWideBlk, kwp = wide_pad(kw);
            TRY(wide_pack_rows(c, input + k0, d.Cin, w.xp, rows, kw, kwp));
            TRY(wide_pack_filter(c, filter, k0, kw, c0, cw, kwp, cwp, w.wp));
            Call<float> cp = c;
            cp.d.Cin = kwp;
            cp.d.Cout = cwp;
            cp.act = false;
            cp.order_ok[0] = have_order;
            set_ld(cp, nullptr, kwp, cwp);
            int rc = CONV3P_ERR_UNSUPPORTED;
#define X(ci, co) if (kwp == ci && cwp == co) rc = deep_forward<ci, co>(cp, w.xp, w.wp, w.zp);
            CONV3P_DEEP_SHAPES(X)
#undef X
            if (rc != CONV3P_OK) return rc;
            have_order = true;
            if (k0 == 0)
                hipLaunchKernelGGL(copy_cols_kernel<float>, dim3(grid_1d(rows * cw)), dim3(256), 0, c.s, w.zp, output + c0, rows, cw,
                                   cwp, d.Cout);
            else
                hipLaunchKernelGGL(add_cols_kernel<float>, dim3(grid_1d(rows * cw)), dim3(256), 0, c.s, w.zp, output + c0, rows, cw,
                                   cwp, d.Cout);
        }
    }
    return hip_ok();
}

// grad_input[:, k-block] = sum over the output-channel blocks (ascending c0) of the block's grad_input; every
// (k-block, c-block) pair gives its own block of grad_filter.
int wide_backward(const Call<float> &c, const float *grad_out, const float *input, const float *filter, float *grad_input,
                  float *grad_filter)
{
    const Dims &d = c.d;
    const size_t rows = (size_t)d.B * d.N;
    const WideScratch w = carve_wide(c);
    bool have_order = c.order_ok[1];
    for (int k0 = 0; k0 < d.Cin; k0 += kWideBlk) {
        const int kw = d.Cin - k0 < kWideBlk ? d.Cin - k0 : kWideBlk, kwp = wide_pad(kw);
        TRY(wide_pack_rows(c, input + k0, d.Cin, w.xp, rows, kw, kwp));
        for (int c0 = 0; c0 < d.Cout; c0 += kWideBlk) {
            const int cw = d.Cout - c0 < kWideBlk ? d.Cout - c0 : kWideBlk, cwp = wide_pad(cw);
            TRY(wide_pack_rows(c, grad_out + c0, d.Cout, w.yp, rows, cw, cwp));
            TRY(wide_pack_filter(c, filter, k0, kw, c0, cw, kwp, cwp, w.wp));
            Call<float> cp = c;
            cp.d.Cin = kwp;
            cp.d.Cout = cwp;
            cp.act = false;
            cp.accum = false;
            cp.addend = nullptr;
            cp.order_ok[1] = have_order;
            set_ld(cp, nullptr, kwp, cwp);
            int rc = CONV3P_ERR_UNSUPPORTED;
#define X(ci, co) if (kwp == ci && cwp == co) rc = deep_backward<ci, co>(cp, w.yp, w.xp, w.wp, w.zp, w.dwp);
            CONV3P_DEEP_SHAPES(X)
#undef X
            if (rc != CONV3P_OK) return rc;
            have_order = true;
            if (c0 == 0)
                hipLaunchKernelGGL(copy_cols_kernel<float>, dim3(grid_1d(rows * kw)), dim3(256), 0, c.s, w.zp, grad_input + k0, rows,
                                   kw, kwp, d.Cin);
            else
                hipLaunchKernelGGL(add_cols_kernel<float>, dim3(grid_1d(rows * kw)), dim3(256), 0, c.s, w.zp, grad_input + k0, rows,
                                   kw, kwp, d.Cin);
            hipLaunchKernelGGL(copy_block_kernel<float>, dim3(grid_1d((size_t)d.ntap * kw * cw)), dim3(256), 0, c.s, w.dwp,
                               grad_filter + (size_t)k0 * d.Cout + c0, d.ntap, kw, cw, (size_t)kwp * cwp, cwp,
                               (size_t)d.Cin * d.Cout, d.Cout);
        }
    }
    return hip_ok();
}

// ----------------------------------------------------------------------------- fp64 layers outside the register-path shapes
struct F64Scratch {
    double *parts;          // the block kernel's grad_filter partials
    double *xp, *yp, *zp;   // packed rows: input block [B N][16], grad_out block / forward result [B N][8], grad_input block [B N][16]
    double *wp, *dwp;       // [ntap][16][8] filter block, its grad_filter block
};
inline F64Scratch carve_f64(const Call<double> &c)
{
    const Dims &d = c.d;
    const size_t rows = (size_t)d.B * d.N, nwb = (size_t)d.ntap * kF64Ki * kF64Co;
    char *p = reinterpret_cast<char *>(c.L.partials);
    F64Scratch w{};
    w.parts = reinterpret_cast<double *>(p); p += up((size_t)grid_of(make_blockmap(d)) * nwb * 8);
    w.xp = reinterpret_cast<double *>(p); p += up(rows * kF64Row * 8);
    w.yp = reinterpret_cast<double *>(p); p += up(rows * kF64Row * 8);
    w.zp = reinterpret_cast<double *>(p); p += up(rows * kF64Row * 8);
    w.wp = reinterpret_cast<double *>(p); p += up(nwb * 8);
    w.dwp = reinterpret_cast<double *>(p);
    return w;
}
inline int f64_pack_rows(const Call<double> &c, const double *src, int ld_s, double *dst, size_t rows, int cols, int ldp)
{
    if (cols < ldp) TRY(zero_async(dst, rows * (size_t)ldp * 8, c.s));
    hipLaunchKernelGGL(copy_cols_kernel<double>, dim3(grid_1d(rows * cols)), dim3(256), 0, c.s, src, dst, rows, cols, ld_s, ldp);
    return CONV3P_OK;
}
inline int f64_pack_filter(const Call<double> &c, const double *filter, int k0, int kw, int c0, int cw, double *wp, int bki, int bco)
{
    const Dims &d = c.d;
    if (kw < bki || cw < bco) TRY(zero_async(wp, (size_t)d.ntap * bki * bco * 8, c.s));
    hipLaunchKernelGGL(copy_block_kernel<double>, dim3(grid_1d((size_t)d.ntap * kw * cw)), dim3(256), 0, c.s,
                       filter + (size_t)k0 * d.Cout + c0, wp, d.ntap, kw, cw, (size_t)d.Cin * d.Cout, d.Cout,
                       (size_t)bki * bco, bco);
    return CONV3P_OK;
}
inline Call<double> f64_block_call(const Call<double> &c, int bki, int bco)
{
    Call<double> cp = c;
    cp.d.Cin = bki;
    cp.d.Cout = bco;
    cp.act = false;
    cp.accum = false;
    cp.addend = nullptr;
    set_ld(cp, nullptr, bki, bco);
    return cp;
}

// out[:, c-block] = sum over the input-channel blocks (ascending) of the block kernel's result: deterministic
int f64_blocked_forward(const Call<double> &c, const double *input, const double *filter, double *output)
{
    const Dims &d = c.d;
    const size_t rows = (size_t)d.B * d.N;
    const F64Scratch w = carve_f64(c);
    const Call<double> cp = f64_block_call(c, kF64FwdKi, kF64FwdCo);
    for (int k0 = 0; k0 < d.Cin; k0 += kF64FwdKi) {
        const int kw = d.Cin - k0 < kF64FwdKi ? d.Cin - k0 : kF64FwdKi;
        TRY(f64_pack_rows(c, input + k0, d.Cin, w.xp, rows, kw, kF64FwdKi));
        for (int c0 = 0; c0 < d.Cout; c0 += kF64FwdCo) {
            const int cw = d.Cout - c0 < kF64FwdCo ? d.Cout - c0 : kF64FwdCo;
            TRY(f64_pack_filter(c, filter, k0, kw, c0, cw, w.wp, kF64FwdKi, kF64FwdCo));
            TRY((launch_forward<double, kF64FwdKi, kF64FwdCo>(cp, w.xp, w.wp, w.yp)));
            if (k0 == 0)
                hipLaunchKernelGGL(copy_cols_kernel<double>, dim3(grid_1d(rows * cw)), dim3(256), 0, c.s, w.yp, output + c0, rows, cw,
                                   kF64FwdCo, d.Cout);
            else
                hipLaunchKernelGGL(add_cols_kernel<double>, dim3(grid_1d(rows * cw)), dim3(256), 0, c.s, w.yp, output + c0, rows, cw,
                                   kF64FwdCo, d.Cout);
        }
    }
    return hip_ok();
}

// grad_input[:, k-block] = sum over the output-channel blocks (ascending) of the block's grad_input; every block pair
// gives its own block of grad_filter (per-workgroup partials, reduced in fixed order)
int f64_blocked_backward(const Call<double> &c, const double *grad_out, const double *input, const double *filter,
                         double *grad_input, double *grad_filter)
{
    const Dims &d = c.d;
    const size_t rows = (size_t)d.B * d.N, nwb = (size_t)d.ntap * kF64BwdKi * kF64BwdCo;
    const int nslots = (int)grid_of(make_blockmap(d));
    const F64Scratch w = carve_f64(c);
    const Call<double> cp = f64_block_call(c, kF64BwdKi, kF64BwdCo);
    for (int k0 = 0; k0 < d.Cin; k0 += kF64BwdKi) {
        const int kw = d.Cin - k0 < kF64BwdKi ? d.Cin - k0 : kF64BwdKi;
        TRY(f64_pack_rows(c, input + k0, d.Cin, w.xp, rows, kw, kF64BwdKi));
        for (int c0 = 0; c0 < d.Cout; c0 += kF64BwdCo) {
            const int cw = d.Cout - c0 < kF64BwdCo ? d.Cout - c0 : kF64BwdCo;
            TRY(f64_pack_rows(c, grad_out + c0, d.Cout, w.yp, rows, cw, kF64BwdCo));
            TRY(f64_pack_filter(c, filter, k0, kw, c0, cw, w.wp, kF64BwdKi, kF64BwdCo));
            TRY((launch_backward<double, kF64BwdKi, kF64BwdCo>(cp, w.yp, w.xp, w.wp, w.zp, w.parts)));
            {
                Scope sc(K_REDUCE, c.s);
                hipLaunchKernelGGL(reduce_partials_kernel<double>, dim3((unsigned)((nwb + kReduceW - 1) / kReduceW)), dim3(1024), 0, c.s, w.parts,
                                   nslots, nwb, w.dwp);
            }
            if (c0 == 0)
                hipLaunchKernelGGL(copy_cols_kernel<double>, dim3(grid_1d(rows * kw)), dim3(256), 0, c.s, w.zp, grad_input + k0, rows,
                                   kw, kF64BwdKi, d.Cin);
            else
                hipLaunchKernelGGL(add_cols_kernel<double>, dim3(grid_1d(rows * kw)), dim3(256), 0, c.s, w.zp, grad_input + k0, rows,
                                   kw, kF64BwdKi, d.Cin);
            hipLaunchKernelGGL(copy_block_kernel<double>, dim3(grid_1d((size_t)d.ntap * kw * cw)), dim3(256), 0, c.s, w.dwp,
                               grad_filter + (size_t)k0 * d.Cout + c0, d.ntap, kw, cw, (size_t)kF64BwdKi * kF64BwdCo, kF64BwdCo,
                               (size_t)d.Cin * d.Cout, d.Cout);
        }
    }
    return hip_ok();
}

int buf_check(const void *p, size_t have, size_t need)
{
    if (need == 0) return CONV3P_OK;
    if (!p || (reinterpret_cast<uintptr_t>(p) % kAlign) != 0 || have < need) return CONV3P_ERR_WORKSPACE;
    return CONV3P_OK;
}

// ----------------------------------------------------------------------------- cache (host side)
// What the host remembers about a cache buffer: only which stencil tag lives in which slot
// (LRU) and a call counter.  Validity itself is decided on the device (CacheCtl).
struct CacheHost {
    int B = -1, N = -1, elem = 0, ntap_max = 0, nslots = 0, ppp = 0;
    std::vector<unsigned long long> tags;
    std::vector<uint64_t> stamp;
    std::vector<uint64_t> built_gen;   // generation in which the slot's search was last enqueued
    struct SlotDesc { int fz = 0, fy = 0, fx = 0; int32_t stride[3] = {0, 0, 0}; double voxel = 0.0; };
    std::vector<SlotDesc> desc;        // what stencil the slot's tag stands for (un-hinted calls rebuild them all together)
    uint64_t clock = 0;
    uint64_t gen = 0;                  // bumped by every call that does not carry CONV3P_CACHE_POINTS_UNCHANGED
    uint32_t epoch = 0;
    // record orders of the matrix-core path left in the scratch region: for which slot, built in which generation
    // (0: none).  Any call that uses the scratch for something else clears them.
    int order_slot[2] = {-1, -1};
    uint64_t order_gen[2] = {0, 0};
    // conv3p_stack_prefetch_*: geometry of `pending_points` enqueued on another stream; ready[l] fires when layer
    // l's lists are complete
    const void *pending_points = nullptr;
    int pending_layers = 0;
    std::vector<hipEvent_t> ready;
    // fused stack launches (conv3p_stack_fused.hpp): the per-cloud arrival counters -- a small device allocation of the
    // library's own (the caller's cache bytes may be garbage; a counter must not be), [kind][clouds][kSyncLineWords]: a
    // cloud's line = {counter, placement word, error bits} -- and the value every counter of a kind holds between launches
    uint32_t *sync = nullptr;
    int sync_clouds = 0;
    uint32_t sync_base[2] = {0u, 0u};
    uint32_t fused_launches[2] = {0u, 0u};
    uint32_t *host_err = nullptr;      // one word of host-mapped memory the fused kernels set on an error (looked at before every launch)
};
std::mutex g_cache_mu;
std::map<void *, CacheHost> g_caches;

void note_deep_order(const Call<float> &c, int which)
{
    if (c.cache_key == nullptr) return;   // per-call workspace: nothing outlives the call
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_caches.find(c.cache_key);
    if (it == g_caches.end()) return;
    it->second.order_slot[which] = c.slot;
    it->second.order_gen[which] = c.gen;
}

// Describes where a call's state lives: a persistent cache or per-call scratch.
struct Where {
    void *buf;
    size_t bytes;
    bool persistent;
    int nslots, ntap_max, ppp;
    size_t scratch_cap;   // persistent: bytes of scratch the layout was sized with
    int flags;            // CONV3P_CACHE_* of this call
};

template <typename T>
int begin_call(Call<T> &c, const Dims &d, const int32_t *stride, T voxel, size_t scratch, const Where &wh,
               hipStream_t s)
{
    c.d = d;
    c.st = make_stencil<T>(d, stride, voxel);
    c.s = s;
    c.sparse_hint = wh.persistent && (wh.flags & CONV3P_CACHE_SPARSE_NEIGHBOURHOODS) != 0;
    c.dense_hint = wh.persistent && !c.sparse_hint && (wh.flags & CONV3P_CACHE_DENSE_NEIGHBOURHOODS) != 0;
    const int ntap_max = wh.persistent ? wh.ntap_max : d.ntap;
    if (d.ntap > ntap_max) return CONV3P_ERR_WORKSPACE;
    if (wh.persistent && scratch > wh.scratch_cap) return CONV3P_ERR_WORKSPACE;
    c.L = carve<T>(d.B, d.N, d.ntiles, ntap_max, wh.nslots, wh.ppp, wh.persistent ? wh.scratch_cap : scratch, wh.buf);
    TRY(buf_check(wh.buf, wh.bytes, c.L.bytes));
    {
        const size_t have = wh.persistent ? wh.scratch_cap : scratch;
        c.deep_scratch_ok = deep_shape((int)sizeof(T), d.Cin, d.Cout) &&
                            have >= deep_scratch_bytes(d, (size_t)d.B * c.L.pairs_per_cloud);
        c.tap_scratch_ok = tap_forward_shape((int)sizeof(T), d.Cin, d.Cout) && have >= tap_forward_bytes(d);
        c.wide_scratch_ok = wide_shape((int)sizeof(T), d.Cin, d.Cout) && have >= wide_scratch_bytes(d, (size_t)d.B * c.L.pairs_per_cloud);
        c.deep_fwd_ok = deep_shape((int)sizeof(T), d.Cin, d.Cout) && have >= deep_forward_bytes(d, (size_t)d.B * c.L.pairs_per_cloud);
        c.wide_fwd_ok = wide_shape((int)sizeof(T), d.Cin, d.Cout) && have >= wide_scratch_bytes(d, (size_t)d.B * c.L.pairs_per_cloud, true);
        c.f64_scratch_ok = f64_blocked_shape((int)sizeof(T), d.Cin, d.Cout) && have >= f64_blocked_bytes(d);
    }
    const unsigned long long tag = stencil_tag(d, stride, (double)voxel, (int)sizeof(T));
    if (!wh.persistent) {
        c.slot = 0;
        c.cc = make_ctl(c.L, 0, tag, 1u, /*force=*/1);
        return CONV3P_OK;
    }
    std::lock_guard<std::mutex> lk(g_cache_mu);
    CacheHost &h = g_caches[wh.buf];
    if (h.B != d.B || h.N != d.N || h.elem != (int)sizeof(T) || h.ntap_max != ntap_max || h.nslots != wh.nslots ||
        h.ppp != wh.ppp) {
        std::vector<hipEvent_t> keep;
        keep.swap(h.ready);            // the stack-level entry points' events outlive a re-shape of the cache
        uint32_t *const ksync = h.sync;   // ... and so do the fused launches' counters
        uint32_t *const kherr = h.host_err;
        const int kclouds = h.sync_clouds;
        const uint32_t kb0 = h.sync_base[0], kb1 = h.sync_base[1], kf0 = h.fused_launches[0], kf1 = h.fused_launches[1];
        h = CacheHost();
        h.ready.swap(keep);
        h.sync = ksync; h.sync_clouds = kclouds; h.sync_base[0] = kb0; h.sync_base[1] = kb1;
        h.fused_launches[0] = kf0; h.fused_launches[1] = kf1; h.host_err = kherr;
        h.B = d.B; h.N = d.N; h.elem = (int)sizeof(T); h.ntap_max = ntap_max; h.nslots = wh.nslots; h.ppp = wh.ppp;
        h.tags.assign(wh.nslots, 0ull);
        h.stamp.assign(wh.nslots, 0ull);
        h.built_gen.assign(wh.nslots, 0ull);
        h.desc.assign(wh.nslots, CacheHost::SlotDesc());
    }
    const bool hinted = (wh.flags & CONV3P_CACHE_POINTS_UNCHANGED) != 0 && h.gen != 0;
    if (!hinted) h.gen += 1;
    int slot = -1;
    for (int i = 0; i < h.nslots; ++i)
        if (h.tags[i] == tag) slot = i;
    if (slot < 0) {   // least recently used slot; the device-side tag check forces its rebuild
        slot = 0;
        for (int i = 1; i < h.nslots; ++i)
            if (h.stamp[i] < h.stamp[slot]) slot = i;
        h.tags[slot] = tag;
        h.built_gen[slot] = 0;
    }
    {
        CacheHost::SlotDesc &sd = h.desc[slot];
        sd.fz = d.fz; sd.fy = d.fy; sd.fx = d.fx;
        sd.stride[0] = stride[0]; sd.stride[1] = stride[1]; sd.stride[2] = stride[2];
        sd.voxel = (double)voxel;
    }
    c.skip_prep = hinted;
    c.evicted_hinted = hinted && h.built_gen[slot] == 0 && c.L.slot[slot].cursor != nullptr;
    c.skip_search = hinted && h.built_gen[slot] == h.gen;
    h.built_gen[slot] = h.gen;
    c.cache_key = wh.buf;
    c.gen = h.gen;
    for (int w = 0; w < 2; ++w)
        c.order_ok[w] = c.skip_search && c.deep_scratch_ok && h.order_slot[w] == slot && h.order_gen[w] == h.gen && h.gen != 0;
    // a call with channels may use the scratch region for anything (partials, Z, ...): the orders count as gone unless
    // the matrix-core path, which leaves them in place, says otherwise (deep_forward / deep_backward re-note them)
    if (d.Cin > 0) h.order_gen[0] = h.order_gen[1] = 0;
    h.stamp[slot] = ++h.clock;
    h.epoch += 1;
    if (h.epoch == 0) h.epoch = 1;
    c.slot = slot;
    c.cc = make_ctl(c.L, slot, tag, h.epoch, /*force=*/0);
    return CONV3P_OK;
}

// An un-hinted call on a persistent cache re-validates the points; if they changed, EVERY stencil the cache holds is
// stale, and the caller (a framework executing a model op by op: 4 strides forward, the same 4 backward) is about to ask
// for each of them in turn.  The call therefore takes the cache's other stencils along as jobs of its own search launch
// (one launch for all strides: 115 us instead of 4 x 42 + launch overheads on cfg2); for clouds whose lists are current
// the extra jobs' workgroups exit at once, as the requested one's do.  Host bookkeeping only: what is rebuilt is still
// decided on the device.
template <typename T> int add_companions(const Call<T> &c, FusedJobs<T> &fj, SchedJobs &sj, int n)
{
    if (c.cache_key == nullptr || c.skip_prep) return n;   // per-call workspace, or the caller vouches for the points
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_caches.find(c.cache_key);
    if (it == g_caches.end()) return n;
    CacheHost &h = it->second;
    for (int sl = 0; sl < h.nslots && n < kFusedMaxJobs; ++sl) {
        if (sl == c.slot || h.tags[sl] == 0ull) continue;
        const CacheHost::SlotDesc &sd = h.desc[sl];
        if (sd.voxel != (double)c.st.voxel) continue;                 // the window tables are built for ONE voxel size
        Dims d2 = c.d;
        d2.fz = sd.fz; d2.fy = sd.fy; d2.fx = sd.fx;
        d2.ntap = sd.fz * sd.fy * sd.fx;
        Call<T> c2;
        c2.d = d2;
        c2.st = make_stencil<T>(d2, sd.stride, c.st.voxel);
        c2.L = c.L;
        c2.slot = sl;
        c2.s = c.s;
        c2.cc = make_ctl(c.L, sl, h.tags[sl], c.cc.epoch, /*force=*/0);
        if (!fused_ok(c2)) continue;
        // the launch's LDS request and the number M of hit masks a wave keeps follow the LARGEST filter among its jobs: a
        // companion must not cost the requested stencil its masks, let alone push the launch past the LDS limit
        if (fused_mask_depth((int)sizeof(T), c.d.ntiles, std::max(c.st.ntap, c2.st.ntap), std::max(c.st.maxfull, c2.st.maxfull)) <
            fused_mask_depth((int)sizeof(T), c.d.ntiles, c.st.ntap, c.st.maxfull))
            continue;
        fj.job[n] = make_fused_job(c2);
        sj.job[n] = make_sched_job(c, sl, true);
        ++n;
    }
    return n;
}
// ... and, once the launch that carries them has been issued without error: a hinted call for one of them later in
// this generation skips its search
template <typename T> void note_companions_built(const Call<T> &c, const FusedJobs<T> &fj, int n)
{
    if (c.cache_key == nullptr || n <= 1) return;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_caches.find(c.cache_key);
    if (it == g_caches.end()) return;
    CacheHost &h = it->second;
    for (int k = 1; k < n; ++k)
        for (int sl = 0; sl < h.nslots; ++sl)
            if (sl != c.slot && h.tags[sl] == fj.job[k].cc.tag) h.built_gen[sl] = h.gen;
}

// Stack-level backward: a layer's grad_filter partials stay in the caller's region and are reduced later, together
// with the other layers', by one reduce_multi_kernel launch.
template <typename T> struct DeferredReduce {
    T *region = nullptr;      // in: where this layer's partials go (>= reduce_region_bytes)
    ReduceJob<T> job{};       // out: what to reduce (nslots == 0: the layer reduced its grad_filter itself)
};

template <typename T> int selu_impl(const T *x, T *y, size_t n, void *stream);
// rows x cols values; lds = {ld_y, ld_dy, ld_b, ld_dx} or nullptr for dense operands
template <typename T>
int selu_grad_impl(const T *y, const T *dy, const T *dy_b, T *dx, size_t rows, int cols, const int *lds, void *stream);

template <typename T>
int forward_impl(const T *points, const T *input, const T *filter, const int32_t *stride, T voxel, int B,
                 int N, int Cin, int Cout, int fz, int fy, int fx, T *output, const Where &wh, void *stream,
                 bool act = false, const RowLd *ldp = nullptr)
{
    Dims d{B, N, Cin, Cout, fz, fy, fx, 0, 0};
    TRY(check(d, stride, (double)voxel, true));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t out_elems = (size_t)B * N * Cout;
    if (out_elems == 0) return CONV3P_OK;
    if (!points || !output || (Cin > 0 && (!input || !filter))) return CONV3P_ERR_INVALID_ARGUMENT;
    if (Cin == 0) return zero_async(output, out_elems * sizeof(T), s);   // empty contraction; selu(0) == 0
    Call<T> c;
    c.act = act;
    set_ld(c, ldp, Cin, Cout);
    if (c.strided && !small_shape((int)sizeof(T), Cin, Cout)) return CONV3P_ERR_UNSUPPORTED;   // dense tensors only
    TRY(begin_call<T>(c, d, stride, voxel, forward_scratch_bytes(d, (int)sizeof(T), wh.ppp, wh.persistent), wh, s));
    TRY(run_prep<T>(points, c));
    TRY(run_cloud_min<T>(points, c));
    TRY(run_search<T>(c, c.L.slot[c.slot].count, true));
#define X(ci, co)                                                                                    \
    if (Cin == ci && Cout == co && small_shape((int)sizeof(T), Cin, Cout)) {                         \
        int rc = launch_forward<T, ci, co>(c, input, filter, output);                                \
        if (rc != CONV3P_ERR_UNSUPPORTED) return rc;                                                 \
    }
    CONV3P_SMALL_SHAPES(X)
#undef X
    if (c.strided) return CONV3P_ERR_UNSUPPORTED;   // (a register-path shape whose LDS did not fit)
    if constexpr (sizeof(T) == 4) {
        int cip = 0, cop = 0;
        if (c.deep_fwd_ok && deep_class(4, Cin, Cout, cip, cop)) {
#define X(ci, co)                                                                                    \
    if (cip == ci && cop == co) {                                                                    \
        int rc = deep_forward<ci, co>(c, input, filter, output);                                     \
        if (rc != CONV3P_ERR_UNSUPPORTED) return rc != CONV3P_OK || !act ? rc : selu_impl<T>(output, output, out_elems, stream); \
    }
            CONV3P_DEEP_SHAPES(X)
#undef X
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (c.wide_fwd_ok) {   // more than 256 channels on a side: blocks of <= 256 x 256 on the matrix-core kernels
            const int rc = wide_forward(c, input, filter, output);
            if (rc != CONV3P_ERR_UNSUPPORTED) return rc != CONV3P_OK || !act ? rc : selu_impl<T>(output, output, out_elems, stream);
        }
    }
    if constexpr (sizeof(T) == 8) {
        if (c.f64_scratch_ok && !c.strided) {   // fp64 outside the register-path shapes: 16 x 8 channel blocks on <double, 16, 8>
            const int rc = f64_blocked_forward(c, input, filter, output);
            if (rc != CONV3P_ERR_UNSUPPORTED) return rc != CONV3P_OK || !act ? rc : selu_impl<T>(output, output, out_elems, stream);
        }
    }
    TRY(zero_async(output, out_elems * sizeof(T), s));                   // .cpp:451
    TRY((launch_forward<T, 0, 0>(c, input, filter, output)));
    return act ? selu_impl<T>(output, output, out_elems, stream) : CONV3P_OK;   // paths without a fused epilogue
}

// geometry only: what a later forward / backward with the same points + stencil will find ready
template <typename T>
int prepare_impl(const T *points, const int32_t *stride, T voxel, int B, int N, int fz, int fy, int fx,
                 const Where &wh, void *stream)
{
    Dims d{B, N, 0, 0, fz, fy, fx, 0, 0};
    TRY(check(d, stride, (double)voxel, false));
    if ((size_t)B * N == 0) return CONV3P_OK;
    if (!points) return CONV3P_ERR_INVALID_ARGUMENT;
    Call<T> c;
    TRY(begin_call<T>(c, d, stride, voxel, 0, wh, static_cast<hipStream_t>(stream)));
    TRY(run_prep<T>(points, c));
    TRY(run_cloud_min<T>(points, c));
    TRY(run_search<T>(c, c.L.slot[c.slot].count, true));
    if constexpr (sizeof(T) == 4) {
        // CONV3P_CACHE_PREPARE_DEEP_ORDERS: also the matrix-core path's two record orders (forward and backward taps),
        // so that the layer's calls on these points find them in the scratch region
        if (wh.persistent && (wh.flags & CONV3P_CACHE_PREPARE_DEEP_ORDERS) != 0) {
            const size_t pair_slots = (size_t)d.B * c.L.pairs_per_cloud;
            if (wh.scratch_cap < deep_order_bytes(c.d, pair_slots)) return CONV3P_ERR_WORKSPACE;
            c.order_ok[0] = c.order_ok[1] = false;
            TRY(launch_deep_order<false>(c, carve_deep(c.d, pair_slots, c.L.partials, 0)));
            TRY(launch_deep_order<true>(c, carve_deep(c.d, pair_slots, c.L.partials, 1)));
        }
    }
    return CONV3P_OK;
}

// geometry of K stencils (same filter extents, K strides) over the same points: one prep, ONE search launch and
// nothing else
template <typename T>
int prepare_multi_impl(const T *points, const int32_t *strides, int K, T voxel, int B, int N, int fz, int fy,
                       int fx, Where wh, void *stream)
{
    if (K <= 0 || K > kMaxJobs || !strides) return CONV3P_ERR_INVALID_ARGUMENT;
    Dims d{B, N, 0, 0, fz, fy, fx, 0, 0};
    for (int k = 0; k < K; ++k) TRY(check(d, strides + 3 * k, (double)voxel, false));
    if ((size_t)B * N == 0) return CONV3P_OK;
    if (!points) return CONV3P_ERR_INVALID_ARGUMENT;
    if (K > wh.nslots) return CONV3P_ERR_WORKSPACE;     // the stencils would evict each other
    hipStream_t s = static_cast<hipStream_t>(stream);
    SearchJobs<T> jobs;
    FusedJobs<T> fjobs;
    SchedJobs sjobs;
    int njobs = 0;
    size_t lds = 0;
    bool any_window = false, all_fused = true;
    Call<T> c;
    for (int k = 0; k < K; ++k) {
        TRY(begin_call<T>(c, d, strides + 3 * k, voxel, 0, wh, s));
        TRY(run_prep<T>(points, c));
        wh.flags |= CONV3P_CACHE_POINTS_UNCHANGED;      // the other stencils see the same points
        if (c.skip_search) continue;
        const Stencil<T> &st = c.st;
        const auto &S = c.L.slot[c.slot];
        TRY(run_cloud_min<T>(points, c));
        any_window |= st.window != 0;
        const size_t l = search_lds_bytes(st, c.L.gtiles);
        if (l > kMaxLds) return CONV3P_ERR_UNSUPPORTED;
        lds = l > lds ? l : lds;
        all_fused &= fused_ok(c) && fused_mask_depth((int)sizeof(T), c.d.ntiles, c.st.ntap, c.st.maxfull) > 0;
        fjobs.job[njobs] = make_fused_job(c);
        SearchJob<T> &j = jobs.job[njobs++];
        j.st = st;
        j.cc = c.cc;
        j.count = S.count;
        j.tcount = S.tcount;
        j.pairs = S.pairs;
        j.segs = S.segs;
        j.qsegs = S.qsegs;
        j.qbm = S.qbm;
        j.qbm_hi = S.qbm_hi;
        sjobs.job[njobs - 1] = make_sched_job(c, c.slot, true);
    }
    if (njobs == 0) return CONV3P_OK;
    if (all_fused) return launch_fused<T>(c, fjobs, sjobs, njobs);
    for (int k = 0; k < njobs; ++k)   // (the tile-pair search lets fewer false positives through: its own threshold)
        sjobs.job[k].limit = kShortListsPerPoint * (unsigned long long)c.d.B * (unsigned long long)c.d.N;
#ifndef CONV3P_DEV_JOBS_IN_ORDER
    // widest stencil first (blockIdx.y ascending is the dispatch order): its boxes meet the most candidate tiles, its
    // workgroups run longest -- started last they would be the launch's tail
    for (int a = 1; a < njobs; ++a)
        for (int b2 = a; b2 > 0; --b2) {
            auto vol = [](const SearchJob<T> &j) { return (long long)j.st.step[0] * j.st.step[1] * j.st.step[2]; };
            if (vol(jobs.job[b2]) <= vol(jobs.job[b2 - 1])) break;
            std::swap(jobs.job[b2], jobs.job[b2 - 1]);
            std::swap(sjobs.job[b2], sjobs.job[b2 - 1]);
        }
#endif
    const BlockMap bm = make_blockmap(c.d);
    {
        Scope sc(K_SEARCH, s);
        auto launch = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(grid_of(bm), njobs), dim3(256), lds, s, c.L.pts, c.L.boxes, c.d.N, c.d.ntiles,
                               c.L.gtiles, c.L.ngroups, bm, jobs, c.L.cmin);
        };
        if (any_window) launch(search_multi_kernel<T, true>);    // window replication is a no-op for odd extents
        else launch(search_multi_kernel<T, false>);
        hipLaunchKernelGGL(tile_sched_kernel, dim3(8, njobs), dim3(1024), 0, s, sjobs, c.d.B, c.d.ntiles, c.L.ngroups,
                           bm.rounds * c.d.ntiles);
    }
    return hip_ok();
}

// fp64 36 -> 13 (the scene_seg head in double precision): the backward kernel's G matrix (27 x 13 rows of 64 doubles)
// plus the transposed filter do not fit LDS, and the generic kernels take 37 ms for it.  The op is linear in the
// output channels, so it runs as three passes of the SAME register-path kernel over column blocks of 5 + 4 + 4
// output channels: grad_out is read through its row stride from a column offset, the filter block is packed into
// scratch, every pass adds its grad_input contribution to the previous ones (the SELU epilogue, if any, in the last
// pass), and the grad_filter blocks are reduced and scattered back.  Deterministic like the single-pass kernel.
template <typename T>
int backward_split_36_13(const Call<T> &c, const T *grad_out, const T *input, const T *filter, T *grad_input,
                         T *grad_filter)
{
    if constexpr (sizeof(T) != 8) {
        return CONV3P_ERR_UNSUPPORTED;
    } else {
        const Dims &d = c.d;
        const int widths[3] = {5, 4, 4};
        const int nslots = (int)grid_of(make_blockmap(d));
        const size_t rows = (size_t)d.ntap * d.Cin;
        const size_t nwp_max = rows * 5;
        T *region = c.L.partials;                         // [nslots][rows * width] per pass (sized for 13 columns)
        T *Wp = region + (size_t)nslots * nwp_max;        // packed filter block
        T *Wd = Wp + nwp_max;                             // its grad_filter block
        int c0 = 0;
        for (int p = 0; p < 3; ++p) {
            const int cw = widths[p];
            const size_t nwp = rows * cw;
            hipLaunchKernelGGL(copy_cols_kernel<T>, dim3((unsigned)((nwp + 255) / 256)), dim3(256), 0, c.s, filter + c0, Wp,
                               rows, cw, d.Cout, cw);
            Call<T> cp = c;
            cp.d.Cout = cw;
            cp.accum = p > 0;
            cp.act = c.act && p == 2;
            cp.addend = p == 2 ? c.addend : nullptr;
            const int rc = cw == 5 ? launch_backward<T, 36, 5>(cp, grad_out + c0, input, Wp, grad_input, region)
                                   : launch_backward<T, 36, 4>(cp, grad_out + c0, input, Wp, grad_input, region);
            if (rc != CONV3P_OK) return rc;
            {
                Scope sc(K_REDUCE, c.s);
                hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3((unsigned)((nwp + kReduceW - 1) / kReduceW)), dim3(1024), 0, c.s, region,
                                   nslots, nwp, Wd);
            }
            hipLaunchKernelGGL(copy_cols_kernel<T>, dim3((unsigned)((nwp + 255) / 256)), dim3(256), 0, c.s, Wd, grad_filter + c0,
                               rows, cw, cw, d.Cout);
            c0 += cw;
        }
        return hip_ok();
    }
}

template <typename T>
int backward_impl(const T *grad_out, const T *points, const T *input, const T *filter,
                  const int32_t *stride, T voxel, int B, int N, int Cin, int Cout, int fz, int fy, int fx,
                  T *grad_input, T *grad_filter, const Where &wh, void *stream, bool act = false,
                  const T *addend = nullptr, const RowLd *ldp = nullptr, DeferredReduce<T> *defer = nullptr)
{
    Dims d{B, N, Cin, Cout, fz, fy, fx, 0, 0};
    TRY(check(d, stride, (double)voxel, true));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nw = (size_t)d.ntap * Cin * Cout;
    const size_t dx_elems = (size_t)B * N * Cin;
    if ((dx_elems && !grad_input) || (nw && !grad_filter)) return CONV3P_ERR_INVALID_ARGUMENT;
    if (ldp && (dx_elems == 0 || Cout == 0)) return CONV3P_ERR_UNSUPPORTED;
    if (dx_elems == 0 || Cout == 0) {                                    // nothing to accumulate
        TRY(zero_async(grad_input, dx_elems * sizeof(T), s));            // .cpp:580
        TRY(zero_async(grad_filter, nw * sizeof(T), s));                 // .cpp:590
        if (act && dx_elems && addend) {
            if (!input) return CONV3P_ERR_INVALID_ARGUMENT;
            return selu_grad_impl<T>(input, addend, nullptr, grad_input, (size_t)B * N, Cin, nullptr, stream);
        }
        return CONV3P_OK;
    }
    if (!points || !input || !filter || !grad_out) return CONV3P_ERR_INVALID_ARGUMENT;
    Call<T> c;
    c.act = act;
    c.addend = addend;
    set_ld(c, ldp, Cin, Cout);
    if (c.strided && !small_shape((int)sizeof(T), Cin, Cout)) return CONV3P_ERR_UNSUPPORTED;   // dense tensors only
    TRY(begin_call<T>(c, d, stride, voxel, backward_scratch_bytes(d, (int)sizeof(T), wh.ppp), wh, s));
    TRY(run_prep<T>(points, c));
    TRY(run_cloud_min<T>(points, c));
    TRY(run_search<T>(c, c.L.slot[c.slot].count, true));
    int rc = CONV3P_ERR_UNSUPPORTED;
    int nslots = (int)grid_of(make_blockmap(d));
    T *region = defer ? defer->region : nullptr;
#define X(ci, co)                                                                                    \
    if (Cin == ci && Cout == co && small_shape((int)sizeof(T), Cin, Cout))                           \
        rc = launch_backward<T, ci, co>(c, grad_out, input, filter, grad_input, region);
    CONV3P_SMALL_SHAPES(X)
#undef X
    if (defer && rc == CONV3P_OK) {
        defer->job = ReduceJob<T>{region, grad_filter, nslots, (unsigned)nw};
        return CONV3P_OK;
    }
    if (rc == CONV3P_ERR_UNSUPPORTED && !defer && sizeof(T) == 8 && Cin == 36 && Cout == 13 &&
        small_shape((int)sizeof(T), Cin, Cout)) {
        // (filters of more than 27 taps do not fit LDS even in column blocks: its first launch says so before anything is
        // written, and the generic kernels below take the call -- found by tools/fuzz_gpu.py, 3 x 5 x 3 taps in fp64)
        const int src = backward_split_36_13<T>(c, grad_out, input, filter, grad_input, grad_filter);
        if (src != CONV3P_ERR_UNSUPPORTED) return src;
    }
    if constexpr (sizeof(T) == 4) {
        int cip = 0, cop = 0;
        if (rc == CONV3P_ERR_UNSUPPORTED && c.deep_scratch_ok && deep_class(4, Cin, Cout, cip, cop)) {
#define X(ci, co)                                                                                    \
    if (cip == ci && cop == co) {                                                                    \
        int drc = deep_backward<ci, co>(c, grad_out, input, filter, grad_input, grad_filter);        \
        if (drc != CONV3P_ERR_UNSUPPORTED)                                                           \
            return drc != CONV3P_OK || !act ? drc                                                    \
                       : selu_grad_impl<T>(input, grad_input, addend, grad_input, (size_t)B * N, Cin, nullptr, stream); \
    }
            CONV3P_DEEP_SHAPES(X)
#undef X
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (rc == CONV3P_ERR_UNSUPPORTED && c.wide_scratch_ok) {
            const int wrc = wide_backward(c, grad_out, input, filter, grad_input, grad_filter);
            if (wrc != CONV3P_ERR_UNSUPPORTED)
                return wrc != CONV3P_OK || !act ? wrc
                           : selu_grad_impl<T>(input, grad_input, addend, grad_input, (size_t)B * N, Cin, nullptr, stream);
        }
    }
    if constexpr (sizeof(T) == 8) {
        if (rc == CONV3P_ERR_UNSUPPORTED && c.f64_scratch_ok && !c.strided && !defer) {
            const int brc = f64_blocked_backward(c, grad_out, input, filter, grad_input, grad_filter);
            if (brc != CONV3P_ERR_UNSUPPORTED)
                return brc != CONV3P_OK || !act ? brc
                           : selu_grad_impl<T>(input, grad_input, addend, grad_input, (size_t)B * N, Cin, nullptr, stream);
        }
    }
    bool generic = false;
    if (rc == CONV3P_ERR_UNSUPPORTED && c.strided) return rc;
    if (rc == CONV3P_ERR_UNSUPPORTED) {
        generic = true;
        nslots = generic_slots(d, (int)sizeof(T));
        if (small_shape((int)sizeof(T), Cin, Cout) && nslots > (int)grid_of(make_blockmap(d))) nslots = (int)grid_of(make_blockmap(d));
        TRY(zero_async(grad_input, dx_elems * sizeof(T), s));
        TRY(zero_async(c.L.partials, nw * (size_t)nslots * sizeof(T), s));
        rc = launch_backward<T, 0, 0>(c, grad_out, input, filter, grad_input, nullptr, nullptr, nslots);
    }
    TRY(rc);
    {
        Scope sc(K_REDUCE, s);
        hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3((unsigned)((nw + kReduceW - 1) / kReduceW)), dim3(1024), 0, s,
                           c.L.partials, nslots, nw, grad_filter);
    }
    TRY(hip_ok());
    if (act && generic)   // generic path has no fused epilogue
        return selu_grad_impl<T>(input, grad_input, addend, grad_input, (size_t)B * N, Cin, nullptr, stream);
    return CONV3P_OK;
}

template <typename T>
int count_impl(const T *points, const int32_t *stride, T voxel, int B, int N, int fz, int fy, int fx,
               int32_t *count, void *ws, size_t ws_bytes, void *stream)
{
    Dims d{B, N, 0, 0, fz, fy, fx, 0, 0};
    TRY(check(d, stride, (double)voxel, false));
    if ((size_t)B * N == 0) return CONV3P_OK;
    if (!points || !count) return CONV3P_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Call<T> c;
    const Where wh{ws, ws_bytes, false, 1, d.ntap, 0, 0, 0};   // populations only: no pair storage
    TRY(begin_call<T>(c, d, stride, voxel, 0, wh, s));
    TRY(run_prep<T>(points, c));
    TRY(run_cloud_min<T>(points, c));
    return run_search<T>(c, count, false);
}

template <typename T> int selu_impl(const T *x, T *y, size_t n, void *stream)
{
    if (n == 0) return CONV3P_OK;
    if (!x || !y) return CONV3P_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(K_SELU, s);
    const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(selu_kernel<T>, dim3(grid), dim3(256), 0, s, x, y, n);
    return hip_ok();
}
template <typename T>
int selu_grad_impl(const T *y, const T *dy, const T *dy_b, T *dx, size_t rows, int cols, const int *lds, void *stream)
{
    const size_t n = rows * (size_t)cols;
    if (n == 0) return CONV3P_OK;
    if (!y || !dy || !dx) return CONV3P_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(K_SELU_GRAD, s);
    const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(selu_grad_kernel<T>, dim3(grid), dim3(256), 0, s, y, dy, dy_b, dx, rows, cols,
                       lds ? lds[0] : cols, lds ? lds[1] : cols, lds ? lds[2] : cols, lds ? lds[3] : cols);
    return hip_ok();
}

size_t layout_bytes(int elem, int B, int N, int ntap, int nslots, int ppp, size_t scratch)
{
    const int ntiles = (N + kTile - 1) / kTile;
    return elem == 4 ? carve<float>(B, N, ntiles, ntap, nslots, ppp, scratch, nullptr).bytes
                     : carve<double>(B, N, ntiles, ntap, nslots, ppp, scratch, nullptr).bytes;
}

size_t cache_scratch_bytes(int elem, int B, int N, int max_taps, int max_Cin, int max_Cout, int ppp)
{
    // the largest need of any (Cin <= max_Cin, Cout <= max_Cout, taps <= max_taps) call: register-path shapes take
    // one partial per workgroup, every other shape the generic path's capped set of partials, deep shapes their own
    Dims d{B, N, max_Cin, max_Cout, 1, 1, max_taps, max_taps, (N + kTile - 1) / kTile};
    const size_t grid = (size_t)grid_of(make_blockmap(d));
    size_t b = (size_t)max_taps * max_Cin * max_Cout * (size_t)generic_slots(d, elem) * (size_t)elem;
#define X(ci, co)                                                                                    \
    if (ci <= max_Cin && co <= max_Cout) {                                                           \
        const size_t need = (size_t)max_taps * ci * co * grid * (size_t)elem;                        \
        if (need > b) b = need;                                                                      \
    }
    CONV3P_SMALL_SHAPES(X)
#undef X
#define X(ci, co)                                                                                    \
    if (ci <= max_Cin && co <= max_Cout && tap_forward_shape(elem, ci, co) && tap_forward_bytes(d) > b) b = tap_forward_bytes(d);
    CONV3P_SMALL_SHAPES(X)
#undef X
#define X(ci, co)                                                                                    \
    if (elem == 4 && max_Cin > 0 && max_Cout > 0 && ci <= deep_pad(max_Cin) && co <= deep_pad(max_Cout)) { \
        Dims dd{B, N, ci, co, 1, 1, max_taps, max_taps, (N + kTile - 1) / kTile};                    \
        const size_t need = deep_scratch_bytes(dd, (size_t)B * N * (size_t)ppp);                     \
        if (need > b) b = need;                                                                      \
    }
    CONV3P_DEEP_SHAPES(X)
#undef X
    if (elem == 8) {   // (any fp64 shape outside the register-path list may come: the blocked path's scratch)
        Dims db{B, N, max_Cin, max_Cout, 1, 1, max_taps, max_taps, (N + kTile - 1) / kTile};
        if (f64_blocked_bytes(db) > b) b = f64_blocked_bytes(db);
    }
    if (wide_shape(elem, max_Cin, max_Cout)) {
        Dims dw{B, N, max_Cin, max_Cout, 1, 1, max_taps, max_taps, (N + kTile - 1) / kTile};
        const size_t need = wide_scratch_bytes(dw, (size_t)B * N * (size_t)ppp);
        if (need > b) b = need;
    }
    return b;
}

Where stateless(void *ws, size_t bytes) { return Where{ws, bytes, false, 1, 0, kDefaultPairsPerPoint, 0, 0}; }

Where persistent(int elem, int B, int N, void *cache, size_t bytes, int slots, int max_taps, int ppp, int max_Cin,
                 int max_Cout, int flags)
{
    return Where{cache, bytes, true, slots, max_taps, ppp > 0 ? ppp : kDefaultPairsPerPoint,
                 cache_scratch_bytes(elem, B, N, max_taps, max_Cin, max_Cout, ppp > 0 ? ppp : kDefaultPairsPerPoint), flags};
}


// ----------------------------------------------------------------------------- the models' stack in one call
bool stack_desc_ok(const conv3p_stack_desc *sd)
{
    if (!sd || sd->n_hidden < 1 || sd->n_hidden > CONV3P_STACK_MAX_LAYERS || sd->in_channels < 1 || sd->hidden < 1 ||
        sd->num_class < 0 || sd->fz < 1 || sd->fy < 1 || sd->fx < 1)
        return false;
    for (int l = 0; l < sd->n_hidden + (sd->num_class > 0 ? 1 : 0); ++l)
        for (int a = 0; a < 3; ++a)
            if (sd->strides[l][a] < 1) return false;
    return true;
}
inline int stack_layers(const conv3p_stack_desc *sd) { return sd->n_hidden + (sd->num_class > 0 ? 1 : 0); }

// geometry of every layer on `s`, ordered after `after`; one event per layer in the cache's host record
template <typename T>
int stack_geometry(const conv3p_stack_desc *sd, const T *points, T voxel, int B, int N, void *cache, size_t cache_bytes,
                   const conv3p_cache_config *cfg, hipStream_t s, hipStream_t after, bool differs, bool one_launch)
{
    const int nl = stack_layers(sd);
    std::vector<hipEvent_t> ev;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        CacheHost &h = g_caches[cache];
        while ((int)h.ready.size() < nl + 1) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return CONV3P_ERR_LAUNCH;
            h.ready.push_back(e);
        }
        ev = h.ready;
    }
    if (differs) {   // `points` is produced on another stream
        if (hipEventRecord(ev[nl], after) != hipSuccess || hipStreamWaitEvent(s, ev[nl], 0) != hipSuccess)
            return CONV3P_ERR_LAUNCH;
    }
    // one_launch (prefetch of the NEXT batch: nobody waits for the first layer's lists): all strides in ONE search
    // launch, whose light tiles fill in behind another stride's heavy ones (0.534 -> 0.527 ms/step on cfg2);
    // otherwise one launch and one event per layer, so that layer 0 can start as soon as its lists exist
#ifdef CONV3P_DEV_STACK_NO_MULTI   // developer A/B build only (-DCONV3P_DEV_STACK_NO_MULTI)
    const bool no_batch = true;
#else
    const bool no_batch = false;
#endif
    if (one_launch && !no_batch && nl <= kMaxJobs) {
        int32_t strides[kMaxJobs * 3];
        int k = 0;
        for (int l = 0; l < nl; ++l) {
            bool dup = false;
            for (int m = 0; m < k; ++m)
                dup |= strides[3 * m] == sd->strides[l][0] && strides[3 * m + 1] == sd->strides[l][1] && strides[3 * m + 2] == sd->strides[l][2];
            if (!dup) { strides[3 * k] = sd->strides[l][0]; strides[3 * k + 1] = sd->strides[l][1]; strides[3 * k + 2] = sd->strides[l][2]; ++k; }
        }
        const Where wh = persistent((int)sizeof(T), B, N, cache, cache_bytes, cfg->slots, cfg->max_taps, cfg->pairs_per_point,
                                    cfg->max_Cin, cfg->max_Cout, 0);
        TRY(prepare_multi_impl<T>(points, strides, k, voxel, B, N, sd->fz, sd->fy, sd->fx, wh, s));
        for (int l = 0; l < nl; ++l)
            if (hipEventRecord(ev[l], s) != hipSuccess) return CONV3P_ERR_LAUNCH;
    } else
    for (int l = 0; l < nl; ++l) {
        conv3p_cache_config c2 = *cfg;
        c2.flags = (l > 0 ? CONV3P_CACHE_POINTS_UNCHANGED : 0) | (cfg->flags & (CONV3P_CACHE_SPARSE_NEIGHBOURHOODS | CONV3P_CACHE_DENSE_NEIGHBOURHOODS));
        const Where wh = persistent((int)sizeof(T), B, N, cache, cache_bytes, c2.slots, c2.max_taps, c2.pairs_per_point,
                                    c2.max_Cin, c2.max_Cout, c2.flags);
        TRY(prepare_impl<T>(points, sd->strides[l], voxel, B, N, sd->fz, sd->fy, sd->fx, wh, s));
        if (hipEventRecord(ev[l], s) != hipSuccess) return CONV3P_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lk(g_cache_mu);
    CacheHost &h = g_caches[cache];
    h.pending_points = points;
    h.pending_layers = nl;
    return CONV3P_OK;
}


// ----------------------------------------------------------------------------- fused stack launches (conv3p_stack_fused.hpp)
// Per device, once: the placement census (is workgroup b of a 2048-workgroup grid on the same XCC for every b of one
// residue mod 8?) and the CU count.  One host synchronisation, at the first stack call of the process.
struct FusedDevice {
    bool ok = false;
    uint32_t xcc_of[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cus = 0;
};
const FusedDevice &fused_device()
{
    static std::mutex mu;
    static std::map<int, FusedDevice> devs;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto it = devs.find(dev);
    if (it != devs.end()) return it->second;
    FusedDevice fd;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        fd.cus = prop.multiProcessorCount;
        constexpr int kCensus = 2048;
        uint32_t *d = nullptr;
        std::vector<uint32_t> h(kCensus, 0xFFFFFFFFu);
        if (hipMalloc(&d, sizeof(uint32_t) * kCensus) == hipSuccess) {
            hipLaunchKernelGGL(xcc_census_kernel, dim3(kCensus), dim3(256), 0, nullptr, d);
            if (hipMemcpy(h.data(), d, sizeof(uint32_t) * kCensus, hipMemcpyDeviceToHost) == hipSuccess) {
                fd.ok = true;
                for (int i = 0; i < 8; ++i) fd.xcc_of[i] = h[i];
                for (int b = 0; b < kCensus; ++b) fd.ok &= h[b] == fd.xcc_of[b & 7];
            }
            (void)hipFree(d);
        }
        (void)hipGetLastError();
    }
    return devs.emplace(dev, fd).first->second;
}
// workgroups of `kern` (256 threads, `lds` bytes of dynamic LDS) the device holds at once
int fused_capacity(const void *kern, size_t lds, int cus)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, size_t>, int> memo;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(kern, lds);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second * cus;
    int nb = 0;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess)
        nb = 0;
    (void)hipGetLastError();
    memo[key] = nb;
    return nb * cus;
}
// the cache's arrival counters (created at the first fused launch on it; zero-filled once, never reset: the host tracks
// their value)
uint32_t *fused_sync(void *cache, int B, int kind, uint32_t arrivals, uint32_t &base, int *status)
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    CacheHost &h = g_caches[cache];
    *status = CONV3P_ERR_LAUNCH;
    // an earlier fused launch on this cache reported an error (a barrier wait that gave up: e.g. two fused launches on one
    // device at the same time, each holding the slots the other's tiles wait for; or a cloud whose tiles did not share an
    // XCC): its results were wrong -- fail loudly now, once
    if (h.host_err != nullptr && *reinterpret_cast<volatile uint32_t *>(h.host_err) != 0u) {
        *reinterpret_cast<volatile uint32_t *>(h.host_err) = 0u;
        return nullptr;
    }
    if (h.sync != nullptr && h.sync_clouds < B) {
        (void)hipFree(h.sync);   // (synchronises: nothing can still be using it)
        h.sync = nullptr;
    }
    if (h.sync == nullptr) {
        const size_t words = (size_t)2 * B * kSyncLineWords;
        if (h.host_err == nullptr) {
            if (hipHostMalloc(reinterpret_cast<void **>(&h.host_err), sizeof(uint32_t), hipHostMallocMapped) != hipSuccess) {
                (void)hipGetLastError();
                h.host_err = nullptr;
                return nullptr;
            }
            *h.host_err = 0u;
        }
        void *dev_err = nullptr;
        std::vector<uint32_t> init(words, 0u);
        if (hipHostGetDevicePointer(&dev_err, h.host_err, 0) != hipSuccess) dev_err = nullptr;
        for (size_t l = 0; l < words; l += kSyncLineWords) {   // words 4-5 of every line: where an error is also reported
            init[l + 4] = (uint32_t)(reinterpret_cast<uintptr_t>(dev_err) & 0xFFFFFFFFu);
            init[l + 5] = (uint32_t)((unsigned long long)reinterpret_cast<uintptr_t>(dev_err) >> 32);
        }
        if (hipMalloc(&h.sync, words * sizeof(uint32_t)) != hipSuccess ||
            hipMemcpy(h.sync, init.data(), words * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            h.sync = nullptr;
            return nullptr;
        }
        h.sync_clouds = B;
        h.sync_base[0] = h.sync_base[1] = 0u;
    }
    base = h.sync_base[kind];
    h.sync_base[kind] += arrivals;
    h.fused_launches[kind] += 1u;
    *status = CONV3P_OK;
    return h.sync + (size_t)kind * h.sync_clouds * kSyncLineWords;
}

template <typename T> size_t forward_lds_bytes(const Stencil<T> &st, int ci, int co)
{
    return lds_common(st) + a16((size_t)st.ntap * fwd_wstr<T>(ci, co) * sizeof(T)) + a16((size_t)st.ntap * kCntStride * sizeof(T)) + 256 +
           a16((size_t)kWavesPerBlock * 192 * 4) + a16((size_t)kWavesPerBlock * co * 64 * sizeof(T));
}

// Can the hidden layers of this stack run as fused launches on this cache?  Decided from the description alone, BEFORE any
// call touches the cache's host record.  (fp32; in_channels 3 or 9 -> 9 -> 9 ...; odd dilated extents; one search group;
// enough slots for the strides not to evict each other; the whole grid resident at once; census passed.)
template <typename T>
bool stack_fusable(const conv3p_stack_desc *sd, T voxel, int B, int N, const conv3p_cache_config *cfg, size_t scratch_cap_needed,
                   size_t scratch_cap)
{
    if constexpr (sizeof(T) != 4) {
        return false;
    } else {
        if ((cfg->flags & CONV3P_CACHE_FUSED_STACK) == 0) return false;   // opt-in, one bit per pass (include/conv3p.h; the callers test theirs)
        if (sd->hidden != 9 || (sd->in_channels != 3 && sd->in_channels != 9) || sd->n_hidden < 2 || sd->n_hidden > kStackMaxFused) return false;
        if (sd->fz * sd->fy * sd->fx > 32 || N <= kTile || N > kFusedMaxPoints) return false;
        if (scratch_cap_needed > scratch_cap) return false;
        int distinct = 0;
        for (int l = 0; l < stack_layers(sd); ++l) {
            bool dup = false;
            for (int m = 0; m < l; ++m) dup |= sd->strides[m][0] == sd->strides[l][0] && sd->strides[m][1] == sd->strides[l][1] && sd->strides[m][2] == sd->strides[l][2];
            distinct += dup ? 0 : 1;
        }
        if (cfg->slots < distinct) return false;
        const int ntiles = (N + kTile - 1) / kTile;
        if (carve<T>(B, N, ntiles, cfg->max_taps, 1, cfg->pairs_per_point > 0 ? cfg->pairs_per_point : kDefaultPairsPerPoint, 0, nullptr).ngroups != 1) return false;
        for (int l = 0; l < sd->n_hidden; ++l) {
            Dims d{B, N, 9, 9, sd->fz, sd->fy, sd->fx, sd->fz * sd->fy * sd->fx, ntiles};
            if (check(d, sd->strides[l], (double)voxel, true) != CONV3P_OK) return false;
            if (make_stencil<T>(d, sd->strides[l], voxel).window) return false;
        }
        return fused_device().ok;
    }
}

// forward of the hidden layers in ONE launch.  CONV3P_ERR_UNSUPPORTED (nothing enqueued, nothing recorded): take the per-layer path.
template <typename T>
int stack_forward_fused(const conv3p_stack_desc *sd, const T *points, const T *input, const T *const *filters, T voxel, int B, int N,
                        T *concat, void *cache, size_t cache_bytes, const conv3p_cache_config *cfg, hipStream_t s, bool geometry_enqueued)
{
    if constexpr (sizeof(T) != 4) {
        return CONV3P_ERR_UNSUPPORTED;
    } else {
        const int nh = sd->n_hidden, H = sd->hidden, CW = nh * H;
        const size_t rows = (size_t)B * N;
        const size_t handoff = up(rows * H * sizeof(T));
        const Where wh0 = persistent((int)sizeof(T), B, N, cache, cache_bytes, cfg->slots, cfg->max_taps, cfg->pairs_per_point,
                                     cfg->max_Cin, cfg->max_Cout, 0);
        if ((cfg->flags & CONV3P_CACHE_FUSED_FORWARD) == 0 || !stack_fusable<T>(sd, voxel, B, N, cfg, handoff * (size_t)(nh - 1), wh0.scratch_cap))
            return CONV3P_ERR_UNSUPPORTED;
        const FusedDevice &fd = fused_device();
        const int ntiles = (N + kTile - 1) / kTile;
        Dims d0{B, N, sd->in_channels, H, sd->fz, sd->fy, sd->fx, sd->fz * sd->fy * sd->fx, ntiles};
        const BlockMap bm = make_blockmap(d0);
        size_t lds = 0;
        for (int l = 0; l < nh; ++l) {
            Dims d = d0;
            d.Cin = l == 0 ? sd->in_channels : H;
            lds = std::max(lds, forward_lds_bytes<T>(make_stencil<T>(d, sd->strides[l], voxel), d.Cin, H));
        }
        const void *kern = sd->in_channels == 3 ? reinterpret_cast<const void *>(stack_forward_kernel<T, 3, 9>)
                                                : reinterpret_cast<const void *>(stack_forward_kernel<T, 9, 9>);
        if (lds > kMaxLds || (int)grid_of(bm) > fused_capacity(kern, lds, fd.cus)) return CONV3P_ERR_UNSUPPORTED;
        for (int l = 0; l < nh; ++l)
            if (!filters[l]) return CONV3P_ERR_INVALID_ARGUMENT;
        // from here on the cache's host record is touched exactly as the per-layer calls would touch it
        StackFwdArgs<T> a{};
        for (int l = 0; l < nh; ++l) {
            const int flags = ((l > 0 || geometry_enqueued) ? CONV3P_CACHE_POINTS_UNCHANGED : 0) |
                              (cfg->flags & (CONV3P_CACHE_SPARSE_NEIGHBOURHOODS | CONV3P_CACHE_DENSE_NEIGHBOURHOODS));
            const Where wh = persistent((int)sizeof(T), B, N, cache, cache_bytes, cfg->slots, cfg->max_taps, cfg->pairs_per_point,
                                        cfg->max_Cin, cfg->max_Cout, flags);
            Dims d = d0;
            d.Cin = l == 0 ? sd->in_channels : H;
            Call<T> c;
            TRY(begin_call<T>(c, d, sd->strides[l], voxel, handoff * (size_t)(nh - 1), wh, s));
            TRY(run_prep<T>(points, c));
            TRY(run_cloud_min<T>(points, c));
            TRY(run_search<T>(c, c.L.slot[c.slot].count, true));
            const auto &S = c.L.slot[c.slot];
            StackFwdLayer<T> &L = a.layer[l];
            L.st = c.st;
            L.count = S.count; L.tcount = S.tcount; L.pairs = S.pairs; L.segs = S.segs; L.qsegs = S.qsegs;
            char *hand = reinterpret_cast<char *>(c.L.partials);
            // layer l gathers the dense hand-off copy of layer l - 1's activation (a buffer of its own: no stale L1 line)
            L.input = l == 0 ? input : reinterpret_cast<const T *>(hand + handoff * (size_t)(l - 1));
            L.filter = filters[l];
            L.output = concat + (size_t)H * l;
            L.out2 = l + 1 < nh ? reinterpret_cast<T *>(hand + handoff * (size_t)l) : nullptr;
            L.ld_out2 = H;
            L.ld = RowLd{l == 0 ? sd->in_channels : H, CW, H, d.Cin, d.Cin};
            if (l == 0) {
                a.pts = c.L.pts; a.boxes = c.L.boxes; a.cmin = nullptr;
            }
        }
        a.N = N; a.ntiles = ntiles; a.nl = nh; a.bm = bm;
        int srcs = CONV3P_OK;
        a.sync = fused_sync(cache, B, 0, (uint32_t)(ntiles * (nh - 1)), a.base, &srcs);
        if (a.sync == nullptr) return srcs;
        Scope sc(K_FORWARD, s);
        if (sd->in_channels == 3) hipLaunchKernelGGL((stack_forward_kernel<T, 3, 9>), dim3(grid_of(bm)), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((stack_forward_kernel<T, 9, 9>), dim3(grid_of(bm)), dim3(256), lds, s, a);
        return hip_ok();
    }
}

template <typename T>
int stack_prefetch_impl(const conv3p_stack_desc *sd, const T *points, T voxel, int B, int N, void *cache,
                        size_t cache_bytes, const conv3p_cache_config *cfg, void *stream, void *after_stream)
{
    if (!stack_desc_ok(sd) || !cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    if ((size_t)B * N == 0) return CONV3P_OK;
    if (!points) return CONV3P_ERR_INVALID_ARGUMENT;
    return stack_geometry<T>(sd, points, voxel, B, N, cache, cache_bytes, cfg, static_cast<hipStream_t>(stream),
                             static_cast<hipStream_t>(after_stream), after_stream != stream, /*one_launch=*/true);
}

template <typename T>
int stack_forward_impl(const conv3p_stack_desc *sd, const T *points, const T *input, const T *const *filters, T voxel,
                       int B, int N, T *concat, T *head_out, void *cache, size_t cache_bytes,
                       const conv3p_cache_config *cfg, void *stream, void *side_stream)
{
    if (!stack_desc_ok(sd) || !cache_cfg_ok(cfg) || !filters) return CONV3P_ERR_INVALID_ARGUMENT;
    if ((size_t)B * N == 0) return CONV3P_OK;
    if (!points || !input || !concat || (sd->num_class > 0 && !head_out)) return CONV3P_ERR_INVALID_ARGUMENT;
    const int nl = stack_layers(sd), CW = sd->n_hidden * sd->hidden;
    hipStream_t main = static_cast<hipStream_t>(stream);
    // geometry: already enqueued by conv3p_stack_prefetch_* for these points, or enqueued now on the side stream
    // (each layer's search then runs while the previous layers accumulate), or built inline by the op calls
    bool events = false, prefetched = false;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_caches.find(cache);
        if (it != g_caches.end() && it->second.pending_points == points && it->second.pending_layers == nl)
            events = prefetched = true;
        if (it != g_caches.end()) it->second.pending_points = nullptr;
    }
    if (!events && side_stream != nullptr && side_stream != stream) {
        TRY(stack_geometry<T>(sd, points, voxel, B, N, cache, cache_bytes, cfg, static_cast<hipStream_t>(side_stream), main, true,
                              /*one_launch=*/false));
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g_caches[cache].pending_points = nullptr;
        events = true;
    }
    std::vector<hipEvent_t> ev;
    if (events) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        ev = g_caches[cache].ready;
    }
    // prefetched geometry (normally finished long ago, under the previous batch's backward): ONE wait on the last
    // layer's event covers them all; geometry enqueued just now: per-layer waits, so that layer 0 starts early
    if (prefetched && hipStreamWaitEvent(main, ev[nl - 1], 0) != hipSuccess) return CONV3P_ERR_LAUNCH;
    // the hidden layers as ONE launch (conv3p_stack_fused.hpp) where the stack, the cache and the device allow it
    int first = 0;
    {
        const Where wh0 = persistent((int)sizeof(T), B, N, cache, cache_bytes, cfg->slots, cfg->max_taps, cfg->pairs_per_point,
                                     cfg->max_Cin, cfg->max_Cout, 0);
        if ((cfg->flags & CONV3P_CACHE_FUSED_FORWARD) != 0 &&
            stack_fusable<T>(sd, voxel, B, N, cfg, up((size_t)B * N * sd->hidden * sizeof(T)) * (size_t)(sd->n_hidden - 1), wh0.scratch_cap)) {
            if (events && !prefetched && hipStreamWaitEvent(main, ev[sd->n_hidden - 1], 0) != hipSuccess) return CONV3P_ERR_LAUNCH;
            const int rc = stack_forward_fused<T>(sd, points, input, filters, voxel, B, N, concat, cache, cache_bytes, cfg, main, events);
            if (rc == CONV3P_OK) first = sd->n_hidden;
            else if (rc != CONV3P_ERR_UNSUPPORTED) return rc;
        }
    }
    for (int l = first; l < nl; ++l) {
        if (events && !prefetched && hipStreamWaitEvent(main, ev[l], 0) != hipSuccess) return CONV3P_ERR_LAUNCH;
        conv3p_cache_config c2 = *cfg;
        c2.flags = ((l > 0 || events) ? CONV3P_CACHE_POINTS_UNCHANGED : 0) |   // the first call of a step re-validates
                   (cfg->flags & (CONV3P_CACHE_SPARSE_NEIGHBOURHOODS | CONV3P_CACHE_DENSE_NEIGHBOURHOODS));
        const bool head = l == sd->n_hidden;
        const int Cin = head ? CW : (l == 0 ? sd->in_channels : sd->hidden);
        const int Cout = head ? sd->num_class : sd->hidden;
        const T *x = head ? concat : (l == 0 ? input : concat + (size_t)sd->hidden * (l - 1));
        T *y = head ? head_out : concat + (size_t)sd->hidden * l;
        const RowLd ld{head ? CW : (l == 0 ? sd->in_channels : CW), head ? sd->num_class : CW, 0, 0, 0};
        const Where wh = persistent((int)sizeof(T), B, N, cache, cache_bytes, c2.slots, c2.max_taps, c2.pairs_per_point,
                                    c2.max_Cin, c2.max_Cout, c2.flags);
        if (!filters[l]) return CONV3P_ERR_INVALID_ARGUMENT;
        TRY(forward_impl<T>(points, x, filters[l], sd->strides[l], voxel, B, N, Cin, Cout, sd->fz, sd->fy, sd->fx, y, wh,
                            stream, /*act=*/true, &ld));
    }
    return CONV3P_OK;
}

// bytes of layer l's grad_filter partials (one per backward workgroup)
template <typename T> size_t stack_region_bytes(const conv3p_stack_desc *sd, int l, int B, int N)
{
    const bool head = l == sd->n_hidden;
    const int Cin = head ? sd->n_hidden * sd->hidden : (l == 0 ? sd->in_channels : sd->hidden);
    const int Cout = head ? sd->num_class : sd->hidden;
    Dims d{B, N, Cin, Cout, sd->fz, sd->fy, sd->fx, sd->fz * sd->fy * sd->fx, (N + kTile - 1) / kTile};
    return up((size_t)d.ntap * Cin * Cout * (size_t)grid_of(make_blockmap(d)) * sizeof(T));
}

template <typename T> size_t stack_scratch_bytes(const conv3p_stack_desc *sd, int B, int N)
{
    const size_t rows = (size_t)B * N;
    const size_t wide = (size_t)(sd->hidden > sd->num_class ? sd->hidden : sd->num_class);
    // two ping-pong gradient buffers + the head's gradient w.r.t. the concat + every layer's grad_filter partials
    size_t b = up(rows * wide * sizeof(T)) * 2 + up(rows * (size_t)sd->n_hidden * sd->hidden * sizeof(T));
    for (int l = 0; l < stack_layers(sd); ++l) b += stack_region_bytes<T>(sd, l, B, N);
    // + the fused launch's gradient hand-off buffers, one per hidden layer (conv3p_stack_fused.hpp: every layer's rows in a
    // buffer of their own)
    b += up(rows * (size_t)sd->hidden * sizeof(T)) * (size_t)sd->n_hidden;
    return b;
}


// backward of the hidden layers nh-1 .. 1 in ONE launch (populated-rows kernel; needs the caller's SPARSE hint: the fused
// launch cannot pick the kernel per layer on the device).  G[l] = gradient w.r.t. the conv output of hidden layer l, dense
// [B][N][H], each in a buffer of its own.  CONV3P_ERR_UNSUPPORTED (nothing enqueued): take the per-layer path.
template <typename T>
int stack_backward_fused(const conv3p_stack_desc *sd, const T *points, const T *const *filters, T voxel, int B, int N, const T *concat,
                         const T *ext, int ld_ext, T *const *G, DeferredReduce<T> *red, T *const *grad_filters, void *cache, size_t cache_bytes,
                         const conv3p_cache_config *cfg, hipStream_t s)
{
    if constexpr (sizeof(T) != 4) {
        return CONV3P_ERR_UNSUPPORTED;
    } else {
        const int nh = sd->n_hidden, H = sd->hidden, CW = nh * H;
        if ((cfg->flags & CONV3P_CACHE_SPARSE_NEIGHBOURHOODS) == 0 || (cfg->flags & CONV3P_CACHE_FUSED_BACKWARD) == 0) return CONV3P_ERR_UNSUPPORTED;
        const Where wh0 = persistent((int)sizeof(T), B, N, cache, cache_bytes, cfg->slots, cfg->max_taps, cfg->pairs_per_point,
                                     cfg->max_Cin, cfg->max_Cout, 0);
        if (!stack_fusable<T>(sd, voxel, B, N, cfg, 0, wh0.scratch_cap)) return CONV3P_ERR_UNSUPPORTED;
        const FusedDevice &fd = fused_device();
        const int ntiles = (N + kTile - 1) / kTile;
        Dims d{B, N, H, H, sd->fz, sd->fy, sd->fx, sd->fz * sd->fy * sd->fx, ntiles};
        const BlockMap bm = make_blockmap(d);
        size_t lds = 0;
        int caps[kStackMaxFused];
        for (int l = nh - 1; l >= 1; --l) {
            size_t sl = 0;
            caps[l] = sparse_cap<T>(make_stencil<T>(d, sd->strides[l], voxel), H, H, sl);
            if (caps[l] <= 0 || sl > 40960) return CONV3P_ERR_UNSUPPORTED;   // (undilated layers: dense G; four workgroups per CU)
            lds = std::max(lds, sl);
        }
        const void *kern = reinterpret_cast<const void *>(stack_backward_kernel<T, 9>);
        if ((int)grid_of(bm) > fused_capacity(kern, lds, fd.cus)) return CONV3P_ERR_UNSUPPORTED;
        StackBwdArgs<T> a{};
        int k = 0;
        for (int l = nh - 1; l >= 1; --l, ++k) {
            const Where wh = persistent((int)sizeof(T), B, N, cache, cache_bytes, cfg->slots, cfg->max_taps, cfg->pairs_per_point,
                                        cfg->max_Cin, cfg->max_Cout,
                                        CONV3P_CACHE_POINTS_UNCHANGED | (cfg->flags & (CONV3P_CACHE_SPARSE_NEIGHBOURHOODS | CONV3P_CACHE_DENSE_NEIGHBOURHOODS)));
            Call<T> c;
            TRY(begin_call<T>(c, d, sd->strides[l], voxel, backward_scratch_bytes(d, (int)sizeof(T), wh.ppp), wh, s));
            TRY(run_prep<T>(points, c));
            TRY(run_cloud_min<T>(points, c));
            TRY(run_search<T>(c, c.L.slot[c.slot].count, true));
            const auto &S = c.L.slot[c.slot];
            StackBwdLayer<T> &L = a.layer[k];
            L.st = c.st;
            L.count = S.count; L.pairs = S.pairs; L.segs = S.segs; L.qsegs = S.qsegs; L.qbm = S.qbm;
            L.grad_out = G[l];
            L.input = concat + (size_t)H * (l - 1);
            L.filter = filters[l];
            L.addend = ext + (size_t)H * (l - 1);
            L.grad_input = G[l - 1];
            L.partials = red[l].region;
            L.ld = RowLd{CW, H, H, H, ld_ext};
            L.cap = caps[l];
            red[l].job = ReduceJob<T>{red[l].region, grad_filters[l], (int)grid_of(bm), (unsigned)((size_t)d.ntap * H * H)};
            if (k == 0) {
                a.pts = c.L.pts; a.boxes = c.L.boxes; a.cmin = nullptr;
            }
        }
        a.N = N; a.ntiles = ntiles; a.nl = k; a.bm = bm;
        a.nw = (size_t)d.ntap * H * H;
        a.top_act = concat + (size_t)H * (nh - 1);
        a.top_ext = ext + (size_t)H * (nh - 1);
        a.top_g = G[nh - 1];
        a.ld_act = CW;
        a.ld_ext = ld_ext;
        int srcs = CONV3P_OK;
        a.sync = fused_sync(cache, B, 1, (uint32_t)(ntiles * k), a.base, &srcs);
        if (a.sync == nullptr) return srcs;
        Scope sc(K_BACKWARD, s);
        (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((stack_backward_kernel<T, 9>), dim3(grid_of(bm)), dim3(256), lds, s, a);
        return hip_ok();
    }
}

template <typename T>
int stack_backward_impl(const conv3p_stack_desc *sd, const T *points, const T *input, const T *const *filters, T voxel,
                        int B, int N, const T *concat, const T *head_out, const T *grad_concat, const T *grad_head,
                        T *grad_input, T *const *grad_filters, void *scratch, size_t scratch_bytes, void *cache,
                        size_t cache_bytes, const conv3p_cache_config *cfg, void *stream)
{
    if (!stack_desc_ok(sd) || !cache_cfg_ok(cfg) || !filters || !grad_filters) return CONV3P_ERR_INVALID_ARGUMENT;
    if ((size_t)B * N == 0) return CONV3P_OK;   // (grad_filters of an empty batch are the caller's zeros)
    const bool has_head = sd->num_class > 0;
    if (!points || !input || !concat || !grad_input || (has_head && (!head_out || !grad_head)) ||
        (!has_head && !grad_concat))
        return CONV3P_ERR_INVALID_ARGUMENT;
    if (has_head && grad_concat) return CONV3P_ERR_UNSUPPORTED;   // (see conv3p.h: not needed by either model) -- before any launch
    TRY(buf_check(scratch, scratch_bytes, stack_scratch_bytes<T>(sd, B, N)));
    const size_t rows = (size_t)B * N;
    const int H = sd->hidden, CW = sd->n_hidden * H, nh = sd->n_hidden;
    const size_t wide = (size_t)(H > sd->num_class ? H : sd->num_class);
    char *sp = static_cast<char *>(scratch);
    T *ga = reinterpret_cast<T *>(sp);
    T *gb = reinterpret_cast<T *>(sp + up(rows * wide * sizeof(T)));
    T *dconcat = reinterpret_cast<T *>(sp + 2 * up(rows * wide * sizeof(T)));
    // every layer leaves its grad_filter partials in its own region; ONE launch reduces them all at the end, so the
    // chain of dependent backward kernels is not interleaved with reductions
    DeferredReduce<T> red[CONV3P_STACK_MAX_LAYERS + 1];
    {
        char *rp = sp + 2 * up(rows * wide * sizeof(T)) + up(rows * (size_t)CW * sizeof(T));
        for (int l = 0; l < stack_layers(sd); ++l) {
            red[l].region = reinterpret_cast<T *>(rp);
            rp += stack_region_bytes<T>(sd, l, B, N);
        }
    }
    auto reduce_all = [&]() -> int {
        ReduceJobs<T> jobs{};
        int nj = 0;
        unsigned gx = 0;
        for (int l = 0; l < stack_layers(sd); ++l)
            if (red[l].job.nslots > 0) {
                jobs.job[nj++] = red[l].job;
                const unsigned g = (red[l].job.nw + (unsigned)kReduceMultiW - 1u) / (unsigned)kReduceMultiW;
                gx = g > gx ? g : gx;
            }
        if (nj == 0) return CONV3P_OK;
        hipStream_t s = static_cast<hipStream_t>(stream);
        Scope sc(K_REDUCE, s);
        hipLaunchKernelGGL(reduce_multi_kernel<T>, dim3(gx, (unsigned)nj), dim3(1024), 0, s, jobs);
        return hip_ok();
    };
    auto where = [&]() {
        return persistent((int)sizeof(T), B, N, cache, cache_bytes, cfg->slots, cfg->max_taps, cfg->pairs_per_point,
                          cfg->max_Cin, cfg->max_Cout,
                          CONV3P_CACHE_POINTS_UNCHANGED | (cfg->flags & (CONV3P_CACHE_SPARSE_NEIGHBOURHOODS | CONV3P_CACHE_DENSE_NEIGHBOURHOODS)));
    };
    // external gradient of the concat's column blocks: the caller's, the head's, or their sum
    const T *ext = grad_concat;
    int ld_ext = CW;
    if (has_head) {
        // g = dL/d(head conv output) = grad_head * selu'(head_out)
        TRY(selu_grad_impl<T>(head_out, grad_head, nullptr, ga, rows, sd->num_class, nullptr, stream));
        TRY(backward_impl<T>(ga, points, concat, filters[nh], sd->strides[nh], voxel, B, N, CW, sd->num_class, sd->fz,
                             sd->fy, sd->fx, dconcat, grad_filters[nh], where(), stream, false, nullptr, nullptr,
                             &red[nh]));
        ext = dconcat;
    }
    // the hidden layers nh-1 .. 1 as ONE launch (conv3p_stack_fused.hpp) where the stack, the cache and the device allow it
    {
        T *G[CONV3P_STACK_MAX_LAYERS];
        char *gp = sp + 2 * up(rows * wide * sizeof(T)) + up(rows * (size_t)CW * sizeof(T));
        for (int l = 0; l < stack_layers(sd); ++l) gp += stack_region_bytes<T>(sd, l, B, N);
        for (int l = 0; l < nh; ++l) G[l] = reinterpret_cast<T *>(gp + up(rows * (size_t)H * sizeof(T)) * (size_t)l);
        const int rc = stack_backward_fused<T>(sd, points, filters, voxel, B, N, concat, ext, ld_ext, G, red, grad_filters, cache, cache_bytes,
                                               cfg, static_cast<hipStream_t>(stream));
        if (rc == CONV3P_OK) {
            TRY(backward_impl<T>(G[0], points, input, filters[0], sd->strides[0], voxel, B, N, sd->in_channels, H, sd->fz, sd->fy,
                                 sd->fx, grad_input, grad_filters[0], where(), stream, false, nullptr, nullptr, &red[0]));
            return reduce_all();
        }
        if (rc != CONV3P_ERR_UNSUPPORTED) return rc;
    }
    // g_l = dL/d(conv output of hidden layer l).  Last hidden layer: only the external gradient reaches its activation.
    T *g = has_head ? gb : ga, *gn = has_head ? ga : gb;
    {
        const int lds[4] = {CW, ld_ext, 0, H};
        TRY(selu_grad_impl<T>(concat + (size_t)H * (nh - 1), ext + (size_t)H * (nh - 1), nullptr, g, rows, H, lds, stream));
    }
    for (int l = nh - 1; l >= 1; --l) {
        // grad wrt the argument of the SELU that produced act_{l-1}: (dX + ext_{l-1}) * selu'(act_{l-1})
        const RowLd ld{CW, 0, H, H, ld_ext};
        const int Cin = H;
        TRY(backward_impl<T>(g, points, concat + (size_t)H * (l - 1), filters[l], sd->strides[l], voxel, B, N, Cin, H,
                             sd->fz, sd->fy, sd->fx, gn, grad_filters[l], where(), stream, /*act=*/true,
                             ext + (size_t)H * (l - 1), &ld, &red[l]));
        T *t = g; g = gn; gn = t;
    }
    TRY(backward_impl<T>(g, points, input, filters[0], sd->strides[0], voxel, B, N, sd->in_channels, H, sd->fz, sd->fy,
                         sd->fx, grad_input, grad_filters[0], where(), stream, false, nullptr, nullptr, &red[0]));
    return reduce_all();
}

bool cache_cfg_ok(const conv3p_cache_config *cfg)
{
    return cfg && cfg->slots > 0 && cfg->slots <= 64 && cfg->max_taps > 0 && cfg->max_taps < (int)kNoTap &&
           cfg->max_Cin >= 0 && cfg->max_Cout >= 0;
}

}  // namespace

extern "C" {

size_t conv3p_workspace_bytes(int pass, int elem_bytes, int B, int N, int Cin, int Cout, int fz, int fy,
                              int fx)
{
    if (pass < 0 || pass > 2 || (elem_bytes != 4 && elem_bytes != 8)) return 0;
    Dims d{B, N, Cin, Cout, fz, fy, fx, 0, 0};
    const int32_t one[3] = {1, 1, 1};
    if (check(d, one, 1.0, pass != CONV3P_PASS_NEIGHBOR_COUNT) != CONV3P_OK) return 0;
    const int ppp = pass == CONV3P_PASS_NEIGHBOR_COUNT ? 0 : kDefaultPairsPerPoint;
    const size_t scratch = pass == CONV3P_PASS_BACKWARD ? backward_scratch_bytes(d, elem_bytes)
                           : pass == CONV3P_PASS_FORWARD ? forward_scratch_bytes(d, elem_bytes) : 0;
    const size_t b = layout_bytes(elem_bytes, B, N, d.ntap, 1, ppp, scratch);
    return b ? b : kAlign;
}

size_t conv3p_cache_bytes(int elem_bytes, int B, int N, const conv3p_cache_config *cfg)
{
    if ((elem_bytes != 4 && elem_bytes != 8) || B < 0 || N < 0 || !cache_cfg_ok(cfg)) return 0;
    const int ppp = cfg->pairs_per_point > 0 ? cfg->pairs_per_point : kDefaultPairsPerPoint;
    return layout_bytes(elem_bytes, B, N, cfg->max_taps, cfg->slots, ppp,
                        cache_scratch_bytes(elem_bytes, B, N, cfg->max_taps, cfg->max_Cin, cfg->max_Cout, ppp));
}

int conv3p_cache_forget(void *cache)
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_caches.find(cache);
    if (it != g_caches.end()) {
        for (hipEvent_t e : it->second.ready) (void)hipEventDestroy(e);
        if (it->second.sync != nullptr) (void)hipFree(it->second.sync);
        if (it->second.host_err != nullptr) (void)hipHostFree(it->second.host_err);
        g_caches.erase(it);
    }
    return CONV3P_OK;
}

int conv3p_cache_fused_status(void *cache, unsigned *forward_launches, unsigned *backward_launches, unsigned *error_bits)
{
    uint32_t *sync = nullptr;
    int clouds = 0;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_caches.find(cache);
        if (forward_launches) *forward_launches = it != g_caches.end() ? it->second.fused_launches[0] : 0u;
        if (backward_launches) *backward_launches = it != g_caches.end() ? it->second.fused_launches[1] : 0u;
        if (it != g_caches.end()) { sync = it->second.sync; clouds = it->second.sync_clouds; }
    }
    if (error_bits) {
        *error_bits = 0u;
        if (sync != nullptr) {
            std::vector<uint32_t> h((size_t)2 * clouds * kSyncLineWords);
            if (hipMemcpy(h.data(), sync, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
                (void)hipGetLastError();
                return CONV3P_ERR_LAUNCH;
            }
            for (size_t l = 0; l < h.size(); l += kSyncLineWords) *error_bits |= h[l + 2];
        }
    }
    return CONV3P_OK;
}

int conv3p_cache_init(void *cache, size_t cache_bytes, void *stream)
{
    if (cache == nullptr) return cache_bytes == 0 ? CONV3P_OK : CONV3P_ERR_INVALID_ARGUMENT;
    (void)conv3p_cache_forget(cache);
    if (hipMemsetAsync(cache, 0, cache_bytes, static_cast<hipStream_t>(stream)) != hipSuccess) {
        (void)hipGetLastError();
        return CONV3P_ERR_LAUNCH;
    }
    return CONV3P_OK;
}

#define FWD_ARGS(T)                                                                                            \
    const T *points, const T *input, const T *filter, const int32_t *stride_xyz, T voxel_size, int B, int N,   \
        int Cin, int Cout, int fz, int fy, int fx, T *output
#define FWD_PASS points, input, filter, stride_xyz, voxel_size, B, N, Cin, Cout, fz, fy, fx, output
#define BWD_ARGS(T)                                                                                            \
    const T *grad_out, const T *points, const T *input, const T *filter, const int32_t *stride_xyz,            \
        T voxel_size, int B, int N, int Cin, int Cout, int fz, int fy, int fx, T *grad_input, T *grad_filter
#define BWD_PASS                                                                                               \
    grad_out, points, input, filter, stride_xyz, voxel_size, B, N, Cin, Cout, fz, fy, fx, grad_input, grad_filter

int conv3p_forward_f32(FWD_ARGS(float), void *workspace, size_t workspace_bytes, void *stream)
{
    return forward_impl<float>(FWD_PASS, stateless(workspace, workspace_bytes), stream);
}
int conv3p_forward_f64(FWD_ARGS(double), void *workspace, size_t workspace_bytes, void *stream)
{
    return forward_impl<double>(FWD_PASS, stateless(workspace, workspace_bytes), stream);
}
int conv3p_backward_f32(BWD_ARGS(float), void *workspace, size_t workspace_bytes, void *stream)
{
    return backward_impl<float>(BWD_PASS, stateless(workspace, workspace_bytes), stream);
}
int conv3p_backward_f64(BWD_ARGS(double), void *workspace, size_t workspace_bytes, void *stream)
{
    return backward_impl<double>(BWD_PASS, stateless(workspace, workspace_bytes), stream);
}

#define CACHE_ARGS void *cache, size_t cache_bytes, const conv3p_cache_config *cfg, void *stream
#define CACHE_WHERE(elem)                                                                                      \
    persistent(elem, B, N, cache, cache_bytes, cfg->slots, cfg->max_taps, cfg->pairs_per_point, cfg->max_Cin,   \
               cfg->max_Cout, cfg->flags)
int conv3p_forward_cached_f32(FWD_ARGS(float), CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return forward_impl<float>(FWD_PASS, CACHE_WHERE(4), stream);
}
int conv3p_forward_cached_f64(FWD_ARGS(double), CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return forward_impl<double>(FWD_PASS, CACHE_WHERE(8), stream);
}
int conv3p_backward_cached_f32(BWD_ARGS(float), CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return backward_impl<float>(BWD_PASS, CACHE_WHERE(4), stream);
}
int conv3p_backward_cached_f64(BWD_ARGS(double), CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return backward_impl<double>(BWD_PASS, CACHE_WHERE(8), stream);
}

int conv3p_layer_forward_cached_f32(FWD_ARGS(float), CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return forward_impl<float>(FWD_PASS, CACHE_WHERE(4), stream, true);
}
int conv3p_layer_forward_cached_f64(FWD_ARGS(double), CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return forward_impl<double>(FWD_PASS, CACHE_WHERE(8), stream, true);
}
#define LAYER_BWD_ARGS(T)                                                                                      \
    const T *grad_out, const T *points, const T *input, const T *filter, const int32_t *stride_xyz,            \
        T voxel_size, int B, int N, int Cin, int Cout, int fz, int fy, int fx, const T *grad_addend,           \
        T *grad_input, T *grad_filter
int conv3p_layer_backward_cached_f32(LAYER_BWD_ARGS(float), CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return backward_impl<float>(BWD_PASS, CACHE_WHERE(4), stream, true, grad_addend);
}
int conv3p_layer_backward_cached_f64(LAYER_BWD_ARGS(double), CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return backward_impl<double>(BWD_PASS, CACHE_WHERE(8), stream, true, grad_addend);
}

int conv3p_cache_prepare_f32(const float *points, const int32_t *stride_xyz, float voxel_size, int B, int N,
                             int fz, int fy, int fx, CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return prepare_impl<float>(points, stride_xyz, voxel_size, B, N, fz, fy, fx, CACHE_WHERE(4), stream);
}
int conv3p_cache_prepare_f64(const double *points, const int32_t *stride_xyz, double voxel_size, int B, int N,
                             int fz, int fy, int fx, CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return prepare_impl<double>(points, stride_xyz, voxel_size, B, N, fz, fy, fx, CACHE_WHERE(8), stream);
}

int conv3p_cache_prepare_multi_f32(const float *points, const int32_t *strides_xyz, int n_strides,
                                   float voxel_size, int B, int N, int fz, int fy, int fx, CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return prepare_multi_impl<float>(points, strides_xyz, n_strides, voxel_size, B, N, fz, fy, fx, CACHE_WHERE(4),
                                     stream);
}
int conv3p_cache_prepare_multi_f64(const double *points, const int32_t *strides_xyz, int n_strides,
                                   double voxel_size, int B, int N, int fz, int fy, int fx, CACHE_ARGS)
{
    if (!cache_cfg_ok(cfg)) return CONV3P_ERR_INVALID_ARGUMENT;
    return prepare_multi_impl<double>(points, strides_xyz, n_strides, voxel_size, B, N, fz, fy, fx, CACHE_WHERE(8),
                                      stream);
}

int conv3p_neighbor_count_f32(const float *points, const int32_t *stride_xyz, float voxel_size, int B,
                              int N, int fz, int fy, int fx, int32_t *count, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    return count_impl<float>(points, stride_xyz, voxel_size, B, N, fz, fy, fx, count, workspace,
                             workspace_bytes, stream);
}
int conv3p_neighbor_count_f64(const double *points, const int32_t *stride_xyz, double voxel_size, int B,
                              int N, int fz, int fy, int fx, int32_t *count, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    return count_impl<double>(points, stride_xyz, voxel_size, B, N, fz, fy, fx, count, workspace,
                              workspace_bytes, stream);
}

int conv3p_selu_f32(const float *x, float *y, size_t n, void *stream) { return selu_impl<float>(x, y, n, stream); }
int conv3p_selu_f64(const double *x, double *y, size_t n, void *stream) { return selu_impl<double>(x, y, n, stream); }
int conv3p_selu_grad_f32(const float *y, const float *dy, float *dx, size_t n, void *stream)
{
    return selu_grad_impl<float>(y, dy, nullptr, dx, n, 1, nullptr, stream);
}
int conv3p_selu_grad_f64(const double *y, const double *dy, double *dx, size_t n, void *stream)
{
    return selu_grad_impl<double>(y, dy, nullptr, dx, n, 1, nullptr, stream);
}
int conv3p_selu_grad_add_f32(const float *y, const float *dy_a, const float *dy_b, float *dx, size_t n,
                             void *stream)
{
    if (n && !dy_b) return CONV3P_ERR_INVALID_ARGUMENT;
    return selu_grad_impl<float>(y, dy_a, dy_b, dx, n, 1, nullptr, stream);
}
int conv3p_selu_grad_add_f64(const double *y, const double *dy_a, const double *dy_b, double *dx, size_t n,
                             void *stream)
{
    if (n && !dy_b) return CONV3P_ERR_INVALID_ARGUMENT;
    return selu_grad_impl<double>(y, dy_a, dy_b, dx, n, 1, nullptr, stream);
}

size_t conv3p_stack_scratch_bytes(const conv3p_stack_desc *desc, int elem_bytes, int B, int N)
{
    if (!stack_desc_ok(desc) || B < 0 || N < 0 || (elem_bytes != 4 && elem_bytes != 8)) return 0;
    const size_t b = elem_bytes == 4 ? stack_scratch_bytes<float>(desc, B, N) : stack_scratch_bytes<double>(desc, B, N);
    return b ? b : kAlign;
}
#define STACK_ENTRY(SFX, T)                                                                                         \
    int conv3p_stack_prefetch_##SFX(const conv3p_stack_desc *desc, const T *points, T voxel_size, int B, int N,     \
                                    void *cache, size_t cache_bytes, const conv3p_cache_config *cfg, void *stream,  \
                                    void *after_stream)                                                             \
    {                                                                                                               \
        return stack_prefetch_impl<T>(desc, points, voxel_size, B, N, cache, cache_bytes, cfg, stream,              \
                                      after_stream);                                                                \
    }                                                                                                               \
    int conv3p_stack_forward_##SFX(const conv3p_stack_desc *desc, const T *points, const T *input,                  \
                                   const T *const *filters, T voxel_size, int B, int N, T *concat, T *head_out,     \
                                   void *cache, size_t cache_bytes, const conv3p_cache_config *cfg, void *stream,   \
                                   void *side_stream)                                                               \
    {                                                                                                               \
        return stack_forward_impl<T>(desc, points, input, filters, voxel_size, B, N, concat, head_out, cache,       \
                                     cache_bytes, cfg, stream, side_stream);                                        \
    }                                                                                                               \
    int conv3p_stack_backward_##SFX(const conv3p_stack_desc *desc, const T *points, const T *input,                 \
                                    const T *const *filters, T voxel_size, int B, int N, const T *concat,           \
                                    const T *head_out, const T *grad_concat, const T *grad_head, T *grad_input,     \
                                    T *const *grad_filters, void *scratch, size_t scratch_bytes, void *cache,       \
                                    size_t cache_bytes, const conv3p_cache_config *cfg, void *stream)               \
    {                                                                                                               \
        return stack_backward_impl<T>(desc, points, input, filters, voxel_size, B, N, concat, head_out,             \
                                      grad_concat, grad_head, grad_input, grad_filters, scratch, scratch_bytes,     \
                                      cache, cache_bytes, cfg, stream);                                             \
    }
STACK_ENTRY(f32, float)
STACK_ENTRY(f64, double)

int conv3p_augment_f32(const float *points_in, const double *cos_sin, const double *noise, double sigma, double clip,
                       int B, int N, float *points_out, void *stream)
{
    if (B < 0 || N < 0 || !(clip > 0.0) || !(sigma >= 0.0)) return CONV3P_ERR_INVALID_ARGUMENT;   // assert(clip > 0), :72
    const size_t total = (size_t)B * N;
    if (total == 0) return CONV3P_OK;
    if (!points_in || !points_out) return CONV3P_ERR_INVALID_ARGUMENT;
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(augment_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), points_in,
                       reinterpret_cast<const double2 *>(cos_sin), noise, sigma, clip, points_out, total, N);
    return hip_ok();
}

int conv3p_sort_xyz_order_f32(const float *data, int B, int N, int row_floats, int32_t *order, void *stream)
{
    if (B < 0 || N < 0 || row_floats < 3) return CONV3P_ERR_INVALID_ARGUMENT;
    if ((size_t)B * N == 0) return CONV3P_OK;
    if (!data || !order) return CONV3P_ERR_INVALID_ARGUMENT;
    if (N > 8192) return CONV3P_ERR_UNSUPPORTED;
    int npad = 64;
    while (npad < N) npad <<= 1;
    const size_t lds = (size_t)npad * 16;
    const int threads = npad / 2 < 1024 ? (npad / 2 < 64 ? 64 : npad / 2) : 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sort_xyz_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(sort_xyz_kernel, dim3((unsigned)B), dim3(threads), lds, static_cast<hipStream_t>(stream), data, N,
                       row_floats, npad, order);
    return hip_ok();
}

int conv3p_gather_rows(const void *src, const int32_t *order, int B, int N, int row_bytes, void *dst, void *stream)
{
    if (B < 0 || N < 0 || row_bytes < 1) return CONV3P_ERR_INVALID_ARGUMENT;
    const size_t rows = (size_t)B * N;
    if (rows == 0) return CONV3P_OK;
    if (!src || !order || !dst || src == dst) return CONV3P_ERR_INVALID_ARGUMENT;
    const size_t work = rows * (size_t)((row_bytes & 3) == 0 ? row_bytes >> 2 : row_bytes);
    const unsigned grid = (unsigned)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char *>(src), order, N, row_bytes, static_cast<char *>(dst), rows);
    return hip_ok();
}

namespace {
struct FcPlan { int kc, chunks, mblocks, nblocks; size_t part_bytes, dz_bytes; };
bool fc_plan(int M, int K, int N, FcPlan &p)
{
    if (M < 1 || K < 1 || N < 1 || (N % 8) != 0 || N > 1024 || M > 128) return false;
    int kc = (K + 255) / 256;                     // ~256 workgroups along K ...
    kc = (kc + 1) & ~1;
    if (kc < 32) kc = 32;
    if (kc > 480) kc = 480;                       // ... with the x chunk [32][kc + 1] within 64 KiB of LDS
    p.kc = kc;
    p.chunks = (K + kc - 1) / kc;
    p.mblocks = (M + 31) / 32;
    p.nblocks = (N + kFcCols - 1) / kFcCols;
    p.part_bytes = up((size_t)p.chunks * p.mblocks * 32 * N * 4);
    p.dz_bytes = up((size_t)M * N * 4);
    return true;
}
}  // namespace

size_t conv3p_fc_workspace_bytes(int M, int K, int N)
{
    FcPlan p;
    if (!fc_plan(M, K, N, p)) return 0;
    return p.part_bytes > p.dz_bytes ? p.part_bytes : p.dz_bytes;
}

int conv3p_fc_forward_f32(const float *x, const float *W, const float *b, int M, int K, int N, int act, float *y,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    if (M < 0 || K < 0 || N < 0 || (act != 0 && act != 1)) return CONV3P_ERR_INVALID_ARGUMENT;
    if ((size_t)M * N == 0) return CONV3P_OK;
    if (!x || !W || !y || K == 0) return CONV3P_ERR_INVALID_ARGUMENT;
    FcPlan p;
    if (!fc_plan(M, K, N, p)) return CONV3P_ERR_UNSUPPORTED;
    TRY(buf_check(workspace, workspace_bytes, p.part_bytes));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *part = static_cast<float *>(workspace);
    {
        Scope sc(K_FC_FWD, s);
        const int psteps = ((p.kc + 1) / 2 + kFcDepth - 1) / kFcDepth * kFcDepth;
        const size_t lds = (size_t)32 * (2 * psteps + 1) * 4;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fc_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(fc_forward_kernel, dim3((unsigned)p.chunks, (unsigned)p.mblocks, (unsigned)p.nblocks), dim3(256), lds, s,
                           x, W, M, K, N, p.kc, part);
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(fc_finish_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, s, part, b, M, N, p.chunks,
                           p.mblocks, act, y);
    }
    return hip_ok();
}

int conv3p_fc_backward_f32(const float *x, const float *W, const float *y, const float *dy, int M, int K, int N,
                           int act, float *dx, float *dW, float *db, void *workspace, size_t workspace_bytes,
                           void *stream)
{
    if (M < 0 || K < 0 || N < 0 || (act != 0 && act != 1)) return CONV3P_ERR_INVALID_ARGUMENT;
    if ((size_t)K * N == 0) return CONV3P_OK;
    if (M == 0) return zero_async(dW, (size_t)K * N * 4, static_cast<hipStream_t>(stream));
    if (!x || !W || !dy || !dW || (act && !y)) return CONV3P_ERR_INVALID_ARGUMENT;
    FcPlan p;
    if (!fc_plan(M, K, N, p)) return CONV3P_ERR_UNSUPPORTED;
    TRY(buf_check(workspace, workspace_bytes, p.dz_bytes));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *dz = static_cast<float *>(workspace);
    const int dw_steps = M <= 32 ? 16 : M <= 64 ? 32 : 64;
    if ((size_t)2 * dw_steps * (N + 1) * 4 > kMaxLds) return CONV3P_ERR_UNSUPPORTED;   // before anything is launched
    hipLaunchKernelGGL(fc_dz_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, y, dy, M, N, act, dz, db);
    TRY(hip_ok());
    // waves per workgroup of the two streaming kernels: 32 rows of W per wave, the whole grid in ONE resident round
    // (two workgroups of ~66 KiB of LDS per CU: 512 slots) -- the model's fc1 (73 728 rows): 5 waves, 461 workgroups
    auto waves_for = [&](int slots) {
        const int tiles = (K + 31) / 32;
        int nw = (tiles + slots - 1) / slots;
        return nw < 4 ? 4 : (nw > 8 ? 8 : nw);
    };
    {
        Scope sc(K_FC_DW, s);
        auto launch = [&](auto kern, int steps) {
            const size_t lds = (size_t)2 * steps * (N + 1) * 4;
            const int nw = waves_for(lds <= 80 * 1024 ? 512 : 256);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3((unsigned)((K + 32 * nw - 1) / (32 * nw))), dim3(64 * nw), lds, s, x, dz, M, K, N, dW);
        };
        if (M <= 32) launch(fc_dw_kernel<16>, 16);
        else if (M <= 64) launch(fc_dw_kernel<32>, 32);
        else launch(fc_dw_kernel<64>, 64);
    }
    TRY(hip_ok());
    if (dx != nullptr) {
        Scope sc(K_FC_DX, s);
        const int psteps = (N / 8 + 7) / 8 * 8;                            // (fc_dx_kernel's kD)
        size_t lds = (size_t)32 * (8 * psteps + 4) * 4;                   // the dz tile; the transpose tiles reuse it
        const int nw = waves_for(lds <= 80 * 1024 ? 512 : 256);
        if (lds < (size_t)nw * 32 * 33 * 4) lds = (size_t)nw * 32 * 33 * 4;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fc_dx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(fc_dx_kernel, dim3((unsigned)((K + 32 * nw - 1) / (32 * nw)), (unsigned)p.mblocks), dim3(64 * nw), lds, s, dz, W,
                           M, K, N, dx);
    }
    return hip_ok();
}

int conv3p_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
    return CONV3P_OK;
}
int conv3p_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (auto &r : g_prof.recs) {
        g_prof.pool.push_back(r.a);
        g_prof.pool.push_back(r.b);
    }
    g_prof.recs.clear();
    return CONV3P_OK;
}
int conv3p_profile_kinds(void) { return K_NKINDS; }
const char *conv3p_profile_name(int kind) { return kind >= 0 && kind < K_NKINDS ? kKindName[kind] : ""; }
int conv3p_profile_read(int kind, uint64_t *launches, double *total_ms)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    uint64_t n = 0;
    double ms = 0.0;
    for (auto &r : g_prof.recs) {
        if (r.kind != kind) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return CONV3P_ERR_LAUNCH;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return CONV3P_ERR_LAUNCH;
        ms += t;
        ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return CONV3P_OK;
}

const char *conv3p_status_string(int status)
{
    switch (status) {
    case CONV3P_OK: return "ok";
    case CONV3P_ERR_INVALID_ARGUMENT: return "invalid argument";
    case CONV3P_ERR_WORKSPACE: return "workspace / cache null, misaligned or too small";
    case CONV3P_ERR_UNSUPPORTED: return "unsupported configuration";
    case CONV3P_ERR_LAUNCH: return "HIP launch error";
    case CONV3P_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
    }
}
int conv3p_abi_version(void) { return CONV3P_ABI_VERSION; }

}  // extern "C"
