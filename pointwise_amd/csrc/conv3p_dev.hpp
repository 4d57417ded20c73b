// conv3p_dev.hpp -- the developer instruments of the kernels, in ONE place (round 6; verdict r5 item 7).
//
// Product builds define none of the switches below: every macro of this file then expands to nothing and every `kDev...`
// test is a compile-time false, so the kernels read (and compile) as if the instruments were not there.  Instrumentation /
// ablation builds go to devlibs/ (tools/build_variants.sh) and are loaded through CONV3P_HIP_LIB; bench.py echoes that.
//
//   -DCONV3P_ABLATE=<bits>       register path and deep path
//        1, 2, 4        backward_kernel: no phase A walk / no phase B / no phase C            (tools/ablate.sh)
//        16, 256, 512   backward_kernel phase A: no read-modify-write variants (timing only)
//        32, 64         search_tile: empty pre-filter masks / no exact stage
//        1024, 2048     forward / backward gathers redirected to L1-resident rows (timing only)
//        65536, 131072, 262144   deep_gemm: no stage 2 / no stage 1 walk / no G store
//        16777216       deep_gemm: per-phase stamps printed by every 400th workgroup
//        33554432       backward_kernel: per-phase stamps (tools/bw_trace.py)
//        134217728      forward_tile: per-phase stamps (tools/phase_trace.py, tools/fused_trace.py)
//   -DCONV3P_SP_ABLATE=<bits>    backward_sparse_tile: 1 / 2 / 4 no phase A / B / C, 32 / 64 / 2048 gathers from L1, 256 / 512 /
//        1024 no read-modify-write, 128 per-phase stamps, 4096 stack_backward_kernel: per-layer spin / tile / arrive
//        stamps, printed once at the end of the kernel (conv3p_stack_fused.hpp; tools/fused_trace.py --backward)
//   -DCONV3P_DEV_FUSED_ABLATE=<bits>   search_fused_kernel (conv3p_search_fused.hpp)
// Stamps are 10-ns ticks of wall_clock64() taken after an s_waitcnt(0), printed by lane 0 of every wave of every 211th
// workgroup.
#pragma once

#ifndef CONV3P_ABLATE
#define CONV3P_ABLATE 0
#endif
#ifndef CONV3P_SP_ABLATE
#define CONV3P_SP_ABLATE 0
#endif

// ---- forward_tile (conv3p_kernels.hpp)
#if CONV3P_ABLATE & 134217728
#define DEV_FWD_DECL() long long ft[8]; int fti = 0, fsteps = 0;
#define FDBG() { __builtin_amdgcn_s_waitcnt(0); ft[fti++] = wall_clock64(); }
#define DEV_FWD_STEP() fsteps++;
#define DEV_FWD_PRINT(CIN_, COUT_, wave_, lane_)                                                                                   \
    FDBG()                                                                                                                         \
    if ((lane_) == 0 && (blockIdx.x % 211) == 7)                                                                                   \
        printf("fwd<%d,%d> wg %d wave %d: loads %lld sync %lld table %lld qsegs %lld first-recs %lld loop %lld (%d steps) epilogue %lld\n", \
               CIN_, COUT_, (int)blockIdx.x, wave_, ft[1] - ft[0], ft[2] - ft[1], ft[3] - ft[2], ft[4] - ft[3], ft[5] - ft[4],     \
               ft[6] - ft[5], fsteps, ft[7] - ft[6]);
#else
#define DEV_FWD_DECL()
#define FDBG()
#define DEV_FWD_STEP()
#define DEV_FWD_PRINT(CIN_, COUT_, wave_, lane_)
#endif

// ---- backward_kernel (conv3p_kernels.hpp)
#if CONV3P_ABLATE & 33554432
#define DEV_BWD_DECL() long long bt[8]; int bti = 0;
#define BDBG() { __builtin_amdgcn_s_waitcnt(0); bt[bti++] = wall_clock64(); }
#define DEV_BWD_PRINT(CIN_, COUT_, wave_, lane_)                                                                                   \
    BDBG()                                                                                                                         \
    if ((lane_) == 0 && (blockIdx.x % 211) == 7)                                                                                   \
        printf("bwd<%d,%d> wg %d wave %d: prologue %lld  phaseA %lld  sync %lld  B %lld  C %lld  reduce+store %lld\n", CIN_, COUT_, \
               (int)blockIdx.x, wave_, bt[1] - bt[0], bt[2] - bt[1], bt[3] - bt[2], bt[4] - bt[3], bt[5] - bt[4], bt[6] - bt[5]);
#else
#define DEV_BWD_DECL()
#define BDBG()
#define DEV_BWD_PRINT(CIN_, COUT_, wave_, lane_)
#endif

// ---- backward_sparse_tile (conv3p_backward_sparse.hpp): stamps of the one-round path (st_), accumulated phase times of the
//      multi-round paths (ra_ / rb_ / rc_ / rs_)
#if CONV3P_SP_ABLATE & 128
#define DEV_SP_DECL() long long st_[10]; int sti_ = 0;
#define SDBG() { __builtin_amdgcn_s_waitcnt(0); st_[sti_++] = wall_clock64(); }
#define DEV_SP_ROUNDS_DECL() long long ra_ = 0, rb_ = 0, rc_ = 0, rs_ = 0, t_; int nrs_ = 0; __builtin_amdgcn_s_waitcnt(0); t_ = wall_clock64();
#define DEV_SP_ACC(v) { __builtin_amdgcn_s_waitcnt(0); const long long n_ = wall_clock64(); v += n_ - t_; t_ = n_; }
#define DEV_SP_ROUND() ++nrs_;
#define DEV_SP_ROUNDS_PRINT(CIN_, COUT_, wave_, lane_, rounds_)                                                                    \
    if ((lane_) == 0 && (blockIdx.x % 211) == 7)                                                                                   \
        printf("bsp<%d,%d> wg %d wave %d: rounds %d  prologue %lld  syncs %lld  phaseA %lld  B %lld  C %lld\n", CIN_, COUT_,      \
               (int)blockIdx.x, wave_, rounds_, st_[1] - st_[0], rs_, ra_, rb_, rc_);
#define DEV_SP_PRINT(CIN_, COUT_, wave_, lane_, one_round_)                                                                        \
    SDBG()                                                                                                                         \
    if ((lane_) == 0 && (one_round_) && (blockIdx.x % 211) == 7)                                                                   \
        printf("bsp<%d,%d> wg %d wave %d: prologue %lld  zero+sync %lld  phaseA %lld  sync %lld  B %lld  C %lld  reduce+store %lld\n", CIN_, COUT_, \
               (int)blockIdx.x, wave_, st_[1] - st_[0], st_[2] - st_[1], st_[3] - st_[2], st_[4] - st_[3], st_[5] - st_[4],        \
               st_[6] - st_[5], st_[7] - st_[6]);
#else
#define DEV_SP_DECL()
#define SDBG()
#define DEV_SP_ROUNDS_DECL()
#define DEV_SP_ACC(v)
#define DEV_SP_ROUND()
#define DEV_SP_ROUNDS_PRINT(CIN_, COUT_, wave_, lane_, rounds_)
#define DEV_SP_PRINT(CIN_, COUT_, wave_, lane_, one_round_)
#endif

// ---- deep_gemm_kernel (conv3p_deep.hpp)
#if CONV3P_ABLATE & 16777216
#define DEV_GEMM_DECL() long long gk[6] = {0, 0, 0, 0, 0, 0}; long long gstart = wall_clock64(); long long glast = wall_clock64();
#define GDBG(i) { const long long t_ = wall_clock64(); gk[i] += t_ - glast; glast = t_; }
#define DEV_GEMM_PRINT(KDIM_, NDIM_, BWD_, wave_, lane_, taps_)                                                                    \
    if ((lane_) == 0 && (blockIdx.x % 400) == 100)                                                                                 \
        printf("gemm<%d,%d,%d> wg %d wave %d dbg (10 ns ticks): total %lld | stage 1 (+ barriers) %lld  gbuf + stage 2 %lld  tail %lld | taps %d\n", \
               KDIM_, NDIM_, (int)(BWD_), (int)blockIdx.x, wave_, wall_clock64() - gstart, gk[0], gk[4], gk[5], taps_);
#else
#define DEV_GEMM_DECL()
#define GDBG(i)
#define DEV_GEMM_PRINT(KDIM_, NDIM_, BWD_, wave_, lane_, taps_)
#endif
