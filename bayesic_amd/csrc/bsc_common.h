// Internal helpers shared by the kernels of libbayesic_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <utility>
#include <vector>

#include "../../include/bayesic_hip.h"

constexpr int BSC_PROF_SLOTS = 3;
constexpr int BSC_EXCHANGE_SLOTS = 16;

struct bsc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void* workspace = nullptr;   // partial-sum slabs; grown on demand
    size_t workspace_bytes = 0;
    int cu_count = 256;
    int blr_tile_rows = 16;      // 16: forward on the MFMA pipe when D == 256 (else 8-row VALU tiles); 8 | 4: VALU variants
    int blr_waves_per_simd = 0;  // tuning: cap resident waves per SIMD (0 = kernel's own limit)
    int fused_map_blocks_per_cu = 8;  // bsc_map_reduce, pure map: grid cap (256-thread blocks)
    int gram_pp = 0;                  // 1: gram256_pp_kernel (the two waves of a SIMD take turns on the matrix pipe: two half-steps, two barriers a step) instead of gram256_bx_kernel -- measured no faster (267 vs 261 us), off
    int gram_dbg = 0;                 // deletion builds of gram256_bx_kernel (BSC_GRAM_DBG + BSC_PROFILING_BUILDS): WRONG results
    int rows_dbg = 0;                 // deletion builds of map_reduce_rows_f32_kernel (BSC_ROWS_DBG + BSC_PROFILING_BUILDS): WRONG results
    int rows_wg_per_cu = 64;          // map_reduce_rows_f32_kernel: workgroups per CU in the grid, eight of them resident (BSC_ROWS_WG; 0 = one step per wave)
    int skinny_nt_dbg = 0;            // deletion builds of gemm_skinny_nt_kernel (BSC_SKINNY_NT_DBG + BSC_PROFILING_BUILDS): WRONG results
    int skinny_nt_wg_per_cu = 1;      // gemm_skinny_nt_kernel: resident workgroups (4 waves, 64 KiB of rings each) per CU (BSC_SKINNY_NT_WG)
    int gemm_nt_c = 1;                // stream GEMM: results of 128 MiB and more leave by non-temporal stores (BSC_GEMM_NT_C=0 for A/B)
    int fused_map_flat = 1;           // pure maps with a contiguous float32 result: map_flat_f32_kernel (BSC_FUSED_MAP_FLAT=0: the round-1/2 kernels, for A/B)
    int fused_map_unroll = 2;         // float4 per operand in flight per lane (1 | 2); 2 is +18% measured
    int gemm_fast = 1;                // GEMM: scalar-base loads for interior tiles (BSC_GEMM_FAST=0 turns them off)
    int gemm_skinny = 1;              // GEMM: the LDS-DMA kernels of csrc/bsc_skinny.hip for products with one tiny extent (BSC_GEMM_SKINNY=0 turns them off)
    int gemm_dma = 2;                 // GEMM: 2 = persistent stream-K on LDS-DMA operand tiles (gemm_f32_stream_kernel), 1 = one tile per workgroup with LDS-DMA operands (gemm_f32_dma_kernel), 0 = the register-staged kernel (BSC_GEMM_DMA)
    int lda_stream = 1;               // LDA statistic, K = 128: the persistent LDS-DMA kernel (BSC_LDA_STREAM=0: one column block per workgroup, register staging)
    int gemm_sym = 1;                 // GEMM: X^T X computes the tiles on and above the diagonal only (BSC_GEMM_SYM=0: all of them)
    int gemm_dbg = 0;                 // profiling only (BSC_GEMM_DBG): 1 = the stream kernel drops its whole-tile stores
    int gemm_pipe = 1;                // GEMM: LDS operand reads one k-pair ahead of the MFMAs
    int fused_nt_store = 1;           // dense map: non-temporal stores of the result
    int bbvi_waves = 4;          // bsc_logreg_bbvi_loglik: waves per workgroup (4: one wave per 16 samples, 8: per (16 samples, 16 rows))
    int bbvi_kernel = 1;         // bsc_logreg_bbvi_loglik: 1 = draws in LDS, X by LDS-DMA strips (S <= 64; S <= 128 X through VGPRs), 2 = X through VGPRs, 0 = first-generation LDS-staged tiles (S == 64)
    int lda_dbg = 0;             // BSC_LDA_DBG: profiling-only deletion builds of the split-operand LDA kernel (1: no DMAs after the first two steps; 2: no arithmetic) -- wrong results
    int bbvi_dbg = 0;            // BSC_BBVI_DBG: profiling-only deletion builds of the xreg kernel (wrong results)
    int csc_fast = 1;            // bsc_lda_sstats_csc: buffer-descriptor gathers (BSC_CSC_FAST=0 turns them off)
    int mfma_split = 0;          // 0: f32 MFMA (exact f32 products; the default and the dtype of every reported line); 2 / 3: f32 operands as sums of two / three bf16 terms on the bf16 MFMA, 3 / 6 products (csrc/bsc_bf16split.h; BSC_MFMA_SPLIT, bsc_set_mfma_split) where a kernel offers it
    int mog_nt = 0;              // bsc_mog_estep: 1 = non-temporal loads of X (BSC_MOG_NT).  Default 0: both half-waves read the same rows and the L2 keeps a row for the second one -- 1.01 x the algorithmic bytes instead of 1.19 x, same time (profiles/r03_pmc_kernels.txt)
    int wo_wg_per_cu = 2;        // bsc_weighted_outer: resident workgroups per CU the grid is sized for
    int fused_waves_per_cu = 16; // bsc_map_reduce: reduce splits target this many waves per CU
    int blr_dma = 1;             // blr_pass_dma_kernel (the tile by LDS-DMA: 166 -> 161 us at 1M x 256) -- BSC_BLR_DMA=0: blr_pass_mfma_kernel (tile through registers), for A/B
    int blr_q = 1;               // blr_pass_q_kernel (both contractions on v_mfma_f32_4x4x1, the tile by LDS-DMA; round 4) -- BSC_BLR_Q=0: blr_pass_dma_kernel, for A/B
    int blr_q_dbg = 0;           // deletion builds of blr_pass_q_kernel (BSC_BLR_Q_DBG + BSC_PROFILING_BUILDS): WRONG results
    int blr_q_bias = 70;         // blr_pass_q_kernel, static schedule: per mille of further windows for the workgroups with an even blockIdx (QSched)
    int blr_q_prio = 1;          // blr_pass_q_kernel: s_setprio 1 from a tile's landing to the next tile's DMAs (1), or during the backward (2)
    int blr_fold = 0;            // 1: blr_pass_q_kernel carries its finish (or the float64 statistics of the N > 1 structure) in its tail -- one launch per update (FoldArgs).  Built and measured in round 4: break-even (162.9-163.1 vs 162.5-162.9 us per 1M-row update, 36.4 vs 35.3-36.4 at 125k rows, profiles/r04_fold_*.txt), so off by default
    unsigned* fold_counters = nullptr;   // [arrivals, roles done] of the folded finish: zero between launches
    int blr_steal = 0;           // blr_pass_q_kernel, streaming sweep: per mille of the tiles left to the queues (StealArgs) the waves draw on after their static share.  Built and measured in round 4 (profiles/r04_ab_pass_q_steal.txt): every workgroup then ends within ~4 us of every other, but the launch gains 0.5-1.5 us of 157 at 1M rows, 2-2.7 of 85 at 500k, loses 0.5 of 28.7 at 125k -- the static schedule's early finishers leave their bandwidth to the late ones, so little was lost -- and the last bits of the sums are no longer reproducible: off by default
    unsigned* steal_heads = nullptr;     // 64 queue heads 256 bytes apart + the workgroups-done counter (inside fold_counters' allocation): zero between launches
    int blr_stamps = 0;          // blr_pass_q_kernel: every workgroup leaves start / end s_memrealtime stamps and its XCD (bsc_blr_read_stamps)
    void* stamps = nullptr;      // 32 bytes per workgroup, allocated when blr_stamps is first used
    int stamp_rows = 0;          // workgroups of the last stamped launch
    int profiling_builds = 0;    // 1: the deletion builds (options marked dbg) may be selected -- WRONG results, timing only
    int blr_pk = 1;              // MFMA pass: backward rank-1 updates as packed FMAs (BSC_BLR_PK=0: scalar; +0.4 % in-process A/B, same bits)
    int blr_finish_block = 1024; // threads per workgroup of blr_fused_update_kernel (BSC_BLR_FINISH_BLOCK = 256 | 512 | 1024)
    int blr_nt_loads = 1;        // non-temporal loads of X (read once per pass): +9% measured
    int blr_keep = -1;           // windows a keeping sweep leaves in the Infinity Cache (BSC_BLR_KEEP; -1 = what fits 256 MiB)
    int blr_mx = 0;              // BSC_BLR_MX: 1 = blr_pass_mx_kernel (backward on 4x4x1 MFMA) with the rotated cached-zone schedule, 2 = that kernel with the plain sweep orders
    int blr_wide = 1;            // BSC_BLR_WIDE: S > 8 runs sixteen draws per pass (blr_pass_mx_kernel<., 4>); 0 = eight per pass
    int blr_rot = 2;             // BSC_BLR_ROT: workgroups per rotation group = 2^rot
    int slab_rows = 0;  // block partials left in `workspace` by bsc_blr_data_pass_partial
    int capturing = 0;  // between bsc_capture_begin and bsc_capture_end: launches are recorded, not run
    // optional per-kernel timing of the dominant kernel (bsc_ctx_profile)
    int profile = 0;        // 0 = off, n = time every n-th launch of a dominant kernel
    // slot 0: the dominant kernel of an entry point; slot 1: the collective; slot 2: the finish kernel
    int profile_tick[BSC_PROF_SLOTS] = {0, 0, 0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events[BSC_PROF_SLOTS];  // recorded pairs
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;                    // reusable pairs
    // data-parallel exchange (csrc/bsc_comm.hip): an ncclComm_t, NULL = a world of one
    void* comm = nullptr;
    int comm_rank = 0;
    int comm_world = 1;
    // the overlapped exchange (bsc_allreduce_sum_begin / _end): collectives of finished pieces of a statistic run on a
    // second stream while the kernels of the next piece run on `stream`; one event pair per slot, created on first use
    hipStream_t comm_stream = nullptr;
    hipEvent_t xch_ready[BSC_EXCHANGE_SLOTS] = {};
    hipEvent_t xch_done[BSC_EXCHANGE_SLOTS] = {};
    int xch_pending[BSC_EXCHANGE_SLOTS] = {};
};

// Records an event pair around one launch when ctx->profile is on.
#if defined(__HIPCC__)
// digamma in float64: recurrence up to x >= 8, then the asymptotic series
//   ln x - 1/2x - 1/12x^2 + 1/120x^4 - 1/252x^6 + 1/240x^8 - 5/660x^10 + 691/32760x^12
// (absolute error < 1e-15 there).  Shared by the Dirichlet / Normal-Gamma
// expectations and the element-wise digamma of the executor.
__device__ inline double bsc_digamma_f64(double x) {
#pragma clang fp contract(off)
    // shift x up to >= 8: psi(x) = psi(x + n) - sum_{i<n} 1/(x + i).  The sum is P'(x)/P(x) for
    // P = prod (x + i), built with multiplies and adds, so it costs ONE division instead of n
    // (float64 division is ~10x a multiply; this halves bsc_dirichlet_expectation)
    double P = 1.0, dP = 0.0;
    while (x < 8.0) {
        dP = dP * x + P;
        P *= x;
        x += 1.0;
    }
    const double acc = -dP / P;
    const double inv = 1.0 / x, inv2 = inv * inv;
    const double series = inv2 * (1.0 / 12.0 - inv2 * (1.0 / 120.0 - inv2 * (1.0 / 252.0 - inv2 *
                          (1.0 / 240.0 - inv2 * (5.0 / 660.0 - inv2 * (691.0 / 32760.0))))));
    return acc + log(x) - 0.5 * inv - series;
}

// lnGamma in float64 by the same route: lnGamma(x) = lnGamma(x + n) - log prod_{i<n} (x + i), x + n >= 8,
// Stirling's series there (next term 1 / (156 y^13) < 2e-14 at y = 8).  x > 0.
__device__ inline double bsc_lgamma_f64(double x) {
#pragma clang fp contract(off)
    double P = 1.0;
    while (x < 8.0) {
        P *= x;
        x += 1.0;
    }
    const double inv = 1.0 / x, inv2 = inv * inv;
    const double series = inv * (1.0 / 12.0 - inv2 * (1.0 / 360.0 - inv2 * (1.0 / 1260.0 - inv2 *
                          (1.0 / 1680.0 - inv2 * (1.0 / 1188.0 - inv2 * (691.0 / 360360.0))))));
    return (x - 0.5) * log(x) - x + 0.91893853320467274178032973640562 + series - log(P);
}
#endif

struct bsc_prof_scope {
    bsc_ctx* ctx;
    int slot;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    hipStream_t on;      // the stream the timed launch goes to (the context's own unless said otherwise)
    explicit bsc_prof_scope(bsc_ctx* c, int slot_ = 0, hipStream_t on_ = nullptr) : ctx(c), slot(slot_), on(on_ ? on_ : c->stream) {
        if (ctx->profile <= 0 || ctx->capturing) return;
        if ((ctx->profile_tick[slot]++ % ctx->profile) != 0) return;
        if (!ctx->prof_pool.empty()) {
            ev = ctx->prof_pool.back();
            ctx->prof_pool.pop_back();
        } else {
            (void)hipEventCreate(&ev.first);
            (void)hipEventCreate(&ev.second);
        }
        (void)hipEventRecord(ev.first, on);
    }
    ~bsc_prof_scope() {
        if (!ev.first) return;
        (void)hipEventRecord(ev.second, on);
        ctx->prof_events[slot].push_back(ev);
    }
};

// thread-local error message; returns `code` so callers can `return bsc_fail(...)`
int bsc_fail(int code, const char* fmt, ...);

// Make sure ctx->workspace holds at least `bytes`; may hipMalloc (synchronous).
int bsc_workspace(bsc_ctx* ctx, size_t bytes, void** out);

// csrc/bsc_skinny.hip: float32 products with one tiny extent that stream their large operand once
// through LDS-DMA.  Sets *handled = 1 when it took the product (then C is written / enqueued).
int bsc_gemm_skinny(bsc_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k,
                    const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t sc_m, int64_t sc_n, int* handled);

#define BSC_HIP(call)                                                            \
    do {                                                                         \
        hipError_t err__ = (call);                                               \
        if (err__ != hipSuccess)                                                 \
            return bsc_fail(BSC_ERR_HIP, "%s failed: %s (%s:%d)", #call,         \
                            hipGetErrorString(err__), __FILE__, __LINE__);       \
    } while (0)

#define BSC_REQUIRE(cond, ...)                                                   \
    do {                                                                         \
        if (!(cond)) return bsc_fail(BSC_ERR_INVALID, __VA_ARGS__);              \
    } while (0)

#define BSC_CHECK_CTX(ctx)                                                       \
    do {                                                                         \
        if ((ctx) == nullptr) return bsc_fail(BSC_ERR_INVALID, "null bsc_ctx");  \
        BSC_HIP(hipSetDevice((ctx)->device));                                    \
    } while (0)

#define BSC_LAUNCH_CHECK()                                                       \
    do {                                                                         \
        hipError_t err__ = hipGetLastError();                                    \
        if (err__ != hipSuccess)                                                 \
            return bsc_fail(BSC_ERR_HIP, "kernel launch failed: %s (%s:%d)",     \
                            hipGetErrorString(err__), __FILE__, __LINE__);       \
    } while (0)

constexpr int BSC_WAVE = 64;

// ---- wave-level helpers ---------------------------------------------------

// DPP controls (cdna4 ISA): quad_perm encodes the source lane of each quad lane.
constexpr int DPP_QUAD_XOR1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_ROR4 = 0x124;
constexpr int DPP_ROW_ROR8 = 0x128;

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// Sum over the 16 lanes of a DPP row, result in every lane of the row.
// Rotation direction does not matter: each stage is applied to a value that is
// already periodic with the stage's period.
__device__ __forceinline__ float row16_allsum(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    v += dpp_f32<DPP_ROW_ROR4>(v);
    v += dpp_f32<DPP_ROW_ROR8>(v);
    return v;
}

// Full 64-lane sum, result in every lane.
__device__ __forceinline__ float wave_allsum(float v) {
    v = row16_allsum(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}


// FOUR 64-lane float64 sums at once, without the LDS crossbar: the 16 lanes of DPP row q end with the sum of a_q over
// the wave.  v_permlane32_swap / v_permlane16_swap halve the lanes a value occupies while two values share a register
// pair (lanes 0..31 fold a0 resp. a1 over (l, l + 32), lanes 32..63 a2 resp. a3; then rows 0..3 fold (l, l + 16)),
// the last four steps are DPP rotations inside a row: 21 vector instructions for four sums, where four butterflies
// of __shfl_xor are 48 ds_bpermute_b32 in chains of six LDS round trips.  Fixed order: reproducible.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double swap_add32_f64(double a, double b) {
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add16_f64(double a, double b) {
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// One 64-lane float64 sum, result in every lane: DPP inside the rows, then the rows by the two swaps (a value swapped
// with itself: rows 0 / 1 and 2 / 3 exchange, then the halves) -- 22 vector instructions and no LDS round trip where
// six steps of __shfl_xor were twelve ds_bpermute_b32 in a dependent chain.
__device__ __forceinline__ double wave_allsum_f64(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_ROW_ROR4>(v);
    v += dpp_f64<DPP_ROW_ROR8>(v);
    v = swap_add16_f64(v, v);
    return swap_add32_f64(v, v);
}
__device__ __forceinline__ double wave_allsum4_f64(double a0, double a1, double a2, double a3) {
    double t = swap_add16_f64(swap_add32_f64(a0, a2), swap_add32_f64(a1, a3));
    t += dpp_f64<DPP_QUAD_XOR1>(t);
    t += dpp_f64<DPP_QUAD_XOR2>(t);
    t += dpp_f64<DPP_ROW_ROR4>(t);
    t += dpp_f64<DPP_ROW_ROR8>(t);
    return t;
}

// Orders this wave's LDS writes before its later LDS reads of other lanes' data
// (wave-private LDS regions, no workgroup barrier).  LDS operations of one wave
// execute in order; this only stops the compiler from moving them.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- LDS-DMA helpers (csrc/bsc_bbvi.hip, csrc/bsc_skinny.hip) -------------------------------
typedef __attribute__((address_space(3))) void* bsc_lds_ptr;

// s_waitcnt immediate (gfx9 layout) that waits for vmcnt <= n only: vmcnt = [3:0] and [15:14],
// expcnt [6:4] and lgkmcnt [11:8] left at "no wait"
constexpr int bsc_vmcnt_only(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }
constexpr int BSC_LGKMCNT0 = 0xC07F;   // lgkmcnt(0) alone

// A descriptor over the rows of a row-major matrix from `row0` on (`rows` in all, `ld` floats
// apart, `width` floats wide): everything past the last row reads as zero.
__device__ __forceinline__ auto bsc_rows_rsrc(const float* X, int64_t ld, int width, int64_t rows, int64_t row0) {
    const int64_t rem = rows - row0;
    uint64_t xb = 0;
    if (rem > 0) xb = ((uint64_t)(rem - 1) * (uint64_t)ld + (uint64_t)width) * 4u;
    const unsigned rec = xb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)xb;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(X + (rem > 0 ? row0 : 0) * ld), 0, rec, 0x00020000);
}
// ... over a vector of 4-byte elements (one per row) from element `row0` on
__device__ __forceinline__ auto bsc_vec_rsrc(const void* base, int64_t n, int64_t row0) {
    const int64_t rem = n - row0;
    const uint64_t b = rem > 0 ? (uint64_t)rem * 4u : 0;
    const unsigned rec = b > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)b;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + (rem > 0 ? row0 : 0) * 4), 0, rec, 0x00020000);
}

__device__ __forceinline__ float readlane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
