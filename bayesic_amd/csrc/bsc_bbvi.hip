// Black-box VI for hierarchical logistic regression (BASELINE config 5).
//
// ABSENT in the reference: spec is README.md:52 ("the gradient estimator from
// Black box variational inference [3] -- ... the control variate one") with
// README.md:69-79 for mini-batching; in bayesic.algebra terms the data-sized
// contraction is dot(W, X.T) (cfg 5).
//
// logreg_loglik_kernel: ONE read of X[N,D], y[N], g[N] gives, for S = 64 Monte
// Carlo draws of (w, b) at once,
//     l_ns  = x_n . Wz[s] + Bz[g_n, s]
//     ell_s = sum_n ( y_n l_ns - softplus(l_ns) )
// 32 flop/B: close to the HBM/FP32 ridge, so the contraction runs on
// v_mfma_f32_16x16x4_f32 (exact f32) and everything else is cheap.  A workgroup
// (4 waves) owns 32-row tiles staged in LDS by coalesced 1-KiB row loads; wave w
// owns samples 16w..16w+15 and keeps its slice of Wz in 64 registers as the MFMA
// B operand for the whole kernel.  The k order inside a contraction is free, so
// k-step s of lane group k reads column 16(s/4) + 4k + (s%4): one ds_read_b128
// feeds four MFMAs.  LDS row stride 264 floats makes those reads conflict-free.
#include "bsc_common.h"
#include "bsc_bf16split.h"

namespace {

constexpr int LS = 64;            // samples
constexpr int LD = 256;           // column capacity
constexpr int LT = 32;            // rows per tile
constexpr int LSTR = LD + 4;      // LDS row stride (floats): row r starts on bank 4 r, so 16 rows x 16 B cover the 64 banks once
constexpr int LR_BLOCK = 256;
constexpr int TILE_FLOATS = LT * LSTR + LT;   // rows + y

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int RW>
struct Stage {   // one wave's share of a tile in registers: RW rows x 16 B per lane
    float4 x[RW];
    float yv;
};

template <bool FULL, int RW>
__device__ __forceinline__ void stage_load(Stage<RW>& st, const float* __restrict__ X, int64_t ldx,
                                           const float* __restrict__ y, const int* __restrict__ g,
                                           int64_t row0, int64_t N, int D, int wave, int lane) {
    const int64_t rem = N - row0;
    uint64_t xb = 0, yb = 0;
    if (rem > 0) {
        xb = ((uint64_t)(rem - 1) * (uint64_t)ldx + (uint64_t)D) * 4u;
        yb = (uint64_t)rem * 4u;
    }
    const unsigned xrec = xb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)xb;
    const unsigned yrec = yb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)yb;
    const int64_t safe0 = rem > 0 ? row0 : 0;
    auto xs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + safe0 * ldx), 0, xrec, 0x00020000);
    auto ys = __builtin_amdgcn_make_buffer_rsrc((void*)(y + safe0), 0, yrec, 0x00020000);
    const int row_bytes = (int)(ldx * 4);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        auto v = __builtin_amdgcn_raw_buffer_load_b128(xs, 16 * lane, (RW * wave + r) * row_bytes, 2);  // nt
        float4 f = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                               __uint_as_float(v[3]));
        if (!FULL && 4 * lane >= D) f = make_float4(0.f, 0.f, 0.f, 0.f);
        st.x[r] = f;
    }
    // wave 0 also brings the tile's y
    st.yv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ys, 4 * (lane & 31), 0, 0));
}

// Byte offsets (group id * 256) into Bz of the rows whose logits this lane ends up holding
// (MFMA result rows 16 (rb0 + rb) + 4 kq + r).  No clamping: the gather below is a buffer
// load, so an id outside [0, n_groups) reads an intercept of 0 instead of faulting, and rows
// past N read id 0 (their result is masked).
template <int RB>
__device__ __forceinline__ void load_groups(int (&gi)[4 * RB], const int* __restrict__ g, int64_t row0,
                                            int64_t N, int kq, int rb0) {
    const int64_t rem = N - row0;
    const uint64_t gb = rem > 0 ? (uint64_t)rem * 4u : 0;
    const unsigned grec = gb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)gb;
    auto gs = __builtin_amdgcn_make_buffer_rsrc((void*)(g + (rem > 0 ? row0 : 0)), 0, grec, 0x00020000);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        auto v = __builtin_amdgcn_raw_buffer_load_b128(gs, 4 * (16 * (rb0 + rb) + 4 * kq), 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) gi[4 * rb + r] = (int)v[r] * (LS * 4);
    }
}

template <int RW>
__device__ __forceinline__ void stage_store(const Stage<RW>& st, float* tile, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < RW; ++r)
        *reinterpret_cast<float4*>(tile + (RW * wave + r) * LSTR + 4 * lane) = st.x[r];
    if (wave == 0 && lane < 32) tile[LT * LSTR + lane] = st.yv;
}

// NW = 4: a wave owns 16 samples for both 16-row blocks of the tile (two accumulators).
// NW = 8: a wave owns 16 samples for ONE row block -- twice the waves per SIMD to cover the
// softplus epilogue, the LDS stores and the barrier of the others.
template <bool FULL, int NW>   // FULL: D == 256, no column masking
__global__ __launch_bounds__(64 * NW, 2) void logreg_loglik_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y,
    const int* __restrict__ g, int64_t N, int D, const float* __restrict__ Wz,
    const float* __restrict__ Bz, int n_groups, float* __restrict__ slab, int n_iter) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    constexpr int RB = 8 / NW;          // row blocks per wave
    constexpr int RW = LT / NW;         // rows a wave stages per tile
    const int sb = wave & 3;            // sample block
    const int rb0 = NW == 8 ? (wave >> 2) : 0;

    // B operand: Wz[sample 16*sb + i16][column of (k-step s, lane group kq)]
    float wreg[LD / 4];
#pragma unroll
    for (int s = 0; s < LD / 4; ++s) {
        // lane group kq contracts columns [64 kq, 64 kq + 64): the four 16-byte A reads of a
        // row then sit 256 B apart, on the same banks, and every ds_read_b128 lane group
        // (each holds all 16 rows once, with two different kq) is conflict-free; with the
        // columns interleaved (4 kq + 16 q) each lane group had one 2-way conflict
        const int col = 64 * kq + s;
        wreg[s] = col < D ? Wz[(int64_t)(16 * sb + i16) * D + col] : 0.f;
    }
    double acc_ll = 0.0;   // per-tile float32 sums enter a float64 accumulator: no drift over the 60 tiles

    int64_t tile = blockIdx.x;
    const int64_t stride = gridDim.x;
    Stage<RW> st;
    stage_load<FULL>(st, X, ldx, y, g, tile * LT, N, D, wave, lane);
    stage_store(st, lds, wave, lane);
    // The intercept b[g_n, s] is a gather that depends on the row's group id: requested
    // when it is needed it costs a full memory round trip per tile with the MFMA pipe idle.
    // So group ids run two tiles ahead and intercepts one tile ahead, in registers.
    // Bz[g, s] gathered through a buffer descriptor: 32-bit offsets, out-of-range ids read 0
    const auto bz_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, (unsigned)n_groups * (LS * 4u), 0x00020000);
    const int bz_off = 4 * (16 * sb + i16);
    auto bz_load = [&](int goff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bz_rsrc, goff + bz_off, 0, 0));
    };
    int gi[4 * RB];
    float bz_a[4 * RB], bz_b[4 * RB];
    load_groups<RB>(gi, g, tile * LT, N, kq, rb0);
#pragma unroll
    for (int e = 0; e < 4 * RB; ++e) bz_a[e] = bz_load(gi[e]);
    load_groups<RB>(gi, g, (tile + stride) * LT, N, kq, rb0);
    __syncthreads();
    int cur = 0;
    // One tile.  bz_cur holds this tile's intercepts (requested a tile ago), bz_next
    // receives the next tile's; the caller alternates the two register sets so that no
    // copy (which would wait for the loads straight away) is needed.
    auto one_tile = [&](const float (&bz_cur)[4 * RB], float (&bz_next)[4 * RB]) {
        stage_load<FULL>(st, X, ldx, y, g, (tile + stride) * LT, N, D, wave, lane);   // prefetch
        const float* t = lds + cur * TILE_FLOATS;
        f32x4 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A operands one k-group ahead in registers: the MFMAs of group q cover the LDS
        // latency of group q+1.  (A single dependent accumulator chain runs at the full
        // 32-cycle rate -- tools/ubench_mfma_mix.hip -- so the row blocks need not alternate
        // for the pipe's sake; they do because one A read then feeds four MFMAs.)
        const float* ta = t + (16 * rb0 + i16) * LSTR + 64 * kq;
        float4 an[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) an[rb] = *reinterpret_cast<const float4*>(ta + 16 * rb * LSTR);
        __builtin_amdgcn_sched_group_barrier(0x100, RB, 0);       // the reads of group 0
#pragma unroll
        for (int q = 0; q < LD / 16; ++q) {
            float4 a[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[rb] = an[rb];
            if (q + 1 < LD / 16) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    an[rb] = *reinterpret_cast<const float4*>(ta + 16 * rb * LSTR + 4 * (q + 1));
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb].x, wreg[4 * q + 0], acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb].y, wreg[4 * q + 1], acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb].z, wreg[4 * q + 2], acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb].w, wreg[4 * q + 3], acc[rb], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RB, 0);       // LDS reads of group q+1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * RB, 0);   // ... then the MFMAs of group q
        }
        // next tile's intercepts (its group ids arrived during the MFMAs), then the ids
        // of the tile after
#pragma unroll
        for (int e = 0; e < 4 * RB; ++e) bz_next[e] = bz_load(gi[e]);
        load_groups<RB>(gi, g, (tile + 2 * stride) * LT, N, kq, rb0);
        // C/D map of 16x16x4: col = lane & 15 (sample), row = 4 * (lane >> 4) + reg
        const int64_t row0 = tile * LT;
        float tile_ll = 0.f;
        // y l - softplus(l),  softplus(l) = max(l,0) + ln2 log2(1 + 2^(-|l| log2e)) on the raw
        // v_exp_f32 / v_log_f32 (the log's argument is in (1, 2], the exponent's <= 0: no
        // denormal handling needed, ~1e-7 ABSOLUTE error, far inside the stated tolerance).
        // Instruction count matters here: VALU work delays the MFMAs of the other waves.
        constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
        const bool whole = row0 + LT <= N;      // uniform: only the last tile masks rows
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (rb0 + rb) + 4 * kq + r;
                const float yv = t[LT * LSTR + row];
                const float l = acc[rb][r] + bz_cur[4 * rb + r];
                const float g2 = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(l) * LOG2E));
                float v = __builtin_fmaf(yv, l, -fmaxf(l, 0.f));
                v = __builtin_fmaf(-LN2, g2, v);
                if (!whole && row0 + row >= N) v = 0.f;
                tile_ll += v;
            }
        acc_ll += (double)tile_ll;
        stage_store(st, lds + (cur ^ 1) * TILE_FLOATS, wave, lane);
        __syncthreads();
        cur ^= 1;
        tile += stride;
    };
    for (int it = 0; it < n_iter; it += 2) {   // n_iter is even (host)
        one_tile(bz_a, bz_b);
        one_tile(bz_b, bz_a);
    }
    // lanes with the same sample (lane & 15) hold different rows: fold bits 4,5
    acc_ll += __shfl_xor(acc_ll, 16);
    acc_ll += __shfl_xor(acc_ll, 32);
    if (lane < 16)
        slab[((int64_t)blockIdx.x * (NW / 4) + (wave >> 2)) * LS + 16 * sb + lane] = (float)acc_ll;
}

// ---- second generation: a wave owns rows, the draws live in LDS -----------------------------
// The kernel above stages X tiles in LDS (a store, a workgroup barrier and an A-operand read per
// tile) and keeps the draws in registers.  What its time is made of was measured this round:
//   * fp32 MFMA and VALU share the SIMD's issue: every VALU instruction between MFMAs costs the
//     matrix pipe ~4.6 cycles (profiles/r01_ubench_mfma_valu_mix.txt);
//   * a vector load that RETURNS TO REGISTERS costs the issuing SIMD ~115 cycles of MFMA time,
//     whatever its width (a dword gather as much as a 1-KiB dwordx4), an LDS-DMA
//     (buffer_load ... lds) ~45 (tools/ubench_mfma_vmem.hip, profiles/r02_ubench_mfma_vmem.txt);
//     the 32-row tile of the first kernel took 8 + 8 such loads per wave and 128 MFMAs.
// Both kernels below give a wave 16-row tiles of X for ALL sample blocks:
//   * the S <= 128 draws Wz (static for the whole launch) are written ONCE to LDS in exactly the
//     order the MFMA B operand is read: block (sb, j) = 1 KiB, lane l's 16 bytes at l*16 --
//     every ds_read_b128 is conflict-free, no padding, no per-tile stores, no barrier per tile;
//   * lane i16 owns the NSB consecutive samples NSB i16 .. NSB i16 + NSB - 1, so the intercepts
//     b[g_n, s] of a row are ONE 4 NSB-byte gather per lane (the first kernel: one dword gather
//     per row and sample block), and they enter as the MFMA's C input (the C layout is the
//     result layout): l = x.w + b costs no add;
//   * epilogue on packed f32 math with two transcendentals per element:
//       y l - softplus(l) = y l - (l + |l|)/2 - ln2 log2(1 + 2^(-|l| log2 e)),
//     the four rows a lane holds per sample share ONE v_log_f32 (log2 of the product of their
//     (1 + t) in (1, 16]); sums of y l, l, |l| run on v_pk_fma/add_f32;
//   * waves never synchronise; a wave leaves the loop as soon as its tiles are done.
constexpr int XT = 16;        // rows per wave tile
constexpr int XW = 8;         // waves per workgroup (2 per SIMD, one workgroup per CU)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef bsc_lds_ptr lds_ptr;
constexpr int vmcnt_only(int n) { return bsc_vmcnt_only(n); }
__device__ __forceinline__ auto x_tile_rsrc(const float* X, int64_t ldx, int D, int64_t N, int64_t row0) {
    return bsc_rows_rsrc(X, ldx, D, N, row0);       // rows past N read as zero
}
__device__ __forceinline__ auto row_vec_rsrc(const void* base, int64_t N, int64_t row0) {   // y or g: 4 B per row
    return bsc_vec_rsrc(base, N, row0);
}

// One tile's epilogue: acc[sb][r] = logit of row 4 kq + r, sample NSB i16 + sb; yv = y of those rows.
// Adds sum_r (y l - softplus(l)) per sample block to acc_ll (float64).
template <int NSB>
__device__ __forceinline__ void loglik_epilogue(const f32x4 (&acc)[NSB], const f32x4 yv, bool whole, int64_t row0,
                                                int64_t N, int kq, double (&acc_ll)[NSB]) {
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    if (whole) {
        const f32x2 y01 = {yv[0], yv[1]}, y23 = {yv[2], yv[3]};
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            const f32x4 l = acc[sb];
            const f32x2 l01 = {l[0], l[1]}, l23 = {l[2], l[3]};
            f32x2 u01, u23, t01, t23;
            u01[0] = -LOG2E * __builtin_fabsf(l[0]); u01[1] = -LOG2E * __builtin_fabsf(l[1]);
            u23[0] = -LOG2E * __builtin_fabsf(l[2]); u23[1] = -LOG2E * __builtin_fabsf(l[3]);
            t01[0] = __builtin_amdgcn_exp2f(u01[0]); t01[1] = __builtin_amdgcn_exp2f(u01[1]);
            t23[0] = __builtin_amdgcn_exp2f(u23[0]); t23[1] = __builtin_amdgcn_exp2f(u23[1]);
            f32x2 p = t01 + 1.0f;                                   // (1 + t0, 1 + t1)
            p = __builtin_elementwise_fma(p, t23, p);               // * (1 + t2), * (1 + t3)
            const float g2 = __builtin_amdgcn_logf(p[0] * p[1]);    // log2 of the product, in (0, 4]
            f32x2 yl = y01 * l01;
            yl = __builtin_elementwise_fma(y23, l23, yl);
            const f32x2 sl = l01 + l23;                             // sum l
            const f32x2 su = u01 + u23;                             // -log2e sum |l|
            // y l - (l + |l|)/2 - ln2 g2
            f32x2 v = __builtin_elementwise_fma(sl, f32x2{-0.5f, -0.5f}, yl);
            v = __builtin_elementwise_fma(su, f32x2{0.5f * LN2, 0.5f * LN2}, v);
            acc_ll[sb] += (double)(__builtin_fmaf(-LN2, g2, v[0] + v[1]));
        }
    } else {       // (wave-uniform) the tile that holds row N: rows past it count nothing
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            float tile_ll = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float l = acc[sb][r];
                const float g2 = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-__builtin_fabsf(l) * LOG2E));
                float v = __builtin_fmaf(yv[r], l, -fmaxf(l, 0.f));
                v = __builtin_fmaf(-LN2, g2, v);
                if (row0 + 4 * kq + r >= N) v = 0.f;
                tile_ll += v;
            }
            acc_ll[sb] += (double)tile_ll;
        }
    }
}

// The wave's per-sample sums -> one float64 row of the slab per workgroup (fixed order).
template <int NSB>
__device__ __forceinline__ void loglik_finish(const double (&acc_ll)[NSB], double (*red)[16 * NSB], int wave,
                                              int lane, int tid, int S, double* __restrict__ slab) {
    // lanes with the same samples (lane & 15) hold different rows: fold bits 4, 5; then the
    // workgroup's waves in a fixed order
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
        double v = acc_ll[sb];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) red[wave][NSB * lane + sb] = v;
    }
    __syncthreads();
    if (tid < 16 * NSB && tid < S) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < XW; ++w) v += red[w][tid];
        slab[(int64_t)blockIdx.x * S + tid] = v;
    }
}

// ---- X through VGPRs (any NSB <= 8; the S > 64 path, and the A/B partner of the DMA kernel) ----
// A wave loads its tile straight into the A operand layout: lane (i, kq) holds
// X[row i][16 j + 4 kq .. +3] for j = 0..15 (sixteen 16-byte buffer loads; the four kq lanes of a
// row read 64 contiguous bytes).  Register set j is refilled for the NEXT tile as soon as its last
// MFMA has issued, so 64 VGPRs hold the tile and the prefetch.
// DBG != 0: deletion builds for profiling only (results wrong): bit 0 drops the X refill loads,
// bit 1 the epilogue, bit 2 the LDS operand reads, bit 3 the intercept / y / id loads.
template <bool FULL, int NSB, int DBG = 0>   // FULL: D == 256; NSB: 16-sample blocks (S <= 16 NSB)
__global__ __launch_bounds__(64 * XW, 2) void logreg_loglik_xreg_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y,
    const int* __restrict__ g, int64_t N, int D, const float* __restrict__ Wz,
    const float* __restrict__ Bz, int n_groups, int S, double* __restrict__ slab, int n_iter) {
    __shared__ __attribute__((aligned(16))) f32x4 wl[NSB * 16 * 64];   // [sb][j][lane] -> 4 columns
    __shared__ double red[XW][16 * NSB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;

    for (int idx = tid; idx < NSB * 16 * 64; idx += 64 * XW) {
        const int ln = idx & 63, j = (idx >> 6) & 15, sb = idx >> 10;
        const int sample = NSB * (ln & 15) + sb, col = 16 * j + 4 * (ln >> 4);
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (sample < S && col < D) w = *reinterpret_cast<const f32x4*>(Wz + (int64_t)sample * D + col);
        wl[idx] = w;
    }
    __syncthreads();

    const int64_t n_waves = (int64_t)gridDim.x * XW;
    int64_t tile = (int64_t)blockIdx.x * XW + wave;
    const int x_voff = i16 * (int)(ldx * 4) + 16 * kq;
    auto x_load = [&](decltype(x_tile_rsrc(X, ldx, D, N, 0)) rs, int j) {
        auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, x_voff, 64 * j, 2);   // nt: X is read once
        f32x4 f = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        if (!FULL && 16 * j + 4 * kq >= D) f = f32x4{0.f, 0.f, 0.f, 0.f};
        return f;
    };
    // Bz[g, s] through a buffer descriptor: 32-bit offsets, ids outside [0, n_groups) read 0
    const auto bz_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, (unsigned)n_groups * (unsigned)S * 4u,
                                                           0x00020000);
    const int row_b = S * 4;
    struct Side {          // what the epilogue / the next tile's C input need, per tile
        f32x4 yv;          // y of rows 4 kq .. 4 kq + 3
        f32x4 bz[NSB];     // intercepts [sample block][row]
    };
    int gi[4];             // group-id byte offsets of the tile after next
    auto g_load = [&](int64_t row0) {
        auto v = __builtin_amdgcn_raw_buffer_load_b128(row_vec_rsrc(g, N, row0), 16 * kq, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) gi[r] = (int)v[r] * row_b;
    };
    auto side_load = [&](Side& sd, int64_t row0) {      // uses gi = ids of this tile
        auto v = __builtin_amdgcn_raw_buffer_load_b128(row_vec_rsrc(y, N, row0), 16 * kq, 0, 0);
        sd.yv = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        // (padded samples >= S read the neighbouring bytes or 0; their W is 0 and they are never stored)
        const int soff = 4 * NSB * i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (NSB == 1) {
                sd.bz[0][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bz_rsrc, gi[r] + soff, 0, 0));
            } else if constexpr (NSB == 2) {
                auto w = __builtin_amdgcn_raw_buffer_load_b64(bz_rsrc, gi[r] + soff, 0, 0);
                sd.bz[0][r] = __uint_as_float(w[0]);
                sd.bz[1][r] = __uint_as_float(w[1]);
            } else {
#pragma unroll
                for (int h = 0; h < NSB / 4; ++h) {
                    auto w = __builtin_amdgcn_raw_buffer_load_b128(bz_rsrc, gi[r] + soff + 16 * h, 0, 0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) sd.bz[4 * h + c][r] = __uint_as_float(w[c]);
                }
            }
        }
    };

    f32x4 A[16];
    {
        const auto rs = x_tile_rsrc(X, ldx, D, N, tile * XT);
#pragma unroll
        for (int j = 0; j < 16; ++j) A[j] = x_load(rs, j);
    }
    Side sa, sb_;
    g_load(tile * XT);
    side_load(sa, tile * XT);
    g_load((tile + n_waves) * XT);

    double acc_ll[NSB];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) acc_ll[sb] = 0.0;

    auto one_tile = [&](const Side& cur, Side& nxt) {
        const int64_t row0 = tile * XT;
        const auto rs_next = x_tile_rsrc(X, ldx, D, N, (tile + n_waves) * XT);
        f32x4 acc[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) acc[sb] = cur.bz[sb];          // C input = intercepts
        // next tile's y and intercepts (its ids arrived a tile ago) and the ids of the tile after,
        // requested FIRST: they are then older than the 16 refill loads below, so the next tile's
        // first MFMA (which takes the intercepts as its C input) waits for nothing recent
        if (!(DBG & 8)) {
            side_load(nxt, (tile + n_waves) * XT);
            g_load((tile + 2 * n_waves) * XT);
        }
        // The draws in LDS never change, so the compiler would hoist all 16 NSB operand reads out
        // of the tile loop (and spill them): the address is made opaque once per tile.  B operands
        // run one k-group ahead of the MFMAs that consume them.
        int wo = lane;          // (an integer, so the reads stay ds_read: an opaque POINTER would turn them flat)
        asm volatile("" : "+v"(wo));
        const f32x4* wp = wl + wo;
        f32x4 bn[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) bn[sb] = wp[(sb * 16) * 64];
        __builtin_amdgcn_sched_group_barrier(0x100, NSB, 0);             // the reads of group 0
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            f32x4 b[NSB];
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) b[sb] = bn[sb];
            if (j + 1 < 16 && !(DBG & 4)) {
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) bn[sb] = wp[(sb * 16 + j + 1) * 64];
            }
            const f32x4 a = A[j];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb)
                    acc[sb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[sb][r], acc[sb], 0, 0, 0);
            if (!(DBG & 1)) A[j] = x_load(rs_next, j);                   // refill for the next tile
            if (j + 1 < 16 && !(DBG & 4)) __builtin_amdgcn_sched_group_barrier(0x100, NSB, 0);   // LDS reads of group j+1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NSB, 0);               // ... the MFMAs of group j ...
            if (!(DBG & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // ... the refill load
        }
        if (DBG & 2) {
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) acc_ll[sb] += (double)(acc[sb][0] + acc[sb][1] + acc[sb][2] + acc[sb][3]);
        } else {
            loglik_epilogue<NSB>(acc, cur.yv, row0 + XT <= N, row0, N, kq, acc_ll);
        }
        tile += n_waves;
    };
    for (int it = 0; it < n_iter; it += 2) {   // n_iter is even (host)
        if (tile * XT >= N) break;             // (wave-uniform) nothing left for this wave
        one_tile(sa, sb_);
        if (tile * XT >= N) break;
        one_tile(sb_, sa);
    }
    loglik_finish<NSB>(acc_ll, red, wave, lane, tid, S, slab);
}

// ---- X by LDS-DMA (NSB <= 4: the default up to S = 64) ------------------------------------------
// Nothing a tile needs passes through a VGPR-returning load.  Each wave owns in LDS
//   * a ring of XR 1-KiB slots: the tile arrives in pieces of 8 rows x 128 bytes (whole lines: pieces
//     shaped like one strip of A operands, 16 rows x 64 bytes, read 23 % slower -- tools/ab_skinny_nt.py),
//     two per block of 32 columns, DMA'd XR pieces ahead of their use into an XOR-permuted image from
//     which strip j of a tile (the A-operand block of k-group j) is one conflict-free ds_read_b128;
//   * y and the group ids of the next tiles (a 256-B dword DMA each: lane l <- row0 + l);
//   * the 4-KiB block of intercepts b[g_row, NSB i16 ..] of the next tile, gathered by four
//     dwordx4 DMAs whose per-lane offsets come from the ids, read back as the MFMA's C input.
// LDS-DMA completes in issue order and is covered ONLY by the issuing wave's vmcnt; the compiler
// cannot tell the slots apart and would wait for vmcnt(0) at every ds_read it can see next to a
// DMA in flight, so (1) every read of DMA'd bytes is inline asm, paired with a hand-placed
// lgkmcnt(0) a k-group later, and (2) strip waits are counted by hand (see DMA_WAIT).  With four
// sample blocks the draws of block 0 stay in 64 registers, so that W (48 KiB) + 8 waves x 13 KiB
// fit the CU's 160 KiB; it also saves one of the four B-operand reads per k-group.
constexpr int XR = 8;         // ring slots per wave (divides 16: the slot of strip j is j % XR)
constexpr int DMA_SIDE = 6;   // vector-memory operations a tile issues before its first strip wait
constexpr int DMA_WAVE_BYTES = XR * 1024 + 4096 + 4 * 256;   // ring | intercepts | y[2] | ids[2]

template <bool FULL, int NSB, int DBG = 0>
__global__ __launch_bounds__(64 * XW, 2) void logreg_loglik_dma_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y,
    const int* __restrict__ g, int64_t N, int D, const float* __restrict__ Wz,
    const float* __restrict__ Bz, int n_groups, int S, double* __restrict__ slab, int n_iter) {
    static_assert(NSB == 1 || NSB == 2 || NSB == 4, "eight sample blocks of draws and the rings do not fit 160 KiB");
    constexpr int WREG = NSB == 4 ? 1 : 0;          // sample blocks whose draws stay in registers
    constexpr int NL = NSB - WREG;                  // sample blocks read from LDS
    __shared__ __attribute__((aligned(16))) f32x4 wl[NL * 16 * 64];   // [sb - WREG][j][lane] -> 4 columns
    __shared__ __attribute__((aligned(16))) char dma[XW * DMA_WAVE_BYTES];
    __shared__ double red[XW][16 * NSB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;

    for (int idx = tid; idx < NL * 16 * 64; idx += 64 * XW) {
        const int ln = idx & 63, j = (idx >> 6) & 15, sb = (idx >> 10) + WREG;
        const int sample = NSB * (ln & 15) + sb, col = 16 * j + 4 * (ln >> 4);
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (sample < S && col < D) w = *reinterpret_cast<const f32x4*>(Wz + (int64_t)sample * D + col);
        wl[idx] = w;
    }
    f32x4 wreg[WREG ? 16 : 1];      // draws of sample NSB i16 + 0: [j] -> columns 16 j + 4 kq .. +3
    if constexpr (WREG) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int sample = NSB * i16, col = 16 * j + 4 * kq;
            wreg[j] = (sample < S && col < D) ? *reinterpret_cast<const f32x4*>(Wz + (int64_t)sample * D + col)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();

    const int64_t n_waves = (int64_t)gridDim.x * XW;
    int64_t tile = (int64_t)blockIdx.x * XW + wave;
    // X arrives in WHOLE 128-byte lines (pieces of 16 rows x 64 bytes read 23 % slower: tools/ab_skinny_nt.py, and see
    // csrc/bsc_skinny.hip): piece s = 2 c + sp is rows 8 sp .. + 7 x columns 32 c .. + 31, DMA lane l = (row l >> 3,
    // 16-byte position l & 7); the chunk fetched into position p of row a is p ^ f, f = (a >> 1) | (sp << 2), which
    // keeps the operand read (lane (i16, kq) <- row i16, chunk kq + 4 t of block c = strip 2 c + t) conflict-free
    const int x_row_bytes = (int)(ldx * 4);
    const int x_voff0 = (lane >> 3) * x_row_bytes + 16 * ((lane & 7) ^ (lane >> 4));
    const int x_voff1 = (lane >> 3) * x_row_bytes + 16 * ((lane & 7) ^ ((lane >> 4) | 4));
    char* const my = dma + wave * DMA_WAVE_BYTES;                       // this wave's DMA region
    const unsigned my_addr = (unsigned)(uintptr_t)(lds_ptr)my;
    const unsigned addr_x0 = my_addr + 1024u * (i16 >> 3) + 128u * (i16 & 7) + 16u * (kq ^ (((i16 & 7) >> 1) | ((i16 >> 3) << 2)));
    const unsigned addr_x1 = addr_x0 ^ 64u;
    constexpr int BZ_OFF = XR * 1024, Y_OFF = BZ_OFF + 4096, G_OFF = Y_OFF + 512;

    // ---- the DMAs (every one counts in vmcnt, in this order) ----
    auto x_dma = [&](decltype(x_tile_rsrc(X, ldx, D, N, 0)) rs, int s) {            // piece s -> slot s % XR
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(my + (s % XR) * 1024), 16, (s & 1) ? x_voff1 : x_voff0,
                                                 (s & 1) * 8 * x_row_bytes + 128 * (s >> 1), 0, 2);
    };
    auto row_dma = [&](const void* base, int64_t row0, int lds_off) {              // lane l <- 4 bytes of row0 + l
        __builtin_amdgcn_raw_ptr_buffer_load_lds(row_vec_rsrc(base, N, row0), (lds_ptr)(my + lds_off), 4, 4 * lane, 0, 0, 0);
    };
    const auto bz_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, (unsigned)n_groups * (unsigned)S * 4u,
                                                           0x00020000);
    const int row_b = S * 4, soff = 4 * NSB * i16;
    auto bz_dma = [&](const f32x4 ids) {    // ids (as int bits) of rows 4 kq .. +3 -> intercepts [r][lane][NSB]
#pragma unroll
        for (int r = 0; r < 4; ++r)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bz_rsrc, (lds_ptr)(my + BZ_OFF + r * 1024), 16,
                                                     __float_as_int(ids[r]) * row_b + soff, 0, 0, 0);
    };
    // ---- reads of DMA'd bytes: inline asm (see the header), valid after lds_ready() ----
    const unsigned addr_lane = my_addr + 16u * lane;     // lane l's 16 bytes of a 1-KiB block
    const unsigned addr_kq = my_addr + 16u * kq;         // the 4 rows 4 kq .. 4 kq + 3 of a row vector
#define BSC_LDS_B128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
    auto a_read = [&](int j) {          // strip j = 2 c + t: both of its pieces sit in slots (2 c) % XR, + 1
        f32x4 f;
        if (j & 1) BSC_LDS_B128(f, addr_x1, ((j & ~1) % XR) * 1024);
        else BSC_LDS_B128(f, addr_x0, ((j & ~1) % XR) * 1024);
        return f;
    };

    // ---- prologue: strips 0 .. XR-1, y, ids and intercepts of the first tile, ids of the second ----
    {
        const auto rs = x_tile_rsrc(X, ldx, D, N, tile * XT);
#pragma unroll
        for (int j = 0; j < XR; ++j) x_dma(rs, j);
    }
    row_dma(y, tile * XT, Y_OFF);
    row_dma(g, tile * XT, G_OFF);
    row_dma(g, (tile + n_waves) * XT, G_OFF + 256);
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
    asm volatile("" ::: "memory");
    {
        f32x4 ids;
        BSC_LDS_B128(ids, addr_kq, G_OFF);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        bz_dma(ids);
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
    asm volatile("" ::: "memory");
    f32x4 an = a_read(0);           // the A operand of the next k-group, read one group ahead

    double acc_ll[NSB];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) acc_ll[sb] = 0.0;

    // LDS -> registers for the tile of parity `par`: its intercepts q[r] (lane l's 16 bytes = samples
    // NSB i16 .., the first NSB count), its y, and the ids of the tile after it.  Valid after the
    // next lgkmcnt(0).
    f32x4 q[4], yv_n, ids_n;
    auto side_read = [&](int par) {
        BSC_LDS_B128(q[0], addr_lane, BZ_OFF);
        BSC_LDS_B128(q[1], addr_lane, BZ_OFF + 1024);
        BSC_LDS_B128(q[2], addr_lane, BZ_OFF + 2048);
        BSC_LDS_B128(q[3], addr_lane, BZ_OFF + 3072);
        if (par) {
            BSC_LDS_B128(yv_n, addr_kq, Y_OFF + 256);
            BSC_LDS_B128(ids_n, addr_kq, G_OFF);
        } else {
            BSC_LDS_B128(yv_n, addr_kq, Y_OFF);
            BSC_LDS_B128(ids_n, addr_kq, G_OFF + 256);
        }
    };
    side_read(0);

    // `par` = parity of the tile within this wave's sequence (y and ids are double-buffered)
    auto one_tile = [&](int par) {
        const int64_t row0 = tile * XT;
        const auto rs_cur = x_tile_rsrc(X, ldx, D, N, row0);
        const auto rs_next = x_tile_rsrc(X, ldx, D, N, (tile + n_waves) * XT);
        // this tile's intercepts (C input) and y and the next tile's ids were requested from LDS
        // before the previous tile's epilogue (side_read): back long ago, the wait is for the compiler
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NSB];
        const f32x4 yv = yv_n, ids_next = ids_n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) acc[sb][r] = q[r][sb];
        }
        // side DMAs of the tiles ahead, FIRST (DMA_SIDE = 6 of them): older than this tile's strips,
        // so the last strip wait of this tile also covers them
        if (!(DBG & 8)) {
            row_dma(y, (tile + n_waves) * XT, Y_OFF + (par ^ 1) * 256);
            row_dma(g, (tile + 2 * n_waves) * XT, G_OFF + par * 256);
            bz_dma(ids_next);
        }
        int wo = lane;          // opaque once per tile: keeps the static B-operand reads inside the loop
        asm volatile("" : "+v"(wo));
        const f32x4* wp = wl + wo;
        f32x4 bn[NL];
#pragma unroll
        for (int sb = 0; sb < NL; ++sb) bn[sb] = wp[(sb * 16) * 64];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // `an` was read a k-group ago, like this group's B operands: the wait finds them done --
            // placed BEFORE the next group's reads are issued, or it would wait for those as well.
            // The builtin (0xC07F = lgkmcnt(0) alone), so that the compiler's own bookkeeping knows
            // its B reads are back: with an asm wait it would add lgkmcnt(N) waits that count the
            // asm read it cannot see, and stall on the reads just issued.
            if (j > 0) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 b[NSB];
            if constexpr (WREG) b[0] = wreg[j];
#pragma unroll
            for (int sb = 0; sb < NL; ++sb) b[sb + WREG] = bn[sb];
            if (j + 1 < 16 && !(DBG & 4)) {
#pragma unroll
                for (int sb = 0; sb < NL; ++sb) bn[sb] = wp[(sb * 16 + j + 1) * 64];
            }
            f32x4 a = an;
            if (!FULL && 16 * j + 4 * kq >= D) a = f32x4{0.f, 0.f, 0.f, 0.f};
            // DMA_WAIT: strip j+1 (of the next tile for j = 15) must have landed.  LDS-DMA completes
            // in issue order, so "at most as many outstanding as were issued after it": a strip issued
            // in the previous tile (j+1 < XR) is followed by the rest of that batch, this tile's side
            // DMAs and this tile's j strips = XR - 2 + DMA_SIDE; one issued in this tile by XR - 2.
            // (pieces: strip j + 1 of an even j lies in the block that strip j was read from; behind an odd j = 2 c + 1
            // block c + 1 is followed by XR / 2 - 2 blocks of two pieces = XR - 4)
            if (DBG & 16) {
                // profiling only: no strip wait at all (reads race the DMAs; the time shows what the
                // waits cost)
            } else if (j & 1) {
                if ((j + 1) / 2 < XR / 2) __builtin_amdgcn_s_waitcnt(vmcnt_only(XR - 4 + ((DBG & 8) ? 0 : DMA_SIDE)));
                else __builtin_amdgcn_s_waitcnt(vmcnt_only(XR - 4));
            }
            asm volatile("" ::: "memory");
            if (!(DBG & 4)) an = a_read((j + 1) % 16);
            // MFMAs have no memory semantics and would float above the asm read (and the next group's
            // lgkmcnt wait would then sit right behind it): pin "reads, then MFMAs, then the DMA"
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb)
                    acc[sb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[sb][r], acc[sb], 0, 0, 0);
            // behind an odd strip both slots of its block are free (strips j - 1 and j are in registers): pieces
            // j - 1 + XR, j + XR
            if (!(DBG & 1) && (j & 1)) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int p = j - 1 + e;
                    if (p + XR < 16) x_dma(rs_cur, p + XR);
                    else x_dma(rs_next, p + XR - 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next tile's side data: its DMAs were this tile's first, and the last strip wait
        // (vmcnt <= XR - 2) has covered them; the epilogue below covers the LDS latency
        if (!(DBG & 8)) side_read(par ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (DBG & 2) {
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) acc_ll[sb] += (double)(acc[sb][0] + acc[sb][1] + acc[sb][2] + acc[sb][3]);
        } else {
            loglik_epilogue<NSB>(acc, yv, row0 + XT <= N, row0, N, kq, acc_ll);
        }
        tile += n_waves;
    };
    for (int it = 0; it < n_iter; it += 2) {   // n_iter is even (host)
        if (tile * XT >= N) break;             // (wave-uniform) nothing left for this wave
        one_tile(0);
        if (tile * XT >= N) break;
        one_tile(1);
    }
    // no LDS-DMA of this wave may still be in flight when the workgroup's LDS is released
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    loglik_finish<NSB>(acc_ll, red, wave, lane, tid, S, slab);
#undef BSC_LDS_B128
}

template <bool FULL>
__global__ __launch_bounds__(64 * XW, 2) void logreg_loglik_dma_bx_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y,
    const int* __restrict__ g, int64_t N, int D, const float* __restrict__ Wz,
    const float* __restrict__ Bz, int n_groups, int S, double* __restrict__ slab, int n_iter) {
    constexpr int NSB = 4, WREG = 1, NL = NSB - WREG, DBG = 0, SPLIT = 2;
    // the draws as two bf16 terms, in B-operand order: [sb - WREG][k-step jj][term][lane] -> the eight columns
    // 32 jj + 4 kq .. + 3 and 32 jj + 16 + 4 kq .. + 3 of sample NSB i16 + sb (the A operand takes the same eight from
    // strips 2 jj and 2 jj + 1)
    __shared__ __attribute__((aligned(16))) bsc_u32x4 wl[NL * 8 * SPLIT * 64];
    __shared__ __attribute__((aligned(16))) char dma[XW * DMA_WAVE_BYTES];
    __shared__ double red[XW][16 * NSB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;

    auto draw_terms = [&](int sample, int jj, int q4, bsc_u32x4 (&t)[SPLIT]) {
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
        const int c0 = 32 * jj + 4 * q4, c1 = c0 + 16;
        if (sample < S && c0 < D) w0 = *reinterpret_cast<const f32x4*>(Wz + (int64_t)sample * D + c0);
        if (sample < S && c1 < D) w1 = *reinterpret_cast<const f32x4*>(Wz + (int64_t)sample * D + c1);
        unsigned pk[4][SPLIT];
        bsc_split_pk<SPLIT>(w0[0], w0[1], pk[0]);
        bsc_split_pk<SPLIT>(w0[2], w0[3], pk[1]);
        bsc_split_pk<SPLIT>(w1[0], w1[1], pk[2]);
        bsc_split_pk<SPLIT>(w1[2], w1[3], pk[3]);
#pragma unroll
        for (int c = 0; c < SPLIT; ++c) t[c] = bsc_u32x4{pk[0][c], pk[1][c], pk[2][c], pk[3][c]};
    };
    for (int idx = tid; idx < NL * 8 * 64; idx += 64 * XW) {
        const int ln = idx & 63, jj = (idx >> 6) & 7, sb = (idx >> 9) + WREG;
        bsc_u32x4 t[SPLIT];
        draw_terms(NSB * (ln & 15) + sb, jj, ln >> 4, t);
#pragma unroll
        for (int c = 0; c < SPLIT; ++c) wl[(((sb - WREG) * 8 + jj) * SPLIT + c) * 64 + ln] = t[c];
    }
    bsc_u32x4 wreg[8][SPLIT];      // draws of sample NSB i16 + 0
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) draw_terms(NSB * i16, jj, kq, wreg[jj]);
    __syncthreads();

    const int64_t n_waves = (int64_t)gridDim.x * XW;
    int64_t tile = (int64_t)blockIdx.x * XW + wave;
    // X arrives in WHOLE 128-byte lines (pieces of 16 rows x 64 bytes read 23 % slower: tools/ab_skinny_nt.py, and see
    // csrc/bsc_skinny.hip): piece s = 2 c + sp is rows 8 sp .. + 7 x columns 32 c .. + 31, DMA lane l = (row l >> 3,
    // 16-byte position l & 7); the chunk fetched into position p of row a is p ^ f, f = (a >> 1) | (sp << 2), which
    // keeps the operand read (lane (i16, kq) <- row i16, chunk kq + 4 t of block c = strip 2 c + t) conflict-free
    const int x_row_bytes = (int)(ldx * 4);
    const int x_voff0 = (lane >> 3) * x_row_bytes + 16 * ((lane & 7) ^ (lane >> 4));
    const int x_voff1 = (lane >> 3) * x_row_bytes + 16 * ((lane & 7) ^ ((lane >> 4) | 4));
    char* const my = dma + wave * DMA_WAVE_BYTES;                       // this wave's DMA region
    const unsigned my_addr = (unsigned)(uintptr_t)(lds_ptr)my;
    const unsigned addr_x0 = my_addr + 1024u * (i16 >> 3) + 128u * (i16 & 7) + 16u * (kq ^ (((i16 & 7) >> 1) | ((i16 >> 3) << 2)));
    const unsigned addr_x1 = addr_x0 ^ 64u;
    constexpr int BZ_OFF = XR * 1024, Y_OFF = BZ_OFF + 4096, G_OFF = Y_OFF + 512;

    // ---- the DMAs (every one counts in vmcnt, in this order) ----
    auto x_dma = [&](decltype(x_tile_rsrc(X, ldx, D, N, 0)) rs, int s) {            // piece s -> slot s % XR
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(my + (s % XR) * 1024), 16, (s & 1) ? x_voff1 : x_voff0,
                                                 (s & 1) * 8 * x_row_bytes + 128 * (s >> 1), 0, 2);
    };
    auto row_dma = [&](const void* base, int64_t row0, int lds_off) {              // lane l <- 4 bytes of row0 + l
        __builtin_amdgcn_raw_ptr_buffer_load_lds(row_vec_rsrc(base, N, row0), (lds_ptr)(my + lds_off), 4, 4 * lane, 0, 0, 0);
    };
    const auto bz_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, (unsigned)n_groups * (unsigned)S * 4u,
                                                           0x00020000);
    const int row_b = S * 4, soff = 4 * NSB * i16;
    auto bz_dma = [&](const f32x4 ids) {    // ids (as int bits) of rows 4 kq .. +3 -> intercepts [r][lane][NSB]
#pragma unroll
        for (int r = 0; r < 4; ++r)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bz_rsrc, (lds_ptr)(my + BZ_OFF + r * 1024), 16,
                                                     __float_as_int(ids[r]) * row_b + soff, 0, 0, 0);
    };
    // ---- reads of DMA'd bytes: inline asm (see the header), valid after lds_ready() ----
    const unsigned addr_lane = my_addr + 16u * lane;     // lane l's 16 bytes of a 1-KiB block
    const unsigned addr_kq = my_addr + 16u * kq;         // the 4 rows 4 kq .. 4 kq + 3 of a row vector
#define BSC_LDS_B128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
    auto a_read = [&](int j) {          // strip j = 2 c + t: both of its pieces sit in slots (2 c) % XR, + 1
        f32x4 f;
        if (j & 1) BSC_LDS_B128(f, addr_x1, ((j & ~1) % XR) * 1024);
        else BSC_LDS_B128(f, addr_x0, ((j & ~1) % XR) * 1024);
        return f;
    };

    // ---- prologue: strips 0 .. XR-1, y, ids and intercepts of the first tile, ids of the second ----
    {
        const auto rs = x_tile_rsrc(X, ldx, D, N, tile * XT);
#pragma unroll
        for (int j = 0; j < XR; ++j) x_dma(rs, j);
    }
    row_dma(y, tile * XT, Y_OFF);
    row_dma(g, tile * XT, G_OFF);
    row_dma(g, (tile + n_waves) * XT, G_OFF + 256);
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
    asm volatile("" ::: "memory");
    {
        f32x4 ids;
        BSC_LDS_B128(ids, addr_kq, G_OFF);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        bz_dma(ids);
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
    asm volatile("" ::: "memory");
    f32x4 an = a_read(0), an2 = a_read(1);      // the A operand of the next k-step, read one step ahead

    double acc_ll[NSB];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) acc_ll[sb] = 0.0;

    // LDS -> registers for the tile of parity `par`: its intercepts q[r] (lane l's 16 bytes = samples
    // NSB i16 .., the first NSB count), its y, and the ids of the tile after it.  Valid after the
    // next lgkmcnt(0).
    f32x4 q[4], yv_n, ids_n;
    auto side_read = [&](int par) {
        BSC_LDS_B128(q[0], addr_lane, BZ_OFF);
        BSC_LDS_B128(q[1], addr_lane, BZ_OFF + 1024);
        BSC_LDS_B128(q[2], addr_lane, BZ_OFF + 2048);
        BSC_LDS_B128(q[3], addr_lane, BZ_OFF + 3072);
        if (par) {
            BSC_LDS_B128(yv_n, addr_kq, Y_OFF + 256);
            BSC_LDS_B128(ids_n, addr_kq, G_OFF);
        } else {
            BSC_LDS_B128(yv_n, addr_kq, Y_OFF);
            BSC_LDS_B128(ids_n, addr_kq, G_OFF + 256);
        }
    };
    side_read(0);

    // `par` = parity of the tile within this wave's sequence (y and ids are double-buffered)
    auto one_tile = [&](int par) {
        const int64_t row0 = tile * XT;
        const auto rs_cur = x_tile_rsrc(X, ldx, D, N, row0);
        const auto rs_next = x_tile_rsrc(X, ldx, D, N, (tile + n_waves) * XT);
        // this tile's intercepts (C input) and y and the next tile's ids were requested from LDS
        // before the previous tile's epilogue (side_read): back long ago, the wait is for the compiler
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NSB];
        const f32x4 yv = yv_n, ids_next = ids_n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) acc[sb][r] = q[r][sb];
        }
        // side DMAs of the tiles ahead, FIRST (DMA_SIDE = 6 of them): older than this tile's strips,
        // so the last strip wait of this tile also covers them
        if (!(DBG & 8)) {
            row_dma(y, (tile + n_waves) * XT, Y_OFF + (par ^ 1) * 256);
            row_dma(g, (tile + 2 * n_waves) * XT, G_OFF + par * 256);
            bz_dma(ids_next);
        }
        int wo = lane;          // opaque once per tile: keeps the static B-operand reads inside the loop
        asm volatile("" : "+v"(wo));
        const bsc_u32x4* wp = wl + wo;
        bsc_u32x4 bn[NL][SPLIT];
#pragma unroll
        for (int sb = 0; sb < NL; ++sb)
#pragma unroll
            for (int c = 0; c < SPLIT; ++c) bn[sb][c] = wp[((sb * 8) * SPLIT + c) * 64];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            // `an`, `an2` (strips 2 jj, 2 jj + 1) were read a k-step ago, like this step's B operands: the wait finds them
            // done -- placed BEFORE the next step's reads are issued (see logreg_loglik_dma_kernel)
            if (jj > 0) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
            }
            bsc_u32x4 b[NSB][SPLIT];
#pragma unroll
            for (int c = 0; c < SPLIT; ++c) b[0][c] = wreg[jj][c];
#pragma unroll
            for (int sb = 0; sb < NL; ++sb)
#pragma unroll
                for (int c = 0; c < SPLIT; ++c) b[sb + WREG][c] = bn[sb][c];
            if (jj + 1 < 8) {
#pragma unroll
                for (int sb = 0; sb < NL; ++sb)
#pragma unroll
                    for (int c = 0; c < SPLIT; ++c) bn[sb][c] = wp[((sb * 8 + jj + 1) * SPLIT + c) * 64];
            }
            f32x4 a0 = an, a1 = an2;
            if (!FULL && 32 * jj + 4 * kq >= D) a0 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!FULL && 32 * jj + 16 + 4 * kq >= D) a1 = f32x4{0.f, 0.f, 0.f, 0.f};
            // DMA_WAIT: strips 2 jj + 2, 2 jj + 3 (of the next tile for jj = 7) must have landed.  LDS-DMA completes in
            // issue order: a strip issued in the previous tile (2 jj + 3 < XR) is followed by the rest of that batch,
            // this tile's side DMAs and this tile's 2 jj strips = XR - 4 + DMA_SIDE; one issued in this tile by XR - 4.
            if (2 * jj + 3 < XR) __builtin_amdgcn_s_waitcnt(vmcnt_only(XR - 4 + DMA_SIDE));
            else __builtin_amdgcn_s_waitcnt(vmcnt_only(XR - 4));
            asm volatile("" ::: "memory");
            an = a_read((2 * jj + 2) % 16);
            an2 = a_read((2 * jj + 3) % 16);
            __builtin_amdgcn_sched_barrier(0);
            // the A operand: eight f32 of the row as two bf16 terms
            bsc_u32x4 ah, am;
            {
                unsigned pk[4][SPLIT];
                bsc_split_pk<SPLIT>(a0[0], a0[1], pk[0]);
                bsc_split_pk<SPLIT>(a0[2], a0[3], pk[1]);
                bsc_split_pk<SPLIT>(a1[0], a1[1], pk[2]);
                bsc_split_pk<SPLIT>(a1[2], a1[3], pk[3]);
                ah = bsc_u32x4{pk[0][0], pk[1][0], pk[2][0], pk[3][0]};
                am = bsc_u32x4{pk[0][1], pk[1][1], pk[2][1], pk[3][1]};
            }
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                acc[sb] = bsc_mfma16_bf16(ah, b[sb][1], acc[sb]);
                acc[sb] = bsc_mfma16_bf16(am, b[sb][0], acc[sb]);
                acc[sb] = bsc_mfma16_bf16(ah, b[sb][0], acc[sb]);
            }
            // slots (2 jj) % XR, (2 jj + 1) % XR are free (their strips are in a0, a1): strips 2 jj + XR, 2 jj + 1 + XR
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * jj + e;
                if (j + XR < 16) x_dma(rs_cur, j + XR);
                else x_dma(rs_next, j + XR - 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next tile's side data: its DMAs were this tile's first, and the last strip wait
        // (vmcnt <= XR - 2) has covered them; the epilogue below covers the LDS latency
        if (!(DBG & 8)) side_read(par ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (DBG & 2) {
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) acc_ll[sb] += (double)(acc[sb][0] + acc[sb][1] + acc[sb][2] + acc[sb][3]);
        } else {
            loglik_epilogue<NSB>(acc, yv, row0 + XT <= N, row0, N, kq, acc_ll);
        }
        tile += n_waves;
    };
    for (int it = 0; it < n_iter; it += 2) {   // n_iter is even (host)
        if (tile * XT >= N) break;             // (wave-uniform) nothing left for this wave
        one_tile(0);
        if (tile * XT >= N) break;
        one_tile(1);
    }
    // no LDS-DMA of this wave may still be in flight when the workgroup's LDS is released
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    loglik_finish<NSB>(acc_ll, red, wave, lane, tid, S, slab);
#undef BSC_LDS_B128
}

// ell[s] = sum over the block partials, float64, fixed order.  One wave per sample (a single
// 1024-thread workgroup walking all 64 columns took 10 us, mostly latency: a third of what the
// whole parameter side of an update costs).
__global__ __launch_bounds__(256) void loglik_reduce_kernel(const float* __restrict__ slab,
                                                            int n_rows, double* __restrict__ ell) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    double sum = 0.0;
    for (int b = lane; b < n_rows; b += 64) sum += (double)slab[(int64_t)b * LS + s];
    sum = wave_allsum_f64(sum);
    if (lane == 0) ell[s] = sum;
}

// The same for the float64 workgroup partials of logreg_loglik_xreg_kernel and any S.
__global__ __launch_bounds__(256) void loglik_reduce_f64_kernel(const double* __restrict__ slab,
                                                                int n_rows, int S,
                                                                double* __restrict__ ell) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    double sum = 0.0;
    for (int b = lane; b < n_rows; b += 64) sum += slab[(int64_t)b * S + s];
    sum = wave_allsum_f64(sum);
    if (lane == 0) ell[s] = sum;
}

// ---- parameter side (float64, tiny) -----------------------------------------

__device__ __forceinline__ void philox4(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

#pragma clang fp contract(off)
// z = mu + exp(rho) * eps (Philox stream 2); writes eps [S,P] f64, Wz [S,D] f32,
// Bz [G,S] f32 (transposed so a row's 64 intercepts are one 256-B run), zeta [S] f64
__global__ void bbvi_sample_kernel(const double* __restrict__ lam, int D, int G, int S,
                                   uint64_t seed, uint32_t step, double* __restrict__ eps,
                                   float* __restrict__ Wz, float* __restrict__ Bz,
                                   double* __restrict__ zeta) {
    const int P = D + G + 1;
    const int n_blocks = (P + 3) / 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * n_blocks) return;
    const int s = idx / n_blocks, b = idx % n_blocks;
    uint32_t c[4] = {(uint32_t)b, (uint32_t)s, 2u, step};
    philox4(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double two_m32 = 2.3283064365386963e-10, two_pi = 6.283185307179586476925286766559;
    const double u0 = ((double)c[0] + 0.5) * two_m32, u1 = ((double)c[1] + 0.5) * two_m32;
    const double u2 = ((double)c[2] + 0.5) * two_m32, u3 = ((double)c[3] + 0.5) * two_m32;
    const double r0 = sqrt(-2.0 * log(u0)), r1 = sqrt(-2.0 * log(u2));
    const double t0 = two_pi * u1, t1 = two_pi * u3;
    const double z4[4] = {r0 * cos(t0), r0 * sin(t0), r1 * cos(t1), r1 * sin(t1)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 4 * b + j;
        if (i >= P) break;
        eps[(int64_t)s * P + i] = z4[j];
        const double sd = exp(lam[P + i]);
        const double z = lam[i] + sd * z4[j];
        if (i < D) Wz[(int64_t)s * D + i] = (float)z;
        else if (i < D + G) Bz[(int64_t)(i - D) * S + s] = (float)z;
        else zeta[s] = z;
    }
}

// f_s, the scalar control variate and the score-function gradient, as three small
// launches (a single workgroup did all of it in ~70 us, a quarter of the whole update):
//   bbvi_f_kernel        one workgroup per sample:  f_s = scale*ell_s + log p(z_s) - log q(z_s)
//   bbvi_moments_kernel  one thread per score component i: its covariance / variance terms,
//                        summed per workgroup (fixed order)
//   bbvi_finish_kernel   a = sum cov / sum var;  grad_i = mean_s (f_s - a) h_si;  elbo
constexpr int BB_BLOCK = 256;
constexpr int BB_WAVES = BB_BLOCK / 64;
constexpr int BB_MAX_S = 64;

__global__ __launch_bounds__(BB_BLOCK) void bbvi_f_kernel(
    const double* __restrict__ lam, const double* __restrict__ eps, const double* __restrict__ ell,
    int D, int G, double scale, double a0, double b0, double log_prior_const,
    double* __restrict__ f, double* __restrict__ f_out) {
    __shared__ double red[BB_WAVES];
    const int P = D + G + 1;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double LOG_2PI = 1.8378770664093454835606594728112;
    const double zeta = lam[P - 1] + exp(lam[2 * P - 1]) * eps[(int64_t)s * P + P - 1];
    const double tau = exp(zeta);
    double part = 0.0;
    for (int i = tid; i < P; i += BB_BLOCK) {
        const double e = eps[(int64_t)s * P + i], rho = lam[P + i];
        const double z = lam[i] + exp(rho) * e;
        double lp;
        if (i < D) lp = -0.5 * LOG_2PI - 0.5 * z * z;
        else if (i < D + G) lp = -0.5 * LOG_2PI + 0.5 * zeta - 0.5 * tau * z * z;
        else lp = log_prior_const + a0 * zeta - b0 * tau;
        const double lq = -0.5 * LOG_2PI - rho - 0.5 * e * e;
        part += lp - lq;
    }
    part = wave_allsum_f64(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double tot = red[0];
        for (int k = 1; k < BB_WAVES; ++k) tot += red[k];
        const double v = scale * ell[s] + tot;
        f[s] = v;
        if (f_out) f_out[s] = v;
    }
}

__device__ __forceinline__ double score_component(const double* __restrict__ eps, int64_t s, int P,
                                                  int i, int p, double inv_sd) {
    const double e = eps[s * P + p];
    return i < P ? e * inv_sd : e * e - 1.0;
}

// A workgroup owns 16 parameters (both score components of each) for all S samples, thread
// (parameter tid % 16, sample group tid / 16) holds its (at most four) draws in registers: means
// first, then the centred sums, each added over the 16 groups in a fixed order.  (One thread per
// component walking all S samples twice -- ten workgroups, 128 dependent loads each -- took 12 us.)
constexpr int BM_PARAMS = 16;

__global__ __launch_bounds__(BB_BLOCK) void bbvi_moments_kernel(
    const double* __restrict__ lam, const double* __restrict__ eps, const double* __restrict__ f,
    int P, int S, double* __restrict__ cov_part, double* __restrict__ var_part) {
    __shared__ double fs[BB_MAX_S];
    __shared__ double part[BM_PARAMS][16][4];
    __shared__ double mean[BM_PARAMS][4];
    __shared__ double cv[BM_PARAMS][2];
    const int tid = threadIdx.x, pp = tid & 15, sg = tid >> 4;
    if (tid < S) fs[tid] = f[tid];
    __syncthreads();
    const int p = blockIdx.x * BM_PARAMS + pp;
    double h0[4], h1[4], fv[4];
    bool on[4];
    const double inv_sd = p < P ? exp(-lam[P + p]) : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                    // S <= 64: at most four draws per thread
        const int s = sg + 16 * k;
        on[k] = p < P && s < S;
        const double e = on[k] ? eps[(int64_t)s * P + p] : 0.0;
        h0[k] = e * inv_sd;
        h1[k] = e * e - 1.0;
        fv[k] = on[k] ? fs[s] : 0.0;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (on[k]) { a0 += h0[k]; a1 += fv[k] * h0[k]; a2 += h1[k]; a3 += fv[k] * h1[k]; }
    part[pp][sg][0] = a0; part[pp][sg][1] = a1; part[pp][sg][2] = a2; part[pp][sg][3] = a3;
    __syncthreads();
    if (tid < 4 * BM_PARAMS) {                       // (parameter tid / 4, quantity tid % 4): over the groups, in order
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += part[tid >> 2][k][tid & 3];
        mean[tid >> 2][tid & 3] = t / S;
    }
    __syncthreads();
    const double mh0 = mean[pp][0], mfh0 = mean[pp][1], mh1 = mean[pp][2], mfh1 = mean[pp][3];
    double c0 = 0.0, v0 = 0.0, c1 = 0.0, v1 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (on[k]) {
            c0 += (fv[k] * h0[k] - mfh0) * (h0[k] - mh0);
            v0 += (h0[k] - mh0) * (h0[k] - mh0);
            c1 += (fv[k] * h1[k] - mfh1) * (h1[k] - mh1);
            v1 += (h1[k] - mh1) * (h1[k] - mh1);
        }
    __syncthreads();                                 // `part` is reused
    part[pp][sg][0] = c0; part[pp][sg][1] = v0; part[pp][sg][2] = c1; part[pp][sg][3] = v1;
    __syncthreads();
    if (tid < BM_PARAMS) {                           // one parameter: cov and var of both its components
        double c = 0.0, v = 0.0;
        for (int q = 0; q < 2; ++q) {
            double cq = 0.0, vq = 0.0;
            for (int k = 0; k < 16; ++k) { cq += part[tid][k][2 * q]; vq += part[tid][k][2 * q + 1]; }
            c += cq / (S - 1);
            v += vq / (S - 1);
        }
        const bool live = blockIdx.x * BM_PARAMS + tid < P;
        cv[tid][0] = live ? c : 0.0;
        cv[tid][1] = live ? v : 0.0;
    }
    __syncthreads();
    if (tid == 0) {
        double c = 0.0, v = 0.0;
        for (int k = 0; k < BM_PARAMS; ++k) { c += cv[k][0]; v += cv[k][1]; }
        cov_part[blockIdx.x] = c;
        var_part[blockIdx.x] = v;
    }
}

__global__ __launch_bounds__(BB_BLOCK) void bbvi_finish_kernel(
    const double* __restrict__ lam, const double* __restrict__ eps, const double* __restrict__ f,
    int P, int S, const double* __restrict__ cov_part, const double* __restrict__ var_part,
    int n_parts, double* __restrict__ elbo, double* __restrict__ grad) {
    __shared__ double fs[BB_MAX_S];
    const int tid = threadIdx.x;
    if (tid < S) fs[tid] = f[tid];
    __syncthreads();
    __shared__ double a_sh;                        // as in bbvi_update_kernel
    if (tid < 64) {
        double c = 0.0, v = 0.0;
        for (int k = tid; k < n_parts; k += 64) { c += cov_part[k]; v += var_part[k]; }
        c = wave_allsum_f64(c);
        v = wave_allsum_f64(v);
        if (tid == 0) a_sh = c / v;
    }
    __syncthreads();
    const double a = a_sh;
    if (blockIdx.x == 0 && tid == 0) {
        double fm = 0.0;
        for (int s = 0; s < S; ++s) fm += fs[s];
        elbo[0] = fm / S;
    }
    const int i = blockIdx.x * BB_BLOCK + tid;
    if (i >= 2 * P) return;
    const int p = i < P ? i : i - P;
    const double inv_sd = exp(-lam[P + p]);
    double gsum = 0.0;
    for (int s = 0; s < S; ++s) gsum += (fs[s] - a) * score_component(eps, s, P, i, p, inv_sd);
    grad[i] = gsum / S;
}

// finish + Adam + the NEXT update's draws in one launch (bsc_bbvi_update): a workgroup owns 16
// parameters (mu_p, rho_p) for all S samples.
//   1. thread (parameter pp = tid % 16, sample group sg = tid / 16): its share of
//      sum_s (f_s - a) h_s for both score components, added over the 16 groups in a fixed order;
//   2. the 16 threads with sg = 0: gradient, Adam (the arithmetic of adam_ascent_kernel, contraction
//      off), the new (mu, rho) to LDS;
//   3. thread (parameter quad tid / 64, sample tid % 64): the Philox draw of bbvi_sample_kernel for the
//      next step, written over eps (this workgroup's 16 columns, which only it has read) and into
//      Wz / Bz / zeta.
// Before: finish, Adam and the sampler were three launches of ~5 us each on a 300 us update.
constexpr int BU_PARAMS = 16;

#pragma clang fp contract(off)
__global__ __launch_bounds__(BB_BLOCK) void bbvi_update_kernel(
    double* __restrict__ lam, double* __restrict__ eps, const double* __restrict__ f, int D, int G, int S,
    const double* __restrict__ cov_part, const double* __restrict__ var_part, int n_parts,
    double* __restrict__ m1, double* __restrict__ m2, double lr, double beta1, double beta2, double eps_adam,
    double corr1, double corr2, uint64_t seed, uint32_t next_step, float* __restrict__ Wz,
    float* __restrict__ Bz, double* __restrict__ zeta, double* __restrict__ elbo, double* __restrict__ grad) {
    __shared__ double fs[BB_MAX_S];
    __shared__ double part[BU_PARAMS][BU_PARAMS][2];
    __shared__ double fresh[BU_PARAMS][2];
    const int P = D + G + 1;
    const int tid = threadIdx.x, pp = tid & 15, sg = tid >> 4;
    const int p = blockIdx.x * BU_PARAMS + pp;
    if (tid < S) fs[tid] = f[tid];
    __syncthreads();
    // a = sum cov / sum var over the moment kernel's workgroups: lane l of the first wave adds parts
    // l, l + 64, ..., the wave adds its lanes in a fixed tree -- the same value in every workgroup
    __shared__ double a_sh;
    if (tid < 64) {
        double c = 0.0, v = 0.0;
        for (int k = tid; k < n_parts; k += 64) { c += cov_part[k]; v += var_part[k]; }
        c = wave_allsum_f64(c);
        v = wave_allsum_f64(v);
        if (tid == 0) a_sh = c / v;
    }
    __syncthreads();
    const double a = a_sh;
    if (blockIdx.x == 0 && tid == 0) {
        double fm = 0.0;
        for (int s = 0; s < S; ++s) fm += fs[s];
        elbo[0] = fm / S;
    }
    double gm = 0.0, gr = 0.0;
    if (p < P) {
        const double inv_sd = exp(-lam[P + p]);
        for (int s = sg; s < S; s += 16) {
            const double e = eps[(int64_t)s * P + p], wgt = fs[s] - a;
            gm += wgt * (e * inv_sd);
            gr += wgt * (e * e - 1.0);
        }
    }
    part[pp][sg][0] = gm;
    part[pp][sg][1] = gr;
    __syncthreads();
    if (sg == 0 && p < P) {
        double g2[2] = {0.0, 0.0};
        for (int k = 0; k < 16; ++k) { g2[0] += part[pp][k][0]; g2[1] += part[pp][k][1]; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t i = (int64_t)h * P + p;
            const double gi = g2[h] / S;
            grad[i] = gi;
            const double am = beta1 * m1[i] + (1.0 - beta1) * gi;
            const double bm = beta2 * m2[i] + (1.0 - beta2) * gi * gi;
            m1[i] = am;
            m2[i] = bm;
            const double mhat = am / corr1;
            const double vhat = bm / corr2;
            const double nv = lam[i] + lr * mhat / (sqrt(vhat) + eps_adam);
            lam[i] = nv;
            fresh[pp][h] = nv;
        }
    }
    __syncthreads();
    // ---- the next step's draws for this workgroup's four parameter quads
    const int s = tid & 63, quad = blockIdx.x * (BU_PARAMS / 4) + (tid >> 6);
    if (s >= S || 4 * quad >= P) return;
    uint32_t cnt[4] = {(uint32_t)quad, (uint32_t)s, 2u, next_step};
    philox4(cnt, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double two_m32 = 2.3283064365386963e-10, two_pi = 6.283185307179586476925286766559;
    const double u0 = ((double)cnt[0] + 0.5) * two_m32, u1 = ((double)cnt[1] + 0.5) * two_m32;
    const double u2 = ((double)cnt[2] + 0.5) * two_m32, u3 = ((double)cnt[3] + 0.5) * two_m32;
    const double r0 = sqrt(-2.0 * log(u0)), r1 = sqrt(-2.0 * log(u2));
    const double t0 = two_pi * u1, t1 = two_pi * u3;
    const double z4[4] = {r0 * cos(t0), r0 * sin(t0), r1 * cos(t1), r1 * sin(t1)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 4 * quad + j;
        if (i >= P) break;
        eps[(int64_t)s * P + i] = z4[j];
        const int lp = 4 * (tid >> 6) + j;
        const double z = fresh[lp][0] + exp(fresh[lp][1]) * z4[j];
        if (i < D) Wz[(int64_t)s * D + i] = (float)z;
        else if (i < D + G) Bz[(int64_t)(i - D) * S + s] = (float)z;
        else zeta[s] = z;
    }
}
#pragma clang fp contract(fast)

}  // namespace

extern "C" {

int bsc_logreg_bbvi_loglik(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y,
                           const int32_t* g, int64_t N, int32_t D, int32_t n_groups,
                           const float* Wz, const float* Bz, int32_t S, double* ell) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(N >= 0 && ((X && y && g) || N == 0) && Wz && Bz && ell,
                "bsc_logreg_bbvi_loglik: null pointer");
    if (S < 1 || S > 128 || D < 4 || D > LD || (D % 4) != 0)
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_logreg_bbvi_loglik: needs 1 <= S <= 128 and D %% 4 == 0 in [4,%d] (got S=%d D=%d)",
                        LD, S, D);
    BSC_REQUIRE(n_groups >= 1, "bsc_logreg_bbvi_loglik: n_groups=%d", n_groups);
    if ((int64_t)n_groups * S > ((int64_t)1 << 28))
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_logreg_bbvi_loglik: n_groups * S = %lld exceeds the 32-bit gather range (2^28)",
                        (long long)n_groups * S);
    BSC_REQUIRE(ldx >= D && ldx % 4 == 0 && ldx < ((int64_t)1 << 26),
                "bsc_logreg_bbvi_loglik: bad ldx=%lld", (long long)ldx);
    BSC_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Wz & 15) == 0 && (N == 0 || ((uintptr_t)y & 15) == 0) &&
                    (N == 0 || ((uintptr_t)g & 15) == 0),
                "bsc_logreg_bbvi_loglik: X, y, g and Wz must be 16-byte aligned");
    if (ctx->bbvi_kernel == 0 && S == LS) {
        // first-generation kernel (X tiles staged in LDS, draws in registers): kept for in-process A/B
        const int64_t n_tiles = (N + LT - 1) / LT;
        const int64_t max_blocks = 2 * (int64_t)ctx->cu_count;
        int n_iter = 0, n_blocks = 1;
        if (n_tiles > 0) {
            const int64_t it = (n_tiles + max_blocks - 1) / max_blocks;
            n_iter = (int)(it + (it & 1));   // even: the kernel alternates two register sets per tile pair
            n_blocks = (int)((n_tiles + it - 1) / it);
        }
        void* ws = nullptr;
        const int nw = ctx->bbvi_waves == 8 ? 8 : 4;   // 8 measured 1 % slower (346 vs 342 us): kept as a knob
        int rc = bsc_workspace(ctx, (size_t)n_blocks * (nw / 4) * LS * sizeof(float), &ws);
        if (rc != BSC_OK) return rc;
        ctx->slab_rows = 0;
        {
            bsc_prof_scope prof(ctx);
#define BSC_LL(FULL, NW)                                                                         \
    hipLaunchKernelGGL((logreg_loglik_kernel<FULL, NW>), dim3(n_blocks), dim3(64 * NW), 0,        \
                       ctx->stream, X, ldx, y, (const int*)g, N, (int)D, Wz, Bz, (int)n_groups,  \
                       (float*)ws, n_iter)
            if (D == LD && nw == 8) BSC_LL(true, 8);
            else if (D == LD) BSC_LL(true, 4);
            else if (nw == 8) BSC_LL(false, 8);
            else BSC_LL(false, 4);
#undef BSC_LL
        }
        BSC_LAUNCH_CHECK();
        hipLaunchKernelGGL(loglik_reduce_kernel, dim3(LS / 4), dim3(256), 0, ctx->stream, (const float*)ws,
                           n_blocks * (nw / 4), ell);
        BSC_LAUNCH_CHECK();
        return BSC_OK;
    }
    // X in registers, draws in LDS: one 512-thread workgroup per CU, waves take 16-row tiles
    const int64_t n_tiles = (N + XT - 1) / XT;
    int n_blocks = (int)((n_tiles + XW - 1) / XW);
    if (n_blocks > ctx->cu_count) n_blocks = ctx->cu_count;
    if (n_blocks < 1) n_blocks = 1;
    const int64_t n_waves = (int64_t)n_blocks * XW;
    int64_t it = (n_tiles + n_waves - 1) / n_waves;
    const int n_iter = (int)(it + (it & 1));   // even: the kernel alternates two register sets per tile pair
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)n_blocks * S * sizeof(double), &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    const int nsb = S <= 16 ? 1 : S <= 32 ? 2 : S <= 64 ? 4 : 8;
    {
        bsc_prof_scope prof(ctx);
#define BSC_LL_ARGS                                                                                   \
    dim3(n_blocks), dim3(64 * XW), 0, ctx->stream, X, ldx, y, (const int*)g, N, (int)D, Wz, Bz,           \
        (int)n_groups, (int)S, (double*)ws, n_iter
#define BSC_LLX_D(NSB)                                                                                \
    do {                                                                                              \
        if (D == LD) hipLaunchKernelGGL((logreg_loglik_xreg_kernel<true, NSB>), BSC_LL_ARGS);         \
        else hipLaunchKernelGGL((logreg_loglik_xreg_kernel<false, NSB>), BSC_LL_ARGS);                \
    } while (0)
        // X by LDS-DMA when the intercept gathers are 16-byte aligned four-sample runs (S in 36..64,
        // S % 4 == 0: config 5's S = 64); X through VGPRs otherwise
        const bool dma_ok = nsb == 4 && S % 4 == 0 && ((uintptr_t)Bz & 15) == 0 && ctx->bbvi_kernel != 2;
        if (dma_ok && ctx->bbvi_dbg && D == LD) {
            // profiling-only deletion builds (BSC_BBVI_DBG): wrong results by construction
            switch (ctx->bbvi_dbg) {
                case 1: hipLaunchKernelGGL((logreg_loglik_dma_kernel<true, 4, 1>), BSC_LL_ARGS); break;
                case 2: hipLaunchKernelGGL((logreg_loglik_dma_kernel<true, 4, 2>), BSC_LL_ARGS); break;
                case 3: hipLaunchKernelGGL((logreg_loglik_dma_kernel<true, 4, 3>), BSC_LL_ARGS); break;
                case 7: hipLaunchKernelGGL((logreg_loglik_dma_kernel<true, 4, 7>), BSC_LL_ARGS); break;
                case 11: hipLaunchKernelGGL((logreg_loglik_dma_kernel<true, 4, 11>), BSC_LL_ARGS); break;
                case 15: hipLaunchKernelGGL((logreg_loglik_dma_kernel<true, 4, 15>), BSC_LL_ARGS); break;
                case 16: hipLaunchKernelGGL((logreg_loglik_dma_kernel<true, 4, 16>), BSC_LL_ARGS); break;
                default: hipLaunchKernelGGL((logreg_loglik_dma_kernel<true, 4>), BSC_LL_ARGS); break;
            }
        } else if (dma_ok && ctx->mfma_split == 2) {
            // the contraction on the bf16 MFMA, X and the draws as two bf16 terms (bsc_ctx_set_mfma_split; three
            // terms of the draws do not fit the LDS next to the rings: that setting takes the f32 route here)
            if (D == LD) hipLaunchKernelGGL((logreg_loglik_dma_bx_kernel<true>), BSC_LL_ARGS);
            else hipLaunchKernelGGL((logreg_loglik_dma_bx_kernel<false>), BSC_LL_ARGS);
        } else if (dma_ok) {
            if (D == LD) hipLaunchKernelGGL((logreg_loglik_dma_kernel<true, 4>), BSC_LL_ARGS);
            else hipLaunchKernelGGL((logreg_loglik_dma_kernel<false, 4>), BSC_LL_ARGS);
        } else if (nsb == 1) BSC_LLX_D(1);
        else if (nsb == 2) BSC_LLX_D(2);
        else if (nsb == 4) BSC_LLX_D(4);
        else BSC_LLX_D(8);
#undef BSC_LLX_D
#undef BSC_LL_ARGS
    }
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(loglik_reduce_f64_kernel, dim3((S + 3) / 4), dim3(256), 0, ctx->stream,
                       (const double*)ws, n_blocks, (int)S, ell);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_bbvi_sample(bsc_ctx* ctx, const double* lam, int32_t D, int32_t G, int32_t S, uint64_t seed,
                    uint32_t step, double* eps, float* Wz, float* Bz, double* zeta) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(lam && eps && Wz && Bz && zeta && D >= 1 && G >= 1 && S >= 1,
                "bsc_bbvi_sample: bad arguments");
    const int n = S * ((D + G + 1 + 3) / 4);
    hipLaunchKernelGGL(bbvi_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, lam,
                       (int)D, (int)G, (int)S, seed, step, eps, Wz, Bz, zeta);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_bbvi_grad(bsc_ctx* ctx, const double* lam, const double* eps, const double* ell, int32_t D,
                  int32_t G, int32_t S, double scale, double a0, double b0, double* elbo,
                  double* grad, double* f_out) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(lam && eps && ell && elbo && grad, "bsc_bbvi_grad: null pointer");
    BSC_REQUIRE(D >= 1 && G >= 1 && S >= 2 && S <= BB_MAX_S, "bsc_bbvi_grad: D=%d G=%d S=%d (2..%d)",
                D, G, S, BB_MAX_S);
    BSC_REQUIRE(a0 > 0 && b0 > 0, "bsc_bbvi_grad: a0, b0 must be positive");
    const double log_prior_const = a0 * log(b0) - lgamma(a0);
    const int P = D + G + 1;
    const int n_parts = (P + BM_PARAMS - 1) / BM_PARAMS;
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)(BB_MAX_S + 2 * n_parts) * sizeof(double), &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    double* f = (double*)ws;
    double* cov_part = f + BB_MAX_S;
    double* var_part = cov_part + n_parts;
    hipLaunchKernelGGL(bbvi_f_kernel, dim3((unsigned)S), dim3(BB_BLOCK), 0, ctx->stream, lam, eps, ell,
                       (int)D, (int)G, scale, a0, b0, log_prior_const, f, f_out);
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(bbvi_moments_kernel, dim3((unsigned)n_parts), dim3(BB_BLOCK), 0, ctx->stream,
                       lam, eps, f, P, (int)S, cov_part, var_part);
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(bbvi_finish_kernel, dim3((unsigned)((2 * P + BB_BLOCK - 1) / BB_BLOCK)), dim3(BB_BLOCK), 0, ctx->stream,
                       lam, eps, f, P, (int)S, cov_part, var_part, n_parts, elbo, grad);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_bbvi_update(bsc_ctx* ctx, double* lam, double* eps, const double* ell, int32_t D, int32_t G, int32_t S,
                    double scale, double a0, double b0, double* m1, double* m2, int64_t t, double lr,
                    double beta1, double beta2, double eps_adam, uint64_t seed, uint32_t next_step, float* Wz,
                    float* Bz, double* zeta, double* elbo, double* grad, double* f_out) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(lam && eps && ell && m1 && m2 && Wz && Bz && zeta && elbo && grad, "bsc_bbvi_update: null pointer");
    BSC_REQUIRE(D >= 1 && G >= 1 && S >= 2 && S <= BB_MAX_S, "bsc_bbvi_update: D=%d G=%d S=%d (2..%d)", D, G, S,
                BB_MAX_S);
    BSC_REQUIRE(a0 > 0 && b0 > 0 && t >= 1, "bsc_bbvi_update: a0, b0 must be positive and t >= 1");
    const double log_prior_const = a0 * log(b0) - lgamma(a0);
    const int P = D + G + 1;
    const int n_parts = (P + BM_PARAMS - 1) / BM_PARAMS;
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)(BB_MAX_S + 2 * n_parts) * sizeof(double), &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    double* f = (double*)ws;
    double* cov_part = f + BB_MAX_S;
    double* var_part = cov_part + n_parts;
    hipLaunchKernelGGL(bbvi_f_kernel, dim3((unsigned)S), dim3(BB_BLOCK), 0, ctx->stream, (const double*)lam,
                       (const double*)eps, ell, (int)D, (int)G, scale, a0, b0, log_prior_const, f, f_out);
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(bbvi_moments_kernel, dim3((unsigned)n_parts), dim3(BB_BLOCK), 0, ctx->stream,
                       (const double*)lam, (const double*)eps, (const double*)f, P, (int)S, cov_part, var_part);
    BSC_LAUNCH_CHECK();
    const double corr1 = 1.0 - pow(beta1, (double)t), corr2 = 1.0 - pow(beta2, (double)t);
    hipLaunchKernelGGL(bbvi_update_kernel, dim3((unsigned)((P + BU_PARAMS - 1) / BU_PARAMS)), dim3(BB_BLOCK), 0,
                       ctx->stream, lam, eps, (const double*)f, (int)D, (int)G, (int)S, (const double*)cov_part,
                       (const double*)var_part, n_parts, m1, m2, lr, beta1, beta2, eps_adam, corr1, corr2, seed,
                       next_step, Wz, Bz, zeta, elbo, grad);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

}  // extern "C"
