// Row softmax of a tall-skinny product: R = softmax_rows(alpha A . B), lse, sum_c R * (alpha A . B).
//
// The local step of a mixture with exponential-family components (README.md:43: the discrete
// latent's factor; bayesic/distribution/base.py:47-69 for the likelihood split): the logits of row
// n are features(x_n) . coefficients -- a [rows, K] x [K, N] product with K, N <= 64 and millions of
// rows -- and only their softmax is wanted.  As two launches (the executor's _tensordot, then
// bsc_softmax_rows) the [rows, N] logits are written, read and written again; here a wave owns
// 32-row tiles, forms their logits with v_mfma_f32_32x32x2_f32 (the forward half of
// csrc/bsc_mog.hip's E-step: A(B^T in registers) * B(row features: lane = row), so that lane
// (row, half) holds 32 of the row's N logits), takes the softmax in-lane plus ONE permlane32 swap,
// and writes the responsibilities through wave-private LDS as whole rows.  HBM-bound: A once,
// R once.
//
// k-steps pair column s with column K/2 + s: the lower half-wave contracts the first half of a
// row, the upper half-wave the second half, so every lane loads exactly the K/2 contiguous floats
// it multiplies (K % 8 == 0).
#include "bsc_common.h"

namespace {

constexpr int RS_T = 32;                 // rows per tile
constexpr int RS_N = 64;                 // columns of the result (padded)
constexpr int RS_KH = 32;                // k-steps (K / 2) at most
constexpr int RS_BLOCK = 256;
constexpr int RS_WAVES = RS_BLOCK / BSC_WAVE;
constexpr int RS_STRIDE = RS_N + 4;      // LDS row stride of the staged tile [row][column]

typedef float rs_f32x16 __attribute__((ext_vector_type(16)));
typedef float rs_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int rs_drow(int q, int lane) {  // C/D row of accumulator register q
    return (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ float rs_swap32_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float rs_swap32_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int KHT>
struct RsRow {
    float x[KHT];
};

// lane (row = l & 31, half = l >> 5) loads columns half * K/2 .. + K/2 - 1 of its row; rows past the
// end read 0 through the descriptor
template <int KHT>
__device__ __forceinline__ void rs_load(RsRow<KHT>& t, const float* __restrict__ A, int64_t lda, int64_t row0,
                                        int64_t rows, int K, int lane) {
    const int64_t rem = rows - row0;
    uint64_t bytes = 0;
    if (rem > 0) bytes = ((uint64_t)(rem - 1) * (uint64_t)lda + (uint64_t)K) * 4u;
    const unsigned rec = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes;
    const int64_t safe0 = rem > 0 ? row0 : 0;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(A + safe0 * lda), 0, rec, 0x00020000);
    const int kh = K >> 1;
    const int off = (lane & 31) * (int)(lda * 4) + (lane >> 5) * kh * 4;
#pragma unroll
    for (int c4 = 0; c4 < KHT / 4; ++c4) {  // unconditional: a load inside a branch makes hipcc wait with vmcnt(0)
        auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16 * c4, 0, 2);   // nt: read once
        t.x[4 * c4 + 0] = __uint_as_float(v[0]);
        t.x[4 * c4 + 1] = __uint_as_float(v[1]);
        t.x[4 * c4 + 2] = __uint_as_float(v[2]);
        t.x[4 * c4 + 3] = __uint_as_float(v[3]);
    }
}

// KHT = K / 2 exactly: one instantiation per K = 8, 16, .. 64, so that neither the loads nor the MFMAs
// sit inside a branch (hipcc answers a conditional load with s_waitcnt vmcnt(0) before every use --
// the prefetch was waited for the moment it was issued, 1.18 ms instead of 0.8 at 10M x 40 -> 64).
template <int KHT>
__global__ __launch_bounds__(RS_BLOCK, 2) void gemm_softmax_rows_kernel(
    const float* __restrict__ A, int64_t lda, int64_t rows, int K, const float* __restrict__ B,
    int64_t ldbk, int64_t ldbn, int N, float alpha, float* __restrict__ R, int64_t ldr,
    float* __restrict__ lse, float* __restrict__ cross, int n_iter) {
    __shared__ __attribute__((aligned(16))) float lds[RS_WAVES * RS_T * RS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    float* rt = lds + wave * (RS_T * RS_STRIDE);
    const int kh = K >> 1;

    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    // A operand of the MFMA: alpha * B[k = half * K/2 + s][column 32 cb + l31], in log2 units
    float wreg[2][KHT];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = 32 * cb + l31;
#pragma unroll
        for (int s = 0; s < KHT; ++s) {
            float v = 0.f;
            if (s < kh && col < N) v = B[(int64_t)(half * kh + s) * ldbk + (int64_t)col * ldbn] * (alpha * LOG2E);
            wreg[cb][s] = v;
        }
    }

    const int64_t stride = (int64_t)gridDim.x * RS_WAVES;
    int64_t tile = (int64_t)blockIdx.x * RS_WAVES + wave;
    // Three row buffers in rotation: the tile being multiplied and the next TWO in flight -- with one
    // tile (5 KB per wave at K = 40) in flight the kernel sat at 3.7 TB/s, bound by memory latency.
    RsRow<KHT> xa, xb, xc;
#pragma unroll
    for (int s = 0; s < KHT; ++s) { xa.x[s] = 0.f; xb.x[s] = 0.f; xc.x[s] = 0.f; }
    rs_load(xa, A, lda, tile * RS_T, rows, K, lane);
    rs_load(xb, A, lda, (tile + stride) * RS_T, rows, K, lane);
    auto one_tile = [&](const RsRow<KHT>& cur, RsRow<KHT>& nxt) {
        rs_load(nxt, A, lda, (tile + 2 * stride) * RS_T, rows, K, lane);   // unconditional prefetch
        const int64_t row0 = tile * RS_T;
        rs_f32x16 logit[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) { logit[0][q] = 0.f; logit[1][q] = 0.f; }
#pragma unroll
        for (int s = 0; s < KHT; ++s) {
            logit[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[0][s], cur.x[s], logit[0], 0, 0, 0);
            logit[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[1][s], cur.x[s], logit[1], 0, 0, 0);
        }
        if (N < RS_N) {                    // padded columns take no part in the softmax
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (32 * cb + rs_drow(q, lane) >= N) logit[cb][q] = -1.0e30f;
        }
        // softmax over the row's columns: 32 in this lane, 32 in lane ^ 32
        // rows of this tile that exist (0 .. 32): everything below is 32-bit
        const int64_t left = rows - row0;
        const int n_valid = left >= RS_T ? RS_T : (left > 0 ? (int)left : 0);
        const bool valid = l31 < n_valid;
        float m = -3.0e38f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; ++q) m = fmaxf(m, logit[cb][q]);
        m = rs_swap32_max(m);
        const rs_f32x2 m2 = {m, m};
        rs_f32x2 ssum2 = {0.f, 0.f}, csum2 = {0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const rs_f32x2 d = rs_f32x2{logit[cb][q], logit[cb][q + 1]} - m2;
                const rs_f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                logit[cb][q] = e[0];
                logit[cb][q + 1] = e[1];
                ssum2 += e;
                // (d = -1e30 on a padded column: e = 0 and 0 * -1e30 = -0)
                csum2 = __builtin_elementwise_fma(e, d, csum2);
            }
        const float ssum = rs_swap32_sum(ssum2[0] + ssum2[1]);
        const float csum = rs_swap32_sum(csum2[0] + csum2[1]);
        const float inv = 1.0f / ssum;
        if (valid && half == 0) {
            // natural-log units: logit = LN2 * log2-logit
            float* lse_t = lse + row0;         // (uniform base, 32-bit lane offset)
            lse_t[l31] = LN2 * (m + __builtin_amdgcn_logf(ssum));
            if (cross) {
                float* cross_t = cross + row0;
                cross_t[l31] = LN2 * (csum * inv + m);
            }
        }
        // responsibilities -> LDS as [row][column] (registers 4g .. 4g+3 are 4 consecutive columns)
        const rs_f32x2 inv2 = {inv, inv};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const rs_f32x2 a = rs_f32x2{logit[cb][4 * gq], logit[cb][4 * gq + 1]} * inv2;
                const rs_f32x2 b = rs_f32x2{logit[cb][4 * gq + 2], logit[cb][4 * gq + 3]} * inv2;
                *reinterpret_cast<float4*>(rt + l31 * RS_STRIDE + 32 * cb + 8 * gq + 4 * half) =
                    make_float4(a[0], a[1], b[0], b[1]);
            }
        wave_lds_sync();
        // whole rows out: a quarter-wave per row (lane & 15 = sixteen-byte piece, lane >> 4 = row of
        // the group of four), eight groups per tile; no division, 32-bit offsets from the tile's base
        {
            float* Rt = R + row0 * ldr;
            const int p = lane & 15, rq = lane >> 4;
            const int ld = (int)ldr;
            const bool piece = 4 * p < N;
#pragma unroll
            for (int it = 0; it < RS_T / 4; ++it) {
                const int r = 4 * it + rq;
                if (piece && r < n_valid) {
                    const float4 v = *reinterpret_cast<const float4*>(rt + r * RS_STRIDE + 4 * p);
                    *reinterpret_cast<float4*>(Rt + r * ld + 4 * p) = v;
                }
            }
        }
        wave_lds_sync();   // the next tile overwrites rt
        tile += stride;
    };
    for (int it = 0; it < n_iter; it += 3) {   // n_iter is a multiple of 3 (host): the buffers rotate
        one_tile(xa, xc);
        one_tile(xb, xa);
        one_tile(xc, xb);
    }
}


// ---- ... and the statistics R^T . A in the same pass ------------------------------------------------
//
// The other half of a mixture's local step: every message the responsibilities send is a contraction of
// them with features of the same rows -- sum_n r_nk T_j(x_n) (bayesic/distribution/base.py:329-332: statistics
// of iid draws add up), i.e. column blocks of R^T . A when A = [T_1(x) | T_2(x) | .. | 1] is the wide operand
// the logits were formed from.  As launches of their own they re-read the [rows, N] responsibilities this
// kernel would have to write first (2.56 GB each way at 10M x 64); here the tile's responsibilities go from
// the softmax registers through wave-private LDS straight into the backward MFMAs
//     S[column c of R][feature f] += r[row][c] * A[row][f]        (v_mfma_f32_32x32x2_f32: A operand = r with
//     lane = column, B operand = features with lane = feature, two rows per instruction)
// -- csrc/bsc_mog.hip's E-step for ANY feature matrix of up to 64 columns.  R itself is written only when the
// caller passes a buffer (something else wants the responsibilities); lse comes back as ONE sum.
// Per 32-row tile: K forward + 32 (K <= 32) or 64 backward MFMAs.  Block partials [64][64] + 1 per workgroup,
// added in float64 in a fixed order by softmax_stats_reduce_kernel.
constexpr int ST_XS = RS_N + 4;          // LDS row stride of the staged features [row][feature], zero beyond K
constexpr int ST_WAVE_LDS = RS_T * RS_STRIDE + RS_T * ST_XS;
constexpr int ST_SLAB = RS_N * RS_N + 1; // [column of R][feature] + sum of lse

//
// BIAS: the logits carry a bias row (alpha * A . B + alpha * bias) -- the wide operand's ones column taken OUT of
// the product: it enters as the C operand of the first forward MFMA, and its statistic, the responsibilities'
// column sums, is accumulated per lane beside the softmax (slab column 63) instead of costing a second
// 32-feature block of backward MFMAs for one useful column.  [X | X^2 | 1 | pad] (40 columns, the diagonal
// Gaussian mixture) is then 32 + 32 MFMAs per tile -- csrc/bsc_mog.hip's count -- instead of 40 + 64.
__device__ __forceinline__ float rs_half32_allsum(float v) {
    v = row16_allsum(v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int KHT, bool WRITE_R, bool BIAS>
__global__ __launch_bounds__(RS_BLOCK, 2) void gemm_softmax_stats_kernel(
    const float* __restrict__ A, int64_t lda, int64_t rows, int K, const float* __restrict__ B,
    int64_t ldbk, int64_t ldbn, int N, float alpha, const float* __restrict__ bias, float* __restrict__ R,
    int64_t ldr, float* __restrict__ slab, int n_iter) {
    constexpr int FB = KHT > 16 ? 2 : 1;             // 32-feature blocks of the statistics
    __shared__ __attribute__((aligned(16))) float lds[RS_WAVES * ST_WAVE_LDS > RS_WAVES * ST_SLAB
                                                          ? RS_WAVES * ST_WAVE_LDS : RS_WAVES * ST_SLAB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    float* rt = lds + wave * ST_WAVE_LDS;
    float* xt = rt + RS_T * RS_STRIDE;
    const int kh = K >> 1;

    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    float wreg[2][KHT];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = 32 * cb + l31;
#pragma unroll
        for (int s = 0; s < KHT; ++s) {
            float v = 0.f;
            if (s < kh && col < N) v = B[(int64_t)(half * kh + s) * ldbk + (int64_t)col * ldbn] * (alpha * LOG2E);
            wreg[cb][s] = v;
        }
    }
    // the feature tile's padding (columns K .. 63) is zero for the whole kernel: only columns < K are staged
    for (int i = lane; i < RS_T * ST_XS; i += BSC_WAVE) xt[i] = 0.f;
    wave_lds_sync();
    // the bias of the column each accumulator register holds: kept in LDS as [column block][half][register] and
    // read back per tile as the forward product's C operand (4 + 4 ds_read_b128) -- 32 registers this kernel does
    // not have (wreg, two row buffers, logits, statistics and column sums are 224 at K = 32)
    __shared__ __attribute__((aligned(16))) float bias_s[BIAS ? 64 : 4];
    rs_f32x16 rsum[BIAS ? 2 : 1];
    if constexpr (BIAS) {
        if (tid < 64) {
            const int cb = tid >> 5, hf = (tid >> 4) & 1, q = tid & 15;
            const int col = 32 * cb + (q & 3) + 8 * (q >> 2) + 4 * hf;
            bias_s[tid] = col < N ? bias[(int64_t)col * ldbn] * (alpha * LOG2E) : -1.0e30f;
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; ++q) rsum[cb][q] = 0.f;
        __syncthreads();
    }

    rs_f32x16 S[2][FB];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
#pragma unroll
            for (int q = 0; q < 16; ++q) S[cb][fb][q] = 0.f;
    float lse_acc = 0.f;

    const int64_t stride = (int64_t)gridDim.x * RS_WAVES;
    int64_t tile = (int64_t)blockIdx.x * RS_WAVES + wave;
    RsRow<KHT> xa, xb;
#pragma unroll
    for (int s = 0; s < KHT; ++s) { xa.x[s] = 0.f; xb.x[s] = 0.f; }
    rs_load(xa, A, lda, tile * RS_T, rows, K, lane);
    auto one_tile = [&](const RsRow<KHT>& cur, RsRow<KHT>& nxt) {
        rs_load(nxt, A, lda, (tile + stride) * RS_T, rows, K, lane);   // unconditional prefetch
        const int64_t row0 = tile * RS_T;
        rs_f32x16 logit[2];
        if constexpr (BIAS) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = *reinterpret_cast<const float4*>(bias_s + 32 * cb + 16 * half + 4 * g);
                    logit[cb][4 * g] = v.x; logit[cb][4 * g + 1] = v.y; logit[cb][4 * g + 2] = v.z; logit[cb][4 * g + 3] = v.w;
                }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) { logit[0][q] = 0.f; logit[1][q] = 0.f; }
        }
#pragma unroll
        for (int s = 0; s < KHT; ++s) {
            logit[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[0][s], cur.x[s], logit[0], 0, 0, 0);
            logit[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[1][s], cur.x[s], logit[1], 0, 0, 0);
        }
        if (!BIAS && N < RS_N) {           // (with a bias row the padded columns carry -1e30 in it)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (32 * cb + rs_drow(q, lane) >= N) logit[cb][q] = -1.0e30f;
        }
        const int64_t left = rows - row0;
        const int n_valid = left >= RS_T ? RS_T : (left > 0 ? (int)left : 0);
        const bool valid = l31 < n_valid;
        float m = -3.0e38f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; ++q) m = fmaxf(m, logit[cb][q]);
        m = rs_swap32_max(m);
        const rs_f32x2 m2 = {m, m};
        rs_f32x2 ssum2 = {0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const rs_f32x2 d = rs_f32x2{logit[cb][q], logit[cb][q + 1]} - m2;
                const rs_f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                logit[cb][q] = e[0];
                logit[cb][q + 1] = e[1];
                ssum2 += e;
            }
        const float ssum = rs_swap32_sum(ssum2[0] + ssum2[1]);
        const float inv = valid ? 1.0f / ssum : 0.f;          // rows past the end contribute nothing
        if (valid && half == 0) lse_acc += LN2 * (m + __builtin_amdgcn_logf(ssum));
        // responsibilities -> LDS as [row][column]; the row's features as [row][feature]
        const rs_f32x2 inv2 = {inv, inv};
        if constexpr (WRITE_R) {
            // R is wanted: the normalised responsibilities are staged (and copied out below), the features as they are
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const rs_f32x2 a = rs_f32x2{logit[cb][4 * gq], logit[cb][4 * gq + 1]} * inv2;
                    const rs_f32x2 b = rs_f32x2{logit[cb][4 * gq + 2], logit[cb][4 * gq + 3]} * inv2;
                    if constexpr (BIAS) {
                        rsum[cb][4 * gq] += a[0];
                        rsum[cb][4 * gq + 1] += a[1];
                        rsum[cb][4 * gq + 2] += b[0];
                        rsum[cb][4 * gq + 3] += b[1];
                    }
                    *reinterpret_cast<float4*>(rt + l31 * RS_STRIDE + 32 * cb + 8 * gq + 4 * half) =
                        make_float4(a[0], a[1], b[0], b[1]);
                }
#pragma unroll
            for (int c4 = 0; c4 < KHT / 4; ++c4)
                *reinterpret_cast<float4*>(xt + l31 * ST_XS + half * kh + 4 * c4) =
                    make_float4(cur.x[4 * c4], cur.x[4 * c4 + 1], cur.x[4 * c4 + 2], cur.x[4 * c4 + 3]);
        } else {
            // only the statistics are wanted: the row's 1 / sum goes into its K / 2 staged features instead of
            // its 32 responsibilities (csrc/bsc_mog.hip), the unnormalised e are staged straight from their registers
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if constexpr (BIAS) {      // column sums: per-lane (per row slot) partials, folded over the lanes at the end
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        const rs_f32x2 r = __builtin_elementwise_fma(rs_f32x2{logit[cb][q], logit[cb][q + 1]}, inv2,
                                                                     rs_f32x2{rsum[cb][q], rsum[cb][q + 1]});
                        rsum[cb][q] = r[0];
                        rsum[cb][q + 1] = r[1];
                    }
                }
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(rt + l31 * RS_STRIDE + 32 * cb + 8 * gq + 4 * half) =
                        make_float4(logit[cb][4 * gq], logit[cb][4 * gq + 1], logit[cb][4 * gq + 2], logit[cb][4 * gq + 3]);
            }
#pragma unroll
            for (int c4 = 0; c4 < KHT / 4; ++c4) {
                const rs_f32x2 fa = rs_f32x2{cur.x[4 * c4], cur.x[4 * c4 + 1]} * inv2;
                const rs_f32x2 fb2 = rs_f32x2{cur.x[4 * c4 + 2], cur.x[4 * c4 + 3]} * inv2;
                *reinterpret_cast<float4*>(xt + l31 * ST_XS + half * kh + 4 * c4) = make_float4(fa[0], fa[1], fb2[0], fb2[1]);
            }
        }
        wave_lds_sync();
        // backward: S[column][feature] += r[row][column] * A[row][feature], two rows per MFMA
#pragma unroll
        for (int t = 0; t < RS_T / 2; ++t) {
            const int row = 2 * t + half;
            const float r0 = rt[row * RS_STRIDE + l31];
            const float r1 = rt[row * RS_STRIDE + 32 + l31];
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) {
                const float b = xt[row * ST_XS + 32 * fb + l31];
                S[0][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0, b, S[0][fb], 0, 0, 0);
                S[1][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(r1, b, S[1][fb], 0, 0, 0);
            }
        }
        if constexpr (WRITE_R) {
            float* Rt = R + row0 * ldr;
            const int p = lane & 15, rq = lane >> 4;
            const int ld = (int)ldr;
            const bool piece = 4 * p < N;
#pragma unroll
            for (int it = 0; it < RS_T / 4; ++it) {
                const int r = 4 * it + rq;
                if (piece && r < n_valid) {
                    const float4 v = *reinterpret_cast<const float4*>(rt + r * RS_STRIDE + 4 * p);
                    *reinterpret_cast<float4*>(Rt + r * ld + 4 * p) = v;
                }
            }
        }
        wave_lds_sync();   // the next tile overwrites rt and xt
        tile += stride;
    };
    for (int it = 0; it < n_iter; it += 2) {   // n_iter is even (host): the row buffers alternate
        one_tile(xa, xb);
        one_tile(xb, xa);
    }

    // block reduction: per wave [column][feature] + the lse sum, then a fixed-order sum over the waves
    __syncthreads();
    float* ep = lds + wave * ST_SLAB;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int col = 32 * cb + rs_drow(q, lane);
                ep[col * RS_N + 32 * fb + l31] = fb < FB ? S[cb][fb < FB ? fb : 0][q] : 0.f;
            }
    if constexpr (BIAS) {                  // column sums into feature slot 63 (K <= 56 with a bias row)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float r = rs_half32_allsum(rsum[cb][q]);
                if (l31 == 0) ep[(32 * cb + rs_drow(q, lane)) * RS_N + 63] = r;
            }
    }
    const float l = wave_allsum(lse_acc);
    if (lane == 0) ep[RS_N * RS_N] = l;
    __syncthreads();
    float* out = slab + (int64_t)blockIdx.x * ST_SLAB;
    for (int i = tid; i < ST_SLAB; i += RS_BLOCK) {
        float v = lds[i];
#pragma unroll
        for (int k = 1; k < RS_WAVES; ++k) v += lds[k * ST_SLAB + i];
        out[i] = v;
    }
}

// float64, fixed-order sum of the block partials; compacts [64][64] to [N][K] (leading dimension lds_)
// (`colsum`: slab feature slot 63 holds the responsibilities' column sums: they go to stats[c, K])
__global__ __launch_bounds__(1024) void softmax_stats_reduce_kernel(const float* __restrict__ slab, int n_rows, int N,
                                                                   int K, float* __restrict__ stats, int64_t ldst,
                                                                   double* __restrict__ lse_sum, int colsum) {
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double sum = 0.0;
    if (i < ST_SLAB) {
        constexpr int BATCH = 16;
        for (int b0 = wave; b0 < n_rows; b0 += 16 * BATCH) {
            float v[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int b = b0 + 16 * j;
                v[j] = b < n_rows ? slab[(int64_t)b * ST_SLAB + i] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) sum += (double)v[j];
        }
    }
    part[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && i < ST_SLAB) {
        double tot = part[0][lane];
#pragma unroll
        for (int k = 1; k < 16; ++k) tot += part[k][lane];
        if (i == RS_N * RS_N) {
            if (lse_sum) lse_sum[0] = tot;
        } else {
            const int col = i / RS_N, f = i % RS_N;
            if (col < N && f < K) stats[(int64_t)col * ldst + f] = (float)tot;
            else if (col < N && colsum && f == 63) stats[(int64_t)col * ldst + K] = (float)tot;
        }
    }
}

}  // namespace

extern "C" {

int bsc_gemm_softmax_rows(bsc_ctx* ctx, const float* A, int64_t lda, int64_t rows, int32_t K,
                          const float* B, int64_t ldbk, int64_t ldbn, int32_t N, float alpha, float* R,
                          int64_t ldr, float* lse, float* cross) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(rows >= 0 && K > 0 && N > 0, "bsc_gemm_softmax_rows: rows=%lld K=%d N=%d",
                (long long)rows, K, N);
    BSC_REQUIRE((A && R && lse) || rows == 0, "bsc_gemm_softmax_rows: null pointer");
    BSC_REQUIRE(B != nullptr, "bsc_gemm_softmax_rows: B is null");
    if (K > 2 * RS_KH || K % 8 != 0 || N > RS_N || N % 4 != 0)
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_gemm_softmax_rows: K=%d (<= 64, multiple of 8) N=%d (<= 64, multiple of 4)", K, N);
    BSC_REQUIRE(lda >= K && lda % 4 == 0 && lda < ((int64_t)1 << 26) && ldr >= N && ldr % 4 == 0,
                "bsc_gemm_softmax_rows: lda=%lld ldr=%lld", (long long)lda, (long long)ldr);
    BSC_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)R) & 15) == 0,
                "bsc_gemm_softmax_rows: A and R must be 16-byte aligned");
    if (rows == 0) return BSC_OK;
    const int64_t n_tiles = (rows + RS_T - 1) / RS_T;
    const int64_t max_waves = (int64_t)2 * 4 * ctx->cu_count;          // two waves per SIMD
    int64_t n_iter = (n_tiles + max_waves - 1) / max_waves;
    n_iter = (n_iter + 2) / 3 * 3;                                       // the three row buffers rotate
    const int64_t waves = (n_tiles + n_iter - 1) / n_iter;
    const int64_t blocks = (waves + RS_WAVES - 1) / RS_WAVES;
    bsc_prof_scope prof(ctx);
#define BSC_RS(KHT_)                                                                                       \
    hipLaunchKernelGGL(gemm_softmax_rows_kernel<KHT_>, dim3((unsigned)blocks), dim3(RS_BLOCK), 0, ctx->stream, \
                       A, lda, rows, (int)K, B, ldbk, ldbn, (int)N, alpha, R, ldr, lse, cross, (int)n_iter)
    switch (K / 2) {
        case 4: BSC_RS(4); break;
        case 8: BSC_RS(8); break;
        case 12: BSC_RS(12); break;
        case 16: BSC_RS(16); break;
        case 20: BSC_RS(20); break;
        case 24: BSC_RS(24); break;
        case 28: BSC_RS(28); break;
        default: BSC_RS(32); break;
    }
#undef BSC_RS
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_gemm_softmax_stats(bsc_ctx* ctx, const float* A, int64_t lda, int64_t rows, int32_t K,
                           const float* B, int64_t ldbk, int64_t ldbn, int32_t N, float alpha, const float* bias,
                           float* R, int64_t ldr, float* stats, int64_t ldst, double* lse_sum) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(rows >= 0 && K > 0 && N > 0, "bsc_gemm_softmax_stats: rows=%lld K=%d N=%d",
                (long long)rows, K, N);
    BSC_REQUIRE((A || rows == 0) && B && stats, "bsc_gemm_softmax_stats: null pointer");
    if (K > 2 * RS_KH || K % 8 != 0 || N > RS_N || N % 4 != 0)
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_gemm_softmax_stats: K=%d (<= 64, multiple of 8) N=%d (<= 64, multiple of 4)", K, N);
    BSC_REQUIRE(lda >= K && lda % 4 == 0 && lda < ((int64_t)1 << 26) && ldst >= K + (bias ? 1 : 0),
                "bsc_gemm_softmax_stats: lda=%lld ldst=%lld", (long long)lda, (long long)ldst);
    if (bias && K > 56)
        return bsc_fail(BSC_ERR_UNSUPPORTED, "bsc_gemm_softmax_stats: a bias row needs K <= 56 (got %d)", K);
    BSC_REQUIRE((((uintptr_t)A) & 15) == 0, "bsc_gemm_softmax_stats: A must be 16-byte aligned");
    BSC_REQUIRE(!R || (ldr >= N && ldr % 4 == 0 && (((uintptr_t)R) & 15) == 0),
                "bsc_gemm_softmax_stats: R must be 16-byte aligned with ldr %% 4 == 0");
    const int64_t n_tiles = (rows + RS_T - 1) / RS_T;
    const int64_t max_waves = (int64_t)2 * 4 * ctx->cu_count;          // two waves per SIMD
    int64_t n_iter = n_tiles > 0 ? (n_tiles + max_waves - 1) / max_waves : 0;
    n_iter += n_iter & 1;                                                // the two row buffers alternate
    int64_t blocks = 1;
    if (n_iter > 0) {
        const int64_t waves = (n_tiles + n_iter - 1) / n_iter;
        blocks = (waves + RS_WAVES - 1) / RS_WAVES;
    }
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)blocks * ST_SLAB * sizeof(float), &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    {
        bsc_prof_scope prof(ctx);
#define BSC_ST1(KHT_, WR_, BI_)                                                                                      \
    hipLaunchKernelGGL((gemm_softmax_stats_kernel<KHT_, WR_, BI_>), dim3((unsigned)blocks), dim3(RS_BLOCK), 0,       \
                       ctx->stream, A, lda, rows, (int)K, B, ldbk, ldbn, (int)N, alpha, bias, R, ldr, (float*)ws,    \
                       (int)n_iter)
#define BSC_ST(KHT_)                                                                                                 \
    do {                                                                                                             \
        if (R && bias) BSC_ST1(KHT_, true, true);                                                                    \
        else if (R) BSC_ST1(KHT_, true, false);                                                                      \
        else if (bias) BSC_ST1(KHT_, false, true);                                                                   \
        else BSC_ST1(KHT_, false, false);                                                                            \
    } while (0)
        switch (K / 2) {
            case 4: BSC_ST(4); break;
            case 8: BSC_ST(8); break;
            case 12: BSC_ST(12); break;
            case 16: BSC_ST(16); break;
            case 20: BSC_ST(20); break;
            case 24: BSC_ST(24); break;
            case 28: BSC_ST(28); break;
            default: BSC_ST(32); break;
        }
#undef BSC_ST
#undef BSC_ST1
    }
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(softmax_stats_reduce_kernel, dim3((ST_SLAB + 63) / 64), dim3(1024), 0, ctx->stream,
                       (const float*)ws, (int)blocks, (int)N, (int)K, stats, ldst, lse_sum, bias ? 1 : 0);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

}  // extern "C"
