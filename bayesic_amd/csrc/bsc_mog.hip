// Mixture of Gaussians (BASELINE config 3): per-row marginalisation of the
// discrete latent by summation, fused with the responsibility-weighted
// sufficient statistics.
//
// ABSENT in the reference: spec is README.md:43,72 ("marginalise them by
// summation", also after mini-batching) with the exponential-family statistics of
// bayesic/distribution/base.py:329-332; in bayesic.algebra terms the two
// contractions are dot(F, W.T) and dot(R.T, F) (cfg 3).
//
// Per row n, with features f_n = [x_n (16) | x_n^2 (16)] and K <= 64 components:
//     logit_nk = c_k + sum_f W_kf f_nf                  (forward,  rows x 32 x 64)
//     r_nk     = softmax_k(logit_nk)
//     S_kf    += r_nk f_nf,  R_k += r_nk,  L += logsumexp_k(logit_nk)
//                                                       (backward, 64 x rows x 32)
// Both contractions are dense with no padding waste at K=64, D=16, so both run on
// v_mfma_f32_32x32x2_f32 (exact f32).  A wave owns 32-row tiles:
//   forward  D[comp][row]  = A(W, in registers) * B(features: lane = row, k = feature pair)
//            so lane (row, half) holds 32 of the row's 64 logits in its own registers
//   softmax  in-lane max/sum over those 32 + ONE permlane32 swap with the other half
//            (PMC showed the first version, with rows in registers and components on
//            lanes, spent 1240 VALU instructions per tile on cross-lane reductions
//            against 64 MFMAs: profiles/r01_e_mog_bbvi_pmc.txt)
//   r is transposed through wave-private LDS ([row][comp], 16-byte writes)
//   backward D[comp][feat] = A(r: lane = comp, k = row of the pair) * B(features from LDS)
#include "bsc_common.h"
#include "bsc_bf16split.h"

namespace {

constexpr int MK = 64;        // components (padded)
constexpr int MD = 16;        // data columns (padded)
constexpr int MF = 2 * MD;    // features
constexpr int MT = 32;        // rows per tile
constexpr int MOG_BLOCK = 256;
constexpr int MOG_WAVES = MOG_BLOCK / BSC_WAVE;
constexpr int XT_STRIDE = MF + 4;                    // LDS row stride of the staged feature tile [row][x/sum | x^2/sum]
constexpr int RT_STRIDE = MK + 4;                    // LDS row stride of the r tile [row][comp]
constexpr int WAVE_LDS = MT * XT_STRIDE + MT * RT_STRIDE;
constexpr int MOG_SLAB = MK * (1 + MF) + 1;          // [comp][R | S(32)] + L

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int drow(int q, int lane) {  // C/D row of accumulator register q
    return (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
}

// sum over the 32 lanes of each half-wave, result in every lane of the half
__device__ __forceinline__ float half32_allsum(float v) {
    v = row16_allsum(v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float swap32_max(float v) {   // max with the lane 32 away
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float swap32_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct XRow {
    float x[MD];
};

// lane l loads row (row0 + (l&31)): 16 floats; rows past the end and columns >= D read 0
// NT: non-temporal loads (the default: X is read once).  Both half-waves load the SAME 32 rows -- the lower half
// contracts x, the upper x^2 -- and with the non-temporal policy the L2 does not keep a line for the second
// requester: FETCH_SIZE reads 1.19 x the algorithmic bytes (profiles/pmc_traffic.json), 1.0 x with NT off
// (BSC_MOG_NT=0, profiles/r03_pmc_mog_nt.txt); the kernel is MFMA-bound either way (HBM time 80 of 710 us).
template <bool FULL, bool NT = true>   // FULL: D == 16, no column masking
__device__ __forceinline__ void load_rows(XRow& t, const float* __restrict__ X, int64_t ldx,
                                          int64_t row0, int64_t N, int D, int lane) {
    const int64_t rem = N - row0;
    uint64_t bytes = 0;
    if (rem > 0) bytes = ((uint64_t)(rem - 1) * (uint64_t)ldx + (uint64_t)D) * 4u;
    const unsigned rec = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes;
    const int64_t safe0 = rem > 0 ? row0 : 0;
    auto xs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + safe0 * ldx), 0, rec, 0x00020000);
    const int off = (lane & 31) * (int)(ldx * 4);
#pragma unroll
    for (int c4 = 0; c4 < MD / 4; ++c4) {
        auto v = __builtin_amdgcn_raw_buffer_load_b128(xs, off + 16 * c4, 0, NT ? 2 : 0);  // nt: read once
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float f = __uint_as_float(v[j]);
            if (!FULL && 4 * c4 + j >= D) f = 0.f;   // D < 16: those bytes belong to the next row
            t.x[4 * c4 + j] = f;
        }
    }
}

// VALU instructions issued between MFMAs cost the matrix pipe their full issue time on this
// part (profiles/r01_ubench_mfma_valu_mix.txt), so the softmax is written for instruction
// count: weights and biases are scaled by log2e once, so exp(l - m) is one subtract and one
// v_exp_f32 (and a one-hot row stays exactly one-hot); the 1/sum of a row is folded into
// the backward B operand when the row is staged (x/sum and x^2/sum, 32 multiplies per tile)
// rather than into its 64 responsibilities; row buffers alternate instead of being copied.
template <bool FULL, bool NT>
__global__ __launch_bounds__(MOG_BLOCK, 2) void mog_estep_kernel(
    const float* __restrict__ X, int64_t ldx, int64_t N, int D, const float* __restrict__ Wmat,
    const float* __restrict__ cvec, int K, float* __restrict__ slab, int n_iter) {
    __shared__ __attribute__((aligned(16))) float lds[MOG_WAVES * WAVE_LDS > MOG_WAVES * MOG_SLAB
                                                          ? MOG_WAVES * WAVE_LDS
                                                          : MOG_WAVES * MOG_SLAB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    float* xt = lds + wave * WAVE_LDS;
    float* rt = xt + MT * XT_STRIDE;

    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    // A operand of the forward product: W[comp = 32cb + l31][feature s + 16 half]
    float wreg[2][MF / 2];
    f32x16 bias_q[2];   // bias of the component each accumulator register holds
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int comp = 32 * cb + drow(q, lane);
            bias_q[cb][q] = comp < K ? cvec[comp] * LOG2E : -1.0e30f;
        }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int comp = 32 * cb + l31;
#pragma unroll
        for (int s = 0; s < MF / 2; ++s) {
            const int f = s + MD * half;           // lower half of the wave: x_s ; upper half: x_s^2
            const int d = f & (MD - 1);
            float v = 0.f;
            if (comp < K && d < D) v = Wmat[(int64_t)comp * 2 * D + (f < MD ? d : D + d)];
            wreg[cb][s] = v * LOG2E;   // logits come out in log2 units: exp is v_exp_f32(l - m)
        }
    }

    f32x16 S[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 16; ++q) S[cb][q] = 0.f;
    f32x16 rsum[2];   // per-lane (i.e. per-row-slot) partial sums of r, folded over lanes at the end
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 16; ++q) rsum[cb][q] = 0.f;
    float lse_acc = 0.f;

    const int64_t stride = (int64_t)gridDim.x * MOG_WAVES;
    int64_t tile = (int64_t)blockIdx.x * MOG_WAVES + wave;
    XRow xa, xb;
    load_rows<FULL, NT>(xa, X, ldx, tile * MT, N, D, lane);
    auto one_tile = [&](const XRow& cur, XRow& nxt) {
        load_rows<FULL, NT>(nxt, X, ldx, (tile + stride) * MT, N, D, lane);   // unconditional prefetch
        const int64_t row0 = tile * MT;
        // forward: logits[comp][row] -- lane (row = l31, half) gets comps 32cb + drow(q, lane);
        // the C operand of the first k-step carries the bias
        f32x16 logit[2];
        // k-step s contracts feature x_s in the lower half of the wave and x_s^2 in the upper:
        // one selector per column gives the operand here AND, times 1/sum, the staged backward
        // operand below (lower half writes x/sum, upper half x^2/sum)
        float feat[MD];
#pragma unroll
        for (int s = 0; s < MD; ++s) feat[s] = cur.x[s] * (half ? cur.x[s] : 1.0f);
#pragma unroll
        for (int s = 0; s < MF / 2; ++s) {
            logit[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[0][s], feat[s], s == 0 ? bias_q[0] : logit[0], 0, 0, 0);
            logit[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[1][s], feat[s], s == 0 ? bias_q[1] : logit[1], 0, 0, 0);
        }
        // softmax over the row's 64 components: 32 in this lane, 32 in lane ^ 32
        const bool valid = row0 + l31 < N;
        float m = -3.0e38f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; ++q) m = fmaxf(m, logit[cb][q]);
        m = swap32_max(m);
        // Pairs of values go through v_pk_add / v_pk_fma / v_pk_mul: a packed instruction takes
        // the same ~5 cycles from the MFMA pipe as a scalar one (profiles/r01_ubench_mfma_valu_mix.txt)
        const f32x2 m2 = {m, m};
        f32x2 ssum2 = {0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const f32x2 d = f32x2{logit[cb][q], logit[cb][q + 1]} - m2;
                const f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                logit[cb][q] = e[0];
                logit[cb][q + 1] = e[1];
                ssum2 += e;
            }
        const float ssum = swap32_sum(ssum2[0] + ssum2[1]);
        const float inv = valid ? 1.0f / ssum : 0.f;
        if (valid && half == 0) lse_acc += LN2 * (m + __builtin_amdgcn_logf(ssum));
        // unnormalised e -> LDS as [row][comp] (registers 4g..4g+3 are 4 consecutive components);
        // R_k accumulates e * inv
        const f32x2 inv2 = {inv, inv};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const f32x2 e = {logit[cb][q], logit[cb][q + 1]};
                const f32x2 r = __builtin_elementwise_fma(e, inv2, f32x2{rsum[cb][q], rsum[cb][q + 1]});
                rsum[cb][q] = r[0];
                rsum[cb][q + 1] = r[1];
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4*>(rt + l31 * RT_STRIDE + 32 * cb + 8 * gq + 4 * half) =
                    make_float4(logit[cb][4 * gq], logit[cb][4 * gq + 1], logit[cb][4 * gq + 2],
                                logit[cb][4 * gq + 3]);
        }
        // the backward B operand of the row, already divided by the row's sum: lanes 0-31 write
        // x / sum (features 0..15), lanes 32-63 x^2 / sum (features 16..31)
#pragma unroll
        for (int c4 = 0; c4 < MD / 4; ++c4) {
            const f32x2 fa = f32x2{feat[4 * c4], feat[4 * c4 + 1]} * inv2;
            const f32x2 fb = f32x2{feat[4 * c4 + 2], feat[4 * c4 + 3]} * inv2;
            *reinterpret_cast<float4*>(xt + l31 * XT_STRIDE + MD * half + 4 * c4) =
                make_float4(fa[0], fa[1], fb[0], fb[1]);
        }
        wave_lds_sync();
        // backward: S[comp][feat] += e[row][comp] * (f[row][feat] / sum[row]), two rows per MFMA
#pragma unroll
        for (int t = 0; t < MT / 2; ++t) {
            const int row = 2 * t + half;
            const float b = xt[row * XT_STRIDE + l31];
            const float r0 = rt[row * RT_STRIDE + l31];
            const float r1 = rt[row * RT_STRIDE + 32 + l31];
            S[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0, b, S[0], 0, 0, 0);
            S[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(r1, b, S[1], 0, 0, 0);
        }
        wave_lds_sync();   // the next tile overwrites xt and rt
        tile += stride;
    };
    for (int it = 0; it < n_iter; it += 2) {   // n_iter is even (host): the row buffers alternate
        one_tile(xa, xb);
        one_tile(xb, xa);
    }

    // block reduction: per wave [comp][R | S] + L, then fixed-order sum over waves
    __syncthreads();
    float* ep = lds + wave * MOG_SLAB;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int comp = 32 * cb + drow(q, lane);
            ep[comp * (1 + MF) + 1 + l31] = S[cb][q];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {            // R[comp]: fold the 32 row-lanes of this half
            const float r = half32_allsum(rsum[cb][q]);
            if (l31 == 0) ep[(32 * cb + drow(q, lane)) * (1 + MF)] = r;
        }
    }
    const float l = wave_allsum(lse_acc);   // one row per lane of the lower half
    if (lane == 0) ep[MK * (1 + MF)] = l;
    __syncthreads();
    float* out = slab + (int64_t)blockIdx.x * MOG_SLAB;
    for (int i = tid; i < MOG_SLAB; i += MOG_BLOCK) {
        float v = lds[i];
#pragma unroll
        for (int k = 1; k < MOG_WAVES; ++k) v += lds[k * MOG_SLAB + i];
        out[i] = v;
    }
}

// ---- the same pass on the operand-split bf16 route (ctx->mfma_split != 0; csrc/bsc_bf16split.h) ----
//
// Forward on v_mfma_f32_32x32x16_bf16 with THREE bf16 terms per operand (6 products) whatever the context asks for:
// a logit is c - tau (x - mu)^2 / 2 written out in x and x^2, a small difference of terms hundreds of times its
// size, and two terms (2^-17 of each TERM) would leave the responsibilities with 1e-3 of error.  Backward with two
// terms (3 products): sums of products of one sign class, r in (0, 1].  36 MFMAs of 32 cycles a tile against 64 of 64.
//   * a lane loads HALF a row (x[8 h ..], 32 bytes): element j of its k-step-0 fragment is x[8 h + j], of k-step 1
//     x[8 h + j]^2 -- the B operand (k = feature, n = row) with no data movement, no row read twice;
//   * the result has the row on the lane and the components in registers, as before: the softmax is unchanged;
//   * the backward sums over rows, i.e. over the LANE index of both its operands: the responsibilities r = e / sum
//     (two terms) and the features (the first two of the forward's three terms) go to wave-private LDS as bf16 terms, [row][32 components or features] with 64-byte rows, and
//     come back transposed by ds_read_b64_tr_b16 (cdna_hip_programming.md "An accumulator tile as the next MFMA's
//     operand").  8-byte chunk c of row r sits at chunk c ^ sw(r): sw = (r >> 1) & 7 for the e images (written 8
//     bytes a lane: two lanes a bank, the floor), 2 ((r >> 1) & 3) for the feature image (written 16 bytes a lane);
//     a transposed read's 32-lane half covers four whole rows = every bank once.
constexpr int BX_E_IMG = 2048;                       // [32 rows][32 comps] bf16
constexpr int BX_WAVE_LDS = (2 * 2 + 2) * BX_E_IMG;  // e: [cb][term]; features: [term]

template <bool FULL>
__global__ __launch_bounds__(MOG_BLOCK, 2) void mog_estep_bx_kernel(
    const float* __restrict__ X, int64_t ldx, int64_t N, int D, const float* __restrict__ Wmat,
    const float* __restrict__ cvec, int K, float* __restrict__ slab, int n_iter) {
    constexpr int LDS_BYTES = MOG_WAVES * BX_WAVE_LDS > MOG_WAVES * MOG_SLAB * 4 ? MOG_WAVES * BX_WAVE_LDS
                                                                                  : MOG_WAVES * MOG_SLAB * 4;
    __shared__ __attribute__((aligned(16))) char lds_raw[LDS_BYTES];
    float* const lds = reinterpret_cast<float*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    char* const img = lds_raw + wave * BX_WAVE_LDS;
    const unsigned img_addr = (unsigned)(uintptr_t)(bsc_lds_ptr)img;

    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    // A operand of the forward product, three terms: W[comp = 32 cb + l31][k-step 0: x columns 8 half + j; 1: x^2 columns]
    bsc_u32x4 wf[2][2][3];
    f32x16 bias_q[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int comp = 32 * cb + drow(q, lane);
            bias_q[cb][q] = comp < K ? cvec[comp] * LOG2E : -1.0e30f;
        }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int comp = 32 * cb + l31;
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 8 * half + j;
                w[j] = (comp < K && d < D) ? Wmat[(int64_t)comp * 2 * D + (ks ? D + d : d)] * LOG2E : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                unsigned pk[3];
                bsc_split_pk<3>(w[2 * t], w[2 * t + 1], pk);
#pragma unroll
                for (int c = 0; c < 3; ++c) wf[cb][ks][c][t] = pk[c];
            }
        }

    f32x16 S[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 16; ++q) S[cb][q] = 0.f;
    f32x16 rsum[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 16; ++q) rsum[cb][q] = 0.f;
    float lse_acc = 0.f;

    // where this lane writes its e registers 4 g .. 4 g + 3 (components 8 g + 4 half ..) and its features, and the
    // transposed-read addresses: lane 4 q + p of a 16-lane group points at row r0 + q, 8-byte chunk c of the row
    const int sw_e = (l31 >> 1) & 7, sw_g = 2 * ((l31 >> 1) & 3);
    unsigned we[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) we[g] = img_addr + (unsigned)(l31 * 64 + 8 * ((2 * g + half) ^ sw_e));
    const unsigned wg0 = img_addr + 4 * BX_E_IMG + (unsigned)(l31 * 64 + 8 * ((2 * half) ^ sw_g));          // x: chunks 2 h, 2 h + 1
    const unsigned wg1 = img_addr + 4 * BX_E_IMG + (unsigned)(l31 * 64 + 8 * ((4 + 2 * half) ^ sw_g));      // x^2
    const int gg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    // rows 8 (gg >> 1) + 4 e + tq (+ 16 s), chunk 4 (gg & 1) + tp
    unsigned re[2], rg[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 8 * (gg >> 1) + 4 * e + tq, ch = 4 * (gg & 1) + tp;
        re[e] = img_addr + (unsigned)(row * 64 + 8 * (ch ^ ((row >> 1) & 7)));
        rg[e] = img_addr + 4 * BX_E_IMG + (unsigned)(row * 64 + 8 * (ch ^ (2 * ((row >> 1) & 3))));
    }

    struct HalfRow { float x[8]; };
    auto load_half = [&](HalfRow& t, int64_t row0) __attribute__((always_inline)) {
        const int64_t rem = N - row0;
        uint64_t bytes = 0;
        if (rem > 0) bytes = ((uint64_t)(rem - 1) * (uint64_t)ldx + (uint64_t)D) * 4u;
        const unsigned rec = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes;
        const int64_t safe0 = rem > 0 ? row0 : 0;
        auto xs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + safe0 * ldx), 0, rec, 0x00020000);
        const int off = l31 * (int)(ldx * 4) + 32 * half;
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            auto v = __builtin_amdgcn_raw_buffer_load_b128(xs, off + 16 * c4, 0, 2);        // nt: every byte is read once
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f = __uint_as_float(v[j]);
                if (!FULL && 8 * half + 4 * c4 + j >= D) f = 0.f;
                t.x[4 * c4 + j] = f;
            }
        }
    };

    const int64_t stride = (int64_t)gridDim.x * MOG_WAVES;
    int64_t tile = (int64_t)blockIdx.x * MOG_WAVES + wave;
    HalfRow xa, xb;
    load_half(xa, tile * MT);
    auto one_tile = [&](const HalfRow& cur, HalfRow& nxt) __attribute__((always_inline)) {
        load_half(nxt, (tile + stride) * MT);
        const int64_t row0 = tile * MT;
        // ---- forward: B = [x | x^2] of the row in three terms
        float x2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x2[j] = cur.x[j] * cur.x[j];
        bsc_u32x4 fb[2][3];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            unsigned pk[3];
            bsc_split_pk<3>(cur.x[2 * t], cur.x[2 * t + 1], pk);
#pragma unroll
            for (int c = 0; c < 3; ++c) fb[0][c][t] = pk[c];
            bsc_split_pk<3>(x2[2 * t], x2[2 * t + 1], pk);
#pragma unroll
            for (int c = 0; c < 3; ++c) fb[1][c][t] = pk[c];
        }
        // the backward's feature operand = the first two of these terms, to the (free: the previous tile's backward ended
        // with a wave barrier) image right away, so that the fragments die with the forward product
#pragma unroll
        for (int c = 0; c < 2; ++c) {       // (fb[.][c][t] packs the columns 2 t, 2 t + 1 of this lane's half row: the image's order)
            asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(wg0), "v"(fb[0][c]), "n"(c * BX_E_IMG) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(wg1), "v"(fb[1][c]), "n"(c * BX_E_IMG) : "memory");
        }
        f32x16 logit[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            logit[cb] = bsc_mfma_split<3>(wf[cb][0], fb[0], bias_q[cb]);
            logit[cb] = bsc_mfma_split<3>(wf[cb][1], fb[1], logit[cb]);
        }
        // ---- softmax over the row's 64 components: 32 in this lane, 32 in lane ^ 32
        const bool valid = row0 + l31 < N;
        float m = -3.0e38f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; q += 2) m = __builtin_fmaxf(__builtin_fmaxf(logit[cb][q], logit[cb][q + 1]), m);
        m = swap32_max(m);
        float ssum = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                logit[cb][q] = __builtin_amdgcn_exp2f(logit[cb][q] - m);
                ssum += logit[cb][q];
            }
        ssum = swap32_sum(ssum);
        const float inv = valid ? 1.0f / ssum : 0.f;
        if (valid && half == 0) lse_acc += LN2 * (m + __builtin_amdgcn_logf(ssum));
        // ---- the responsibilities r = e / sum (two terms) and the features (the first two of the forward's three terms:
        // nothing to compute) to LDS.  Round 4: the 1 / sum went from the 16 features of a lane -- where it cost a second
        // split of x and x^2, 64 vector instructions -- to its 32 responsibilities (32 multiplies; packed ones were tried:
        // no faster, and MI355X_MICROARCH.md counts packed f32 arithmetic beside MFMAs as an anti-lever).
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                logit[cb][q] *= inv;
                rsum[cb][q] += logit[cb][q];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned p0[2], p1[2];
                bsc_split_pk<2>(logit[cb][4 * g], logit[cb][4 * g + 1], p0);
                bsc_split_pk<2>(logit[cb][4 * g + 2], logit[cb][4 * g + 3], p1);
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(we[g]), "v"(bsc_u32x2{p0[c], p1[c]}),
                                 "n"((2 * cb + c) * BX_E_IMG) : "memory");
            }
        }

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_lds_sync();
        // ---- backward: S[comp][feat] += sum over rows e[row][comp] (f[row][feat] / sum[row]), 16 rows a k-step
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bsc_u32x2 g0[2], g1[2], e0[2][2], e1[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                BSC_LDS_TR_B64(g0[c], rg[0], s * 1024 + c * BX_E_IMG);
                BSC_LDS_TR_B64(g1[c], rg[1], s * 1024 + c * BX_E_IMG);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    BSC_LDS_TR_B64(e0[cb][c], re[0], s * 1024 + (2 * cb + c) * BX_E_IMG);
                    BSC_LDS_TR_B64(e1[cb][c], re[1], s * 1024 + (2 * cb + c) * BX_E_IMG);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(g0[0]), "+v"(g0[1]), "+v"(g1[0]), "+v"(g1[1]), "+v"(e0[0][0]), "+v"(e0[0][1]), "+v"(e0[1][0]),
                           "+v"(e0[1][1]), "+v"(e1[0][0]), "+v"(e1[0][1]), "+v"(e1[1][0]), "+v"(e1[1][1]));
            bsc_u32x4 gb[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) gb[c] = bsc_u32x4{g0[c][0], g0[c][1], g1[c][0], g1[c][1]};
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                bsc_u32x4 ea[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) ea[c] = bsc_u32x4{e0[cb][c][0], e0[cb][c][1], e1[cb][c][0], e1[cb][c][1]};
                S[cb] = bsc_mfma_split<2>(ea, gb, S[cb]);
            }
        }
        wave_lds_sync();   // the next tile overwrites the images
        tile += stride;
    };
    for (int it = 0; it < n_iter; it += 2) {   // n_iter is even (host): the row buffers alternate
        one_tile(xa, xb);
        one_tile(xb, xa);
    }

    // block reduction: per wave [comp][R | S] + L, then fixed-order sum over waves (as mog_estep_kernel)
    __syncthreads();
    float* ep = lds + wave * MOG_SLAB;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int comp = 32 * cb + drow(q, lane);
            ep[comp * (1 + MF) + 1 + l31] = S[cb][q];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float r = half32_allsum(rsum[cb][q]);
            if (l31 == 0) ep[(32 * cb + drow(q, lane)) * (1 + MF)] = r;
        }
    }
    const float l = wave_allsum(lse_acc);
    if (lane == 0) ep[MK * (1 + MF)] = l;
    __syncthreads();
    float* out = slab + (int64_t)blockIdx.x * MOG_SLAB;
    for (int i = tid; i < MOG_SLAB; i += MOG_BLOCK) {
        float v = lds[i];
#pragma unroll
        for (int k = 1; k < MOG_WAVES; ++k) v += lds[k * MOG_SLAB + i];
        out[i] = v;
    }
}

// float64, fixed-order sum of the block partials; compacts [64][1+32] to [K][1+2D]
__global__ __launch_bounds__(1024) void mog_reduce_kernel(const float* __restrict__ slab, int n_rows,
                                                          int K, int D, double* __restrict__ stats,
                                                          double* __restrict__ lse) {
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double sum = 0.0;
    if (i < MOG_SLAB) {
        constexpr int BATCH = 16;
        for (int b0 = wave; b0 < n_rows; b0 += 16 * BATCH) {
            float v[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int b = b0 + 16 * j;
                v[j] = b < n_rows ? slab[(int64_t)b * MOG_SLAB + i] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) sum += (double)v[j];
        }
    }
    part[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && i < MOG_SLAB) {
        double tot = part[0][lane];
#pragma unroll
        for (int k = 1; k < 16; ++k) tot += part[k][lane];
        if (i == MK * (1 + MF)) {
            lse[0] = tot;
        } else {
            const int comp = i / (1 + MF), col = i % (1 + MF);
            if (comp < K) {
                if (col == 0) stats[(int64_t)comp * (1 + 2 * D)] = tot;
                else {
                    const int f = col - 1, d = f & (MD - 1);
                    if (d < D) stats[(int64_t)comp * (1 + 2 * D) + 1 + (f < MD ? d : D + d)] = tot;
                }
            }
        }
    }
}

#pragma clang fp contract(off)
__device__ __forceinline__ double digamma_f64(double x) { return bsc_digamma_f64(x); }

// eta layout: [alpha-1 (K) | kappa*m (K*D) | kappa (K*D) | 2a-1 (K*D) | 2b+kappa*m^2 (K*D)]
// One workgroup; thread k owns component k.
// 16 lanes per component, one column each (float64 digamma / log / divisions per column: a
// thread per component walking its D columns took 15 us); a component's terms are then added
// in a fixed butterfly order inside its 16-lane group.
//
// With eta0 (the prior's natural parameters, same layout) the kernel also forms the global part of
// the evidence lower bound (README.md:30-37; factor decomposition of bayesic/distribution/base.py:47-69),
//     bound = E_q[log p(pi, mu, tau)] - E_q[log q(pi, mu, tau)]
//           = sum over factors of  <eta0 - eta, E_q[T]> - A(eta0) + A(eta),
// from the very digamma / log values the logit coefficients need (oracle.svi.mog_global_bound):
// Dirichlet  T = log pi_k,  A = sum lnGamma(alpha_k) - lnGamma(sum alpha);
// NormalGamma T = (tau mu, -tau mu^2/2, log(tau)/2, -tau/2), A = lnGamma(a) - a log b - log(kappa)/2 + log(2 pi)/2.
// Fixed-order sums: 16-lane butterfly per component, components in index order by one thread.
// (psi(x), lnGamma(x)) of up to six arguments through ONE copy of the series' code: this kernel runs once per
// update behind a 0.7 ms pass, its instructions are in no cache, and straight-line float64 log / lgamma code is
// fetched at ~0.5 us per 64-byte line ("the finish kernel starts instruction-cache cold") -- with
// every call inlined at its own site the bound cost 15 us of fetch for 1 us of arithmetic.
struct PsiLg {
    double psi, lg;
};
__device__ __forceinline__ PsiLg psi_lgamma_f64(double x) {
#pragma clang fp contract(off)
    double P = 1.0, dP = 0.0;
    while (x < 8.0) {
        dP = dP * x + P;
        P *= x;
        x += 1.0;
    }
    const double rden = 1.0 / (P * x);
    const double inv = P * rden, inv2 = inv * inv, logy = log(x), logP = log(P);
    const double ps = inv2 * (1.0 / 12.0 - inv2 * (1.0 / 120.0 - inv2 * (1.0 / 252.0 - inv2 *
                      (1.0 / 240.0 - inv2 * (5.0 / 660.0 - inv2 * (691.0 / 32760.0))))));
    const double ls = inv * (1.0 / 12.0 - inv2 * (1.0 / 360.0 - inv2 * (1.0 / 1260.0 - inv2 *
                      (1.0 / 1680.0 - inv2 * (1.0 / 1188.0 - inv2 * (691.0 / 360360.0))))));
    PsiLg r;
    r.psi = logy - 0.5 * inv - ps - dP * x * rden;
    r.lg = (x - 0.5) * logy - x + 0.91893853320467274178032973640562 + ls - logP;
    return r;
}

// `prior_A` = A(eta0), the prior's log-normaliser: a constant of the model, taken ONCE (this kernel with
// eta = eta0 and `A_out`), not per update.  Per update a lane then evaluates (psi, lnGamma) twice -- its
// cell's a, and alpha_k (the component's first lane) or sum alpha (every other lane; the first lane takes
// psi(sum alpha) from its neighbour) -- and two logs.
__global__ __launch_bounds__(1024) void mog_expected_params_kernel(const double* __restrict__ eta,
                                                                   const double* __restrict__ eta0,
                                                                   const double* __restrict__ prior_A,
                                                                   int K, int D,
                                                                   float* __restrict__ Wmat,
                                                                   float* __restrict__ cvec,
                                                                   double* __restrict__ bound,
                                                                   double* __restrict__ A_out) {
    __shared__ double alpha_sum;
    __shared__ double comp_bound[1024], comp_A[1024];
    const int dl = threadIdx.x & 15;
    const double LOG_2PI = 1.8378770664093454835606594728112;
    const bool with_bound = eta0 != nullptr;
    if (threadIdx.x < 64) {   // one wave: the Dirichlet's total, fixed order
        double a = 0.0;
        for (int j = threadIdx.x; j < K; j += 64) a += eta[j] + 1.0;
        const double tot = wave_allsum_f64(a);
        if (threadIdx.x == 0) alpha_sum = tot;
    }
    __syncthreads();
    const int64_t KD = (int64_t)K * D;
    double lg_sum = 0.0;      // lnGamma(sum alpha), as seen by the second lane of every group
    // 16 lanes per component, 64 components per sweep (the trip count is uniform over a wave's
    // four 16-lane groups only up to the tail, so the shuffles below stay inside a group)
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + (threadIdx.x >> 4);
        double c = 0.0, gb = 0.0, ga = 0.0;
        for (int d0 = 0; d0 < D; d0 += 16) {              // (uniform trip count: the series' loop below is shared)
            const int d = d0 + dl;
            const bool cell = k < K && d < D;             // this lane has a (component, column) cell in this trip
            const bool head = k < K && dl == 0 && d0 == 0;    // ... and the component's Dirichlet entry
            const int64_t i = cell ? (int64_t)k * D + d : 0;
            const double e1 = eta[K + i], kappa = eta[K + KD + i], e3 = eta[K + 2 * KD + i], e4 = eta[K + 3 * KD + i];
            const double m = e1 / kappa, a = 0.5 * (e3 + 1.0), b = 0.5 * (e4 - kappa * m * m), T = a / b;
            const double al = head ? eta[k] + 1.0 : alpha_sum;
            PsiLg r_a = {0.0, 0.0}, r_al = r_a;
#pragma unroll 1
            for (int q = 0; q < 2; ++q) {
                const PsiLg r = psi_lgamma_f64(q == 0 ? a : al);
                if (q == 0) r_a = r;
                else r_al = r;
            }
            double log_b = 0.0, log_k = 0.0;
#pragma unroll 1
            for (int q = 0; q < 2; ++q) {
                const double l = log(q == 0 ? b : kappa);
                if (q == 0) log_b = l;
                else log_k = l;
            }
            // psi(sum alpha) for the group's first lane: from its neighbour (lane 1 of the 16-lane group)
            const double psi_sum = __shfl(r_al.psi, (threadIdx.x & 63 & ~15) | 1);
            if (dl == 1 && d0 == 0) lg_sum = r_al.lg;
            const double elog_tau = r_a.psi - log_b;
            if (cell) {
                c += 0.5 * elog_tau - 0.5 * LOG_2PI - 0.5 * T * m * m - 0.5 / kappa;
                Wmat[(int64_t)k * 2 * D + d] = (float)(T * m);
                Wmat[(int64_t)k * 2 * D + D + d] = (float)(-0.5 * T);
                ga += r_a.lg - a * log_b - 0.5 * log_k + 0.5 * LOG_2PI;
                if (with_bound) {
                    const double p1 = eta0[K + i], kappa0 = eta0[K + KD + i], p3 = eta0[K + 2 * KD + i],
                                 p4 = eta0[K + 3 * KD + i];
                    gb += (p1 - e1) * (T * m) + (kappa0 - kappa) * (-0.5 * (1.0 / kappa + m * m * T)) +
                          (p3 - e3) * (0.5 * elog_tau) + (p4 - e4) * (-0.5 * T);
                }
            }
            if (head) {
                const double elog_pi = r_al.psi - psi_sum;
                c += elog_pi;
                ga += r_al.lg;
                if (with_bound) gb += (eta0[k] - eta[k]) * elog_pi;
            }
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            c += __shfl_xor(c, off);
            gb += __shfl_xor(gb, off);
            ga += __shfl_xor(ga, off);
        }
        if (k < K && dl == 0) {
            cvec[k] = (float)c;
            comp_bound[k] = gb;
            comp_A[k] = ga;
        }
    }
    if (bound || A_out) {
        __syncthreads();
        if (threadIdx.x < 64) {       // components in index order per lane, lanes by the fixed butterfly
            double cross = 0.0, A = 0.0;
            for (int k = threadIdx.x; k < K; k += 64) {
                cross += comp_bound[k];
                A += comp_A[k];
            }
            cross = wave_allsum_f64(cross);
            A = wave_allsum_f64(A);
            const double lgs = __shfl(lg_sum, 1);     // thread 1 holds lnGamma(sum alpha)
            if (threadIdx.x == 0) {
                A -= lgs;
                if (A_out) A_out[0] = A;
                if (bound) bound[0] = cross + A - (prior_A ? prior_A[0] : 0.0);
            }
        }
    }
}

// eta <- (1-rho) eta + rho (eta0 + scale * message(stats)), stats = [K][R | Sx | Sxx]
// With `elbo`: elbo[0] = scale * lse[0] + bound[0] -- the mini-batch estimate of the evidence lower
// bound AT the natural parameters the statistics were taken with (before this step moves them).
__global__ void mog_natgrad_kernel(double* __restrict__ eta, const double* __restrict__ eta0,
                                   const double* __restrict__ stats, int K, int D, double scale,
                                   double rho, const double* __restrict__ lse,
                                   const double* __restrict__ bound, double* __restrict__ elbo) {
    const int64_t KD = (int64_t)K * D;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && elbo) elbo[0] = scale * lse[0] + bound[0];
    if (i >= K + 4 * KD) return;
    double msg;
    if (i < K) {
        msg = stats[i * (1 + 2 * D)];
    } else {
        const int64_t j = i - K;
        const int part = (int)(j / KD);
        const int64_t kd = j % KD;
        const int64_t k = kd / D, d = kd % D;
        const double* row = stats + k * (1 + 2 * D);
        msg = part == 0 ? row[1 + d] : (part == 3 ? row[1 + D + d] : row[0]);
    }
    eta[i] = (1.0 - rho) * eta[i] + rho * (eta0[i] + scale * msg);
}

}  // namespace

extern "C" {

int bsc_mog_expected_params(bsc_ctx* ctx, const double* eta, int32_t K, int32_t D, float* Wmat,
                            float* c) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eta && Wmat && c, "bsc_mog_expected_params: null pointer");
    BSC_REQUIRE(K >= 1 && D >= 1, "bsc_mog_expected_params: K=%d D=%d", K, D);
    hipLaunchKernelGGL(mog_expected_params_kernel, dim3(1), dim3(1024), 0, ctx->stream, eta,
                       (const double*)nullptr, (const double*)nullptr, (int)K, (int)D, Wmat, c, (double*)nullptr,
                       (double*)nullptr);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_mog_log_normalizer(bsc_ctx* ctx, const double* eta, int32_t K, int32_t D, double* A_out) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eta && A_out && K >= 1 && D >= 1, "bsc_mog_log_normalizer: bad arguments");
    if (K > 1024) return bsc_fail(BSC_ERR_UNSUPPORTED, "bsc_mog_log_normalizer: K=%d > 1024 components", K);
    void* ws = nullptr;      // the coefficients this launch also forms are not wanted: they go to the workspace
    int rc = bsc_workspace(ctx, ((size_t)K * 2 * D + K) * sizeof(float), &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    hipLaunchKernelGGL(mog_expected_params_kernel, dim3(1), dim3(1024), 0, ctx->stream, eta,
                       (const double*)nullptr, (const double*)nullptr, (int)K, (int)D, (float*)ws,
                       (float*)ws + (size_t)K * 2 * D, (double*)nullptr, A_out);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_mog_expected_params_bound(bsc_ctx* ctx, const double* eta, const double* eta0, const double* prior_A,
                                  int32_t K, int32_t D, float* Wmat, float* c, double* bound) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eta && eta0 && prior_A && Wmat && c && bound, "bsc_mog_expected_params_bound: null pointer");
    BSC_REQUIRE(K >= 1 && D >= 1, "bsc_mog_expected_params_bound: K=%d D=%d", K, D);
    if (K > 1024)
        return bsc_fail(BSC_ERR_UNSUPPORTED, "bsc_mog_expected_params_bound: K=%d > 1024 components", K);
    hipLaunchKernelGGL(mog_expected_params_kernel, dim3(1), dim3(1024), 0, ctx->stream, eta, eta0, prior_A, (int)K,
                       (int)D, Wmat, c, bound, (double*)nullptr);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_mog_natgrad(bsc_ctx* ctx, double* eta, const double* eta0, const double* stats, int32_t K,
                    int32_t D, double scale, double rho) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eta && eta0 && stats && K >= 1 && D >= 1, "bsc_mog_natgrad: bad arguments");
    const int64_t n = K + 4 * (int64_t)K * D;
    hipLaunchKernelGGL(mog_natgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, eta, eta0, stats, (int)K, (int)D, scale, rho, (const double*)nullptr,
                       (const double*)nullptr, (double*)nullptr);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_mog_natgrad_elbo(bsc_ctx* ctx, double* eta, const double* eta0, const double* stats, int32_t K,
                         int32_t D, double scale, double rho, const double* lse, const double* bound,
                         double* elbo) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eta && eta0 && stats && lse && bound && elbo && K >= 1 && D >= 1,
                "bsc_mog_natgrad_elbo: bad arguments");
    const int64_t n = K + 4 * (int64_t)K * D;
    hipLaunchKernelGGL(mog_natgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, eta, eta0, stats, (int)K, (int)D, scale, rho, lse, bound, elbo);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_mog_estep(bsc_ctx* ctx, const float* X, int64_t ldx, int64_t N, int32_t D, int32_t K,
                  const float* Wmat, const float* c, double* stats, double* lse) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(N >= 0 && ((X != nullptr) || N == 0) && Wmat && c && stats && lse,
                "bsc_mog_estep: null pointer");
    if (D < 1 || D > MD || K < 1 || K > MK)
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_mog_estep: D=%d K=%d outside the MFMA tile limits (D<=%d, K<=%d)", D,
                        K, MD, MK);
    BSC_REQUIRE(ldx >= D && ldx < ((int64_t)1 << 26), "bsc_mog_estep: bad ldx=%lld", (long long)ldx);
    BSC_REQUIRE(((uintptr_t)X & 3) == 0, "bsc_mog_estep: X must be 4-byte aligned");
    const int64_t n_tiles = (N + MT - 1) / MT;
    const int64_t max_waves = 2 * 4 * (int64_t)ctx->cu_count;
    int n_iter = 0, n_blocks = 1;
    if (n_tiles > 0) {
        const int64_t it = (n_tiles + max_waves - 1) / max_waves;
        const int64_t waves = (n_tiles + it - 1) / it;
        n_iter = (int)(it + (it & 1));   // even: the kernel alternates two row buffers per tile pair
        n_blocks = (int)((waves + MOG_WAVES - 1) / MOG_WAVES);
    }
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)n_blocks * MOG_SLAB * sizeof(float), &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    {
        bsc_prof_scope prof(ctx);
        if (ctx->mfma_split) {
            // bf16 MFMA: forward on three terms, backward on two (whichever of 2 / 3 the context asks for)
            if (D == MD) hipLaunchKernelGGL((mog_estep_bx_kernel<true>), dim3(n_blocks), dim3(MOG_BLOCK), 0, ctx->stream, X,
                                            ldx, N, (int)D, Wmat, c, (int)K, (float*)ws, n_iter);
            else hipLaunchKernelGGL((mog_estep_bx_kernel<false>), dim3(n_blocks), dim3(MOG_BLOCK), 0, ctx->stream, X,
                                    ldx, N, (int)D, Wmat, c, (int)K, (float*)ws, n_iter);
        } else if (D == MD && ctx->mog_nt)
            hipLaunchKernelGGL((mog_estep_kernel<true, true>), dim3(n_blocks), dim3(MOG_BLOCK), 0, ctx->stream, X,
                               ldx, N, (int)D, Wmat, c, (int)K, (float*)ws, n_iter);
        else if (D == MD)
            hipLaunchKernelGGL((mog_estep_kernel<true, false>), dim3(n_blocks), dim3(MOG_BLOCK), 0, ctx->stream, X,
                               ldx, N, (int)D, Wmat, c, (int)K, (float*)ws, n_iter);
        else
            hipLaunchKernelGGL((mog_estep_kernel<false, true>), dim3(n_blocks), dim3(MOG_BLOCK), 0, ctx->stream, X,
                               ldx, N, (int)D, Wmat, c, (int)K, (float*)ws, n_iter);
    }
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(mog_reduce_kernel, dim3((MOG_SLAB + 63) / 64), dim3(1024), 0, ctx->stream,
                       (const float*)ws, n_blocks, (int)K, (int)D, stats, lse);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

}  // extern "C"
