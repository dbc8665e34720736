// Bayesian-linear-regression reparameterised ELBO path (BASELINE config 2).
//
// ABSENT in the reference: spec is README.md:51 (reparameterisation trick,
// refs [10][11][12]) and README.md:69-79 (mini-batch SVI), with the
// log-likelihood split of bayesic/distribution/base.py:47-69.
//
// blr_pass_kernel is the HBM-bound hot kernel: ONE read of X[B,D] and y[B]
// produces, for S<=8 Monte-Carlo weight draws at once,
//     Q[s]   = sum_n (y_n - x_n.w_s)^2
//     G[s,:] = sum_n (y_n - x_n.w_s) x_n
// Layout: a wave owns 8-row tiles; lane l holds columns 4l..4l+3 of every row
// (one 16-byte load per row per lane = 1 KiB coalesced per wave instruction at
// D=256).  Forward dot products are reduced across the 64 lanes with a
// transposing butterfly (v_permlane32_swap / v_permlane16_swap / DPP), so that
// lane l ends up holding the residual of (row l>>3, sample l&7).  The backward
// rank-1 updates broadcast each residual through an SGPR (v_readlane) into
// per-lane accumulators acc[s][4] that live in registers for the whole kernel.
// Block partials go to a slab; a second kernel sums slabs in float64 in a fixed
// order (bitwise reproducible, no float atomics).
#include "bsc_common.h"

namespace {

constexpr int TILE_ROWS = 8;
constexpr int SG = 8;  // samples per pass
constexpr int PASS_BLOCK = 256;
constexpr int PASS_WAVES = PASS_BLOCK / BSC_WAVE;
constexpr int GCOLS = 256;                       // column capacity of the lane layout
constexpr int SLAB_STRIDE = SG * GCOLS + SG;     // floats per block partial

struct Tile {
    float4 x[TILE_ROWS];
    float yv;
};

template <bool CHECK>
__device__ __forceinline__ void load_tile(Tile& t, const float* __restrict__ X, int64_t ldx,
                                          const float* __restrict__ y, int64_t row0,
                                          int64_t B, int lane, bool lane_active) {
    const float* base = X + row0 * ldx + 4 * lane;
#pragma unroll
    for (int r = 0; r < TILE_ROWS; ++r) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool ok = lane_active;
        if (CHECK) ok = ok && (row0 + r < B);
        if (ok) v = *reinterpret_cast<const float4*>(base + (int64_t)r * ldx);
        t.x[r] = v;
    }
    int64_t yr = row0 + (lane >> 3);
    float yv = 0.f;
    if (!CHECK || yr < B) yv = y[yr];
    t.yv = yv;
}

__device__ __forceinline__ void swap_add32(float a, float b, float& out) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false,
                                              false);
    out = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ void swap_add16(float a, float b, float& out) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false,
                                              false);
    out = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct LaneSel {
    bool q1, q2, q3;  // ((lane>>2)&3) == 1,2,3
    bool l1, l2, l3;  // (lane&3) == 1,2,3
};

__device__ __forceinline__ float sel4(float v0, float v1, float v2, float v3, bool is1,
                                      bool is2, bool is3) {
    float r = is1 ? v1 : v0;
    r = is2 ? v2 : r;
    r = is3 ? v3 : r;
    return r;
}

// Forward + backward for one tile.  After the butterfly, lane l holds
// dot(x[l>>3], w[l&7]).
__device__ __forceinline__ void compute_tile(const Tile& t, const float4 (&w)[SG],
                                             float4 (&acc)[SG], float& qacc,
                                             const LaneSel& ls) {
    float p[TILE_ROWS * SG];
#pragma unroll
    for (int r = 0; r < TILE_ROWS; ++r) {
#pragma unroll
        for (int s = 0; s < SG; ++s) {
            float v = t.x[r].x * w[s].x;
            v = fmaf(t.x[r].y, w[s].y, v);
            v = fmaf(t.x[r].z, w[s].z, v);
            v = fmaf(t.x[r].w, w[s].w, v);
            p[r * SG + s] = v;
        }
    }
    // 64 values/lane -> 32: lanes 0-31 keep value i, lanes 32-63 value i+32.
    float a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) swap_add32(p[i], p[i + 32], a[i]);
    // 32 -> 16: rows with lane bit 4 clear keep i, set keep i+16.
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) swap_add16(a[i], a[i + 16], b[i]);
    // 16 -> 4: sum over the 4 lanes of the row sharing lane&3 (rotation
    // direction is irrelevant: ror8 then ror4 of a period-8 value), then keep
    // the value selected by lane bits 2-3.
    float c[4];
    {
        float u[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float h = b[i] + dpp_f32<DPP_ROW_ROR8>(b[i]);
            u[i] = h + dpp_f32<DPP_ROW_ROR4>(h);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            c[i] = sel4(u[i], u[i + 4], u[i + 8], u[i + 12], ls.q1, ls.q2, ls.q3);
    }
    // 4 -> 1 inside the quad.
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float h = c[i] + dpp_f32<DPP_QUAD_XOR1>(c[i]);
        v[i] = h + dpp_f32<DPP_QUAD_XOR2>(h);
    }
    float dot = sel4(v[0], v[1], v[2], v[3], ls.l1, ls.l2, ls.l3);
    float resid = t.yv - dot;
    qacc = fmaf(resid, resid, qacc);
    // backward: acc[s] += resid(r,s) * x[r]
#pragma unroll
    for (int r = 0; r < TILE_ROWS; ++r) {
#pragma unroll
        for (int s = 0; s < SG; ++s) {
            float cst = readlane_f32(resid, r * SG + s);
            acc[s].x = fmaf(cst, t.x[r].x, acc[s].x);
            acc[s].y = fmaf(cst, t.x[r].y, acc[s].y);
            acc[s].z = fmaf(cst, t.x[r].z, acc[s].z);
            acc[s].w = fmaf(cst, t.x[r].w, acc[s].w);
        }
    }
}

template <bool FULL>  // FULL: D == 256, every lane owns four live columns
__global__ __launch_bounds__(PASS_BLOCK, 2) void blr_pass_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y, int64_t B, int D,
    const float* __restrict__ W, int S, float* __restrict__ slab) {
    __shared__ float lds[PASS_WAVES][SLAB_STRIDE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bool lane_active = FULL ? true : (4 * lane < D);

    LaneSel ls;
    ls.q1 = ((lane >> 2) & 3) == 1;
    ls.q2 = ((lane >> 2) & 3) == 2;
    ls.q3 = ((lane >> 2) & 3) == 3;
    ls.l1 = (lane & 3) == 1;
    ls.l2 = (lane & 3) == 2;
    ls.l3 = (lane & 3) == 3;

    float4 w[SG], acc[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s) {
        w[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_active && s < S)
            w[s] = *reinterpret_cast<const float4*>(W + (int64_t)s * D + 4 * lane);
        acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float qacc = 0.f;

    const int64_t n_tiles = (B + TILE_ROWS - 1) / TILE_ROWS;
    const int64_t n_full = B / TILE_ROWS;  // tiles [0, n_full) need no row check
    const int64_t stride = (int64_t)gridDim.x * PASS_WAVES;
    int64_t tile = (int64_t)blockIdx.x * PASS_WAVES + wave;

    Tile ta, tb;
    if (tile < n_tiles) {
        if (tile < n_full) load_tile<false>(ta, X, ldx, y, tile * TILE_ROWS, B, lane, lane_active);
        else load_tile<true>(ta, X, ldx, y, tile * TILE_ROWS, B, lane, lane_active);
    }
    while (tile < n_tiles) {
        int64_t nxt = tile + stride;
        if (nxt < n_tiles) {
            if (nxt < n_full) load_tile<false>(tb, X, ldx, y, nxt * TILE_ROWS, B, lane, lane_active);
            else load_tile<true>(tb, X, ldx, y, nxt * TILE_ROWS, B, lane, lane_active);
        }
        compute_tile(ta, w, acc, qacc, ls);
        tile = nxt;
        if (tile >= n_tiles) break;
        nxt = tile + stride;
        if (nxt < n_tiles) {
            if (nxt < n_full) load_tile<false>(ta, X, ldx, y, nxt * TILE_ROWS, B, lane, lane_active);
            else load_tile<true>(ta, X, ldx, y, nxt * TILE_ROWS, B, lane, lane_active);
        }
        compute_tile(tb, w, acc, qacc, ls);
        tile = nxt;
    }

    // block reduction through LDS, fixed order over waves
#pragma unroll
    for (int s = 0; s < SG; ++s)
        *reinterpret_cast<float4*>(&lds[wave][s * GCOLS + 4 * lane]) = acc[s];
    float q = qacc;  // lane l: sample l&7, rows l>>3
    q += __shfl_xor(q, 8);
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (lane < SG) lds[wave][SG * GCOLS + lane] = q;
    __syncthreads();
    float* out = slab + (int64_t)blockIdx.x * SLAB_STRIDE;
    for (int i = tid; i < SLAB_STRIDE; i += PASS_BLOCK) {
        float v = lds[0][i];
#pragma unroll
        for (int k = 1; k < PASS_WAVES; ++k) v += lds[k][i];
        out[i] = v;
    }
}

// Sum block partials in float64, fixed order.  One output per lane; the 16
// waves of a block split the slab rows, then combine through LDS in wave order.
constexpr int RED_BLOCK = 1024;
constexpr int RED_WAVES = RED_BLOCK / BSC_WAVE;

__global__ __launch_bounds__(RED_BLOCK) void blr_slab_reduce_kernel(
    const float* __restrict__ slab, int n_blocks, int D, int S, int s_base,
    double* __restrict__ Q, double* __restrict__ G) {
    __shared__ double part[RED_WAVES][BSC_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = blockIdx.x * BSC_WAVE + lane;
    double sum = 0.0;
    if (i < SLAB_STRIDE) {
        for (int b = wave; b < n_blocks; b += RED_WAVES)
            sum += (double)slab[(int64_t)b * SLAB_STRIDE + i];
    }
    part[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && i < SLAB_STRIDE) {
        double tot = part[0][lane];
#pragma unroll
        for (int k = 1; k < RED_WAVES; ++k) tot += part[k][lane];
        if (i < SG * GCOLS) {
            int s = i / GCOLS, d = i % GCOLS;
            if (s_base + s < S && d < D) G[(int64_t)(s_base + s) * D + d] = tot;
        } else {
            int s = i - SG * GCOLS;
            if (s_base + s < S) Q[s_base + s] = tot;
        }
    }
}

// ---- sampler and ELBO/gradient finish (tiny, float64) ----------------------

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

#pragma clang fp contract(off)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint32_t block, uint32_t sample,
                                               uint32_t stream, uint32_t step, double (&z)[4]) {
    uint32_t c[4] = {block, sample, stream, step};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double two_m32 = 2.3283064365386963e-10;  // 2^-32
    const double two_pi = 6.283185307179586476925286766559;
    double u0 = ((double)c[0] + 0.5) * two_m32;
    double u1 = ((double)c[1] + 0.5) * two_m32;
    double u2 = ((double)c[2] + 0.5) * two_m32;
    double u3 = ((double)c[3] + 0.5) * two_m32;
    double r0 = sqrt(-2.0 * log(u0));
    double r1 = sqrt(-2.0 * log(u2));
    double t0 = two_pi * u1;
    double t1 = two_pi * u3;
    z[0] = r0 * cos(t0);
    z[1] = r0 * sin(t0);
    z[2] = r1 * cos(t1);
    z[3] = r1 * sin(t1);
}

__global__ void philox_normal_kernel(uint64_t seed, uint32_t stream, uint32_t step,
                                     int n_samples, int n_params, double* __restrict__ eps) {
    const int n_blocks = (n_params + 3) / 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_samples * n_blocks) return;
    const int s = idx / n_blocks, b = idx % n_blocks;
    double z[4];
    philox_normal4(seed, (uint32_t)b, (uint32_t)s, stream, step, z);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * b + j < n_params) eps[(int64_t)s * n_params + 4 * b + j] = z[j];
}

__global__ void blr_sample_kernel(const double* __restrict__ lam, int D, int S, uint64_t seed,
                                  uint32_t step, double* __restrict__ eps,
                                  float* __restrict__ W, double* __restrict__ xi) {
    const int n_blocks = (D + 3) / 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < S * n_blocks) {
        const int s = idx / n_blocks, b = idx % n_blocks;
        double z[4];
        philox_normal4(seed, (uint32_t)b, (uint32_t)s, 0u, step, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = 4 * b + j;
            if (d < D) {
                eps[(int64_t)s * (D + 1) + d] = z[j];
                double sd = exp(lam[D + d]);
                double wv = lam[d] + sd * z[j];
                W[(int64_t)s * D + d] = (float)wv;
            }
        }
    } else if (idx < S * n_blocks + S) {
        const int s = idx - S * n_blocks;
        double z[4];
        philox_normal4(seed, 0u, (uint32_t)s, 1u, step, z);
        eps[(int64_t)s * (D + 1) + D] = z[0];
        double sd = exp(lam[2 * D + 1]);
        xi[s] = lam[2 * D] + sd * z[0];
    }
}

// One workgroup.  Thread d owns column d (strided when D > blockDim).
constexpr int FIN_BLOCK = 256;
constexpr int FIN_MAX_S = 64;

__global__ __launch_bounds__(FIN_BLOCK) void blr_elbo_grad_kernel(
    const double* __restrict__ lam, const double* __restrict__ eps,
    const float* __restrict__ W, const double* __restrict__ xi, const double* __restrict__ Q,
    const double* __restrict__ G, int D, int S, double batch_rows, double scale, double alpha0,
    double beta0, double* __restrict__ elbo, double* __restrict__ grad) {
    __shared__ double red[FIN_BLOCK / BSC_WAVE][FIN_MAX_S + 1];
    __shared__ double wsq[FIN_MAX_S];
    __shared__ double e_inv[FIN_MAX_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double LOG_2PI = 1.8378770664093454835606594728112;

    // |w_s|^2 for every sample and sum(rho), fixed-order block reduction
    for (int s = 0; s <= S; ++s) {
        double part = 0.0;
        for (int d = tid; d < D; d += FIN_BLOCK) {
            if (s < S) {
                double wv = (double)W[(int64_t)s * D + d];
                part += wv * wv;
            } else {
                part += lam[D + d];
            }
        }
        part = wave_allsum_f64(part);
        if (lane == 0) red[wave][s] = part;
    }
    __syncthreads();
    if (tid < S) {
        double t = 0.0;
        for (int k = 0; k < FIN_BLOCK / BSC_WAVE; ++k) t += red[k][tid];
        wsq[tid] = t;
        e_inv[tid] = exp(-xi[tid]);
    }
    __syncthreads();

    const double inv_S = 1.0 / (double)S;
    for (int d = tid; d < D; d += FIN_BLOCK) {
        double gm = 0.0, gr = 0.0;
        for (int s = 0; s < S; ++s) {
            double wv = (double)W[(int64_t)s * D + d];
            double dw = e_inv[s] * (scale * G[(int64_t)s * D + d] - wv);
            gm += dw;
            gr += dw * eps[(int64_t)s * (D + 1) + d];
        }
        grad[d] = gm * inv_S;
        grad[D + d] = gr * inv_S * exp(lam[D + d]) + 1.0;
    }
    if (tid == 0) {
        double sum_rho = 0.0;
        for (int k = 0; k < FIN_BLOCK / BSC_WAVE; ++k) sum_rho += red[k][S];
        const double b = lam[2 * D + 1];
        double fa = 0.0, fb = 0.0, fsum = 0.0;
        for (int s = 0; s < S; ++s) {
            const double e = e_inv[s], x = xi[s];
            double dxi = -0.5 * (scale * batch_rows + (double)D) - alpha0 +
                         e * (0.5 * scale * Q[s] + 0.5 * wsq[s] + beta0);
            fa += dxi;
            fb += dxi * eps[(int64_t)s * (D + 1) + D];
            double loglik = scale * (-0.5 * batch_rows * (LOG_2PI + x) - 0.5 * e * Q[s]);
            double logpw = -0.5 * (double)D * (LOG_2PI + x) - 0.5 * e * wsq[s];
            double logpxi = alpha0 * log(beta0) - lgamma(alpha0) - alpha0 * x - beta0 * e;
            fsum += loglik + logpw + logpxi;
        }
        grad[2 * D] = fa * inv_S;
        grad[2 * D + 1] = fb * inv_S * exp(b) + 1.0;
        elbo[0] = fsum * inv_S + sum_rho + b + 0.5 * (double)(D + 1) * (1.0 + LOG_2PI);
    }
}

}  // namespace

extern "C" {

int bsc_philox_normal(bsc_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t step,
                      int32_t n_samples, int32_t n_params, double* eps) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eps && n_samples > 0 && n_params > 0, "bsc_philox_normal: bad arguments");
    const int n = n_samples * ((n_params + 3) / 4);
    hipLaunchKernelGGL(philox_normal_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                       seed, stream, step, n_samples, n_params, eps);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_blr_sample(bsc_ctx* ctx, const double* lam, int32_t D, int32_t S, uint64_t seed,
                   uint32_t step, double* eps, float* W, double* xi) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(lam && eps && W && xi && D > 0 && S > 0, "bsc_blr_sample: bad arguments");
    const int n = S * ((D + 3) / 4) + S;
    hipLaunchKernelGGL(blr_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                       lam, D, S, seed, step, eps, W, xi);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_blr_data_pass(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B,
                      int32_t D, const float* W, int32_t S, double* Q, double* G) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(B >= 0, "bsc_blr_data_pass: B=%lld", (long long)B);
    BSC_REQUIRE(((X && y) || B == 0) && W && Q && G, "bsc_blr_data_pass: null pointer");
    BSC_REQUIRE(D > 0 && D <= GCOLS && D % 4 == 0,
                "bsc_blr_data_pass: D=%d must be a multiple of 4 in [4,%d]", D, GCOLS);
    BSC_REQUIRE(S >= 1 && S <= FIN_MAX_S, "bsc_blr_data_pass: S=%d must be in [1,%d]", S,
                FIN_MAX_S);
    BSC_REQUIRE(ldx >= D && ldx % 4 == 0, "bsc_blr_data_pass: ldx=%lld must be >= D and %% 4 == 0",
                (long long)ldx);
    BSC_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0,
                "bsc_blr_data_pass: X and W must be 16-byte aligned");
    const int64_t n_tiles = (B + TILE_ROWS - 1) / TILE_ROWS;
    int64_t want = (n_tiles + PASS_WAVES - 1) / PASS_WAVES;
    int n_blocks = (int)(want < 1 ? 1 : (want > 2 * ctx->cu_count ? 2 * ctx->cu_count : want));
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)n_blocks * SLAB_STRIDE * sizeof(float), &ws);
    if (rc != BSC_OK) return rc;
    float* slab = (float*)ws;
    for (int s0 = 0; s0 < S; s0 += SG) {
        const int sg = (S - s0 < SG) ? (S - s0) : SG;
        {
        bsc_prof_scope prof(ctx);  // times the pass kernel alone
        if (D == GCOLS)
            hipLaunchKernelGGL(blr_pass_kernel<true>, dim3(n_blocks), dim3(PASS_BLOCK), 0,
                               ctx->stream, X, ldx, y, B, (int)D, W + (int64_t)s0 * D, sg, slab);
        else
            hipLaunchKernelGGL(blr_pass_kernel<false>, dim3(n_blocks), dim3(PASS_BLOCK), 0,
                               ctx->stream, X, ldx, y, B, (int)D, W + (int64_t)s0 * D, sg, slab);
        }
        BSC_LAUNCH_CHECK();
        hipLaunchKernelGGL(blr_slab_reduce_kernel, dim3((SLAB_STRIDE + BSC_WAVE - 1) / BSC_WAVE),
                           dim3(RED_BLOCK), 0, ctx->stream, slab, n_blocks, (int)D, (int)S, s0, Q,
                           G);
        BSC_LAUNCH_CHECK();
    }
    return BSC_OK;
}

int bsc_blr_elbo_grad(bsc_ctx* ctx, const double* lam, const double* eps, const float* W,
                      const double* xi, const double* Q, const double* G, int32_t D, int32_t S,
                      double batch_rows, double scale, double alpha0, double beta0, double* elbo,
                      double* grad) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(lam && eps && W && xi && Q && G && elbo && grad, "bsc_blr_elbo_grad: null pointer");
    BSC_REQUIRE(D > 0 && S >= 1 && S <= FIN_MAX_S, "bsc_blr_elbo_grad: D=%d S=%d (S<=%d)", D, S,
                FIN_MAX_S);
    BSC_REQUIRE(alpha0 > 0 && beta0 > 0, "bsc_blr_elbo_grad: alpha0, beta0 must be positive");
    hipLaunchKernelGGL(blr_elbo_grad_kernel, dim3(1), dim3(FIN_BLOCK), 0, ctx->stream, lam, eps, W,
                       xi, Q, G, (int)D, (int)S, batch_rows, scale, alpha0, beta0, elbo, grad);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

}  // extern "C"
