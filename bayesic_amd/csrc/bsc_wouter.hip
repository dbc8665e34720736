// Responsibility-weighted second moments: out[k,d,e] = scale * sum_n R[n,k] X[n,d] Y[n,e].
//
// With Y = X this is the full-covariance sufficient statistic sum_n r_nk x_n x_n^T of a
// mixture component (t(x) = (x, x x^T), bayesic/distribution/core.py:41-44, summed over the
// iid axis as bayesic/distribution/base.py:329-332 asks).  The reference's front end lowers
// einsum(out_kde = sum_n R_nk X_nd X_ne) to
//   _tensordot(_mul(_dimshuffle(R,1,'x',0), _dimshuffle(X,'x',1,0)), X, [2], [0])
// (bayesic/algebra.py:632-636: the last factor holding the index goes to the right, the rest
// are multiplied on the left), which materialises a K x D x N intermediate -- 41 GB at
// BASELINE config 3's size.  Here it is one pass over R and X with nothing materialised:
//
//   D[comp][pair] += A(R: lane = component, k = row of the pair of rows)
//                  * B(z: lane = (d,e) pair,  z = X[row][d] * Y[row][e], formed in registers)
//
// on v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate).  When Y is X only the
// D(D+1)/2 pairs d <= e are computed and mirrored on output.  A workgroup stages 64 rows of
// R, X (and Y) in LDS; each of its waves owns one 32-pair tile for both component tiles, so a
// k-step of a wave is three LDS reads and one multiply against two MFMAs.  Bound: fp32 MFMA.
// Workgroups are summed by a fixed-order float64 finish.
#include "bsc_common.h"

namespace {

constexpr int WO_TR = 64;       // rows per stage
constexpr int WO_BLOCK = 256;
constexpr int WO_MAXK = 64;
constexpr int WO_MAXD = 32;
constexpr int WO_MAXCT = 8;     // pair tiles (= waves) per workgroup
constexpr int WO_TILE = 1024;   // floats of one 32 x 32 accumulator tile

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WOArgs {
    const float* R;
    const float* X;
    const float* Y;
    int64_t ldr, ldx, ldy, N;
    int K, D, E, sym, P;   // P = number of (d,e) pairs computed
    int iters;             // row stages per workgroup (interleaved over blockIdx.x)
    float* part;           // [gridDim.y][gridDim.x][KT*CT][1024]
};

// pair index -> (d, e); symmetric: d <= e, rows of the upper triangle in order
__device__ __forceinline__ void wo_pair(int p, int D, int E, int sym, int& d, int& e) {
    if (!sym) {
        d = p / E;
        e = p - d * E;
        return;
    }
    d = 0;
    int len = D;
    while (p >= len) {
        p -= len;
        --len;
        ++d;
    }
    e = d + p;
}

// blockDim.x = 64 * CT: wave w owns pair tile blockIdx.y * CT + w for every component tile
// and walks all rows of the stage; no wave shares an accumulator, so nothing is combined
// inside the workgroup.  ~60 VGPRs: several workgroups per CU overlap one's loads with the
// others' MFMAs, which is all the latency hiding there is (no register prefetch).
template <int KT, bool SYM>
__global__ __launch_bounds__(512) void weighted_outer_kernel(WOArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[WO_TR * (WO_MAXK + 2 * WO_MAXD)];
    constexpr int RS = KT * 32;
    float* Rs = lds;
    float* Xs = lds + WO_TR * WO_MAXK;
    float* Ys = SYM ? Xs : Xs + WO_TR * WO_MAXD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_thr = blockDim.x;
    const int CT = n_thr >> 6;
    const int col = lane & 31, half = lane >> 5;

    int xo = 0, yo = 0;
    {
        const int p = ((int)blockIdx.y * CT + wave) * 32 + col;
        if (p < a.P) wo_pair(p, a.D, a.E, SYM, xo, yo);
    }
    f32x16 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[kt][q] = 0.f;

    const int k4 = a.K >> 2, d4 = a.D >> 2, e4 = a.E >> 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < a.iters; ++it) {
        const int64_t row0 = ((int64_t)blockIdx.x + (int64_t)it * gridDim.x) * WO_TR;
        // a stage: R has 16 float4 slots per row (zero beyond K), X / Y have 8
        for (int s = tid; s < WO_TR * 16; s += n_thr) {
            const int r = s >> 4, c = s & 15;
            const int64_t row = row0 + r;
            const float4 v = (c < k4 && row < a.N) ? *(const float4*)(a.R + row * a.ldr + c * 4) : zero4;
            if (c < KT * 8) *(float4*)(Rs + r * RS + c * 4) = v;
        }
        for (int s = tid; s < WO_TR * 8; s += n_thr) {
            const int r = s >> 3, c = s & 7;
            const int64_t row = row0 + r;
            *(float4*)(Xs + r * WO_MAXD + c * 4) =
                (c < d4 && row < a.N) ? *(const float4*)(a.X + row * a.ldx + c * 4) : zero4;
            if (!SYM)
                *(float4*)(Ys + r * WO_MAXD + c * 4) =
                    (c < e4 && row < a.N) ? *(const float4*)(a.Y + row * a.ldy + c * 4) : zero4;
        }
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < WO_TR / 2; ++j) {
            const int r = 2 * j + half;
            const float z = Xs[r * WO_MAXD + xo] * Ys[r * WO_MAXD + yo];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Rs[r * RS + kt * 32 + col], z, acc[kt], 0, 0, 0);
        }
        __syncthreads();
    }

    float* out = a.part + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (KT * CT)) * WO_TILE;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) out[(kt * CT + wave) * WO_TILE + q * 64 + lane] = acc[kt][q];
}

// out[k][d][e] (and [k][e][d] when symmetric) = scale * sum over workgroups, float64, fixed
// order.  A block owns one (component, 32 pairs) line: 8 groups of 32 lanes walk the
// workgroups interleaved, then combine in group order.
__global__ __launch_bounds__(256) void weighted_outer_finish_kernel(const float* __restrict__ part,
                                                                    int gx, int KT, int CT, int K,
                                                                    int D, int E, int sym, int P,
                                                                    double scale,
                                                                    float* __restrict__ out) {
    __shared__ double comb[8][32];
    const int n_ct = (P + 31) / 32;
    const int k = blockIdx.x / n_ct, ctg = blockIdx.x - k * n_ct;
    const int j = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int kt = k >> 5, i = k & 31;
    const int q = (i & 3) + 4 * (i >> 3), half = (i >> 2) & 1;
    const int gy = ctg / CT, ct = ctg - gy * CT;
    const int64_t wg_stride = (int64_t)KT * CT * WO_TILE;
    const float* src = part + (int64_t)gy * gx * wg_stride + (int64_t)(kt * CT + ct) * WO_TILE +
                       q * 64 + half * 32 + j;
    double s = 0.0;
    for (int b = grp; b < gx; b += 8) s += (double)src[(int64_t)b * wg_stride];
    comb[grp][j] = s;
    __syncthreads();
    if (grp != 0) return;
    double tot = comb[0][j];
#pragma unroll
    for (int g = 1; g < 8; ++g) tot += comb[g][j];
    const int p = ctg * 32 + j;
    if (p >= P) return;
    int d, e;
    wo_pair(p, D, E, sym, d, e);
    const float v = (float)(scale * tot);
    out[((int64_t)k * D + d) * E + e] = v;
    if (sym && d != e) out[((int64_t)k * D + e) * E + d] = v;
}

}  // namespace

extern "C" int bsc_weighted_outer(bsc_ctx* ctx, const float* R, int64_t ldr, const float* X,
                                  int64_t ldx, const float* Y, int64_t ldy, int64_t N, int32_t K,
                                  int32_t D, int32_t E, double scale, float* out) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(N >= 0 && K >= 1 && D >= 1 && E >= 1 && out, "bsc_weighted_outer: bad arguments");
    BSC_REQUIRE(N == 0 || (R && X && Y), "bsc_weighted_outer: null pointer");
    if (K > WO_MAXK || D > WO_MAXD || E > WO_MAXD || (K | D | E) % 4 != 0)
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_weighted_outer: K=%d D=%d E=%d (need multiples of 4, K<=%d, D,E<=%d)", K,
                        D, E, WO_MAXK, WO_MAXD);
    if (N > 0 && ((ldr | ldx | ldy) % 4 != 0 || (((uintptr_t)R | (uintptr_t)X | (uintptr_t)Y) & 15) != 0))
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_weighted_outer: operands must be 16-byte aligned with leading "
                        "dimensions that are multiples of 4");
    BSC_REQUIRE(N == 0 || (ldr >= K && ldx >= D && ldy >= E), "bsc_weighted_outer: leading dimension");
    WOArgs a{};
    a.R = R; a.X = X; a.Y = Y;
    a.ldr = ldr; a.ldx = ldx; a.ldy = ldy; a.N = N;
    a.K = K; a.D = D; a.E = E;
    a.sym = (X == Y && ldx == ldy && D == E) ? 1 : 0;
    a.P = a.sym ? D * (D + 1) / 2 : D * E;
    const int KT = (K + 31) / 32;
    const int n_ct = (a.P + 31) / 32;
    const int gy = (n_ct + WO_MAXCT - 1) / WO_MAXCT;
    const int CT = (n_ct + gy - 1) / gy;
    const int64_t stages = (N + WO_TR - 1) / WO_TR;
    // about 20 waves per CU (LDS allows five workgroups), spread over the pair-tile groups
    const int wg_per_cu = std::max(1, std::min(5, 20 / CT));
    int64_t gx = std::max<int64_t>(1, (int64_t)wg_per_cu * ctx->cu_count / gy);
    gx = std::min(gx, std::max<int64_t>(stages, 1));
    a.iters = (int)((stages + gx - 1) / gx);
    if (stages > 0) gx = (stages + a.iters - 1) / a.iters;   // no workgroup without a stage
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)gy * gx * KT * CT * WO_TILE * sizeof(float), &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    a.part = (float*)ws;
    if (stages > 0) {
        bsc_prof_scope prof(ctx);
        const dim3 grid((unsigned)gx, (unsigned)gy);
        const dim3 block((unsigned)(64 * CT));
        if (KT == 1 && a.sym) hipLaunchKernelGGL((weighted_outer_kernel<1, true>), grid, block, 0, ctx->stream, a);
        else if (KT == 1) hipLaunchKernelGGL((weighted_outer_kernel<1, false>), grid, block, 0, ctx->stream, a);
        else if (a.sym) hipLaunchKernelGGL((weighted_outer_kernel<2, true>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((weighted_outer_kernel<2, false>), grid, block, 0, ctx->stream, a);
        BSC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(weighted_outer_finish_kernel, dim3((unsigned)(K * n_ct)), dim3(256), 0,
                       ctx->stream, (const float*)ws, stages > 0 ? (int)gx : 0, KT, CT, (int)K, (int)D,
                       (int)E, a.sym, a.P, scale, out);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}
