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
// R, X (and Y) in LDS; a wave takes one component tile and a part of the rows for all pair
// tiles of the workgroup, so its k-step is 1 + 2 CT LDS reads and CT multiplies against CT
// MFMAs.  Bound: fp32 MFMA.  Partial sums are added by a fixed-order float64 finish.
#include "bsc_common.h"

namespace {

constexpr int WO_TR = 64;       // rows per stage
constexpr int WO_BLOCK = 256;
constexpr int WO_MAXK = 64;
constexpr int WO_MAXD = 32;
constexpr int WO_MAXCT = 8;     // pair tiles per workgroup
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

// Four waves per workgroup, one per SIMD; each takes a quarter of the stage's rows for EVERY
// (component tile, pair tile) of the workgroup -- KT * CT accumulators of 16 registers, which
// the compiler keeps in AGPRs.  On this part VALU instructions between MFMAs cost the MFMA
// pipe their full issue time (profiles/r01_ubench_mfma_valu_mix.txt), so the kernel is built
// to issue as few as possible: one multiply per pair tile and k-step, shared by both
// component tiles; rows arrive through buffer loads whose per-stage offsets are scalar
// (rows past N read as zero, no per-lane address arithmetic or compares).
// Measured on the way here (10M x 16, K = 64): a wave per pair tile (5 waves) 1.90 ms -- one
// SIMD carries two of them; wave = (component tile, row half) with a multiply per MFMA 1.77 ms.
struct WoRows {   // buffer descriptor over the rows of one stage
    __amdgpu_buffer_rsrc_t rsrc;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wo_rsrc(const float* base, int64_t ld, int cols,
                                                          int64_t row0, int64_t N) {
    const int64_t rem = N - row0;
    uint64_t bytes = 0;
    if (rem > 0) bytes = ((uint64_t)(rem - 1) * (uint64_t)ld + (uint64_t)cols) * 4u;
    const unsigned rec = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (rem > 0 ? row0 : 0) * ld), 0, rec, 0x00020000);
}

// lanes that are not `ok` keep what they hold (zero, set once before the loop)
__device__ __forceinline__ void wo_bload(float4& f, __amdgpu_buffer_rsrc_t r, int voff, int soff, bool ok) {
    if (ok) {
        auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        f = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                        __uint_as_float(v[3]));
    }
}

template <int KT, int CT>
__global__ __launch_bounds__(WO_BLOCK, 2) void weighted_outer_kernel(WOArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[WO_TR * (WO_MAXK + 2 * WO_MAXD)];
    constexpr int RS = KT * 32;
    constexpr int ROWS = WO_TR / 4;            // rows of a stage per wave
    float* Rs = lds;
    float* Xs = lds + WO_TR * WO_MAXK;
    float* Ys = a.sym ? Xs : Xs + WO_TR * WO_MAXD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;

    int xo[CT], yo[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int p = ((int)blockIdx.y * CT + ct) * 32 + col;
        int d = 0, e = 0;
        if (p < a.P) wo_pair(p, a.D, a.E, a.sym, d, e);
        xo[ct] = d;
        yo[ct] = e;
    }
    f32x16 acc[KT][CT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[kt][ct][q] = 0.f;

    // a stage: R has 16 float4 slots per row (zero beyond K), X / Y have 8
    constexpr int NR = WO_TR * 16 / WO_BLOCK, NX = WO_TR * 8 / WO_BLOCK;
    const int rr = tid >> 4, rc = tid & 15, xr = tid >> 3, xc = tid & 7;
    const bool r_ok = rc < (a.K >> 2), x_ok = xc < (a.D >> 2), y_ok = !a.sym && xc < (a.E >> 2);
    const int r_voff = (int)(rr * a.ldr + rc * 4) * 4, r_step = (int)(a.ldr * 4) * (WO_BLOCK / 16);
    const int x_voff = (int)(xr * a.ldx + xc * 4) * 4, x_step = (int)(a.ldx * 4) * (WO_BLOCK / 8);
    const int y_voff = (int)(xr * a.ldy + xc * 4) * 4, y_step = (int)(a.ldy * 4) * (WO_BLOCK / 8);
    float4 pr[NR], px[NX], py[NX];
#pragma unroll
    for (int i = 0; i < NR; ++i) pr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NX; ++i) px[i] = py[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#define WO_FETCH(IT)                                                                          \
    {                                                                                         \
        const int64_t row0_ = ((int64_t)blockIdx.x + (int64_t)(IT) * gridDim.x) * WO_TR;       \
        const auto rd = wo_rsrc(a.R, a.ldr, a.K, row0_, a.N);                                 \
        const auto xd = wo_rsrc(a.X, a.ldx, a.D, row0_, a.N);                                 \
        const auto yd = wo_rsrc(a.Y, a.ldy, a.E, row0_, a.N);                                 \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) wo_bload(pr[i], rd, r_voff, i * r_step, r_ok); \
        _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                      \
            wo_bload(px[i], xd, x_voff, i * x_step, x_ok);                                    \
            wo_bload(py[i], yd, y_voff, i * y_step, y_ok);                                    \
        }                                                                                     \
    }
    const float* rrow = Rs + (wave * ROWS + half) * RS + col;
    const float* xrow = Xs + (wave * ROWS + half) * WO_MAXD;
    const float* yrow = Ys + (wave * ROWS + half) * WO_MAXD;
    if (a.iters > 0) WO_FETCH(0)
    for (int it = 0; it < a.iters; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (rc < KT * 8) *(float4*)(Rs + (rr + i * (WO_BLOCK / 16)) * RS + rc * 4) = pr[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            *(float4*)(Xs + (xr + i * (WO_BLOCK / 8)) * WO_MAXD + xc * 4) = px[i];
            if (!a.sym) *(float4*)(Ys + (xr + i * (WO_BLOCK / 8)) * WO_MAXD + xc * 4) = py[i];
        }
        __syncthreads();
        // the next stage's rows travel through registers while this one is multiplied
        if (it + 1 < a.iters) WO_FETCH(it + 1)
#undef WO_FETCH
#pragma unroll
        for (int j = 0; j < ROWS / 2; ++j) {
            float av[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) av[kt] = rrow[2 * j * RS + kt * 32];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float z = xrow[2 * j * WO_MAXD + xo[ct]] * yrow[2 * j * WO_MAXD + yo[ct]];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    acc[kt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kt], z, acc[kt][ct], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // the four waves hold sums over disjoint rows: add them through LDS tile by tile, every
    // thread finishing four elements of the tile
    float* red = lds;   // 4 x 1024 floats
    float* out = a.part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (KT * CT) * WO_TILE;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int q = 0; q < 16; ++q) red[wave * WO_TILE + q * 64 + lane] = acc[kt][ct][q];
            __syncthreads();
            for (int i = tid; i < WO_TILE; i += WO_BLOCK)
                out[(kt * CT + ct) * WO_TILE + i] =
                    (red[i] + red[WO_TILE + i]) + (red[2 * WO_TILE + i] + red[3 * WO_TILE + i]);
            __syncthreads();
        }
}

template <int KT>
void launch_wo(int CT, dim3 grid, hipStream_t stream, const WOArgs& a) {
    const dim3 block(WO_BLOCK);
    switch (CT) {
        case 1: hipLaunchKernelGGL((weighted_outer_kernel<KT, 1>), grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((weighted_outer_kernel<KT, 2>), grid, block, 0, stream, a); break;
        case 3: hipLaunchKernelGGL((weighted_outer_kernel<KT, 3>), grid, block, 0, stream, a); break;
        case 4: hipLaunchKernelGGL((weighted_outer_kernel<KT, 4>), grid, block, 0, stream, a); break;
        case 5: hipLaunchKernelGGL((weighted_outer_kernel<KT, 5>), grid, block, 0, stream, a); break;
        case 6: hipLaunchKernelGGL((weighted_outer_kernel<KT, 6>), grid, block, 0, stream, a); break;
        case 7: hipLaunchKernelGGL((weighted_outer_kernel<KT, 7>), grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL((weighted_outer_kernel<KT, 8>), grid, block, 0, stream, a); break;
    }
}

// out[k][d][e] (and [k][e][d] when symmetric) = scale * sum over workgroups, float64, fixed
// order.  A block owns one (component, 32 pairs) line: 32 groups of 32 lanes walk the
// workgroups interleaved, then combine in group order.
__global__ __launch_bounds__(1024) void weighted_outer_finish_kernel(const float* __restrict__ part,
                                                                     int gx, int KT, int CT, int K,
                                                                     int D, int E, int sym, int P,
                                                                     double scale,
                                                                     float* __restrict__ out) {
    constexpr int G = 32;
    __shared__ double comb[G][32];
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
    int b = grp;
    for (; b + 3 * G < gx; b += 4 * G) {
        const float v0 = src[(int64_t)b * wg_stride], v1 = src[(int64_t)(b + G) * wg_stride];
        const float v2 = src[(int64_t)(b + 2 * G) * wg_stride], v3 = src[(int64_t)(b + 3 * G) * wg_stride];
        s += (double)v0;
        s += (double)v1;
        s += (double)v2;
        s += (double)v3;
    }
    for (; b < gx; b += G) s += (double)src[(int64_t)b * wg_stride];
    comb[grp][j] = s;
    __syncthreads();
    if (grp != 0) return;
    double tot = comb[0][j];
#pragma unroll
    for (int g = 1; g < G; ++g) tot += comb[g][j];
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
    if ((ldr | ldx | ldy) >= ((int64_t)1 << 22))   // 64 rows x ld x 4 B must fit a 32-bit buffer offset
        return bsc_fail(BSC_ERR_UNSUPPORTED, "bsc_weighted_outer: leading dimensions must be below 2^22");
    WOArgs a{};
    a.R = R; a.X = X; a.Y = Y;
    a.ldr = ldr; a.ldx = ldx; a.ldy = ldy; a.N = N;
    a.K = K; a.D = D; a.E = E;
    a.sym = (X == Y && ldx == ldy && D == E) ? 1 : 0;
    a.P = a.sym ? D * (D + 1) / 2 : D * E;
    const int KT = (K + 31) / 32;
    const int n_ct = (a.P + 31) / 32;
    const int max_ct = KT == 2 ? 5 : WO_MAXCT;      // KT * CT accumulators of 16 registers each
    const int gy = (n_ct + max_ct - 1) / max_ct;
    const int CT = (n_ct + gy - 1) / gy;
    const int64_t stages = (N + WO_TR - 1) / WO_TR;
    // two workgroups per CU (up to 160 accumulator registers per wave)
    const int wg_per_cu = ctx->wo_wg_per_cu;
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
        if (KT == 1) launch_wo<1>(CT, grid, ctx->stream, a);
        else launch_wo<2>(CT, grid, ctx->stream, a);
        BSC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(weighted_outer_finish_kernel, dim3((unsigned)(K * n_ct)), dim3(1024), 0,
                       ctx->stream, (const float*)ws, stages > 0 ? (int)gx : 0, KT, CT, (int)K, (int)D,
                       (int)E, a.sym, a.P, scale, out);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}
