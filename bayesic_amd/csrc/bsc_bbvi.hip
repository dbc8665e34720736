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

__global__ __launch_bounds__(BB_BLOCK) void bbvi_moments_kernel(
    const double* __restrict__ lam, const double* __restrict__ eps, const double* __restrict__ f,
    int P, int S, double* __restrict__ cov_part, double* __restrict__ var_part) {
    __shared__ double fs[BB_MAX_S];
    __shared__ double red[BB_WAVES][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < S) fs[tid] = f[tid];
    __syncthreads();
    const int i = blockIdx.x * BB_BLOCK + tid;
    double cov = 0.0, var = 0.0;
    if (i < 2 * P) {
        const int p = i < P ? i : i - P;
        const double inv_sd = exp(-lam[P + p]);
        double mh = 0.0, mfh = 0.0;
        for (int s = 0; s < S; ++s) {
            const double h = score_component(eps, s, P, i, p, inv_sd);
            mh += h;
            mfh += fs[s] * h;
        }
        mh /= S;
        mfh /= S;
        double c = 0.0, v = 0.0;
        for (int s = 0; s < S; ++s) {
            const double h = score_component(eps, s, P, i, p, inv_sd);
            c += (fs[s] * h - mfh) * (h - mh);
            v += (h - mh) * (h - mh);
        }
        cov = c / (S - 1);
        var = v / (S - 1);
    }
    cov = wave_allsum_f64(cov);
    var = wave_allsum_f64(var);
    if (lane == 0) { red[wave][0] = cov; red[wave][1] = var; }
    __syncthreads();
    if (tid == 0) {
        double c = 0.0, v = 0.0;
        for (int k = 0; k < BB_WAVES; ++k) { c += red[k][0]; v += red[k][1]; }
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
    double c = 0.0, v = 0.0;                       // every thread: the same fixed-order sums
    for (int k = 0; k < n_parts; ++k) { c += cov_part[k]; v += var_part[k]; }
    const double a = c / v;
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

}  // namespace

extern "C" {

int bsc_logreg_bbvi_loglik(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y,
                           const int32_t* g, int64_t N, int32_t D, int32_t n_groups,
                           const float* Wz, const float* Bz, int32_t S, double* ell) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(N >= 0 && ((X && y && g) || N == 0) && Wz && Bz && ell,
                "bsc_logreg_bbvi_loglik: null pointer");
    if (S != LS || D < 4 || D > LD || (D % 4) != 0)
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_logreg_bbvi_loglik: needs S == %d and D %% 4 == 0 in [4,%d] (got S=%d D=%d)",
                        LS, LD, S, D);
    BSC_REQUIRE(n_groups >= 1, "bsc_logreg_bbvi_loglik: n_groups=%d", n_groups);
    if (n_groups > (1 << 22))
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_logreg_bbvi_loglik: n_groups=%d exceeds the 32-bit gather range (4M groups)",
                        n_groups);
    BSC_REQUIRE(ldx >= D && ldx % 4 == 0 && ldx < ((int64_t)1 << 26),
                "bsc_logreg_bbvi_loglik: bad ldx=%lld", (long long)ldx);
    BSC_REQUIRE(((uintptr_t)X & 15) == 0, "bsc_logreg_bbvi_loglik: X must be 16-byte aligned");
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
    const int n_parts = (2 * P + BB_BLOCK - 1) / BB_BLOCK;
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
    hipLaunchKernelGGL(bbvi_finish_kernel, dim3((unsigned)n_parts), dim3(BB_BLOCK), 0, ctx->stream,
                       lam, eps, f, P, (int)S, cov_part, var_part, n_parts, elbo, grad);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

}  // extern "C"
