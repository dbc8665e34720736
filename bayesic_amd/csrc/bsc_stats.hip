// Summed sufficient statistics of iid draws and the parameter-update kernels.
//
// Reference contract: ExpFamIndependentObservations.sufficient_statistics sums
// the per-datum statistics over the iid-draw axes
// (bayesic/distribution/base.py:328-332); Normal t(x) = (x, x^2)
// (bayesic/distribution/core.py:16-17).  The natural-gradient step is the
// README.md:36 "unit-step natural gradient" VMP update generalised to a step
// rho (Hoffman et al., ref [4], README.md:75-77).
#include "bsc_common.h"

namespace {

constexpr int STAT_BLOCK = 256;
constexpr int STAT_WAVES = STAT_BLOCK / BSC_WAVE;

// Stage 1: grid-stride, 16 B/lane loads, float64 accumulation per lane, wave
// and block reduction in fixed order -> partial[block][2].
__global__ __launch_bounds__(STAT_BLOCK) void normal_stats_partial_kernel(
    const float* __restrict__ x, int64_t n, double* __restrict__ partial) {
    __shared__ double red[STAT_WAVES][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s1 = 0.0, s2 = 0.0;
    // x may be only 4-byte aligned: peel to a 16-byte boundary
    const int64_t head = (int64_t)((16 - ((uintptr_t)x & 15)) & 15) / 4;
    const int64_t h = head < n ? head : n;
    const int64_t nvec = (n - h) / 4;
    const float4* xv = reinterpret_cast<const float4*>(x + h);
    const int64_t gstride = (int64_t)gridDim.x * STAT_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * STAT_BLOCK + tid; i < nvec; i += gstride) {
        float4 v = xv[i];
        double a = v.x, b = v.y, c = v.z, d = v.w;
        s1 += (a + b) + (c + d);
        s2 += (a * a + b * b) + (c * c + d * d);
    }
    if (blockIdx.x == 0) {
        // scalar head and tail elements
        const int64_t tail0 = h + nvec * 4;
        for (int64_t i = tid; i < h; i += STAT_BLOCK) { double a = x[i]; s1 += a; s2 += a * a; }
        for (int64_t i = tail0 + tid; i < n; i += STAT_BLOCK) { double a = x[i]; s1 += a; s2 += a * a; }
    }
    s1 = wave_allsum_f64(s1);
    s2 = wave_allsum_f64(s2);
    if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; }
    __syncthreads();
    if (tid == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int k = 0; k < STAT_WAVES; ++k) { t1 += red[k][0]; t2 += red[k][1]; }
        partial[2 * blockIdx.x] = t1;
        partial[2 * blockIdx.x + 1] = t2;
    }
}

__global__ __launch_bounds__(BSC_WAVE) void normal_stats_final_kernel(
    const double* __restrict__ partial, int n_blocks, int64_t n, double* __restrict__ stats) {
    const int lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < n_blocks; b += BSC_WAVE) {
        s1 += partial[2 * b];
        s2 += partial[2 * b + 1];
    }
    s1 = wave_allsum_f64(s1);
    s2 = wave_allsum_f64(s2);
    if (lane == 0) {
        stats[0] = (double)n;
        stats[1] = s1;
        stats[2] = s2;
    }
}

#pragma clang fp contract(off)
__global__ void adam_ascent_kernel(double* __restrict__ lam, const double* __restrict__ grad,
                                   double* __restrict__ m1, double* __restrict__ m2, int64_t n,
                                   double lr, double beta1, double beta2, double eps,
                                   double corr1, double corr2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double g = grad[i];
    const double a = beta1 * m1[i] + (1.0 - beta1) * g;
    const double b = beta2 * m2[i] + (1.0 - beta2) * g * g;
    m1[i] = a;
    m2[i] = b;
    const double mhat = a / corr1;
    const double vhat = b / corr2;
    lam[i] = lam[i] + lr * mhat / (sqrt(vhat) + eps);
}

__global__ void natgrad_update_kernel(double* __restrict__ eta, const double* __restrict__ eta0,
                                      const double* __restrict__ message, int64_t n,
                                      double scale, double rho) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    eta[i] = (1.0 - rho) * eta[i] + rho * (eta0[i] + scale * message[i]);
}

// ---- Dirichlet expectation for LDA-style nodes (config 4) ---------------------
// out[r, c] = exp( E[log theta_rc] ) = exp( digamma(lam[r,c]) - digamma(sum_c lam[r,c]) )
// (Hoffman, Blei, Bach 2010; the local/global expectations of README.md:69-79's
// "SVI [4]" applied to a Dirichlet-Multinomial model).  Row sums in float64, fixed
// order: one workgroup per row for the sum, then an element-wise pass.

__device__ __forceinline__ double digamma_f64_stats(double x) { return bsc_digamma_f64(x); }

__global__ __launch_bounds__(1024) void row_sum_kernel(const float* __restrict__ lam, int64_t cols,
                                                       int64_t ld, double* __restrict__ row_psi) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = lam + (int64_t)blockIdx.x * ld;
    double acc = 0.0;
    if ((ld & 3) == 0 && (((uintptr_t)lam) & 15) == 0) {
        // 16-byte loads, four in flight per thread (a row of 100 000 floats took 47 us with
        // one 4-byte load at a time); the order stays fixed
        const int64_t n4 = cols >> 2;
        const float4* row4 = reinterpret_cast<const float4*>(row);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int64_t c = tid;
        for (; c + 3 * 1024 < n4; c += 4 * 1024) {
            const float4 v0 = row4[c], v1 = row4[c + 1024], v2 = row4[c + 2048], v3 = row4[c + 3072];
            a0 += ((double)v0.x + (double)v0.y) + ((double)v0.z + (double)v0.w);
            a1 += ((double)v1.x + (double)v1.y) + ((double)v1.z + (double)v1.w);
            a2 += ((double)v2.x + (double)v2.y) + ((double)v2.z + (double)v2.w);
            a3 += ((double)v3.x + (double)v3.y) + ((double)v3.z + (double)v3.w);
        }
        for (; c < n4; c += 1024) {
            const float4 v0 = row4[c];
            a0 += ((double)v0.x + (double)v0.y) + ((double)v0.z + (double)v0.w);
        }
        acc = (a0 + a1) + (a2 + a3);
        for (int64_t t = 4 * n4 + tid; t < cols; t += 1024) acc += (double)row[t];
    } else {
        for (int64_t c = tid; c < cols; c += 1024) acc += (double)row[c];
    }
    acc = wave_allsum_f64(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k];
        row_psi[blockIdx.x] = digamma_f64_stats(t);
    }
}

// exp(psi(x) - c) without the log/exp round trip: with x shifted up to y >= 8,
//   psi(x) = log y - 1/(2y) - series(1/y^2) - P'(x)/P(x),  P = prod_{i<n} (x + i),
// so exp(psi(x) - c) = y * exp(-(1/(2y) + P'/P) - series - c); the two quotients share one
// division.  float64 throughout (the result is rounded to float32 once): the kernel is bound by
// float64 instruction count, not by its 8 bytes per element.
__device__ __forceinline__ double exp_digamma_minus(double x, double c) {
#pragma clang fp contract(off)
    double P = 1.0, dP = 0.0;
    while (x < 8.0) {
        dP = dP * x + P;
        P *= x;
        x += 1.0;
    }
    const double den = 2.0 * P * x;
    const double rden = 1.0 / den;
    const double quot = (2.0 * dP * x + P) * rden;        // 1/(2y) + P'/P
    const double inv = 2.0 * P * rden, inv2 = inv * inv;  // 1/y
    const double series = inv2 * (1.0 / 12.0 - inv2 * (1.0 / 120.0 - inv2 * (1.0 / 252.0 - inv2 *
                          (1.0 / 240.0 - inv2 * (5.0 / 660.0 - inv2 * (691.0 / 32760.0))))));
    return x * exp(-quot - series - c);
}

__global__ __launch_bounds__(256) void dirichlet_expect_kernel(const float* __restrict__ lam,
                                                               int64_t rows, int64_t cols,
                                                               int64_t ld,
                                                               const double* __restrict__ row_psi,
                                                               float* __restrict__ out) {
    const int64_t n = rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / cols, c = i - r * cols;
        out[i] = (float)exp_digamma_minus((double)lam[r * ld + c], row_psi[r]);
    }
}

// ---- ... and the factor's part of the evidence lower bound ----------------------------------------
// bound = sum_r -KL(Dir(lam_r) || Dir(prior)) = sum_rc [(prior - lam_rc) E[log theta_rc] + lnGamma(lam_rc)]
//         + sum_r [-lnGamma(sum_c lam_rc) + lnGamma(cols prior) - cols lnGamma(prior)]
// (README.md:30-37; <eta0 - eta, E[T]> - A(eta0) + A(eta) with T = log theta, eta = lam - 1, the
// decomposition of bayesic/distribution/base.py:47-69; oracle.svi.dirichlet_neg_kl).  psi and lnGamma of
// an element share the shift to y >= 8 and log y, so the bound costs one more log and a short series
// per element on top of the expectation.  Per-block float64 partials, added in block order.
__global__ __launch_bounds__(1024) void row_sum_bound_kernel(const float* __restrict__ lam, int64_t cols,
                                                             int64_t ld, double prior,
                                                             double* __restrict__ row_psi,
                                                             double* __restrict__ row_const) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = lam + (int64_t)blockIdx.x * ld;
    double acc = 0.0;
    if ((ld & 3) == 0 && (((uintptr_t)lam) & 15) == 0) {       // (the loop of row_sum_kernel: same order, same sums)
        const int64_t n4 = cols >> 2;
        const float4* row4 = reinterpret_cast<const float4*>(row);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int64_t c = tid;
        for (; c + 3 * 1024 < n4; c += 4 * 1024) {
            const float4 v0 = row4[c], v1 = row4[c + 1024], v2 = row4[c + 2048], v3 = row4[c + 3072];
            a0 += ((double)v0.x + (double)v0.y) + ((double)v0.z + (double)v0.w);
            a1 += ((double)v1.x + (double)v1.y) + ((double)v1.z + (double)v1.w);
            a2 += ((double)v2.x + (double)v2.y) + ((double)v2.z + (double)v2.w);
            a3 += ((double)v3.x + (double)v3.y) + ((double)v3.z + (double)v3.w);
        }
        for (; c < n4; c += 1024) {
            const float4 v0 = row4[c];
            a0 += ((double)v0.x + (double)v0.y) + ((double)v0.z + (double)v0.w);
        }
        acc = (a0 + a1) + (a2 + a3);
        for (int64_t t = 4 * n4 + tid; t < cols; t += 1024) acc += (double)row[t];
    } else {
        for (int64_t c = tid; c < cols; c += 1024) acc += (double)row[c];
    }
    acc = wave_allsum_f64(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k];
        row_psi[blockIdx.x] = digamma_f64_stats(t);
        // -lnGamma(sum_c lam) + lnGamma(cols prior) - cols lnGamma(prior): one Stirling body for the three
        double args[3] = {t, (double)cols * prior, prior}, lg[3];
#pragma unroll 1
        for (int q = 0; q < 3; ++q) lg[q] = bsc_lgamma_f64(args[q]);
        row_const[blockIdx.x] = -lg[0] + lg[1] - (double)cols * lg[2];
    }
}

// One log more than the expectation alone: psi and lnGamma of an element share the shift to y >= 8,
//   psi(x)     = log y - 1/(2y) - series_psi(1/y^2) - P'/P
//   lnGamma(x) = (y - 1/2) log y - y + log(2 pi)/2 + series_lg(1/y) - log P,      P = prod_{i<n} (x + i),
// the two quotients come from ONE division, and the log P of successive elements are taken together:
// a running product, renormalised (one log) only when it nears the float64 range.
__global__ __launch_bounds__(256) void dirichlet_expect_bound_kernel(const float* __restrict__ lam,
                                                                     int64_t rows, int64_t cols, int64_t ld,
                                                                     double prior,
                                                                     const double* __restrict__ row_psi,
                                                                     float* __restrict__ out,
                                                                     double* __restrict__ partial) {
#pragma clang fp contract(off)
    __shared__ double red[4];
    const int64_t n = rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0, prodP = 1.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / cols, c = i - r * cols;
        const double x0 = (double)lam[r * ld + c];
        double x = x0, P = 1.0, dP = 0.0;
        while (x < 8.0) {
            dP = dP * x + P;
            P *= x;
            x += 1.0;
        }
        const double rden = 1.0 / (P * x);
        const double inv = P * rden, inv2 = inv * inv;           // 1 / y
        const double logy = log(x);
        const double psi_series = inv2 * (1.0 / 12.0 - inv2 * (1.0 / 120.0 - inv2 * (1.0 / 252.0 - inv2 *
                                  (1.0 / 240.0 - inv2 * (5.0 / 660.0 - inv2 * (691.0 / 32760.0))))));
        const double elog = logy - 0.5 * inv - psi_series - dP * x * rden - row_psi[r];     // E[log theta_rc]
        out[i] = (float)exp(elog);
        const double lg_series = inv * (1.0 / 12.0 - inv2 * (1.0 / 360.0 - inv2 * (1.0 / 1260.0 - inv2 *
                                 (1.0 / 1680.0 - inv2 * (1.0 / 1188.0 - inv2 * (691.0 / 360360.0))))));
        acc += (prior - x0) * elog + ((x - 0.5) * logy - x + 0.91893853320467274178032973640562 + lg_series);
        prodP *= P;
        if (__builtin_expect(prodP > 1.0e250 || prodP < 1.0e-250, 0)) {
            acc -= log(prodP);
            prodP = 1.0;
        }
    }
    acc -= log(prodP);
    acc = wave_allsum_f64(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = scale * (sum of a[0..n) and b[0..m)), fixed order: lane partials in index order (eight independent
// loads in flight per lane), lanes by the fixed butterfly, waves in order
__global__ __launch_bounds__(256) void ordered_sum2_kernel(const double* __restrict__ a, int64_t n,
                                                           const double* __restrict__ b, int64_t m,
                                                           double scale, double* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        const double* src = pass ? b : a;
        const int64_t len = pass ? m : n;
        for (int64_t i0 = 0; i0 < len; i0 += 256 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + u * 256 + threadIdx.x;
                v[u] = i < len ? src[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    acc = wave_allsum_f64(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = scale * ((red[0] + red[1]) + (red[2] + red[3]));
}

// With `elbo`: elbo[0] = scale * (ll[0] + local_bound[0]) + global_bound[0] before the step (the bound at
// the lambda the statistics were taken with).
__global__ void natgrad_update_f32_kernel(float* __restrict__ eta, float eta0,
                                          const float* __restrict__ message, int64_t n,
                                          float scale, float rho, const double* __restrict__ ll,
                                          const double* __restrict__ local_bound,
                                          const double* __restrict__ global_bound, double scale64,
                                          double* __restrict__ elbo) {
    if (elbo && blockIdx.x == 0 && threadIdx.x == 0)
        elbo[0] = scale64 * (ll[0] + local_bound[0]) + global_bound[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        eta[i] = (1.0f - rho) * eta[i] + rho * (eta0 + scale * message[i]);
}

// The same step on a [rows, cols] block of eta and a message with leading dimensions of their own: the piece of lambda a
// finished piece of the statistic belongs to (svi/lda.py: the statistic is taken and all-reduced in column ranges, each
// staged contiguously).  `n_ll` words' terms are added in index order into the bound.
__global__ void natgrad_update_f32_2d_kernel(float* __restrict__ eta, int64_t ld_eta, float eta0,
                                             const float* __restrict__ message, int64_t ld_msg, int64_t rows,
                                             int64_t cols, float scale, float rho, const double* __restrict__ ll,
                                             int n_ll, const double* __restrict__ local_bound,
                                             const double* __restrict__ global_bound, double scale64,
                                             double* __restrict__ elbo) {
    if (elbo && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        double words = 0.0;
        for (int i = 0; i < n_ll; ++i) words += ll[i];
        elbo[0] = scale64 * (words + local_bound[0]) + global_bound[0];
    }
    // blockIdx.y walks the rows, the x extent of the grid a row's columns: no division per element
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        float* e = eta + r * ld_eta;
        const float* m = message + r * ld_msg;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += (int64_t)gridDim.x * blockDim.x)
            e[c] = (1.0f - rho) * e[c] + rho * (eta0 + scale * m[c]);
    }
}

// ---- row softmax: the expectation of a Categorical node (responsibilities) --------------------
// out[r, :] = softmax(in[r, :]), lse[r] = log sum exp in[r, :]; float32 in/out, the row's max and
// sum in float32 with exp2 on pre-scaled values.  cols <= 64: a row occupies P = 2^ceil(log2 cols)
// lanes and a wave takes 64 / P rows per step (coalesced when the rows are contiguous); wider
// rows: one wave per row, values held in registers between the two reductions (cols <= 1024).
template <int P>
__global__ __launch_bounds__(256) void softmax_rows_small_kernel(const float* __restrict__ in, int64_t rows,
                                                                 int cols, int64_t ld_in, float* __restrict__ out,
                                                                 int64_t ld_out, float* __restrict__ lse) {
    constexpr int RPW = 64 / P;                     // rows per wave step
    const int lane = threadIdx.x & 63;
    const int c = lane % P, sub = lane / P;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    for (int64_t r0 = wave * RPW; r0 < rows; r0 += n_waves * RPW) {
        const int64_t r = r0 + sub;
        const bool live = r < rows && c < cols;
        const float v = live ? in[r * ld_in + c] * LOG2E : -INFINITY;
        float m = v;
#pragma unroll
        for (int off = 1; off < P; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
        const float e = live ? __builtin_amdgcn_exp2f(v - m) : 0.f;
        float z = e;
#pragma unroll
        for (int off = 1; off < P; off <<= 1) z += __shfl_xor(z, off);
        if (live) out[r * ld_out + c] = e * __builtin_amdgcn_rcpf(z) * (2.0f - z * __builtin_amdgcn_rcpf(z));
        if (lse && r < rows && c == 0) lse[r] = (m + __builtin_amdgcn_logf(z)) * LN2;
    }
}

// cols % 4 == 0, 16-byte aligned rows, cols <= 256: a lane takes four consecutive columns, a row
// occupies L = 2^ceil(log2(cols / 4)) lanes, a wave load covers 64 / L rows (1 KiB of a dense matrix
// at cols = 64, where one lane per column moved 256 B per load and shuffled 12 times per row); two
// row groups in flight per wave.
template <int L>
__global__ __launch_bounds__(256) void softmax_rows_vec4_kernel(const float* __restrict__ in, int64_t rows,
                                                                int cols, int64_t ld_in, float* __restrict__ out,
                                                                int64_t ld_out, float* __restrict__ lse) {
    constexpr int RPW = 64 / L;
    const int lane = threadIdx.x & 63;
    const int c = 4 * (lane % L), sub = lane / L;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    // (round 3: four row groups in flight instead of two, streaming loads and stores -- 3.0 -> see DESIGN 4)
    constexpr int U = 4;
    typedef float sm_f32x4 __attribute__((ext_vector_type(4)));
    for (int64_t r0 = wave * U * RPW; r0 < rows; r0 += n_waves * U * RPW) {
        float4 x[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * RPW + sub;
            live[u] = r < rows && c < cols;
            if (live[u]) {
                const sm_f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const sm_f32x4*>(in + r * ld_in + c));
                x[u] = make_float4(w.x, w.y, w.z, w.w);
            } else {
                x[u] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * RPW + sub;
            float4 v = make_float4(x[u].x * LOG2E, x[u].y * LOG2E, x[u].z * LOG2E, x[u].w * LOG2E);
            float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
            for (int off = 1; off < L; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live[u])
                e = make_float4(__builtin_amdgcn_exp2f(v.x - m), __builtin_amdgcn_exp2f(v.y - m),
                                __builtin_amdgcn_exp2f(v.z - m), __builtin_amdgcn_exp2f(v.w - m));
            float z = (e.x + e.y) + (e.z + e.w);
#pragma unroll
            for (int off = 1; off < L; off <<= 1) z += __shfl_xor(z, off);
            const float rz = __builtin_amdgcn_rcpf(z), inv = rz * (2.0f - z * rz);
            if (live[u]) {
                const sm_f32x4 w = {e.x * inv, e.y * inv, e.z * inv, e.w * inv};
                __builtin_nontemporal_store(w, reinterpret_cast<sm_f32x4*>(out + r * ld_out + c));
            }
            if (lse && r < rows && c == 0) lse[r] = (m + __builtin_amdgcn_logf(z)) * LN2;
        }
    }
}

__global__ __launch_bounds__(256) void softmax_rows_wide_kernel(const float* __restrict__ in, int64_t rows, int cols,
                                                                int64_t ld_in, float* __restrict__ out,
                                                                int64_t ld_out, float* __restrict__ lse) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    for (int64_t r = wave; r < rows; r += n_waves) {
        float v[16];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < cols ? in[r * ld_in + c] * LOG2E : -INFINITY;
            m = fmaxf(m, v[j]);
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v[j] = lane + 64 * j < cols ? __builtin_amdgcn_exp2f(v[j] - m) : 0.f;
            z += v[j];
        }
        z = wave_allsum(z);
        const float inv = 1.0f / z;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = lane + 64 * j;
            if (c < cols) out[r * ld_out + c] = v[j] * inv;
        }
        if (lse && lane == 0) lse[r] = (m + __builtin_amdgcn_logf(z)) * LN2;
    }
}

}  // namespace

extern "C" {

int bsc_dirichlet_expectation(bsc_ctx* ctx, const float* lam, int64_t rows, int64_t cols,
                              int64_t ld, float* out) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(lam && out && rows >= 1 && cols >= 1 && ld >= cols && rows <= 65535 * 32,
                "bsc_dirichlet_expectation: bad arguments");
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)rows * sizeof(double), &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)rows), dim3(1024), 0, ctx->stream, lam, cols,
                       ld, (double*)ws);
    BSC_LAUNCH_CHECK();
    int64_t blocks = (rows * cols + 255) / 256;
    if (blocks > 8 * (int64_t)ctx->cu_count) blocks = 8 * (int64_t)ctx->cu_count;
    hipLaunchKernelGGL(dirichlet_expect_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       lam, rows, cols, ld, (const double*)ws, out);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_softmax_rows(bsc_ctx* ctx, const float* in, int64_t rows, int64_t cols, int64_t ld_in, float* out,
                     int64_t ld_out, float* lse) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(rows >= 0 && cols >= 1 && ld_in >= cols && ld_out >= cols && ((in && out) || rows == 0),
                "bsc_softmax_rows: bad arguments");
    if (cols > 1024)
        return bsc_fail(BSC_ERR_UNSUPPORTED, "bsc_softmax_rows: cols=%lld > 1024", (long long)cols);
    if (rows == 0) return BSC_OK;
    const int p = cols <= 1 ? 1 : cols <= 2 ? 2 : cols <= 4 ? 4 : cols <= 8 ? 8 : cols <= 16 ? 16 : cols <= 32 ? 32 : 64;
    const int64_t rows_per_block = cols <= 64 ? 4 * (64 / p) : 4;
    int64_t blocks = (rows + rows_per_block - 1) / rows_per_block;
    if (blocks > 16 * (int64_t)ctx->cu_count) blocks = 16 * (int64_t)ctx->cu_count;
#define BSC_SM(P) hipLaunchKernelGGL(softmax_rows_small_kernel<P>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, \
                                     in, rows, (int)cols, ld_in, out, ld_out, lse)
    const bool vec4 = cols % 4 == 0 && cols >= 8 && cols <= 256 && ld_in % 4 == 0 && ld_out % 4 == 0 &&
                      (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    if (vec4) {
        const int l = cols <= 8 ? 2 : cols <= 16 ? 4 : cols <= 32 ? 8 : cols <= 64 ? 16 : cols <= 128 ? 32 : 64;
        int64_t vb = (rows + 16 * (64 / l) - 1) / (16 * (64 / l));
        if (vb > 16 * (int64_t)ctx->cu_count) vb = 16 * (int64_t)ctx->cu_count;
#define BSC_SM4(L_) hipLaunchKernelGGL(softmax_rows_vec4_kernel<L_>, dim3((unsigned)vb), dim3(256), 0, ctx->stream, \
                                       in, rows, (int)cols, ld_in, out, ld_out, lse)
        if (l == 2) BSC_SM4(2); else if (l == 4) BSC_SM4(4); else if (l == 8) BSC_SM4(8); else if (l == 16) BSC_SM4(16);
        else if (l == 32) BSC_SM4(32); else BSC_SM4(64);
#undef BSC_SM4
        BSC_LAUNCH_CHECK();
        return BSC_OK;
    }
    if (cols > 64)
        hipLaunchKernelGGL(softmax_rows_wide_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, in, rows,
                           (int)cols, ld_in, out, ld_out, lse);
    else if (p == 1) BSC_SM(1); else if (p == 2) BSC_SM(2); else if (p == 4) BSC_SM(4); else if (p == 8) BSC_SM(8);
    else if (p == 16) BSC_SM(16); else if (p == 32) BSC_SM(32); else BSC_SM(64);
#undef BSC_SM
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_natgrad_update_f32(bsc_ctx* ctx, float* eta, float eta0, const float* message, int64_t n,
                           float scale, float rho) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eta && message && n > 0, "bsc_natgrad_update_f32: bad arguments");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8 * (int64_t)ctx->cu_count) blocks = 8 * (int64_t)ctx->cu_count;
    hipLaunchKernelGGL(natgrad_update_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       eta, eta0, message, n, scale, rho, (const double*)nullptr, (const double*)nullptr,
                       (const double*)nullptr, 0.0, (double*)nullptr);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_natgrad_update_f32_elbo(bsc_ctx* ctx, float* eta, float eta0, const float* message, int64_t n,
                                float scale, float rho, const double* ll, const double* local_bound,
                                const double* global_bound, double* elbo) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eta && message && n > 0 && ll && local_bound && global_bound && elbo,
                "bsc_natgrad_update_f32_elbo: bad arguments");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8 * (int64_t)ctx->cu_count) blocks = 8 * (int64_t)ctx->cu_count;
    hipLaunchKernelGGL(natgrad_update_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       eta, eta0, message, n, scale, rho, ll, local_bound, global_bound, (double)scale, elbo);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_natgrad_update_f32_2d(bsc_ctx* ctx, float* eta, int64_t ld_eta, float eta0, const float* message,
                              int64_t ld_msg, int64_t rows, int64_t cols, float scale, float rho, const double* ll,
                              int32_t n_ll, const double* local_bound, const double* global_bound, double* elbo) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eta && message && rows > 0 && cols > 0 && ld_eta >= cols && ld_msg >= cols,
                "bsc_natgrad_update_f32_2d: bad arguments");
    BSC_REQUIRE(!elbo || (ll && n_ll >= 1 && local_bound && global_bound),
                "bsc_natgrad_update_f32_2d: the bound needs ll[n_ll], local_bound and global_bound");
    int64_t bx = (cols + 255) / 256;
    if (bx > 16) bx = 16;
    const int64_t by = rows < 65535 ? rows : 65535;
    hipLaunchKernelGGL(natgrad_update_f32_2d_kernel, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, ctx->stream, eta, ld_eta, eta0,
                       message, ld_msg, rows, cols, scale, rho, ll, (int)n_ll, local_bound, global_bound, (double)scale,
                       elbo);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_dirichlet_expectation_bound(bsc_ctx* ctx, const float* lam, int64_t rows, int64_t cols, int64_t ld,
                                    double prior, float* out, double* bound) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(lam && out && bound && rows >= 1 && cols >= 1 && ld >= cols && rows <= 65535 * 32 && prior > 0.0,
                "bsc_dirichlet_expectation_bound: bad arguments");
    int64_t blocks = (rows * cols + 255) / 256;
    if (blocks > 8 * (int64_t)ctx->cu_count) blocks = 8 * (int64_t)ctx->cu_count;
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)(2 * rows + blocks) * sizeof(double), &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    double* row_psi = (double*)ws;
    double* row_const = row_psi + rows;
    double* partial = row_const + rows;
    hipLaunchKernelGGL(row_sum_bound_kernel, dim3((unsigned)rows), dim3(1024), 0, ctx->stream, lam, cols, ld,
                       prior, row_psi, row_const);
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(dirichlet_expect_bound_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, lam,
                       rows, cols, ld, prior, (const double*)row_psi, out, partial);
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(ordered_sum2_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double*)partial, blocks,
                       (const double*)row_const, rows, 1.0, bound);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_suffstats_normal(bsc_ctx* ctx, const float* x, int64_t n, double* stats) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(stats && (x || n == 0) && n >= 0, "bsc_suffstats_normal: bad arguments");
    BSC_REQUIRE(((uintptr_t)x & 3) == 0, "bsc_suffstats_normal: x must be 4-byte aligned");
    int64_t want = (n / 4 + STAT_BLOCK - 1) / STAT_BLOCK;
    int n_blocks = (int)(want < 1 ? 1 : (want > 8 * ctx->cu_count ? 8 * ctx->cu_count : want));
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)n_blocks * 2 * sizeof(double), &ws);
    if (rc != BSC_OK) return rc;
    hipLaunchKernelGGL(normal_stats_partial_kernel, dim3(n_blocks), dim3(STAT_BLOCK), 0,
                       ctx->stream, x, n, (double*)ws);
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(normal_stats_final_kernel, dim3(1), dim3(BSC_WAVE), 0, ctx->stream,
                       (const double*)ws, n_blocks, n, stats);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_adam_ascent(bsc_ctx* ctx, double* lam, const double* grad, double* m1, double* m2,
                    int64_t n, int64_t t, double lr, double beta1, double beta2, double eps) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(lam && grad && m1 && m2 && n > 0 && t >= 1, "bsc_adam_ascent: bad arguments");
    const double corr1 = 1.0 - pow(beta1, (double)t);
    const double corr2 = 1.0 - pow(beta2, (double)t);
    hipLaunchKernelGGL(adam_ascent_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, lam, grad, m1, m2, n, lr, beta1, beta2, eps, corr1, corr2);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_natgrad_update(bsc_ctx* ctx, double* eta, const double* eta0, const double* message,
                       int64_t n, double scale, double rho) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eta && eta0 && message && n > 0, "bsc_natgrad_update: bad arguments");
    hipLaunchKernelGGL(natgrad_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, eta, eta0, message, n, scale, rho);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

}  // extern "C"
