/* Plain-C restatement of the data-sized statistics of the Bayesic SVI hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): an independent second oracle beside the
 * numpy one, fast enough (OpenMP) to check the HIP kernels at BASELINE's full sizes and to
 * serve as a compiled CPU baseline.  float32 operands, float64 arithmetic, fixed reduction
 * order (static row blocks, partials combined in thread order).
 *
 * PARITY UNPINNED, like oracle/svi.py: the reference has no code for this path
 * (README.md:24-80 is prose); every function restates the formula cited at its head and is
 * itself checked against the numpy oracle in tests/test_oracle_c.py.
 *
 *   gcc -O2 -fopenmp -shared -fPIC oracle/c/oracle_kernels.c -o oracle/_build/liboracle.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_max_threads(void) { return 1; }
static int omp_get_thread_num(void) { return 0; }
#endif

static void row_block(long n, int t, int nt, long* lo, long* hi) {
    const long per = (n + nt - 1) / nt;
    *lo = (long)t * per;
    *hi = *lo + per < n ? *lo + per : n;
    if (*lo > n) *lo = n;
}

/* Config 2 (README.md:51, reparameterised ELBO of Bayesian linear regression):
 *   r_ns = y_n - x_n . w_s,   Q[s] = sum_n r_ns^2,   G[s,d] = sum_n r_ns x_nd. */
void oracle_blr_data_pass(const float* X, long ldx, const float* y, long B, int D, const float* W,
                          int S, double* Q, double* G) {
    const int nt = omp_get_max_threads();
    double* part = (double*)calloc((size_t)nt * (S + (size_t)S * D), sizeof(double));
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        double* q = part + (size_t)t * (S + (size_t)S * D);
        double* g = q + S;
        long lo, hi;
        row_block(B, t, nt, &lo, &hi);
        for (long n = lo; n < hi; ++n) {
            const float* x = X + n * ldx;
            for (int s = 0; s < S; ++s) {
                const float* w = W + (long)s * D;
                double dot = 0.0;
                for (int d = 0; d < D; ++d) dot += (double)x[d] * (double)w[d];
                const double r = (double)y[n] - dot;
                q[s] += r * r;
                double* gs = g + (long)s * D;
                for (int d = 0; d < D; ++d) gs[d] += r * (double)x[d];
            }
        }
    }
    memset(Q, 0, sizeof(double) * S);
    memset(G, 0, sizeof(double) * (size_t)S * D);
    for (int t = 0; t < nt; ++t) {
        const double* q = part + (size_t)t * (S + (size_t)S * D);
        for (int s = 0; s < S; ++s) Q[s] += q[s];
        for (long i = 0; i < (long)S * D; ++i) G[i] += q[S + i];
    }
    free(part);
}

/* The same statistics as ONE fused float32 pass, the way a host implementation that wanted to be fast would write
 * them (bench.py's cpu_baseline.fused_f32: the honest host figure beside the numpy / BLAS leg and the float64 scalar
 * leg above).  Every thread streams its row block once; per row the S dot products and the S rank-1 updates run on
 * eight-float vectors (AVX2 + FMA: both this container's Xeon and the GPU box's EPYC run it), eight draws' dot products
 * as eight independent chains, four rows per accumulator update, against accumulators of S x D floats that stay in
 * the core's L1 (8 KiB at S = 8, D = 256) and are flushed into float64 totals every 256 rows, which keeps the float32
 * sums as short as the device's per-wave ones.  Same formula, README.md:51; S <= 16, D <= 1024. */
#define BLR_F32_MAX_S 16
#define BLR_F32_MAX_D 1024
typedef float blr_v8 __attribute__((vector_size(32), aligned(4)));   /* eight floats, unaligned loads allowed */

/* Rows [n0, n1) into acc[S][D] (float32) and qa[S]; NS = S as a compile-time constant keeps the NS dot-product
 * chains in registers.  Four rows at a time: one load and one store of an accumulator vector per four updates. */
#define BLR_F32_BLOCK(NS)                                                                                   \
    static inline __attribute__((always_inline, optimize("O3"), target("avx2,fma")))                          \
    void blr_f32_block_##NS(const float* X, long ldx, const float* y, long n0, long n1, int D, const float* W, \
                            float* acc, float* qa) {                                                          \
        const int D8 = D & ~7;                                                                                \
        long n = n0;                                                                                          \
        for (; n + 4 <= n1; n += 4) {                                                                         \
            float r[4][NS];                                                                                   \
            for (int k = 0; k < 4; ++k) {                                                                     \
                const float* x = X + (n + k) * ldx;                                                           \
                blr_v8 dv[NS];                                                                                \
                for (int s = 0; s < NS; ++s) dv[s] = (blr_v8){0, 0, 0, 0, 0, 0, 0, 0};                        \
                for (int d = 0; d < D8; d += 8) {                                                             \
                    const blr_v8 xv = *(const blr_v8*)(x + d);                                                \
                    for (int s = 0; s < NS; ++s) dv[s] += xv * *(const blr_v8*)(W + (long)s * D + d);         \
                }                                                                                             \
                for (int s = 0; s < NS; ++s) {                                                                \
                    float dot = ((dv[s][0] + dv[s][4]) + (dv[s][1] + dv[s][5])) +                             \
                                ((dv[s][2] + dv[s][6]) + (dv[s][3] + dv[s][7]));                              \
                    for (int d = D8; d < D; ++d) dot += x[d] * W[(long)s * D + d];                            \
                    r[k][s] = y[n + k] - dot;                                                                 \
                    qa[s] += r[k][s] * r[k][s];                                                               \
                }                                                                                             \
            }                                                                                                 \
            const float *x0 = X + n * ldx, *x1 = x0 + ldx, *x2 = x1 + ldx, *x3 = x2 + ldx;                    \
            for (int d = 0; d < D8; d += 8) {                                                                 \
                const blr_v8 a = *(const blr_v8*)(x0 + d), b = *(const blr_v8*)(x1 + d);                      \
                const blr_v8 c = *(const blr_v8*)(x2 + d), e = *(const blr_v8*)(x3 + d);                      \
                for (int s = 0; s < NS; ++s) {                                                                \
                    blr_v8* as = (blr_v8*)(acc + (long)s * D + d);                                            \
                    *as += (r[0][s] * a + r[1][s] * b) + (r[2][s] * c + r[3][s] * e);                         \
                }                                                                                             \
            }                                                                                                 \
            for (int d = D8; d < D; ++d)                                                                      \
                for (int s = 0; s < NS; ++s)                                                                  \
                    acc[(long)s * D + d] += (r[0][s] * x0[d] + r[1][s] * x1[d]) + (r[2][s] * x2[d] + r[3][s] * x3[d]); \
        }                                                                                                     \
        for (; n < n1; ++n) {                                                                                 \
            const float* x = X + n * ldx;                                                                     \
            for (int s = 0; s < NS; ++s) {                                                                    \
                float dot = 0.f;                                                                              \
                for (int d = 0; d < D; ++d) dot += x[d] * W[(long)s * D + d];                                 \
                const float rr = y[n] - dot;                                                                  \
                qa[s] += rr * rr;                                                                             \
                for (int d = 0; d < D; ++d) acc[(long)s * D + d] += rr * x[d];                                \
            }                                                                                                 \
        }                                                                                                     \
    }
BLR_F32_BLOCK(8)
BLR_F32_BLOCK(1)

__attribute__((optimize("O3"), target("avx2,fma")))
int oracle_blr_data_pass_f32(const float* X, long ldx, const float* y, long B, int D, const float* W,
                             int S, double* Q, double* G) {
    if (S < 1 || S > BLR_F32_MAX_S || D < 1 || D > BLR_F32_MAX_D) return 1;
    const int nt = omp_get_max_threads();
    double* part = (double*)calloc((size_t)nt * (S + (size_t)S * D), sizeof(double));
    if (!part) return 2;
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        double* q = part + (size_t)t * (S + (size_t)S * D);
        double* g = q + S;
        float* acc = (float*)aligned_alloc(64, sizeof(float) * (size_t)BLR_F32_MAX_S * BLR_F32_MAX_D);
        float qa[BLR_F32_MAX_S];
        long lo, hi;
        row_block(B, t, nt, &lo, &hi);
        for (long n0 = lo; n0 < hi; n0 += 256) {
            const long n1 = n0 + 256 < hi ? n0 + 256 : hi;
            memset(acc, 0, sizeof(float) * (size_t)S * D);
            for (int s = 0; s < S; ++s) qa[s] = 0.f;
            int s0 = 0;
            for (; s0 + 8 <= S; s0 += 8) blr_f32_block_8(X, ldx, y, n0, n1, D, W + (long)s0 * D, acc + (long)s0 * D, qa + s0);
            for (; s0 < S; ++s0) blr_f32_block_1(X, ldx, y, n0, n1, D, W + (long)s0 * D, acc + (long)s0 * D, qa + s0);
            for (int s = 0; s < S; ++s) q[s] += (double)qa[s];
            for (long i = 0; i < (long)S * D; ++i) g[i] += (double)acc[i];
        }
        free(acc);
    }
    memset(Q, 0, sizeof(double) * S);
    memset(G, 0, sizeof(double) * (size_t)S * D);
    for (int t = 0; t < nt; ++t) {
        const double* q = part + (size_t)t * (S + (size_t)S * D);
        for (int s = 0; s < S; ++s) Q[s] += q[s];
        for (long i = 0; i < (long)S * D; ++i) G[i] += q[S + i];
    }
    free(part);
    return 0;
}

/* Copies `rows` rows of `row_bytes` each with the row blocks (and threads) the passes above use.  Into a freshly mapped
 * destination this is a parallel FIRST TOUCH: every page lands on the NUMA node of the thread that will stream it, which
 * is how a host implementation that cared would lay a mini-batch out (bench.py's fused_f32 leg times the pass on such a
 * copy; the numpy legs read the array as numpy allocated it). */
void oracle_parallel_copy(void* dst, const void* src, long rows, long row_bytes) {
    const int nt = omp_get_max_threads();
#pragma omp parallel num_threads(nt)
    {
        long lo, hi;
        row_block(rows, omp_get_thread_num(), nt, &lo, &hi);
        if (hi > lo) memcpy((char*)dst + lo * row_bytes, (const char*)src + lo * row_bytes, (size_t)(hi - lo) * row_bytes);
    }
}

/* Config 5 (README.md:52, BBVI): ell[s] = sum_n [ y_n l_ns - softplus(l_ns) ],
 *   l_ns = x_n . Wz[s] + Bz[g_n, s]   (Bz is [G, S]). */
void oracle_logreg_loglik(const float* X, long ldx, const float* y, const int* g, long N, int D,
                          int G, const float* Wz, const float* Bz, int S, double* ell) {
    const int nt = omp_get_max_threads();
    double* part = (double*)calloc((size_t)nt * S, sizeof(double));
    (void)G;
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        double* e = part + (size_t)t * S;
        long lo, hi;
        row_block(N, t, nt, &lo, &hi);
        for (long n = lo; n < hi; ++n) {
            const float* x = X + n * ldx;
            const float* b = Bz + (long)g[n] * S;
            for (int s = 0; s < S; ++s) {
                const float* w = Wz + (long)s * D;
                double l = (double)b[s];
                for (int d = 0; d < D; ++d) l += (double)x[d] * (double)w[d];
                const double sp = (l > 0.0 ? l : 0.0) + log1p(exp(-fabs(l)));
                e[s] += (double)y[n] * l - sp;
            }
        }
    }
    for (int s = 0; s < S; ++s) {
        ell[s] = 0.0;
        for (int t = 0; t < nt; ++t) ell[s] += part[(size_t)t * S + s];
    }
    free(part);
}

/* Config 3 (README.md:43,72: discrete latent marginalised by summation):
 *   logit_nk = c_k + sum_d (Wmat[k,d] x_nd + Wmat[k,D+d] x_nd^2),  r_nk = softmax_k,
 *   stats[k] = (sum_n r_nk, sum_n r_nk x_n, sum_n r_nk x_n^2),  lse = sum_n logsumexp_k. */
void oracle_mog_estep(const float* X, long ldx, long N, int D, int K, const float* Wmat,
                      const float* c, double* stats, double* lse) {
    const int nt = omp_get_max_threads();
    const size_t per = (size_t)K * (1 + 2 * D) + 1;
    double* part = (double*)calloc((size_t)nt * per, sizeof(double));
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        double* st = part + (size_t)t * per;
        double* logit = (double*)malloc(sizeof(double) * K);
        long lo, hi;
        row_block(N, t, nt, &lo, &hi);
        for (long n = lo; n < hi; ++n) {
            const float* x = X + n * ldx;
            double m = -1e300;
            for (int k = 0; k < K; ++k) {
                const float* w = Wmat + (long)k * 2 * D;
                double l = (double)c[k];
                for (int d = 0; d < D; ++d) {
                    const double xd = (double)x[d];
                    l += (double)w[d] * xd + (double)w[D + d] * xd * xd;
                }
                logit[k] = l;
                if (l > m) m = l;
            }
            double z = 0.0;
            for (int k = 0; k < K; ++k) { logit[k] = exp(logit[k] - m); z += logit[k]; }
            st[per - 1] += m + log(z);
            for (int k = 0; k < K; ++k) {
                const double r = logit[k] / z;
                double* sk = st + (size_t)k * (1 + 2 * D);
                sk[0] += r;
                for (int d = 0; d < D; ++d) {
                    const double xd = (double)x[d];
                    sk[1 + d] += r * xd;
                    sk[1 + D + d] += r * xd * xd;
                }
            }
        }
        free(logit);
    }
    memset(stats, 0, sizeof(double) * (per - 1));
    *lse = 0.0;
    for (int t = 0; t < nt; ++t) {
        const double* st = part + (size_t)t * per;
        for (size_t i = 0; i + 1 < per; ++i) stats[i] += st[i];
        *lse += st[per - 1];
    }
    free(part);
}

/* Config 4 (README.md:36,75-77; fixed-gamma local step):
 *   sstats[k,v] = Bt[k,v] * sum_d Th[d,k] C[d,v] / (sum_k' Th[d,k'] Bt[k',v]).
 * Zero counts contribute nothing and are skipped.  Threads own column blocks, so every
 * output has one writer and the document order of each sum is fixed. */
void oracle_lda_sstats(const float* C, long ldc, long docs, long V, int K, const float* Th,
                       const float* Bt, double* out) {
    const int nt = omp_get_max_threads();
    memset(out, 0, sizeof(double) * (size_t)K * V);
#pragma omp parallel num_threads(nt)
    {
        long lo, hi;
        row_block(V, omp_get_thread_num(), nt, &lo, &hi);
        for (long d = 0; d < docs; ++d) {
            const float* th = Th + d * K;
            for (long v = lo; v < hi; ++v) {
                const double cnt = (double)C[d * ldc + v];
                if (cnt == 0.0) continue;
                double p = 0.0;
                for (int k = 0; k < K; ++k) p += (double)th[k] * (double)Bt[(long)k * V + v];
                const double ratio = cnt / p;
                for (int k = 0; k < K; ++k) out[(long)k * V + v] += (double)th[k] * ratio;
            }
        }
        for (int k = 0; k < K; ++k)
            for (long v = lo; v < hi; ++v) out[(long)k * V + v] *= (double)Bt[(long)k * V + v];
    }
}

/* The words' term of config 4's evidence lower bound (README.md:30-37; Hoffman, Blei, Bach 2010 eq. 7 with the
 * per-word assignments at their optimum): sum_dv C[d,v] log(sum_k Th[d,k] Bt[k,v]) -- oracle.svi.lda_local_bound
 * restated.  Threads own document blocks; their partials are added in thread order. */
double oracle_lda_local_bound(const float* C, long ldc, long docs, long V, int K, const float* Th,
                              const float* Bt) {
    const int nt = omp_get_max_threads();
    double* part = (double*)calloc((size_t)nt, sizeof(double));
#pragma omp parallel num_threads(nt)
    {
        long lo, hi;
        row_block(docs, omp_get_thread_num(), nt, &lo, &hi);
        double acc = 0.0;
        for (long d = lo; d < hi; ++d) {
            const float* th = Th + d * K;
            for (long v = 0; v < V; ++v) {
                const double cnt = (double)C[d * ldc + v];
                if (cnt == 0.0) continue;
                double p = 0.0;
                for (int k = 0; k < K; ++k) p += (double)th[k] * (double)Bt[(long)k * V + v];
                acc += cnt * log(p);
            }
        }
        part[omp_get_thread_num()] = acc;
    }
    double total = 0.0;
    for (int t = 0; t < nt; ++t) total += part[t];
    free(part);
    return total;
}

/* Full-covariance mixture statistic (t(x) = (x, x x^T), core.py:41-44, summed over rows):
 *   out[k,d,e] = sum_n R[n,k] X[n,d] Y[n,e].  Threads own row blocks; partials in thread order. */
void oracle_weighted_outer(const float* R, const float* X, const float* Y, long N, int K, int D,
                           int E, double* out) {
    const int nt = omp_get_max_threads();
    const size_t sz = (size_t)K * D * E;
    double* part = (double*)calloc((size_t)nt * sz, sizeof(double));
#pragma omp parallel num_threads(nt)
    {
        double* o = part + (size_t)omp_get_thread_num() * sz;
        long lo, hi;
        row_block(N, omp_get_thread_num(), nt, &lo, &hi);
        for (long n = lo; n < hi; ++n)
            for (int d = 0; d < D; ++d)
                for (int e = 0; e < E; ++e) {
                    const double z = (double)X[n * D + d] * (double)Y[n * E + e];
                    double* oo = o + ((size_t)d * E + e) * K;      /* [d][e][k]: k contiguous */
                    const float* r = R + n * K;
                    for (int k = 0; k < K; ++k) oo[k] += (double)r[k] * z;
                }
    }
    memset(out, 0, sizeof(double) * sz);
    for (int t = 0; t < nt; ++t)
        for (int d = 0; d < D; ++d)
            for (int e = 0; e < E; ++e)
                for (int k = 0; k < K; ++k)
                    out[((size_t)k * D + d) * E + e] += part[(size_t)t * sz + ((size_t)d * E + e) * K + k];
    free(part);
}

int oracle_threads(void) { return omp_get_max_threads(); }
/* Threads of the passes above from now on (bench.py: the host's CPU SHARE, not its CPU count -- a container that may
 * use 16 of 256 logical CPUs runs 128 spinning threads far slower than 16). */
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
