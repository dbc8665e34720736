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
