// Strided N-d kernels behind the algebra front end's executable IR:
// element-wise / n-ary add and mul (elemwise, add, _mul: bayesic/algebra.py:195-233,
// 1297-1309, 1435-1448), axis sums (_sum :1284-1294), conversion/materialisation
// of views (_dimshuffle :1312-1326 and _diagonal :1398-1414 are stride views and
// need no kernel of their own) and eye (:236-258).  All memory-bound.
#include "bsc_common.h"

namespace {

constexpr int MAXR = BSC_MAX_RANK;
constexpr int MAXIN = 8;

struct Dims {
    int rank;
    int64_t shape[MAXR];
};

struct ElemArgs {
    Dims d;
    int64_t total;
    int n_in;
    int64_t out_strides[MAXR];
    int64_t in_strides[MAXIN][MAXR];
    const void* in[MAXIN];
    void* out;
};

__device__ __forceinline__ void unravel(int64_t flat, const Dims& d, int64_t (&idx)[MAXR]) {
#pragma unroll
    for (int a = MAXR - 1; a >= 0; --a) {
        if (a < d.rank) {
            const int64_t s = d.shape[a];
            const int64_t q = flat / s;
            idx[a] = flat - q * s;
            flat = q;
        } else {
            idx[a] = 0;
        }
    }
}

__device__ __forceinline__ int64_t dot_strides(const int64_t (&idx)[MAXR], const int64_t* strides,
                                               int rank) {
    int64_t off = 0;
#pragma unroll
    for (int a = 0; a < MAXR; ++a)
        if (a < rank) off += idx[a] * strides[a];
    return off;
}

template <typename T>
__device__ __forceinline__ T op_unary(int op, T x) {
    switch (op) {
        case BSC_OP_LOG: return log(x);
        case BSC_OP_EXP: return exp(x);
        case BSC_OP_ABS: return fabs(x);
        default: return x;
    }
}

template <typename T, int OP>
__global__ __launch_bounds__(256) void elemwise_kernel(ElemArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t flat = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; flat < a.total;
         flat += stride) {
        int64_t idx[MAXR];
        unravel(flat, a.d, idx);
        T v = static_cast<const T*>(a.in[0])[dot_strides(idx, a.in_strides[0], a.d.rank)];
        if (OP == BSC_OP_ADD || OP == BSC_OP_MUL) {
            for (int k = 1; k < a.n_in; ++k) {
                const T u = static_cast<const T*>(a.in[k])[dot_strides(idx, a.in_strides[k], a.d.rank)];
                v = (OP == BSC_OP_ADD) ? v + u : v * u;
            }
        } else if (OP == BSC_OP_POW) {
            const T u = static_cast<const T*>(a.in[1])[dot_strides(idx, a.in_strides[1], a.d.rank)];
            v = pow(v, u);
        } else {
            v = op_unary<T>(OP, v);
        }
        static_cast<T*>(a.out)[dot_strides(idx, a.out_strides, a.d.rank)] = v;
    }
}

// Contiguous fast path for the common case: every operand dense in the same
// order (or a broadcast scalar), 16 bytes per lane.
template <int OP>
__global__ __launch_bounds__(256) void elemwise_dense_f32_kernel(int64_t n4, int n_in,
                                                                 const float* in0, const float* in1,
                                                                 const float* in2, const float* in3,
                                                                 int scalar_mask, float* out) {
    const float* ins[4] = {in0, in1, in2, in3};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v;
        if (scalar_mask & 1) { const float s = ins[0][0]; v = make_float4(s, s, s, s); }
        else v = reinterpret_cast<const float4*>(ins[0])[i];
        if (OP == BSC_OP_ADD || OP == BSC_OP_MUL) {
            for (int k = 1; k < n_in; ++k) {
                float4 u;
                if (scalar_mask & (1 << k)) { const float s = ins[k][0]; u = make_float4(s, s, s, s); }
                else u = reinterpret_cast<const float4*>(ins[k])[i];
                if (OP == BSC_OP_ADD) { v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
                else { v.x *= u.x; v.y *= u.y; v.z *= u.z; v.w *= u.w; }
            }
        } else if (OP == BSC_OP_POW) {
            float4 u;
            if (scalar_mask & 2) { const float s = ins[1][0]; u = make_float4(s, s, s, s); }
            else u = reinterpret_cast<const float4*>(ins[1])[i];
            v.x = powf(v.x, u.x); v.y = powf(v.y, u.y); v.z = powf(v.z, u.z); v.w = powf(v.w, u.w);
        } else {
            v.x = op_unary<float>(OP, v.x); v.y = op_unary<float>(OP, v.y);
            v.z = op_unary<float>(OP, v.z); v.w = op_unary<float>(OP, v.w);
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

template <typename S, typename T>
__global__ __launch_bounds__(256) void convert_kernel(Dims d, int64_t total, const S* src,
                                                      ElemArgs strides, T* dst) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t flat = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; flat < total;
         flat += stride) {
        int64_t idx[MAXR];
        unravel(flat, d, idx);
        dst[dot_strides(idx, strides.out_strides, d.rank)] =
            (T)src[dot_strides(idx, strides.in_strides[0], d.rank)];
    }
}

template <typename T>
__global__ void eye_kernel(T* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * n) out[i] = (i / n == i % n) ? (T)1 : (T)0;
}

// log det of a symmetric positive-definite matrix by an in-place float64 Cholesky
// in the workspace, one workgroup per matrix (parameter-side: D x D, once per
// update -- not a data-sized operation).  Non-SPD input yields NaN.
template <typename T>
__global__ __launch_bounds__(1024) void logdet_spd_kernel(const T* __restrict__ A, int64_t n,
                                                          int64_t s_b, int64_t s_r, int64_t s_c,
                                                          double* __restrict__ work,
                                                          T* __restrict__ out) {
    __shared__ double pivot;
    __shared__ double red[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t b = blockIdx.x;
    double* L = work + b * n * n;
    const T* src = A + b * s_b;
    for (int64_t i = tid; i < n * n; i += nt) L[i] = (double)src[(i / n) * s_r + (i % n) * s_c];
    __syncthreads();
    double logsum = 0.0;
    for (int64_t j = 0; j < n; ++j) {
        if (tid == 0) pivot = sqrt(L[j * n + j]);
        __syncthreads();
        const double d = pivot;
        if (tid == 0) logsum += log(d);
        for (int64_t i = j + 1 + tid; i < n; i += nt) L[i * n + j] /= d;
        __syncthreads();
        const int64_t m = n - j - 1;   // trailing block is m x m, lower triangle only
        for (int64_t t = tid; t < m * m; t += nt) {
            const int64_t i = j + 1 + t / m, k = j + 1 + t % m;
            if (k <= i) L[i * n + k] -= L[i * n + j] * L[k * n + j];
        }
        __syncthreads();
    }
    if (tid == 0) out[b] = (T)(2.0 * logsum);
    (void)red;
}

// Inverse (and log-determinant) of symmetric positive definite matrices, one workgroup each, float64 inside:
// Cholesky A = L L^T in `work`, then L^-1 by forward substitution (column j of the inverse by thread j),
// then A^-1 = L^-T L^-1.  Parameter-sized: what a resident multivariate-normal or Wishart factor needs to turn
// its natural parameters (precision, V^-1) into expectations without leaving the device (inference/vmp.py).
template <typename T>
__global__ __launch_bounds__(1024) void inverse_spd_kernel(const T* __restrict__ A, int64_t n, int64_t s_b, int64_t s_r,
                                                           int64_t s_c, double* __restrict__ work, T* __restrict__ out,
                                                           T* __restrict__ logdet) {
    __shared__ double pivot;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t b = blockIdx.x;
    double* L = work + b * 2 * n * n;
    double* Li = L + n * n;
    const T* src = A + b * s_b;
    for (int64_t i = tid; i < n * n; i += nt) {
        const int64_t r = i / n, c = i % n;
        // (a message is symmetric up to rounding: take the mean of the two triangles)
        L[i] = 0.5 * ((double)src[r * s_r + c * s_c] + (double)src[c * s_r + r * s_c]);
    }
    __syncthreads();
    double logsum = 0.0;
    for (int64_t j = 0; j < n; ++j) {
        if (tid == 0) pivot = sqrt(L[j * n + j]);
        __syncthreads();
        const double d = pivot;
        if (tid == 0) {
            logsum += log(d);
            L[j * n + j] = d;
        }
        for (int64_t i = j + 1 + tid; i < n; i += nt) L[i * n + j] /= d;
        __syncthreads();
        const int64_t m = n - j - 1;
        for (int64_t t = tid; t < m * m; t += nt) {
            const int64_t i = j + 1 + t / m, k = j + 1 + t % m;
            if (k <= i) L[i * n + k] -= L[i * n + j] * L[k * n + j];
        }
        __syncthreads();
    }
    // column c of L^-1 (lower triangular): x_c = 1 / L_cc, x_i = -(sum_{k=c..i-1} L_ik x_k) / L_ii
    for (int64_t c = tid; c < n; c += nt) {
        for (int64_t i = 0; i < c; ++i) Li[i * n + c] = 0.0;
        Li[c * n + c] = 1.0 / L[c * n + c];
        for (int64_t i = c + 1; i < n; ++i) {
            double acc = 0.0;
            for (int64_t k = c; k < i; ++k) acc += L[i * n + k] * Li[k * n + c];
            Li[i * n + c] = -acc / L[i * n + i];
        }
    }
    __syncthreads();
    T* dst = out + b * n * n;
    for (int64_t t = tid; t < n * n; t += nt) {
        const int64_t i = t / n, j = t % n;
        double acc = 0.0;
        for (int64_t k = (i > j ? i : j); k < n; ++k) acc += Li[k * n + i] * Li[k * n + j];
        dst[t] = (T)acc;
    }
    if (tid == 0 && logdet) logdet[b] = (T)(2.0 * logsum);
}

int fill_dims(Dims& d, int rank, const int64_t* shape, int64_t* total, const char* who) {
    BSC_REQUIRE(rank >= 0 && rank <= MAXR, "%s: rank %d exceeds %d", who, rank, MAXR);
    d.rank = rank;
    int64_t t = 1;
    for (int a = 0; a < MAXR; ++a) {
        d.shape[a] = a < rank ? shape[a] : 1;
        BSC_REQUIRE(d.shape[a] >= 0, "%s: negative extent", who);
        t *= d.shape[a];
    }
    *total = t;
    return BSC_OK;
}

int grid_for(int64_t work_items, int cu_count) {
    int64_t blocks = (work_items + 255) / 256;
    const int64_t cap = (int64_t)cu_count * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

template <typename T>
void launch_elemwise(bsc_ctx* ctx, int op, const ElemArgs& a, int grid) {
#define BSC_CASE(OPV)                                                                      \
    case OPV:                                                                              \
        hipLaunchKernelGGL((elemwise_kernel<T, OPV>), dim3(grid), dim3(256), 0, ctx->stream, a); \
        break;
    switch (op) {
        BSC_CASE(BSC_OP_ADD)
        BSC_CASE(BSC_OP_MUL)
        BSC_CASE(BSC_OP_LOG)
        BSC_CASE(BSC_OP_EXP)
        BSC_CASE(BSC_OP_POW)
        BSC_CASE(BSC_OP_ABS)
        BSC_CASE(BSC_OP_COPY)
    }
#undef BSC_CASE
}

}  // namespace

extern "C" {

int bsc_elemwise(bsc_ctx* ctx, int op, int dtype, int rank, const int64_t* host_shape, void* out,
                 const int64_t* host_out_strides, int n_in, const void* const* host_in,
                 const int64_t* host_in_strides) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(op >= BSC_OP_ADD && op <= BSC_OP_COPY, "bsc_elemwise: unknown op %d", op);
    BSC_REQUIRE(dtype == BSC_F32 || dtype == BSC_F64, "bsc_elemwise: unknown dtype %d", dtype);
    BSC_REQUIRE(out && host_in && n_in >= 1 && n_in <= MAXIN, "bsc_elemwise: bad operands");
    const int arity = (op == BSC_OP_ADD || op == BSC_OP_MUL) ? n_in : (op == BSC_OP_POW ? 2 : 1);
    BSC_REQUIRE(arity == n_in, "bsc_elemwise: op %d takes %d inputs, got %d", op, arity, n_in);
    ElemArgs a{};
    int rc = fill_dims(a.d, rank, host_shape, &a.total, "bsc_elemwise");
    if (rc != BSC_OK) return rc;
    if (a.total == 0) return BSC_OK;
    a.n_in = n_in;
    a.out = out;
    bool dense = dtype == BSC_F32 && n_in <= 4 && (a.total % 4) == 0 && (((uintptr_t)out) & 15) == 0;
    int scalar_mask = 0;
    int64_t expect = 1;
    for (int ax = MAXR - 1; ax >= 0; --ax) {
        a.out_strides[ax] = ax < rank ? host_out_strides[ax] : 0;
        if (ax < rank && a.d.shape[ax] != 1) {
            if (a.out_strides[ax] != expect) dense = false;
            expect *= a.d.shape[ax];
        }
    }
    for (int k = 0; k < n_in; ++k) {
        BSC_REQUIRE(host_in[k] != nullptr, "bsc_elemwise: input %d is null", k);
        a.in[k] = host_in[k];
        bool all_zero = true, same = true;
        for (int ax = 0; ax < MAXR; ++ax) {
            a.in_strides[k][ax] = ax < rank ? host_in_strides[k * rank + ax] : 0;
            if (ax < rank && a.d.shape[ax] != 1) {
                if (a.in_strides[k][ax] != 0) all_zero = false;
                if (a.in_strides[k][ax] != a.out_strides[ax]) same = false;
            }
        }
        if (all_zero) scalar_mask |= 1 << k;
        else if (!same || (((uintptr_t)host_in[k]) & 15) != 0) dense = false;
    }
    if (dense && scalar_mask == (1 << n_in) - 1) dense = false;
    if (dense) {
        const int64_t n4 = a.total / 4;
        const int grid = grid_for(n4, ctx->cu_count);
        const float* p[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < n_in; ++k) p[k] = (const float*)host_in[k];
#define BSC_DENSE(OPV)                                                                        \
    case OPV:                                                                                 \
        hipLaunchKernelGGL((elemwise_dense_f32_kernel<OPV>), dim3(grid), dim3(256), 0,        \
                           ctx->stream, n4, n_in, p[0], p[1], p[2], p[3], scalar_mask, (float*)out); \
        break;
        switch (op) {
            BSC_DENSE(BSC_OP_ADD)
            BSC_DENSE(BSC_OP_MUL)
            BSC_DENSE(BSC_OP_LOG)
            BSC_DENSE(BSC_OP_EXP)
            BSC_DENSE(BSC_OP_POW)
            BSC_DENSE(BSC_OP_ABS)
            BSC_DENSE(BSC_OP_COPY)
        }
#undef BSC_DENSE
    } else {
        const int grid = grid_for(a.total, ctx->cu_count);
        if (dtype == BSC_F32) launch_elemwise<float>(ctx, op, a, grid);
        else launch_elemwise<double>(ctx, op, a, grid);
    }
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_convert(bsc_ctx* ctx, int src_dtype, int dst_dtype, int rank, const int64_t* host_shape,
                const void* src, const int64_t* host_src_strides, void* dst,
                const int64_t* host_dst_strides) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE((src_dtype == BSC_F32 || src_dtype == BSC_F64) &&
                    (dst_dtype == BSC_F32 || dst_dtype == BSC_F64),
                "bsc_convert: unknown dtype");
    BSC_REQUIRE(src && dst, "bsc_convert: null pointer");
    ElemArgs a{};
    int64_t total;
    int rc = fill_dims(a.d, rank, host_shape, &total, "bsc_convert");
    if (rc != BSC_OK) return rc;
    if (total == 0) return BSC_OK;
    for (int ax = 0; ax < MAXR; ++ax) {
        a.in_strides[0][ax] = ax < rank ? host_src_strides[ax] : 0;
        a.out_strides[ax] = ax < rank ? host_dst_strides[ax] : 0;
    }
    const int grid = grid_for(total, ctx->cu_count);
    if (src_dtype == BSC_F32 && dst_dtype == BSC_F32)
        hipLaunchKernelGGL((convert_kernel<float, float>), dim3(grid), dim3(256), 0, ctx->stream,
                           a.d, total, (const float*)src, a, (float*)dst);
    else if (src_dtype == BSC_F32)
        hipLaunchKernelGGL((convert_kernel<float, double>), dim3(grid), dim3(256), 0, ctx->stream,
                           a.d, total, (const float*)src, a, (double*)dst);
    else if (dst_dtype == BSC_F32)
        hipLaunchKernelGGL((convert_kernel<double, float>), dim3(grid), dim3(256), 0, ctx->stream,
                           a.d, total, (const double*)src, a, (float*)dst);
    else
        hipLaunchKernelGGL((convert_kernel<double, double>), dim3(grid), dim3(256), 0, ctx->stream,
                           a.d, total, (const double*)src, a, (double*)dst);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_sum(bsc_ctx* ctx, int dtype, int rank_keep, const int64_t* host_keep_shape,
            const int64_t* host_in_keep_strides, int rank_red, const int64_t* host_red_shape,
            const int64_t* host_in_red_strides, const void* in, void* out) {
    // the one-operand case of the fused map-reduce (csrc/bsc_fused.hip)
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(rank_keep >= 0 && rank_keep <= MAXR && rank_red >= 0 && rank_red <= MAXR,
                "bsc_sum: rank exceeds %d", MAXR);
    BSC_REQUIRE(out != nullptr, "bsc_sum: out is null");
    int64_t out_strides[MAXR + 1], n_out = 1, n_red = 1;
    for (int a = rank_keep - 1; a >= 0; --a) {
        out_strides[a] = n_out;
        n_out *= host_keep_shape[a];
    }
    for (int a = 0; a < rank_red; ++a) n_red *= host_red_shape[a];
    BSC_REQUIRE(in != nullptr || n_out * n_red == 0, "bsc_sum: in is null");
    const void* ins[1] = {in};
    const int32_t pre_op[1] = {BSC_OP_COPY};
    const double pre_arg[1] = {0.0};
    return bsc_map_reduce(ctx, dtype, BSC_OP_ADD, rank_keep, host_keep_shape, rank_red,
                          host_red_shape, 1, ins, host_in_keep_strides, host_in_red_strides, pre_op,
                          pre_arg, 1.0, 0.0, BSC_OP_COPY, 0.0, out, out_strides);
}

int bsc_logdet_spd(bsc_ctx* ctx, int dtype, int64_t batch, int64_t n, const void* A, int64_t s_b,
                   int64_t s_r, int64_t s_c, void* out) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(dtype == BSC_F32 || dtype == BSC_F64, "bsc_logdet_spd: unknown dtype %d", dtype);
    BSC_REQUIRE(batch >= 0 && n >= 0 && out, "bsc_logdet_spd: bad arguments");
    if (batch == 0) return BSC_OK;
    BSC_REQUIRE(A || n == 0, "bsc_logdet_spd: A is null");
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)batch * n * n * sizeof(double) + 8, &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    if (dtype == BSC_F32)
        hipLaunchKernelGGL(logdet_spd_kernel<float>, dim3((unsigned)batch), dim3(1024), 0,
                           ctx->stream, (const float*)A, n, s_b, s_r, s_c, (double*)ws, (float*)out);
    else
        hipLaunchKernelGGL(logdet_spd_kernel<double>, dim3((unsigned)batch), dim3(1024), 0,
                           ctx->stream, (const double*)A, n, s_b, s_r, s_c, (double*)ws, (double*)out);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_inverse_spd(bsc_ctx* ctx, int dtype, int64_t batch, int64_t n, const void* A, int64_t s_b, int64_t s_r,
                    int64_t s_c, void* out, void* logdet) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(dtype == BSC_F32 || dtype == BSC_F64, "bsc_inverse_spd: unknown dtype %d", dtype);
    BSC_REQUIRE(batch >= 0 && n >= 0 && (out || batch * n == 0), "bsc_inverse_spd: bad arguments");
    if (batch == 0 || n == 0) return BSC_OK;
    BSC_REQUIRE(A, "bsc_inverse_spd: A is null");
    void* ws = nullptr;
    int rc = bsc_workspace(ctx, (size_t)batch * 2 * n * n * sizeof(double) + 8, &ws);
    if (rc != BSC_OK) return rc;
    ctx->slab_rows = 0;
    if (dtype == BSC_F32)
        hipLaunchKernelGGL(inverse_spd_kernel<float>, dim3((unsigned)batch), dim3(1024), 0, ctx->stream, (const float*)A, n,
                           s_b, s_r, s_c, (double*)ws, (float*)out, (float*)logdet);
    else
        hipLaunchKernelGGL(inverse_spd_kernel<double>, dim3((unsigned)batch), dim3(1024), 0, ctx->stream, (const double*)A,
                           n, s_b, s_r, s_c, (double*)ws, (double*)out, (double*)logdet);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_eye(bsc_ctx* ctx, int dtype, void* out, int64_t n) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(out && n >= 0, "bsc_eye: bad arguments");
    BSC_REQUIRE(dtype == BSC_F32 || dtype == BSC_F64, "bsc_eye: unknown dtype %d", dtype);
    if (n == 0) return BSC_OK;
    const unsigned blocks = (unsigned)((n * n + 255) / 256);
    if (dtype == BSC_F32)
        hipLaunchKernelGGL(eye_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, (float*)out, n);
    else
        hipLaunchKernelGGL(eye_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, (double*)out, n);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

}  // extern "C"
