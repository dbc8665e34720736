// Fused pointwise map with an optional trailing reduction: the executor's fusion
// of element-wise chains (elemwise / add / _mul: bayesic/algebra.py:195-233,
// 1297-1309, 1435-1448) with a following _sum (:1284-1294) into ONE pass over the
// operands.  The reference leaves this to Theano's graph
// optimiser; here the host plan folds unary ops into the consumer's operand list:
//
//   v(keep, red) = post( scale * COMBINE_i pre_i( in_i[keep, red] ) + shift )
//   out[keep]    = sum_red v(keep, red)        (rank_red == 0: out[keep] = v(keep))
//
// All HBM-bound: every operand element is read once, nothing intermediate is
// written.  Reductions accumulate in float64 in a fixed order (lane-strided
// partial -> butterfly -> split partials summed in split order).
#include "bsc_common.h"

namespace {

constexpr int MAXR = BSC_MAX_RANK;
constexpr int MAXIN = 8;

struct Dims {
    int rank;
    int64_t shape[MAXR];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// streaming 16-byte load: the operands of a fused map are used once
__device__ __forceinline__ float4 load4_nt(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

struct MapArgs {
    Dims keep, red;
    int64_t n_out, n_red;
    int n_in, combine, post_op;
    double scale, shift, post_arg;
    int pre_op[MAXIN];
    double pre_arg[MAXIN];
    int64_t keep_strides[MAXIN][MAXR];
    int64_t red_strides[MAXIN][MAXR];
    int64_t out_strides[MAXR];
    const void* in[MAXIN];
    void* out;
    double* partial;  // [splits][n_out] when splits > 1
    int splits;
    int nt_store;     // dense map: streaming stores
    int bcast;        // dense kernels: bit k = operand k is constant along the fast axis (one
                      // scalar load, splat) instead of a 16-byte load
    int lanes_per_row;  // lane-dense reduce with n_out < 256: n_out / 4 lanes take one row, 64 / that rows per wave load (0: a wave per row)
    int64_t col_chunk;  // map_rows: columns per blockIdx.y (a multiple of 256); short-and-wide
                        // matrices (8 x 1M) are split along the columns as well as the rows
};

__device__ __forceinline__ void unravel(int64_t flat, const Dims& d, int64_t (&idx)[MAXR]) {
    if (d.rank <= 1) {  // the host coalesces axes, so this is the common case
#pragma unroll
        for (int a = 0; a < MAXR; ++a) idx[a] = 0;
        idx[0] = flat;
        return;
    }
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

// SP: with the special functions (lgamma, digamma).  They are compiled only into the
// generic kernels' SP instantiations: inlined into every kernel they cost 100+ VGPRs and
// scratch, which made the plain streaming kernels 3x slower (measured).
template <typename T, bool SP = false>
__device__ __forceinline__ T apply_unary(int op, T x, double arg) {
    if (SP) {
        if (op == BSC_OP_LGAMMA) return (T)lgamma((double)x);
        if (op == BSC_OP_DIGAMMA) return (T)bsc_digamma_f64((double)x);
    }
    switch (op) {
        case BSC_OP_LOG: return log(x);
        case BSC_OP_EXP: return exp(x);
        case BSC_OP_ABS: return fabs(x);
        case BSC_OP_SCALE: return x * (T)arg;
        case BSC_OP_POW:
            if (arg == -1.0) return (T)1 / x;
            if (arg == 2.0) return x * x;
            if (arg == 0.5) return sqrt(x);
            if (arg == 1.0) return x;
            return pow(x, (T)arg);
        default: return x;
    }
}

template <typename T, bool SP = false>
__device__ __forceinline__ T finish_value(const MapArgs& a, T v) {
    if (a.scale != 1.0) v *= (T)a.scale;
    if (a.shift != 0.0) v += (T)a.shift;
    return apply_unary<T, SP>(a.post_op, v, a.post_arg);
}

// value at (kept offsets koff[i], reduce index ridx)
template <typename T, bool SP>
__device__ __forceinline__ T map_value(const MapArgs& a, const int64_t (&koff)[MAXIN],
                                       const int64_t (&ridx)[MAXR]) {
    T v = a.combine == BSC_OP_MUL ? (T)1 : (T)0;
#pragma unroll
    for (int k = 0; k < MAXIN; ++k) {
        if (k < a.n_in) {
            const int64_t off = koff[k] + dot_strides(ridx, a.red_strides[k], a.red.rank);
            const T x = apply_unary<T, SP>(a.pre_op[k], static_cast<const T*>(a.in[k])[off], a.pre_arg[k]);
            v = a.combine == BSC_OP_MUL ? v * x : v + x;
        }
    }
    return finish_value<T, SP>(a, v);
}

// ---- pure map ------------------------------------------------------------------

template <typename T, bool SP>
__global__ __launch_bounds__(256) void map_strided_kernel(MapArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t zero[MAXR] = {0, 0, 0, 0, 0, 0};
    for (int64_t flat = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; flat < a.n_out;
         flat += stride) {
        int64_t idx[MAXR], koff[MAXIN];
        unravel(flat, a.keep, idx);
#pragma unroll
        for (int k = 0; k < MAXIN; ++k)
            koff[k] = k < a.n_in ? dot_strides(idx, a.keep_strides[k], a.keep.rank) : 0;
        static_cast<T*>(a.out)[dot_strides(idx, a.out_strides, a.keep.rank)] =
            map_value<T, SP>(a, koff, zero);
    }
}

// Parameter-sized strided maps (a [64, 16] coefficient block transposed into a stacked operand, a column of a
// statistics matrix): rank <= 2, float32, at most 2^20 elements.  Behind a data-sized pass these run with cold
// instruction caches, and the generic kernel above -- six 64-bit divisions per element, operand loops over
// MAXIN x MAXR -- is several KB of code: 28 us per launch for 1 024 elements (profiles/r03_derived_mog_*).  Here
// the index arithmetic is one 32-bit division and the operand loop has its real trip count.
template <int NIN>
__global__ __launch_bounds__(256) void map_small_f32_kernel(MapArgs a) {
    const unsigned n = (unsigned)a.n_out;
    const unsigned n1 = a.keep.rank == 2 ? (unsigned)a.keep.shape[1] : 1u;
    const int s_out0 = (int)a.out_strides[0], s_out1 = a.keep.rank == 2 ? (int)a.out_strides[1] : 0;
    for (unsigned flat = blockIdx.x * 256u + threadIdx.x; flat < n; flat += gridDim.x * 256u) {
        const unsigned i = a.keep.rank == 2 ? flat / n1 : flat, j = a.keep.rank == 2 ? flat - i * n1 : 0u;
        float v = a.combine == BSC_OP_MUL ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int off = (int)i * (int)a.keep_strides[k][0] + (int)j * (int)a.keep_strides[k][1];
            const float x = apply_unary<float, false>(a.pre_op[k], static_cast<const float*>(a.in[k])[off], a.pre_arg[k]);
            v = a.combine == BSC_OP_MUL ? v * x : v + x;
        }
        static_cast<float*>(a.out)[(int)i * s_out0 + (int)j * s_out1] = finish_value<float, false>(a, v);
    }
}

// every operand dense in the output's order (or one broadcast value): 16 B per lane,
// U float4 per operand in flight
// (period_mask: operand k is a row vector of `period4` float4 repeated down the rows -- a bias
// added to every row of a narrow [R, C] matrix: its float4 index is i mod period4)
template <int U>
__global__ __launch_bounds__(256) void map_dense_f32_kernel(MapArgs a, int64_t n4, int scalar_mask,
                                                            int period_mask = 0, int period4 = 1) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float id = a.combine == BSC_OP_MUL ? 1.f : 0.f;
    // float4 index of the periodic operands: i mod period4, kept incrementally (a 64-bit modulo per
    // operand and element held this kernel at 3 TB/s on a five-operand add of a [10M, 64] matrix)
    int prem[U];
    int pstep = 0;
    if (period_mask) {
        const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
        for (int j = 0; j < U; ++j) prem[j] = (int)((first + stride * j) % period4);
        pstep = (int)((stride * U) % period4);
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) prem[j] = 0;
    }
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
        float4 u[MAXIN][U];
#pragma unroll
        for (int k = 0; k < MAXIN; ++k) {   // all loads first
            if (k >= a.n_in) continue;
            if (scalar_mask & (1 << k)) {
                const float s = static_cast<const float*>(a.in[k])[0];
#pragma unroll
                for (int j = 0; j < U; ++j) u[k][j] = make_float4(s, s, s, s);
            } else if (period_mask & (1 << k)) {
#pragma unroll
                for (int j = 0; j < U; ++j)     // (past the end: any index of the vector is a valid load)
                    u[k][j] = *reinterpret_cast<const float4*>(static_cast<const float*>(a.in[k]) + 4 * prem[j]);
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int64_t i = i0 + stride * j;
                    u[k][j] = load4_nt(static_cast<const float*>(a.in[k]) + 4 * (i < n4 ? i : n4 - 1));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            float4 v = make_float4(id, id, id, id);
#pragma unroll
            for (int k = 0; k < MAXIN; ++k) {
                if (k >= a.n_in) continue;
                const int op = a.pre_op[k];
                const double arg = a.pre_arg[k];
                const float x0 = apply_unary<float>(op, u[k][j].x, arg);
                const float x1 = apply_unary<float>(op, u[k][j].y, arg);
                const float x2 = apply_unary<float>(op, u[k][j].z, arg);
                const float x3 = apply_unary<float>(op, u[k][j].w, arg);
                if (a.combine == BSC_OP_MUL) { v.x *= x0; v.y *= x1; v.z *= x2; v.w *= x3; }
                else { v.x += x0; v.y += x1; v.z += x2; v.w += x3; }
            }
            v.x = finish_value<float>(a, v.x); v.y = finish_value<float>(a, v.y);
            v.z = finish_value<float>(a, v.z); v.w = finish_value<float>(a, v.w);
            const int64_t i = i0 + stride * j;
            if (i < n4) {
                const f32x4 w = {v.x, v.y, v.z, v.w};
                if (a.nt_store) __builtin_nontemporal_store(w, static_cast<f32x4*>(a.out) + i);
                else static_cast<f32x4*>(a.out)[i] = w;
            }
        }
        if (period_mask) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                prem[j] += pstep;
                if (prem[j] >= period4) prem[j] -= period4;
            }
        }
    }
}

// The flat map (round 3): a contiguous float32 result of n4 float4 -- a flat array, or a row-major [R, C] with
// C % 4 == 0 -- whose operands are each dense like the result, ONE value, a C-vector repeated down the rows
// (dimshuffle(v, 'x', 0): float4 index i % C4) or one value per row (dimshuffle(u, 0, 'x'): row i / C4).
// Not persistent: a thread owns U float4, 4 KiB apart (a block's pieces are whole 4-KiB runs), every load of
// every operand issued before the first use; operand count and "all unary ops are cheap" are template parameters,
// so there is no loop over absent operands and no switch per element.  What the generic kernels above reached on
// 1M x 256 (profiles/r03_bench_map_vs_torch.txt): X * v[None, :] 3.7 TB/s, X / Y 4.9, X * 2 5.1 of read + write,
// where a tuned element-wise kernel reaches 6.0-6.3 on the same box.
struct FlatArgs {
    const float* in[4];
    float* out;
    unsigned n4, c4;           // float4 count; float4 per row (kinds 2, 3)
    int kind[4];               // 0 dense, 1 one value, 2 row vector (period c4), 3 one value per row
    int op[4];                 // pre op (LINEAR: copy / scale / pow -1 / pow 2 / pow 1 only)
    float arg[4];
    int combine, post_op, nt_store;
    float scale, shift, post_arg;
};

template <bool LINEAR>
__device__ __forceinline__ float flat_unary(int op, float x, float arg) {
    if (LINEAR) {      // (uniform selects; the compiler turns them into scalar branches around one instruction)
        if (op == BSC_OP_SCALE) return x * arg;
        if (op == BSC_OP_POW) return arg == 2.0f ? x * x : arg == -1.0f ? 1.0f / x : x;
        if (op == BSC_OP_ABS) return __builtin_fabsf(x);
        return x;
    }
    return apply_unary<float>(op, x, (double)arg);
}

// One operand's pre-op applied to a whole step's values (U x 16 bytes per lane) under ONE dispatch on `op`.  Called per
// ELEMENT, apply_unary's switch is a chain of scalar branches in front of every value: a reduction of exp(X) * Y spent
// more issue slots there than on expf itself (sum(exp(X) * Y, axis=0) at 1M x 256: 428 us against 316 for X * Y).
template <int U, class F>
__device__ __forceinline__ void each_value(float4 (&u)[U], F f) {
#pragma unroll
    for (int j = 0; j < U; ++j) { u[j].x = f(u[j].x); u[j].y = f(u[j].y); u[j].z = f(u[j].z); u[j].w = f(u[j].w); }
}
template <int U, class F>
__device__ __forceinline__ void each_value(f32x4 (&u)[U], F f) {
#pragma unroll
    for (int j = 0; j < U; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) u[j][e] = f(u[j][e]);
}
template <class V, int U>
__device__ __forceinline__ void apply_unary_step(int op, double arg, V (&u)[U]) {
    if (op == BSC_OP_COPY) return;
    if (op == BSC_OP_EXP) each_value(u, [](float x) { return expf(x); });
    else if (op == BSC_OP_LOG) each_value(u, [](float x) { return logf(x); });
    else if (op == BSC_OP_SCALE) { const float c = (float)arg; each_value(u, [c](float x) { return x * c; }); }
    else if (op == BSC_OP_ABS) each_value(u, [](float x) { return __builtin_fabsf(x); });
    else if (op == BSC_OP_POW && arg == 2.0) each_value(u, [](float x) { return x * x; });
    else if (op == BSC_OP_POW && arg == -1.0) each_value(u, [](float x) { return 1.0f / x; });
    else if (op == BSC_OP_POW && arg == 1.0) return;
    else each_value(u, [op, arg](float x) { return apply_unary<float>(op, x, arg); });
}
// ... and the tail of a value (scale, shift, post op) with the post op's dispatch taken once per call site
__device__ __forceinline__ bool post_is_plain(const MapArgs& a) {
    return a.post_op == BSC_OP_COPY;
}

template <int NIN, bool LINEAR>
__global__ __launch_bounds__(256) void map_flat_f32_kernel(FlatArgs a) {
    constexpr int U = 4;
    const unsigned base = (blockIdx.x * U) * 256u + threadIdx.x;
    const float id = a.combine == BSC_OP_MUL ? 1.f : 0.f;
    f32x4 u[NIN][U];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int kind = a.kind[k];
        if (kind == 1) {
            const float sv = a.in[k][0];
#pragma unroll
            for (int j = 0; j < U; ++j) u[k][j] = f32x4{sv, sv, sv, sv};
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                unsigned i = base + 256u * j;
                i = i < a.n4 ? i : a.n4 - 1;
                if (kind == 0) {
                    u[k][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.in[k]) + i);
                } else {
                    const unsigned r = i / a.c4;
                    if (kind == 2) {
                        u[k][j] = reinterpret_cast<const f32x4*>(a.in[k])[i - r * a.c4];
                    } else {
                        const float sv = a.in[k][r];
                        u[k][j] = f32x4{sv, sv, sv, sv};
                    }
                }
            }
        }
    }
    // (values are consumed and stored load by load -- the waits count down with the loads still in flight; one op
    // dispatch per 16 bytes, not per value: apply_unary_step)
#pragma unroll
    for (int j = 0; j < U; ++j) {
        f32x4 v = {id, id, id, id};
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int op = a.op[k];
            const float arg = a.arg[k];
            f32x4 x1[1] = {u[k][j]};
            if (!LINEAR) apply_unary_step(op, (double)arg, x1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = LINEAR ? flat_unary<true>(op, x1[0][e], arg) : x1[0][e];
                v[e] = a.combine == BSC_OP_MUL ? v[e] * x : v[e] + x;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float w = v[e];
            if (a.scale != 1.0f) w *= a.scale;
            if (a.shift != 0.0f) w += a.shift;
            v[e] = w;
        }
        f32x4 v1[1] = {v};
        if (!LINEAR) apply_unary_step(a.post_op, (double)a.post_arg, v1);
        const unsigned i = base + 256u * j;
        if (i < a.n4) {
            if (a.nt_store) __builtin_nontemporal_store(v1[0], reinterpret_cast<f32x4*>(a.out) + i);
            else reinterpret_cast<f32x4*>(a.out)[i] = v1[0];
        }
    }
}

// Two kept axes [R, C], every operand either dense along C (16-byte loads) or constant
// along it (one scalar per row, splat) -- column scalings, row weights, biases: the
// broadcasts of dimshuffle('x', ...).  A wave takes rows (interleaved over the grid: one
// moving window of memory), lanes take 16-byte column chunks; U rows in flight.
template <int N>
__global__ __launch_bounds__(256) void map_rows_f32_kernel(MapArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t gwave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int64_t R = a.keep.shape[0], C = a.keep.shape[1];
    const float id = a.combine == BSC_OP_MUL ? 1.f : 0.f;
    constexpr int U = 2;
    const int64_t c_begin = (int64_t)blockIdx.y * a.col_chunk;
    const int64_t c_end = c_begin + a.col_chunk < C ? c_begin + a.col_chunk : C;
    for (int64_t rb = gwave; rb < R; rb += n_waves * U) {
        for (int64_t c = c_begin + 4 * lane; c < c_end; c += 256) {
            float4 u[N][U];
#pragma unroll
            for (int k = 0; k < N; ++k) {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int64_t r = rb + n_waves * j < R ? rb + n_waves * j : R - 1;
                    const float* p = static_cast<const float*>(a.in[k]) + r * a.keep_strides[k][0];
                    if ((a.bcast >> k) & 1) {
                        const float sv = p[0];
                        u[k][j] = make_float4(sv, sv, sv, sv);
                    } else {
                        u[k][j] = load4_nt(p + c);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                float4 v = make_float4(id, id, id, id);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const int op = a.pre_op[k];
                    const double arg = a.pre_arg[k];
                    const float x0 = apply_unary<float>(op, u[k][j].x, arg);
                    const float x1 = apply_unary<float>(op, u[k][j].y, arg);
                    const float x2 = apply_unary<float>(op, u[k][j].z, arg);
                    const float x3 = apply_unary<float>(op, u[k][j].w, arg);
                    if (a.combine == BSC_OP_MUL) { v.x *= x0; v.y *= x1; v.z *= x2; v.w *= x3; }
                    else { v.x += x0; v.y += x1; v.z += x2; v.w += x3; }
                }
                const int64_t r = rb + n_waves * j;
                if (r < R) {
                    const f32x4 w = {finish_value<float>(a, v.x), finish_value<float>(a, v.y),
                                     finish_value<float>(a, v.z), finish_value<float>(a, v.w)};
                    f32x4* dst = reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + r * a.out_strides[0] + c);
                    if (a.nt_store) __builtin_nontemporal_store(w, dst);
                    else *dst = w;
                }
            }
        }
    }
}

// ---- map + reduce ----------------------------------------------------------------

// Variant A: the fastest-varying operand axis is a REDUCED one.  One wave per
// (output, split); lanes stride over the flattened reduce index, 4 elements in
// flight per lane.
template <typename T, bool SP>
__global__ __launch_bounds__(256) void map_reduce_wave_kernel(MapArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t job = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (job >= a.n_out * a.splits) return;
    const int64_t o = job % a.n_out;
    const int split = (int)(job / a.n_out);
    int64_t kidx[MAXR], koff[MAXIN];
    unravel(o, a.keep, kidx);
#pragma unroll
    for (int k = 0; k < MAXIN; ++k)
        koff[k] = k < a.n_in ? dot_strides(kidx, a.keep_strides[k], a.keep.rank) : 0;
    const int64_t chunk = (a.n_red + a.splits - 1) / a.splits;
    const int64_t r0 = split * chunk;
    const int64_t r1 = (r0 + chunk < a.n_red) ? r0 + chunk : a.n_red;
    double acc = 0.0;
    int64_t r = r0 + lane;
    for (; r + 192 < r1; r += 256) {
        T v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t ridx[MAXR];
            unravel(r + 64 * j, a.red, ridx);
            v[j] = map_value<T, SP>(a, koff, ridx);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += (double)v[j];
    }
    for (; r < r1; r += 64) {
        int64_t ridx[MAXR];
        unravel(r, a.red, ridx);
        acc += (double)map_value<T, SP>(a, koff, ridx);
    }
    acc = wave_allsum_f64(acc);
    if (lane == 0) {
        if (a.splits > 1) a.partial[(int64_t)split * a.n_out + o] = acc;
        else static_cast<T*>(a.out)[dot_strides(kidx, a.out_strides, a.keep.rank)] = (T)acc;
    }
}

// Variant A, dense: every operand is a dense run along the single reduce axis and
// 16-byte aligned there; 16 B per lane per load.
template <int N, int U, bool LINEAR = false>
__global__ __launch_bounds__(256) void map_reduce_wave_dense_f32_kernel(MapArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t job = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (job >= a.n_out * a.splits) return;
    const int64_t o = job % a.n_out;
    const int split = (int)(job / a.n_out);
    int64_t kidx[MAXR];
    unravel(o, a.keep, kidx);
    const float* base[N];
#pragma unroll
    for (int k = 0; k < N; ++k)
        base[k] = static_cast<const float*>(a.in[k]) + dot_strides(kidx, a.keep_strides[k], a.keep.rank);
    const int64_t n4 = a.n_red / 4;                       // host guarantees n_red % 4 == 0
    double acc = 0.0;
    const float id = a.combine == BSC_OP_MUL ? 1.f : 0.f;
    // U float4 per operand in flight per lane.  The splits of one output interleave
    // (split s takes every splits-th 4-KB piece) so that all waves in flight read one
    // moving window of memory -- DRAM pages stay open; private chunks per wave halve the
    // bandwidth.  Pieces past the end are clamped to a valid address and dropped.
    // (U = 1 for rows of up to 64 float4, where deeper unrolling would only re-read.)
    const int64_t r1 = n4;
    float fixed[N];                                   // operands constant along the reduce axis
#pragma unroll
    for (int k = 0; k < N; ++k) fixed[k] = (a.bcast >> k) & 1 ? base[k][0] : 0.f;
    for (int64_t rb = (int64_t)split * 64 * U + lane; rb < r1; rb += (int64_t)a.splits * 64 * U) {
        float4 u[N][U];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if ((a.bcast >> k) & 1) {
#pragma unroll
                for (int j = 0; j < U; ++j) u[k][j] = make_float4(fixed[k], fixed[k], fixed[k], fixed[k]);
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int64_t r = rb + 64 * j;
                    u[k][j] = load4_nt(base[k] + 4 * (r < r1 ? r : r1 - 1));
                }
            }
        }
        if (!LINEAR) {
#pragma unroll
            for (int k = 0; k < N; ++k) apply_unary_step(a.pre_op[k], a.pre_arg[k], u[k]);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            float4 v = make_float4(id, id, id, id);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int op = a.pre_op[k];
                const double arg = a.pre_arg[k];
                // (not LINEAR: apply_unary_step has been over the step's values already)
                const float x0 = LINEAR ? flat_unary<true>(op, u[k][j].x, (float)arg) : u[k][j].x;
                const float x1 = LINEAR ? flat_unary<true>(op, u[k][j].y, (float)arg) : u[k][j].y;
                const float x2 = LINEAR ? flat_unary<true>(op, u[k][j].z, (float)arg) : u[k][j].z;
                const float x3 = LINEAR ? flat_unary<true>(op, u[k][j].w, (float)arg) : u[k][j].w;
                if (a.combine == BSC_OP_MUL) { v.x *= x0; v.y *= x1; v.z *= x2; v.w *= x3; }
                else { v.x += x0; v.y += x1; v.z += x2; v.w += x3; }
            }
            if (rb + 64 * j < r1) {
                if (LINEAR || post_is_plain(a)) {
                    const float sc = (float)a.scale, sh = (float)a.shift;
                    acc += (double)(v.x * sc + sh); acc += (double)(v.y * sc + sh);
                    acc += (double)(v.z * sc + sh); acc += (double)(v.w * sc + sh);
                } else {
                    acc += (double)finish_value<float>(a, v.x);
                    acc += (double)finish_value<float>(a, v.y);
                    acc += (double)finish_value<float>(a, v.z);
                    acc += (double)finish_value<float>(a, v.w);
                }
            }
        }
    }
    acc = wave_allsum_f64(acc);
    if (lane == 0) {
        if (a.splits > 1) a.partial[(int64_t)split * a.n_out + o] = acc;
        else static_cast<float*>(a.out)[dot_strides(kidx, a.out_strides, a.keep.rank)] = (float)acc;
    }
}

// Row sums of a matrix with SHORT rows (n_red <= 1024: sum(X * Y, axis=1) at 1M x 256): the kernel above gives every
// row a wave of its own -- a million waves that each live for one 1-KiB load, and the dispatcher, not HBM, sets the
// pace (4.1 TB/s).  Here the waves are persistent and take U rows per step, every load before the first use.
// DBG (timing only, BSC_ROWS_DBG behind BSC_PROFILING_BUILDS): bit 0 float32 lane sums, bit 1 no sums across lanes,
// bit 2 plain loads, bit 3 the four ds_bpermute butterflies this kernel had before wave_allsum4_f64 (results right)
template <int N, bool LINEAR, int DBG = 0>
__global__ __launch_bounds__(256) void map_reduce_rows_f32_kernel(MapArgs a) {
    constexpr int U = 4;      // (eight rows in flight: 776 us instead of 407 -- the float64 butterflies of eight rows at once)
    const int lane = threadIdx.x & 63;
    const int64_t gwave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    const int c4 = (int)(a.n_red / 4);
    const float id = a.combine == BSC_OP_MUL ? 1.f : 0.f;
    const float scale = (float)a.scale, shift = (float)a.shift;
    for (int64_t r0 = gwave * U; r0 < a.n_out; r0 += n_waves * U) {
        double acc[U];
#pragma unroll
        for (int j = 0; j < U; ++j) acc[j] = 0.0;
        for (int c = lane; c < c4; c += 64) {
            f32x4 u[N][U];
#pragma unroll
            for (int k = 0; k < N; ++k) {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int64_t r = r0 + j < a.n_out ? r0 + j : a.n_out - 1;
                    const float* p = static_cast<const float*>(a.in[k]) + r * a.keep_strides[k][0];
                    if ((a.bcast >> k) & 1) {
                        const float sv = p[0];
                        u[k][j] = f32x4{sv, sv, sv, sv};
                    } else {
                        u[k][j] = (DBG & 4) ? reinterpret_cast<const f32x4*>(p)[c]
                                            : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + c);
                    }
                }
            }
            if (!LINEAR) {
#pragma unroll
                for (int k = 0; k < N; ++k) apply_unary_step(a.pre_op[k], a.pre_arg[k], u[k]);
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                f32x4 v = {id, id, id, id};
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const int op = a.pre_op[k];
                    const float arg = (float)a.pre_arg[k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = LINEAR ? flat_unary<true>(op, u[k][j][e], arg) : u[k][j][e];
                        v[e] = a.combine == BSC_OP_MUL ? v[e] * x : v[e] + x;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float w = v[e];
                    if (LINEAR || post_is_plain(a)) {
                        if (scale != 1.0f) w *= scale;
                        if (shift != 0.0f) w += shift;
                    } else {
                        w = finish_value<float>(a, w);
                    }
                    if (DBG & 1) acc[j] = (double)((float)acc[j] + w);
                    else acc[j] += (double)w;
                }
            }
        }
        if (DBG & 2) {
            if (lane < U && r0 + lane < a.n_out) static_cast<float*>(a.out)[(r0 + lane) * a.out_strides[0]] = (float)acc[0];
        } else if (DBG & 8) {      // the round-3 form: four butterflies through the LDS crossbar
#pragma unroll
            for (int j = 0; j < U; ++j)
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_xor(acc[j], off);
            if (lane < U && r0 + lane < a.n_out) {
                double mine = acc[0];
#pragma unroll
                for (int j = 1; j < U; ++j) mine = lane == j ? acc[j] : mine;
                static_cast<float*>(a.out)[(r0 + lane) * a.out_strides[0]] = (float)mine;
            }
        } else {
            static_assert(U == 4, "wave_allsum4_f64 folds four rows");
            const double mine = wave_allsum4_f64(acc[0], acc[1], acc[2], acc[3]);     // row (lane >> 4)'s sum
            const int64_t r = r0 + (lane >> 4);
            if ((lane & 15) == 0 && r < a.n_out) static_cast<float*>(a.out)[r * a.out_strides[0]] = (float)mine;
        }
    }
}

// Variant B, dense: one kept axis and one reduce axis, every operand dense along
// the kept axis and 16-byte aligned there.  Lane <-> 4 consecutive
// outputs, a wave covers 256 outputs per row; the four waves of a block and
// `splits` blocks divide the reduce range, 4 rows in flight per lane.
template <int N, bool LINEAR = false>
__global__ __launch_bounds__(256) void map_reduce_lane_dense_f32_kernel(MapArgs a) {
    __shared__ double red[4][64][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // rows are wave-uniform
    const int64_t n_groups = (a.n_out + 255) / 256;
    const int64_t group = blockIdx.x % n_groups;
    const int split = (int)(blockIdx.x / n_groups);
    const int64_t o = group * 256 + 4 * lane;           // n_out % 4 == 0 (host)
    const int64_t r1 = a.n_red;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const float id = a.combine == BSC_OP_MUL ? 1.f : 0.f;
    if (o < a.n_out) {
        // a block takes 4*U consecutive rows per step (wave w row w, w+4, ...), the splits
        // of one output group interleave: one moving window of memory for the whole grid
        constexpr int U = 4;
        for (int64_t rb = (int64_t)split * 4 * U + wave; rb < r1; rb += (int64_t)a.splits * 4 * U) {
            float4 u[N][U];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float* base = static_cast<const float*>(a.in[k]) + o * a.keep_strides[k][0];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int64_t r = rb + 4 * j < r1 ? rb + 4 * j : r1 - 1;
                    if ((a.bcast >> k) & 1) {       // one value per row, the same for every column:
                        // a wave-uniform address, so this is a scalar (SMEM) load
                        const float sv = static_cast<const float*>(a.in[k])[r * a.red_strides[k][0]];
                        u[k][j] = make_float4(sv, sv, sv, sv);
                    } else {
                        u[k][j] = load4_nt(base + r * a.red_strides[k][0]);
                    }
                }
            }
            if (!LINEAR) {
#pragma unroll
                for (int k = 0; k < N; ++k) apply_unary_step(a.pre_op[k], a.pre_arg[k], u[k]);
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                float4 v = make_float4(id, id, id, id);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const int op = a.pre_op[k];
                    const double arg = a.pre_arg[k];
                    // (not LINEAR: apply_unary_step has been over the step's values already)
                    const float x0 = LINEAR ? flat_unary<true>(op, u[k][j].x, (float)arg) : u[k][j].x;
                    const float x1 = LINEAR ? flat_unary<true>(op, u[k][j].y, (float)arg) : u[k][j].y;
                    const float x2 = LINEAR ? flat_unary<true>(op, u[k][j].z, (float)arg) : u[k][j].z;
                    const float x3 = LINEAR ? flat_unary<true>(op, u[k][j].w, (float)arg) : u[k][j].w;
                    if (a.combine == BSC_OP_MUL) { v.x *= x0; v.y *= x1; v.z *= x2; v.w *= x3; }
                    else { v.x += x0; v.y += x1; v.z += x2; v.w += x3; }
                }
                if (rb + 4 * j < r1) {
                    if (LINEAR || post_is_plain(a)) {       // (post op is a copy: scale and shift only)
                        const float sc = (float)a.scale, sh = (float)a.shift;
                        acc[0] += (double)(v.x * sc + sh); acc[1] += (double)(v.y * sc + sh);
                        acc[2] += (double)(v.z * sc + sh); acc[3] += (double)(v.w * sc + sh);
                    } else {
                        acc[0] += (double)finish_value<float>(a, v.x);
                        acc[1] += (double)finish_value<float>(a, v.y);
                        acc[2] += (double)finish_value<float>(a, v.z);
                        acc[3] += (double)finish_value<float>(a, v.w);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave][lane][c] = acc[c];
    __syncthreads();
    if (wave == 0 && o < a.n_out) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double tot = ((red[0][lane][c] + red[1][lane][c]) + red[2][lane][c]) + red[3][lane][c];
            if (a.splits > 1) a.partial[(int64_t)split * a.n_out + o + c] = tot;
            else static_cast<float*>(a.out)[(o + c) * a.out_strides[0]] = (float)tot;
        }
    }
}

// ... with fewer than 256 outputs (a [10M x 64] matrix summed over its rows): a wave per row would
// keep a quarter of its lanes (64 outputs) and move 256 B per load.  Here n_out / 4 lanes take one
// row and a wave load covers 64 / that many CONSECUTIVE rows (1 KiB of a dense matrix); the
// sub-rows of a wave and the waves of a block are added in a fixed order at the end.
template <int N>
__global__ __launch_bounds__(256) void map_reduce_lane_narrow_f32_kernel(MapArgs a) {
    __shared__ double red[4][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = a.lanes_per_row, R = 64 / L;
    const int sub = lane / L, o = 4 * (lane - sub * L);
    const bool active = sub < R;
    const int split = blockIdx.x;
    const int64_t r1 = a.n_red;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const float id = a.combine == BSC_OP_MUL ? 1.f : 0.f;
    if (active) {
        constexpr int U = 4;
        const int64_t step = (int64_t)4 * U * R;
        for (int64_t rb = (int64_t)split * step; rb < r1; rb += (int64_t)a.splits * step) {
            float4 u[N][U];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float* base = static_cast<const float*>(a.in[k]) + o * a.keep_strides[k][0];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int64_t row = rb + (int64_t)(4 * j + wave) * R + sub;
                    const int64_t r = row < r1 ? row : r1 - 1;
                    if ((a.bcast >> k) & 1) {
                        const float sv = static_cast<const float*>(a.in[k])[r * a.red_strides[k][0]];
                        u[k][j] = make_float4(sv, sv, sv, sv);
                    } else {
                        u[k][j] = load4_nt(base + r * a.red_strides[k][0]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < N; ++k) apply_unary_step(a.pre_op[k], a.pre_arg[k], u[k]);      // one dispatch per step
#pragma unroll
            for (int j = 0; j < U; ++j) {
                float4 v = make_float4(id, id, id, id);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const float x0 = u[k][j].x, x1 = u[k][j].y, x2 = u[k][j].z, x3 = u[k][j].w;
                    if (a.combine == BSC_OP_MUL) { v.x *= x0; v.y *= x1; v.z *= x2; v.w *= x3; }
                    else { v.x += x0; v.y += x1; v.z += x2; v.w += x3; }
                }
                if (rb + (int64_t)(4 * j + wave) * R + sub < r1) {
                    if (post_is_plain(a)) {
                        const float sc = (float)a.scale, sh = (float)a.shift;
                        acc[0] += (double)(v.x * sc + sh); acc[1] += (double)(v.y * sc + sh);
                        acc[2] += (double)(v.z * sc + sh); acc[3] += (double)(v.w * sc + sh);
                    } else {
                        acc[0] += (double)finish_value<float>(a, v.x);
                        acc[1] += (double)finish_value<float>(a, v.y);
                        acc[2] += (double)finish_value<float>(a, v.z);
                        acc[3] += (double)finish_value<float>(a, v.w);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave][lane][c] = acc[c];
    __syncthreads();
    if (wave == 0 && lane < L) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double tot = 0.0;
            for (int w = 0; w < 4; ++w)
                for (int s2 = 0; s2 < R; ++s2) tot += red[w][s2 * L + lane][c];
            if (a.splits > 1) a.partial[(int64_t)split * a.n_out + 4 * lane + c] = tot;
            else static_cast<float*>(a.out)[(4 * lane + c) * a.out_strides[0]] = (float)tot;
        }
    }
}

// Variant B: the fastest-varying operand axis is a KEPT one.  Lane <-> output, so a
// wave reads 64 consecutive elements per reduce step; the four waves of a block
// and `splits` blocks divide the reduce range.
template <typename T, bool SP>
__global__ __launch_bounds__(256) void map_reduce_lane_kernel(MapArgs a) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_groups = (a.n_out + 63) / 64;
    const int64_t group = blockIdx.x % n_groups;
    const int split = (int)(blockIdx.x / n_groups);
    const int64_t o = group * 64 + lane;
    const int64_t chunk = (a.n_red + a.splits - 1) / a.splits;
    const int64_t r0 = split * chunk;
    const int64_t r1 = (r0 + chunk < a.n_red) ? r0 + chunk : a.n_red;
    double acc = 0.0;
    int64_t kidx[MAXR] = {0, 0, 0, 0, 0, 0};
    if (o < a.n_out) {
        int64_t koff[MAXIN];
        unravel(o, a.keep, kidx);
        #pragma unroll
    for (int k = 0; k < MAXIN; ++k)
        koff[k] = k < a.n_in ? dot_strides(kidx, a.keep_strides[k], a.keep.rank) : 0;
        int64_t r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            T v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int64_t ridx[MAXR];
                unravel(r + 4 * j, a.red, ridx);
                v[j] = map_value<T, SP>(a, koff, ridx);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += (double)v[j];
        }
        for (; r < r1; r += 4) {
            int64_t ridx[MAXR];
            unravel(r, a.red, ridx);
            acc += (double)map_value<T, SP>(a, koff, ridx);
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && o < a.n_out) {
        const double tot = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        if (a.splits > 1) a.partial[(int64_t)split * a.n_out + o] = tot;
        else static_cast<T*>(a.out)[dot_strides(kidx, a.out_strides, a.keep.rank)] = (T)tot;
    }
}

// Sum of the split partials, fixed order.  Few outputs with many splits (a full
// reduction has 1 output and thousands of splits): one wave per output, lanes
// stride over the splits with 4 loads in flight, float64 butterfly.  Many outputs:
// one thread per output (the splits are few then).
template <typename T>
__global__ __launch_bounds__(256) void map_reduce_finish_kernel(MapArgs a, int wave_per_output) {
    int64_t o;
    double tot = 0.0;
    if (wave_per_output) {
        const int lane = threadIdx.x & 63;
        o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (o >= a.n_out) return;
        const double* p = a.partial + o;
        int s = lane;
        for (; s + 192 < a.splits; s += 256) {
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = p[(int64_t)(s + 64 * j) * a.n_out];
#pragma unroll
            for (int j = 0; j < 4; ++j) tot += v[j];
        }
        for (; s < a.splits; s += 64) tot += p[(int64_t)s * a.n_out];
        tot = wave_allsum_f64(tot);
        if (lane != 0) return;
    } else {
        o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (o >= a.n_out) return;
        for (int s = 0; s < a.splits; ++s) tot += a.partial[(int64_t)s * a.n_out + o];
    }
    int64_t kidx[MAXR];
    unravel(o, a.keep, kidx);
    static_cast<T*>(a.out)[dot_strides(kidx, a.out_strides, a.keep.rank)] = (T)tot;
}

// ---- host: axis coalescing ----------------------------------------------------------

// One group of axes (kept or reduced) with the strides of every operand (and of
// the output for the kept group).  Size-1 axes are dropped; neighbours (outer,
// inner) merge when every stride row satisfies outer == inner * extent(inner).
struct AxisGroup {
    int rank = 0;
    int64_t shape[MAXR];
    int64_t strides[MAXIN + 1][MAXR];
};

void coalesce(AxisGroup& g, int n_rows) {
    int out = 0;
    for (int a = 0; a < g.rank; ++a) {
        if (g.shape[a] == 1) continue;
        bool merge = out > 0;
        if (merge)
            for (int k = 0; k < n_rows; ++k)
                if (g.strides[k][out - 1] != g.strides[k][a] * g.shape[a]) merge = false;
        if (merge) {
            g.shape[out - 1] *= g.shape[a];
            for (int k = 0; k < n_rows; ++k) g.strides[k][out - 1] = g.strides[k][a];
        } else {
            g.shape[out] = g.shape[a];
            for (int k = 0; k < n_rows; ++k) g.strides[k][out] = g.strides[k][a];
            ++out;
        }
    }
    g.rank = out;
}

int64_t iabs(int64_t v) { return v < 0 ? -v : v; }

// Order the axes of a group by decreasing |stride| of row `key` (an element-wise map does not
// care in which order its index space is walked; a sum only changes its -- still fixed --
// summation order).  A transposed view then coalesces like its row-major original.
void sort_axes(AxisGroup& g, int n_rows, int key) {
    for (int a = 1; a < g.rank; ++a)          // insertion sort, stable
        for (int b = a; b > 0 && iabs(g.strides[key][b - 1]) < iabs(g.strides[key][b]); --b) {
            int64_t t = g.shape[b]; g.shape[b] = g.shape[b - 1]; g.shape[b - 1] = t;
            for (int k = 0; k < n_rows; ++k) {
                t = g.strides[k][b]; g.strides[k][b] = g.strides[k][b - 1]; g.strides[k][b - 1] = t;
            }
        }
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// every offset of the small map fits 32 bits, and pow (if any) is one of the cheap cases
bool small_ok(const MapArgs& m, int rank, int n_in) {
    auto span = [&](const int64_t* strides) {
        int64_t lo = 0, hi = 0;
        for (int a = 0; a < rank; ++a) {
            const int64_t e = (m.keep.shape[a] - 1) * strides[a];
            if (e < 0) lo += e; else hi += e;
        }
        return lo > -(int64_t(1) << 30) && hi < (int64_t(1) << 30);
    };
    auto pow_ok = [](int op, double arg) {
        return op != BSC_OP_POW || arg == -1.0 || arg == 2.0 || arg == 0.5 || arg == 1.0;
    };
    if (!span(m.out_strides) || !pow_ok(m.post_op, m.post_arg)) return false;
    for (int k = 0; k < n_in; ++k)
        if (!span(m.keep_strides[k]) || !pow_ok(m.pre_op[k], m.pre_arg[k])) return false;
    return true;
}
}  // namespace

extern "C" {

int bsc_map_reduce(bsc_ctx* ctx, int dtype, int combine, int rank_keep,
                   const int64_t* host_keep_shape, int rank_red, const int64_t* host_red_shape,
                   int n_in, const void* const* host_in, const int64_t* host_in_keep_strides,
                   const int64_t* host_in_red_strides, const int32_t* host_pre_op,
                   const double* host_pre_arg, double scale, double shift, int post_op,
                   double post_arg, void* out, const int64_t* host_out_strides) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(dtype == BSC_F32 || dtype == BSC_F64, "bsc_map_reduce: unknown dtype %d", dtype);
    BSC_REQUIRE(combine == BSC_OP_ADD || combine == BSC_OP_MUL,
                "bsc_map_reduce: combine must be BSC_OP_ADD or BSC_OP_MUL, got %d", combine);
    BSC_REQUIRE(rank_keep >= 0 && rank_keep <= MAXR && rank_red >= 0 && rank_red <= MAXR,
                "bsc_map_reduce: rank exceeds %d", MAXR);
    BSC_REQUIRE(n_in >= 1 && n_in <= MAXIN && host_in && host_pre_op && host_pre_arg && out,
                "bsc_map_reduce: bad operands");
    auto unary_ok = [](int op) {
        return op == BSC_OP_COPY || op == BSC_OP_SCALE || op == BSC_OP_LOG || op == BSC_OP_EXP || op == BSC_OP_ABS ||
               op == BSC_OP_POW || op == BSC_OP_LGAMMA || op == BSC_OP_DIGAMMA;
    };
    BSC_REQUIRE(unary_ok(post_op), "bsc_map_reduce: post op %d is not unary", post_op);
    AxisGroup keep, red;
    keep.rank = rank_keep;
    red.rank = rank_red;
    int64_t n_out = 1, n_red = 1;
    for (int a = 0; a < rank_keep; ++a) {
        BSC_REQUIRE(host_keep_shape[a] >= 0, "bsc_map_reduce: negative extent");
        keep.shape[a] = host_keep_shape[a];
        n_out *= keep.shape[a];
        keep.strides[n_in][a] = host_out_strides[a];
    }
    for (int a = 0; a < rank_red; ++a) {
        BSC_REQUIRE(host_red_shape[a] >= 0, "bsc_map_reduce: negative extent");
        red.shape[a] = host_red_shape[a];
        n_red *= red.shape[a];
    }
    MapArgs m{};
    for (int k = 0; k < n_in; ++k) {
        BSC_REQUIRE(host_in[k] != nullptr || n_out * n_red == 0, "bsc_map_reduce: input %d is null", k);
        BSC_REQUIRE(unary_ok(host_pre_op[k]), "bsc_map_reduce: pre op %d is not unary", host_pre_op[k]);
        m.in[k] = host_in[k];
        m.pre_op[k] = host_pre_op[k];
        m.pre_arg[k] = host_pre_arg[k];
        for (int a = 0; a < rank_keep; ++a) keep.strides[k][a] = host_in_keep_strides[k * rank_keep + a];
        for (int a = 0; a < rank_red; ++a) red.strides[k][a] = host_in_red_strides[k * rank_red + a];
    }
    if (n_out == 0) return BSC_OK;
    bool special = post_op == BSC_OP_LGAMMA || post_op == BSC_OP_DIGAMMA;
    for (int k = 0; k < n_in; ++k)
        special = special || host_pre_op[k] == BSC_OP_LGAMMA || host_pre_op[k] == BSC_OP_DIGAMMA;
    sort_axes(keep, n_in + 1, n_in);                  // by the output's strides
    if (red.rank > 1) {                               // by the strides of the biggest operand
        int key = 0;
        int64_t best = -1;
        for (int k = 0; k < n_in; ++k) {
            int64_t span = 0;
            for (int a = 0; a < red.rank; ++a) span += iabs(red.strides[k][a]) * (red.shape[a] - 1);
            if (span > best) { best = span; key = k; }
        }
        sort_axes(red, n_in, key);
    }
    coalesce(keep, n_in + 1);
    coalesce(red, n_in);
    m.keep.rank = keep.rank;
    m.red.rank = red.rank;
    for (int a = 0; a < MAXR; ++a) {
        m.keep.shape[a] = a < keep.rank ? keep.shape[a] : 1;
        m.red.shape[a] = a < red.rank ? red.shape[a] : 1;
        m.out_strides[a] = a < keep.rank ? keep.strides[n_in][a] : 0;
        for (int k = 0; k < n_in; ++k) {
            m.keep_strides[k][a] = a < keep.rank ? keep.strides[k][a] : 0;
            m.red_strides[k][a] = a < red.rank ? red.strides[k][a] : 0;
        }
    }
    m.n_out = n_out;
    m.n_red = n_red;
    m.n_in = n_in;
    m.combine = combine;
    m.post_op = post_op;
    m.post_arg = post_arg;
    m.scale = scale;
    m.shift = shift;
    m.out = out;
    m.splits = 1;
    m.partial = nullptr;
    m.nt_store = ctx->fused_nt_store;

    if (rank_red == 0) {
        // ---- pure map ----
        // the flat kernel: contiguous float32 result, operands dense / one value / row vector / one value per row
        if (!special && dtype == BSC_F32 && n_in <= 4 && n_out >= 4096 && n_out % 4 == 0 && n_out / 4 < ((int64_t)1 << 31) &&
            (((uintptr_t)out) & 15) == 0 && ctx->fused_map_flat &&
            (keep.rank == 1 ? m.out_strides[0] == 1
                            : keep.rank == 2 && m.out_strides[1] == 1 && m.out_strides[0] == keep.shape[1] && keep.shape[1] % 4 == 0)) {
            FlatArgs f{};
            const int64_t C = keep.rank == 2 ? keep.shape[1] : n_out;
            bool ok = true, linear = post_op == BSC_OP_COPY;
            for (int k = 0; k < n_in && ok; ++k) {
                const int64_t s0 = keep.rank == 2 ? m.keep_strides[k][0] : 0, s1 = m.keep_strides[k][keep.rank - 1];
                int kind = -1;
                if (keep.rank == 1) kind = s1 == 1 ? 0 : s1 == 0 ? 1 : -1;
                else if (s0 == C && s1 == 1) kind = 0;
                else if (s0 == 0 && s1 == 0) kind = 1;
                else if (s0 == 0 && s1 == 1) kind = 2;
                else if (s0 == 1 && s1 == 0) kind = 3;
                if (kind < 0 || ((kind == 0 || kind == 2) && (((uintptr_t)m.in[k]) & 15) != 0)) ok = false;
                f.in[k] = (const float*)m.in[k];
                f.kind[k] = kind;
                f.op[k] = m.pre_op[k];
                f.arg[k] = (float)m.pre_arg[k];
                const int op = m.pre_op[k];
                if (!(op == BSC_OP_COPY || op == BSC_OP_SCALE || op == BSC_OP_ABS ||
                      (op == BSC_OP_POW && (m.pre_arg[k] == 2.0 || m.pre_arg[k] == -1.0 || m.pre_arg[k] == 1.0))))
                    linear = false;
            }
            if (ok) {
                f.out = (float*)out;
                f.n4 = (unsigned)(n_out / 4);
                f.c4 = (unsigned)(C / 4);
                f.combine = combine; f.post_op = post_op; f.nt_store = m.nt_store;
                f.scale = (float)scale; f.shift = (float)shift; f.post_arg = (float)post_arg;
                const unsigned blocks = (f.n4 + 1023u) / 1024u;
#define BSC_FLAT(N_)                                                                                                   \
    do {                                                                                                               \
        if (linear) hipLaunchKernelGGL((map_flat_f32_kernel<N_, true>), dim3(blocks), dim3(256), 0, ctx->stream, f);   \
        else hipLaunchKernelGGL((map_flat_f32_kernel<N_, false>), dim3(blocks), dim3(256), 0, ctx->stream, f);         \
    } while (0)
                switch (n_in) {
                    case 1: BSC_FLAT(1); break;
                    case 2: BSC_FLAT(2); break;
                    case 3: BSC_FLAT(3); break;
                    default: BSC_FLAT(4); break;
                }
#undef BSC_FLAT
                BSC_LAUNCH_CHECK();
                return BSC_OK;
            }
        }
        bool dense = !special && dtype == BSC_F32 && keep.rank <= 1 && (n_out % 4) == 0 &&
                     (((uintptr_t)out) & 15) == 0 && (keep.rank == 0 || m.out_strides[0] == 1);
        int scalar_mask = 0;
        for (int k = 0; k < n_in && dense; ++k) {
            const int64_t s = keep.rank ? m.keep_strides[k][0] : 0;
            if (s == 0) scalar_mask |= 1 << k;
            else if (s != 1 || (((uintptr_t)m.in[k]) & 15) != 0) dense = false;
        }
        // a narrow [R, C] result (or more than three operands) whose operands are dense, one scalar,
        // or a C-vector repeated down the rows (biases: dimshuffle(v, 'x', 0)): the flat dense kernel
        // with periodic operands -- the row kernel below gives a whole wave to 64 columns
        if (!dense && !special && dtype == BSC_F32 && keep.rank == 2 && m.out_strides[1] == 1 &&
            m.out_strides[0] == keep.shape[1] && keep.shape[1] % 4 == 0 && keep.shape[1] <= (1 << 20) &&
            (keep.shape[1] < 256 || n_in > 3) && (((uintptr_t)out) & 15) == 0) {
            const int64_t C = keep.shape[1];
            int smask = 0, pmask = 0;
            bool ok = true;
            for (int k = 0; k < n_in && ok; ++k) {
                const int64_t s0 = m.keep_strides[k][0], s1 = m.keep_strides[k][1];
                if (s0 == 0 && s1 == 0) smask |= 1 << k;
                else if ((((uintptr_t)m.in[k]) & 15) != 0) ok = false;
                else if (s0 == 0 && s1 == 1) pmask |= 1 << k;
                else if (!(s0 == C && s1 == 1)) ok = false;
            }
            if (ok) {
                const int64_t n4 = n_out / 4;
                int64_t blocks = (n4 + 255) / 256;
                const int64_t cap = (int64_t)ctx->cu_count * ctx->fused_map_blocks_per_cu;
                if (blocks > cap) blocks = cap;
                hipLaunchKernelGGL(map_dense_f32_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, m,
                                   n4, smask, pmask, (int)(C / 4));
                BSC_LAUNCH_CHECK();
                return BSC_OK;
            }
        }
        // two kept axes with row / column broadcasts
        bool rows2d = !dense && !special && dtype == BSC_F32 && keep.rank == 2 && n_in <= 3 &&
                      keep.shape[1] % 4 == 0 && keep.shape[1] >= 64 && m.out_strides[1] == 1 &&
                      m.out_strides[0] % 4 == 0 && (((uintptr_t)out) & 15) == 0;
        int bmask = 0;
        for (int k = 0; k < n_in && rows2d; ++k) {
            const int64_t sc = m.keep_strides[k][1];
            if (sc == 0) bmask |= 1 << k;
            else if (sc != 1 || m.keep_strides[k][0] % 4 != 0 || (((uintptr_t)m.in[k]) & 15) != 0)
                rows2d = false;
        }
        if (rows2d) {
            m.bcast = bmask;
            int64_t blocks = (keep.shape[0] + 7) / 8;          // 4 waves x 2 rows per step
            const int64_t cap = (int64_t)ctx->cu_count * ctx->fused_map_blocks_per_cu;
            if (blocks > cap) blocks = cap;
            if (blocks < 1) blocks = 1;
            // few rows (an 8 x 1M matrix of per-sample values): also split the columns, or
            // eight waves would walk 32 MB alone (10 ms instead of ~20 us)
            int64_t chunks = 1;
            if (blocks < cap) {
                chunks = (cap + blocks - 1) / blocks;
                const int64_t max_chunks = (keep.shape[1] + 1023) / 1024;      // >= 1024 columns per chunk
                if (chunks > max_chunks) chunks = max_chunks;
                if (chunks > 65535) chunks = 65535;
                if (chunks < 1) chunks = 1;
            }
            m.col_chunk = ((keep.shape[1] + chunks - 1) / chunks + 255) / 256 * 256;
            chunks = (keep.shape[1] + m.col_chunk - 1) / m.col_chunk;
            const dim3 rgrid((unsigned)blocks, (unsigned)chunks);
            switch (n_in) {
                case 1: hipLaunchKernelGGL(map_rows_f32_kernel<1>, rgrid, dim3(256), 0, ctx->stream, m); break;
                case 2: hipLaunchKernelGGL(map_rows_f32_kernel<2>, rgrid, dim3(256), 0, ctx->stream, m); break;
                default: hipLaunchKernelGGL(map_rows_f32_kernel<3>, rgrid, dim3(256), 0, ctx->stream, m); break;
            }
            BSC_LAUNCH_CHECK();
            return BSC_OK;
        }
        if (dense && n_out >= 4) {
            const int64_t n4 = n_out / 4;
            int64_t blocks = (n4 + 255) / 256;
            const int64_t cap = (int64_t)ctx->cu_count * ctx->fused_map_blocks_per_cu;
            if (blocks > cap) blocks = cap;
            if (ctx->fused_map_unroll == 1)
                hipLaunchKernelGGL(map_dense_f32_kernel<1>, dim3((unsigned)blocks), dim3(256), 0,
                                   ctx->stream, m, n4, scalar_mask);
            else
                hipLaunchKernelGGL(map_dense_f32_kernel<2>, dim3((unsigned)blocks), dim3(256), 0,
                                   ctx->stream, m, n4, scalar_mask);
        } else if (!special && dtype == BSC_F32 && keep.rank <= 2 && n_out <= (1 << 20) && n_in <= 4 && small_ok(m, keep.rank, n_in)) {
            const unsigned blocks = (unsigned)((n_out + 255) / 256);
            switch (n_in) {
                case 1: hipLaunchKernelGGL(map_small_f32_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, m); break;
                case 2: hipLaunchKernelGGL(map_small_f32_kernel<2>, dim3(blocks), dim3(256), 0, ctx->stream, m); break;
                case 3: hipLaunchKernelGGL(map_small_f32_kernel<3>, dim3(blocks), dim3(256), 0, ctx->stream, m); break;
                default: hipLaunchKernelGGL(map_small_f32_kernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, m); break;
            }
        } else {
            int64_t blocks = (n_out + 255) / 256;
            const int64_t cap = (int64_t)ctx->cu_count * 8;
            if (blocks > cap) blocks = cap;
#define BSC_GENERIC(KERNEL, GRID)                                                              \
    do {                                                                                       \
        if (dtype == BSC_F32 && special)                                                       \
            hipLaunchKernelGGL((KERNEL<float, true>), dim3((unsigned)(GRID)), dim3(256), 0, ctx->stream, m);  \
        else if (dtype == BSC_F32)                                                             \
            hipLaunchKernelGGL((KERNEL<float, false>), dim3((unsigned)(GRID)), dim3(256), 0, ctx->stream, m); \
        else if (special)                                                                      \
            hipLaunchKernelGGL((KERNEL<double, true>), dim3((unsigned)(GRID)), dim3(256), 0, ctx->stream, m); \
        else                                                                                   \
            hipLaunchKernelGGL((KERNEL<double, false>), dim3((unsigned)(GRID)), dim3(256), 0, ctx->stream, m); \
    } while (0)
            BSC_GENERIC(map_strided_kernel, blocks);
        }
        BSC_LAUNCH_CHECK();
        return BSC_OK;
    }

    // ---- map + reduce: which axis of the BIGGEST operand varies fastest? ----
    // (a broadcast vector riding along -- sum_n f(X)_nd u_n -- must not decide the access
    // pattern of the matrix it multiplies)
    int big = 0;
    double big_elems = -1.0;
    for (int k = 0; k < n_in; ++k) {
        double elems = 1.0;
        for (int a = 0; a < keep.rank; ++a) if (m.keep_strides[k][a] != 0) elems *= (double)m.keep.shape[a];
        for (int a = 0; a < red.rank; ++a) if (m.red_strides[k][a] != 0) elems *= (double)m.red.shape[a];
        if (elems > big_elems) { big_elems = elems; big = k; }
    }
    int64_t min_keep = INT64_MAX, min_red = INT64_MAX;
    for (int a = 0; a < keep.rank; ++a) {
        const int64_t st = iabs(m.keep_strides[big][a]);
        if (st != 0 && st < min_keep) min_keep = st;
    }
    for (int a = 0; a < red.rank; ++a) {
        const int64_t st = iabs(m.red_strides[big][a]);
        if (st != 0 && st < min_red) min_red = st;
    }
    const bool lanes_over_outputs = min_keep < min_red && n_out >= 16;
    // dense variants (16 B per lane): one reduce axis, and for the lane variant one kept axis
    bool dense_lane = !special && lanes_over_outputs && dtype == BSC_F32 && n_in <= 3 && keep.rank == 1 && red.rank == 1 &&
                      (n_out % 4) == 0 && m.out_strides[0] == 1;
    bool dense_wave = !special && !lanes_over_outputs && dtype == BSC_F32 && n_in <= 3 && red.rank == 1 &&
                      (n_red % 4) == 0;
    int bmask = 0;
    for (int k = 0; k < n_in; ++k) {
        if (dense_lane) {
            const int64_t sk = m.keep_strides[k][0];
            if (sk == 0) bmask |= 1 << k;                         // one value per reduced row
            else if (sk != 1 || (((uintptr_t)m.in[k]) & 15) != 0 || m.red_strides[k][0] % 4 != 0)
                dense_lane = false;
        }
        if (dense_wave) {
            const int64_t sr = m.red_strides[k][0];
            if (sr == 0) bmask |= 1 << k;                         // one value per output
            else {
                if (sr != 1 || (((uintptr_t)m.in[k]) & 15) != 0) dense_wave = false;
                for (int a = 0; a < keep.rank; ++a)
                    if (m.keep_strides[k][a] % 4 != 0) dense_wave = false;
            }
        }
    }
    if (bmask == (1 << n_in) - 1) dense_lane = dense_wave = false;   // nothing streams
    m.bcast = (dense_lane || dense_wave) ? bmask : 0;
    // split the reduce range until about fused_waves_per_cu waves per CU are in flight
    // (a block is 4 waves: 4 (output, split) jobs in the wave kernels, one output group
    // in the lane kernels)
    const bool narrow_lane = dense_lane && n_out <= 128;       // several rows per wave load
    m.lanes_per_row = narrow_lane ? (int)(n_out / 4) : 0;
    const int64_t outs_per_block = lanes_over_outputs ? (dense_lane ? 256 : 64) : 4;
    const int64_t jobs = (n_out + outs_per_block - 1) / outs_per_block;
    int64_t splits = 1;
    const int64_t want_blocks = (int64_t)ctx->cu_count * ctx->fused_waves_per_cu / 4;
    if (jobs < want_blocks) {
        splits = lanes_over_outputs ? want_blocks / jobs : (want_blocks * 4) / n_out;
        const int64_t max_splits = n_red / (narrow_lane ? 16 * (64 / m.lanes_per_row) * 4 : lanes_over_outputs ? 64 : 4096);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        if (splits > 16384) splits = 16384;
    }
    m.splits = (int)splits;
    if (splits > 1) {
        void* ws = nullptr;
        int rc = bsc_workspace(ctx, (size_t)splits * n_out * sizeof(double), &ws);
        if (rc != BSC_OK) return rc;
        m.partial = (double*)ws;
        ctx->slab_rows = 0;
    }
    if (lanes_over_outputs) {
        const int64_t blocks = jobs * splits;
        if (narrow_lane) {
#define BSC_NARROW(NV)                                                                            \
    case NV:                                                                                     \
        hipLaunchKernelGGL(map_reduce_lane_narrow_f32_kernel<NV>, dim3((unsigned)blocks), dim3(256), \
                           0, ctx->stream, m);                                                   \
        break;
            switch (n_in) { BSC_NARROW(1) BSC_NARROW(2) BSC_NARROW(3) }
#undef BSC_NARROW
        } else if (dense_lane) {
            bool lin = post_op == BSC_OP_COPY && ctx->fused_map_flat;
            for (int k = 0; k < n_in; ++k) {
                const int op = m.pre_op[k];
                if (!(op == BSC_OP_COPY || op == BSC_OP_SCALE || op == BSC_OP_ABS ||
                      (op == BSC_OP_POW && (m.pre_arg[k] == 2.0 || m.pre_arg[k] == -1.0 || m.pre_arg[k] == 1.0))))
                    lin = false;
            }
#define BSC_LANE(NV)                                                                             \
    case NV:                                                                                     \
        if (lin) hipLaunchKernelGGL((map_reduce_lane_dense_f32_kernel<NV, true>), dim3((unsigned)blocks), dim3(256), \
                                    0, ctx->stream, m);                                           \
        else hipLaunchKernelGGL((map_reduce_lane_dense_f32_kernel<NV, false>), dim3((unsigned)blocks), dim3(256), \
                                0, ctx->stream, m);                                               \
        break;
            switch (n_in) { BSC_LANE(1) BSC_LANE(2) BSC_LANE(3) }
#undef BSC_LANE
        } else {
            BSC_GENERIC(map_reduce_lane_kernel, blocks);
        }
    } else {
        // the wave kernels run four (output, split) jobs per block
        const int64_t blocks = (n_out * splits + 3) / 4;
        if (dense_wave && splits == 1 && keep.rank == 1 && n_red <= 1024 && n_out >= 4096 && ctx->fused_map_flat) {
            // many short rows: persistent waves, four rows per step
            bool linear = post_op == BSC_OP_COPY;
            for (int k = 0; k < n_in; ++k) {
                const int op = m.pre_op[k];
                if (!(op == BSC_OP_COPY || op == BSC_OP_SCALE || op == BSC_OP_ABS ||
                      (op == BSC_OP_POW && (m.pre_arg[k] == 2.0 || m.pre_arg[k] == -1.0 || m.pre_arg[k] == 1.0))))
                    linear = false;
            }
            int64_t rblocks = (n_out + 15) / 16;
            const int64_t cap = (int64_t)ctx->cu_count * ctx->rows_wg_per_cu;        // (0: one step per wave, not persistent)
            if (cap > 0 && rblocks > cap) rblocks = cap;
            if (ctx->rows_dbg && n_in == 2 && linear) {
#define BSC_ROWSD(DBG) case DBG: hipLaunchKernelGGL((map_reduce_rows_f32_kernel<2, true, DBG>), dim3((unsigned)rblocks), dim3(256), 0, ctx->stream, m); break;
                switch (ctx->rows_dbg) { BSC_ROWSD(2) BSC_ROWSD(3) BSC_ROWSD(4) BSC_ROWSD(8) default: return bsc_fail(BSC_ERR_INVALID, "BSC_ROWS_DBG: 2, 3, 4 or 8"); }
#undef BSC_ROWSD
                BSC_LAUNCH_CHECK();
                return BSC_OK;
            }
#define BSC_ROWS(NV)                                                                                                  \
    case NV:                                                                                                          \
        if (linear) hipLaunchKernelGGL((map_reduce_rows_f32_kernel<NV, true>), dim3((unsigned)rblocks), dim3(256), 0, ctx->stream, m);  \
        else hipLaunchKernelGGL((map_reduce_rows_f32_kernel<NV, false>), dim3((unsigned)rblocks), dim3(256), 0, ctx->stream, m);        \
        break;
            switch (n_in) { BSC_ROWS(1) BSC_ROWS(2) BSC_ROWS(3) }
#undef BSC_ROWS
        } else if (dense_wave) {
            bool wlin = post_op == BSC_OP_COPY && ctx->fused_map_flat;
            for (int k = 0; k < n_in; ++k) {
                const int op = m.pre_op[k];
                if (!(op == BSC_OP_COPY || op == BSC_OP_SCALE || op == BSC_OP_ABS ||
                      (op == BSC_OP_POW && (m.pre_arg[k] == 2.0 || m.pre_arg[k] == -1.0 || m.pre_arg[k] == 1.0))))
                    wlin = false;
            }
#define BSC_WAVE_CASE(NV)                                                                        \
    case NV:                                                                                     \
        if (n_red <= 256 * splits)                                                               \
            hipLaunchKernelGGL((map_reduce_wave_dense_f32_kernel<NV, 1>), dim3((unsigned)blocks), \
                               dim3(256), 0, ctx->stream, m);                                    \
        else if (wlin)                                                                           \
            hipLaunchKernelGGL((map_reduce_wave_dense_f32_kernel<NV, 4, true>), dim3((unsigned)blocks), \
                               dim3(256), 0, ctx->stream, m);                                    \
        else                                                                                     \
            hipLaunchKernelGGL((map_reduce_wave_dense_f32_kernel<NV, 4>), dim3((unsigned)blocks), \
                               dim3(256), 0, ctx->stream, m);                                    \
        break;
            switch (n_in) { BSC_WAVE_CASE(1) BSC_WAVE_CASE(2) BSC_WAVE_CASE(3) }
#undef BSC_WAVE_CASE
        } else {
            BSC_GENERIC(map_reduce_wave_kernel, blocks);
        }
    }
    BSC_LAUNCH_CHECK();
    if (splits > 1) {
        const int wave_per_output = n_out < 4096;
        const unsigned blocks = (unsigned)(wave_per_output ? (n_out + 3) / 4 : (n_out + 255) / 256);
        if (dtype == BSC_F32)
            hipLaunchKernelGGL(map_reduce_finish_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream,
                               m, wave_per_output);
        else
            hipLaunchKernelGGL(map_reduce_finish_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream,
                               m, wave_per_output);
        BSC_LAUNCH_CHECK();
    }
    return BSC_OK;
}

}  // extern "C"
