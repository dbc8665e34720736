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
// D=256).  The 64 per-lane partial dot products (8 rows x 8 samples) are
// transposed and summed through a wave-private LDS region (no workgroup
// barrier), the residuals are broadcast back through LDS, and the backward
// rank-1 updates accumulate into per-lane registers acc[s][4] that live for the
// whole kernel.  Measured VALU issue costs on gfx950 (profiles/
// r01_ubench_valu_issue_rates.txt: v_permlane*_swap 2.9x, DPP 1.5x, SGPR-operand
// FMA 1.55x a plain FMA) are why the cross-lane work rides the LDS pipe and
// every FMA is all-VGPR.  Block partials go to a slab; they are summed in
// float64 in a fixed order (bitwise reproducible, no float atomics).
#include "bsc_common.h"

namespace {

constexpr int SG = 8;  // samples per pass
constexpr int PASS_BLOCK = 256;
constexpr int PASS_WAVES = PASS_BLOCK / BSC_WAVE;
constexpr int GCOLS = 256;                    // column capacity of the lane layout
constexpr int SLAB_G = SG * GCOLS;            // slab[b][d*8 + s], then Q at [SLAB_G + s]
constexpr int SLAB_STRIDE = SLAB_G + SG;      // floats per block partial

// Geometry of one wave's tile, for ROWS = 8 (2 waves/SIMD) or 4 (3 waves/SIMD).
// A lane writes its ROWS*8 partial dots as one LDS row of PSTR floats; PSTR = 4 mod 32
// keeps the 16-byte writes conflict-free, and the row blocks read by the four (eight)
// lane groups start a multiple of 32 (64) floats apart, so the 16-byte column reads are
// conflict-free as well.
template <int ROWS>
struct Geo {
    static constexpr int NVAL = ROWS * SG;        // values per lane: 64 or 32
    static constexpr int PSTR = NVAL + 4;         // 68 or 36 floats
    static constexpr int NGRP = NVAL / 4;         // lanes per value group set: 16 or 8
    static constexpr int NQ = BSC_WAVE / NGRP;    // row subsets: 4 or 8
    static constexpr int RPQ = BSC_WAVE / NQ;     // lane-rows per subset: 16 or 8
    static constexpr int WAVE_LDS = BSC_WAVE * PSTR + NVAL;  // + residual broadcast buffer
    static constexpr int OCC = ROWS == 8 ? 2 : 3;  // waves per SIMD (VGPR budget 256 / 168)
};

template <int ROWS>
struct Tile {
    float4 x[ROWS];
    float yv;
};

// After the transposing reduction lane k holds value v(k) = row*8 + sample.
template <int ROWS>
__device__ __forceinline__ int lane_value(int lane) {
    return 4 * (lane & (Geo<ROWS>::NGRP - 1)) + 2 * ((lane >> 5) & 1) + ((lane >> 4) & 1);
}

// One tile = ROWS rows starting at row0, through buffer loads: the descriptor
// (SGPRs) covers exactly the rows [row0, B), so rows past the end -- and whole
// tiles past the end -- read as zeros without touching memory.  No ragged-tail
// code path and a trip count that is the same for every wave.  Everything but
// the 16*lane byte offset is wave-uniform.
template <int ROWS, bool FULL, bool NT>
__device__ __forceinline__ void load_tile(Tile<ROWS>& t, const float* __restrict__ X,
                                          int64_t ldx, const float* __restrict__ y,
                                          int64_t row0, int64_t B, int D, int lane) {
    const int64_t rem = B - row0;  // rows left; <= 0 for a tile past the end
    uint64_t xbytes = 0, ybytes = 0;
    if (rem > 0) {
        xbytes = ((uint64_t)(rem - 1) * (uint64_t)ldx + (uint64_t)D) * 4u;
        ybytes = (uint64_t)rem * 4u;
    }
    const unsigned xrec = xbytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)xbytes;
    const unsigned yrec = ybytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)ybytes;
    const int64_t safe0 = rem > 0 ? row0 : 0;  // keep the base pointer inside the allocation
    auto xs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + safe0 * ldx), 0, xrec, 0x00020000);
    auto ys = __builtin_amdgcn_make_buffer_rsrc((void*)(y + safe0), 0, yrec, 0x00020000);
    const int lane_off = 16 * lane;
    const int row_bytes = (int)(ldx * 4);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        // NT: non-temporal (aux = 2) -- X is read exactly once per pass
        auto v = __builtin_amdgcn_raw_buffer_load_b128(xs, lane_off, r * row_bytes, NT ? 2 : 0);
        float4 f = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]),
                               __uint_as_float(v[2]), __uint_as_float(v[3]));
        if (!FULL && 4 * lane >= D) f = make_float4(0.f, 0.f, 0.f, 0.f);  // next row's bytes
        t.x[r] = f;
    }
    t.yv = __uint_as_float(
        __builtin_amdgcn_raw_buffer_load_b32(ys, 4 * (lane_value<ROWS>(lane) >> 3), 0, 0));
}

__device__ __forceinline__ float swap_add32(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false,
                                              false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float swap_add16(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false,
                                              false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    float v = a.x * b.x;
    v = fmaf(a.y, b.y, v);
    v = fmaf(a.z, b.z, v);
    return fmaf(a.w, b.w, v);
}

__device__ __forceinline__ void axpy4(float4& acc, float c, const float4& x) {
    acc.x = fmaf(c, x.x, acc.x);
    acc.y = fmaf(c, x.y, acc.y);
    acc.z = fmaf(c, x.z, acc.z);
    acc.w = fmaf(c, x.w, acc.w);
}

// The same four FMAs as two packed ones (v_pk_fma_f32, the scalar broadcast through op_sel):
// a packed FMA costs the MFMA pipe of the other wave what ONE scalar FMA does
// (profiles/r01_ubench_mfma_valu_mix.txt), and the results are the same bits.
typedef float pass_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void axpy4_pk(float4& acc, float c, const float4& x) {
    const pass_f32x2 c2 = {c, c};
    const pass_f32x2 lo = __builtin_elementwise_fma(c2, pass_f32x2{x.x, x.y}, pass_f32x2{acc.x, acc.y});
    const pass_f32x2 hi = __builtin_elementwise_fma(c2, pass_f32x2{x.z, x.w}, pass_f32x2{acc.z, acc.w});
    acc.x = lo[0]; acc.y = lo[1]; acc.z = hi[0]; acc.w = hi[1];
}

// Forward + backward for one tile; `wl` is this wave's LDS region.
template <int ROWS>
__device__ __forceinline__ void compute_tile(const Tile<ROWS>& t, const float4 (&w)[SG],
                                             float4 (&acc)[SG], float& qacc, float* wl,
                                             int lane) {
    using G = Geo<ROWS>;
    // 1. per-lane partial dots, row by row, into this lane's row of the buffer
    float* mine = wl + lane * G::PSTR;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float4 lo, hi;
        lo.x = dot4(t.x[r], w[0]); lo.y = dot4(t.x[r], w[1]);
        lo.z = dot4(t.x[r], w[2]); lo.w = dot4(t.x[r], w[3]);
        hi.x = dot4(t.x[r], w[4]); hi.y = dot4(t.x[r], w[5]);
        hi.z = dot4(t.x[r], w[6]); hi.w = dot4(t.x[r], w[7]);
        *reinterpret_cast<float4*>(mine + r * SG) = lo;
        *reinterpret_cast<float4*>(mine + r * SG + 4) = hi;
    }
    wave_lds_sync();
    // 2. lane k sums values 4g..4g+3 (g = k % NGRP) over the lane-rows q*RPQ .. q*RPQ+RPQ-1
    const int g = lane & (G::NGRP - 1), q = lane / G::NGRP;
    const float* col = wl + q * G::RPQ * G::PSTR + 4 * g;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int RB = ROWS == 8 ? 8 : 4;  // reads in flight before the first add
#pragma unroll
    for (int part = 0; part < G::RPQ / RB; ++part) {
        float4 v[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i)
            v[i] = *reinterpret_cast<const float4*>(col + (RB * part + i) * G::PSTR);
#pragma unroll
        for (int h = RB / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int i = 0; i < h; ++i) {
                v[i].x += v[i + h].x; v[i].y += v[i + h].y;
                v[i].z += v[i + h].z; v[i].w += v[i + h].w;
            }
        }
        s4.x += v[0].x; s4.y += v[0].y; s4.z += v[0].z; s4.w += v[0].w;
    }
    // 3. fold the row subsets (lane bits 4,5; and bit 3 when ROWS == 4): lane k ends
    //    with value lane_value(k)
    float t0 = swap_add32(s4.x, s4.z);
    float t1 = swap_add32(s4.y, s4.w);
    float dot = swap_add16(t0, t1);
    if (ROWS == 4) dot += dpp_f32<DPP_ROW_ROR8>(dot);  // lanes k and k^8 hold the same value
    float resid = t.yv - dot;
    qacc = fmaf(resid, resid, qacc);
    float* rb = wl + BSC_WAVE * G::PSTR;
    rb[lane_value<ROWS>(lane)] = resid;
    wave_lds_sync();
    // 4. backward: acc[s] += resid(r,s) * x[r]; residuals arrive by LDS broadcast
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        // keep at most four rows of broadcast reads in flight (register budget)
        if (r == 4) asm volatile("" ::: "memory");
        float4 c0 = *reinterpret_cast<const float4*>(rb + r * SG);
        float4 c1 = *reinterpret_cast<const float4*>(rb + r * SG + 4);
        axpy4(acc[0], c0.x, t.x[r]); axpy4(acc[1], c0.y, t.x[r]);
        axpy4(acc[2], c0.z, t.x[r]); axpy4(acc[3], c0.w, t.x[r]);
        axpy4(acc[4], c1.x, t.x[r]); axpy4(acc[5], c1.y, t.x[r]);
        axpy4(acc[6], c1.z, t.x[r]); axpy4(acc[7], c1.w, t.x[r]);
    }
}

// FULL: D == 256, every lane owns four live columns.  ROWS: tile height.
// n_iter: tiles per wave (same for every wave; tiles past the end read zeros).
template <bool FULL, int ROWS, bool NT>
__global__ __launch_bounds__(PASS_BLOCK, Geo<ROWS>::OCC) void blr_pass_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y, int64_t B, int D,
    const float* __restrict__ W, int S, float* __restrict__ slab, int n_iter) {
    using G = Geo<ROWS>;
    // one array: wave-private regions during the loop, [wave][SLAB_STRIDE] in the epilogue
    constexpr int LDS_FLOATS = PASS_WAVES * (G::WAVE_LDS > SLAB_STRIDE ? G::WAVE_LDS : SLAB_STRIDE);
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wave-uniform by construction; telling the compiler keeps tile indices and
    // buffer descriptors in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* wl = lds + wave * G::WAVE_LDS;

    float4 w[SG], acc[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s) {
        w[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * lane < D && s < S)
            w[s] = *reinterpret_cast<const float4*>(W + (int64_t)s * D + 4 * lane);
        acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float qacc = 0.f;

    // This wave owns tiles first, first+stride, ... (n_iter of them).  Every
    // prefetch is unconditional -- a conditional one makes the compiler wait with
    // vmcnt(0), i.e. no prefetch at all -- and out-of-range tiles cost no traffic.
    const int64_t stride = (int64_t)gridDim.x * PASS_WAVES;
    int64_t tile = (int64_t)blockIdx.x * PASS_WAVES + wave;
    Tile<ROWS> ta, tb;
    load_tile<ROWS, FULL, NT>(ta, X, ldx, y, tile * ROWS, B, D, lane);
    for (int k = 0; k + 1 < n_iter; k += 2) {
        load_tile<ROWS, FULL, NT>(tb, X, ldx, y, (tile + stride) * ROWS, B, D, lane);
        compute_tile<ROWS>(ta, w, acc, qacc, wl, lane);
        load_tile<ROWS, FULL, NT>(ta, X, ldx, y, (tile + 2 * stride) * ROWS, B, D, lane);
        compute_tile<ROWS>(tb, w, acc, qacc, wl, lane);
        tile += 2 * stride;
    }
    if (n_iter & 1) compute_tile<ROWS>(ta, w, acc, qacc, wl, lane);

    // block reduction through LDS, fixed order over waves
    __syncthreads();  // every wave is done with its private region
    float* ep = lds + wave * SLAB_STRIDE;
#pragma unroll
    for (int s = 0; s < SG; ++s)
        *reinterpret_cast<float4*>(ep + s * GCOLS + 4 * lane) = acc[s];
    // qacc of lane k belongs to sample lane_value(k)&7; fold the tile rows (lane bits 1..)
    // -- with ROWS == 4 lanes k and k^8 carry the same residuals, hence the 0.5
    float qv = qacc;
    qv += __shfl_xor(qv, 2);
    qv += __shfl_xor(qv, 4);
    qv += __shfl_xor(qv, 8);
    if (ROWS == 4) qv *= 0.5f;
    if ((lane & 14) == 0) ep[SLAB_G + (lane_value<ROWS>(lane) & 7)] = qv;
    __syncthreads();
    float* out = slab + (int64_t)blockIdx.x * SLAB_STRIDE;
    for (int i = tid; i < SLAB_STRIDE; i += PASS_BLOCK) {
        // slab order is [d][s] so that a finisher reads 8 columns x 8 samples as one 256-B run
        const int src = i < SLAB_G ? (i & 7) * GCOLS + (i >> 3) : i;
        float v = lds[src];
#pragma unroll
        for (int k = 1; k < PASS_WAVES; ++k) v += lds[k * SLAB_STRIDE + src];
        out[i] = v;
    }
}

// ---- variant with the forward pass on the MFMA pipe (D == 256) -------------------
//
// The kernel above keeps ~83 % of the VALU issue slots busy behind the HBM stream,
// so it slows down on boxes whose sustained shader clock is lower (measured: pure-read
// ceiling equal or higher, pass 10-20 % slower).  fp32 MFMA has the same peak as fp32
// VALU on this part, but it is a SEPARATE pipe: putting the forward x.w products there
// halves the VALU work and removes the transpose-reduce altogether.
//
// A wave owns 16-row tiles.  The tile is written to the wave's LDS region row-major
// (stride 260 floats) and read back twice:
//   forward   A operand of v_mfma_f32_16x16x4_f32: lane (i = l&15, kq = l>>4) reads
//             X[row i][16 j + 4 kq .. +3] (one ds_read_b128 feeds 4 MFMAs); B operand
//             = W[sample l&15][same columns], 64 registers loaded once; the result
//             D[row 4 kq + reg][sample l&15] is the 16 x 8 block of dot products
//             (columns 8..15 of the MFMA are idle).
//   backward  lane l reads X[row][4l..4l+3] again (conflict-free) and the residuals by
//             LDS broadcast: acc[s] += resid(row, s) * x, as above.
// Only the prefetched NEXT tile lives in registers (64 VGPRs); 2 waves per SIMD.
constexpr int MT_ROWS = 16;
constexpr int MT_RS = GCOLS + 4;                         // LDS row stride (floats)
constexpr int MT_WAVE_LDS = MT_ROWS * MT_RS + MT_ROWS * SG;   // tile + residual buffer

typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));

struct MTile {
    float4 x[MT_ROWS];
    float4 yv;     // y[row0 + 4 kq .. + 3]: the rows of this lane's MFMA result registers
};

template <int AUX>
__device__ __forceinline__ void load_mtile_policy(MTile& t, const float* __restrict__ X, int64_t ldx,
                                                  const float* __restrict__ y, int64_t row0, int64_t B,
                                                  int lane) {
    const int64_t rem = row0 < 0 ? 0 : B - row0;   // a window before row 0 is as empty as one past B
    uint64_t xbytes = 0, ybytes = 0;
    if (rem > 0) {
        xbytes = ((uint64_t)(rem - 1) * (uint64_t)ldx + (uint64_t)GCOLS) * 4u;
        ybytes = (uint64_t)rem * 4u;
    }
    const unsigned xrec = xbytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)xbytes;
    const unsigned yrec = ybytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)ybytes;
    const int64_t safe0 = rem > 0 ? row0 : 0;
    auto xs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + safe0 * ldx), 0, xrec, 0x00020000);
    auto ys = __builtin_amdgcn_make_buffer_rsrc((void*)(y + safe0), 0, yrec, 0x00020000);
    const int lane_off = 16 * lane;
    const int row_bytes = (int)(ldx * 4);
#pragma unroll
    for (int r = 0; r < MT_ROWS; ++r) {
        auto v = __builtin_amdgcn_raw_buffer_load_b128(xs, lane_off, r * row_bytes, AUX);
        t.x[r] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                             __uint_as_float(v[3]));
    }
    auto v = __builtin_amdgcn_raw_buffer_load_b128(ys, 16 * (lane >> 4), 0, 0);
    t.yv = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                       __uint_as_float(v[3]));
}

// One iteration of a wave: the tile in `t` goes to LDS and `t` takes the tile at next_row0 with
// the cache policy STREAM (non-temporal) or not (allocating); then forward, residuals, backward.
template <int AUX, bool PK>
__device__ __forceinline__ void mfma_tile_step(MTile& t, float* __restrict__ tl, float* __restrict__ rb,
                                               const float (&wreg)[GCOLS / 4], float4 (&acc)[SG],
                                               float& qacc, const float* __restrict__ X, int64_t ldx,
                                               const float* __restrict__ y, int64_t next_row0,
                                               int64_t B, int lane) {
    const int i16 = lane & 15, kq = lane >> 4;
    const bool live = i16 < SG;                 // lanes whose MFMA column is a sample
    // the tile to LDS, then its registers take the next tile (unconditional prefetch:
    // tiles outside the mini-batch read zeros without touching memory)
#pragma unroll
    for (int r = 0; r < MT_ROWS; ++r)
        *reinterpret_cast<float4*>(tl + r * MT_RS + 4 * lane) = t.x[r];
    const float4 yv = t.yv;
    load_mtile_policy<AUX>(t, X, ldx, y, next_row0, B, lane);
    wave_lds_sync();

    // forward on the MFMA pipe; two accumulators so that no MFMA waits on its predecessor
    mfma_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    const float* arow = tl + i16 * MT_RS + 64 * kq;
#pragma unroll
    for (int j = 0; j < GCOLS / 16; j += 2) {
        const float4 a0 = *reinterpret_cast<const float4*>(arow + 4 * j);
        const float4 a1 = *reinterpret_cast<const float4*>(arow + 4 * j + 4);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wreg[4 * j + 0], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wreg[4 * j + 4], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wreg[4 * j + 1], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wreg[4 * j + 5], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wreg[4 * j + 2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wreg[4 * j + 6], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wreg[4 * j + 3], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wreg[4 * j + 7], d1, 0, 0, 0);
    }
    // result register reg of lane (i16, kq) = dot(row 4 kq + reg, sample i16)
    if (live) {
        const float r0 = yv.x - (d0[0] + d1[0]), r1 = yv.y - (d0[1] + d1[1]);
        const float r2 = yv.z - (d0[2] + d1[2]), r3 = yv.w - (d0[3] + d1[3]);
        qacc = fmaf(r0, r0, qacc); qacc = fmaf(r1, r1, qacc);
        qacc = fmaf(r2, r2, qacc); qacc = fmaf(r3, r3, qacc);
        float* dst = rb + (4 * kq) * SG + i16;
        dst[0] = r0; dst[SG] = r1; dst[2 * SG] = r2; dst[3 * SG] = r3;
    }
    wave_lds_sync();

    // backward on the VALU: rows from LDS (row-major again), residuals by broadcast
#pragma unroll
    for (int r = 0; r < MT_ROWS; ++r) {
        if ((r & 3) == 0 && r) asm volatile("" ::: "memory");   // four rows of reads in flight
        const float4 x4 = *reinterpret_cast<const float4*>(tl + r * MT_RS + 4 * lane);
        const float4 c0 = *reinterpret_cast<const float4*>(rb + r * SG);
        const float4 c1 = *reinterpret_cast<const float4*>(rb + r * SG + 4);
        if (PK) {
            axpy4_pk(acc[0], c0.x, x4); axpy4_pk(acc[1], c0.y, x4);
            axpy4_pk(acc[2], c0.z, x4); axpy4_pk(acc[3], c0.w, x4);
            axpy4_pk(acc[4], c1.x, x4); axpy4_pk(acc[5], c1.y, x4);
            axpy4_pk(acc[6], c1.z, x4); axpy4_pk(acc[7], c1.w, x4);
        } else {
            axpy4(acc[0], c0.x, x4); axpy4(acc[1], c0.y, x4);
            axpy4(acc[2], c0.z, x4); axpy4(acc[3], c0.w, x4);
            axpy4(acc[4], c1.x, x4); axpy4(acc[5], c1.y, x4);
            axpy4(acc[6], c1.z, x4); axpy4(acc[7], c1.w, x4);
        }
    }
    wave_lds_sync();   // the next iteration overwrites the tile
}

// Sweep order.  Iteration k of every wave reads one contiguous WINDOW of gridDim.x * 4 tiles (33 MB at
// 1M x 256 on 505 workgroups); `rev` walks the windows from the end of the mini-batch to its start.
// The last `keep` windows of a sweep are read with the allocating cache policy, everything before them
// non-temporal: they are then still in the 256 MiB Infinity Cache when the NEXT sweep over the same
// mini-batch -- run in the other direction -- starts with exactly those windows.  The policy changes
// once per sweep, so the loop is split in two straight-line bodies rather than branching per tile
// (a per-tile branch around the loads cost 6 us of the 164).
template <bool NT, bool PK>
__global__ __launch_bounds__(PASS_BLOCK, 2) void blr_pass_mfma_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y, int64_t B,
    const float* __restrict__ W, int S, float* __restrict__ slab, int n_iter, int rev, int keep) {
    constexpr int LDS_FLOATS = PASS_WAVES * (MT_WAVE_LDS > SLAB_STRIDE ? MT_WAVE_LDS : SLAB_STRIDE);
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    float* tl = lds + wave * MT_WAVE_LDS;      // this wave's tile
    float* rb = tl + MT_ROWS * MT_RS;          // residuals [row][sample]

    // B operand: W[sample i16][64 kq + 4 j + c]; samples >= S (and MFMA columns 8..15) are zero.
    // Lane group kq contracts columns 64 kq .. 64 kq + 63, so the four 16-byte A reads of a row
    // lie 256 B apart on the same banks and each ds_read_b128 lane group (all 16 rows once,
    // two different kq) is conflict-free; interleaved columns (16 j + 4 kq) gave every lane
    // group a 2-way conflict (SQ_LDS_BANK_CONFLICT 18 % of the LDS cycles).
    float wreg[GCOLS / 4];
#pragma unroll
    for (int j = 0; j < GCOLS / 16; ++j) {
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i16 < S) w4 = *reinterpret_cast<const float4*>(W + (int64_t)i16 * GCOLS + 64 * kq + 4 * j);
        wreg[4 * j + 0] = w4.x; wreg[4 * j + 1] = w4.y;
        wreg[4 * j + 2] = w4.z; wreg[4 * j + 3] = w4.w;
    }
    float4 acc[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    float qacc = 0.f;
    const bool live = i16 < SG;

    const int64_t stride0 = (int64_t)gridDim.x * PASS_WAVES;
    const int64_t stride = rev ? -stride0 : stride0;
    int64_t tile = (int64_t)blockIdx.x * PASS_WAVES + wave + (rev ? (int64_t)(n_iter - 1) * stride0 : 0);
    if (!NT) keep = n_iter;                    // every load allocating
    const int n_stream = n_iter - keep > 0 ? n_iter - keep : 0;   // windows read non-temporal
    MTile t;
    if (n_stream > 0) load_mtile_policy<2>(t, X, ldx, y, n_iter > 0 ? tile * MT_ROWS : B, B, lane);
    else load_mtile_policy<0>(t, X, ldx, y, n_iter > 0 ? tile * MT_ROWS : B, B, lane);
    int k = 0;
    for (; k + 1 < n_stream; ++k) {            // this window and the next one streamed
        tile += stride;
        mfma_tile_step<2, PK>(t, tl, rb, wreg, acc, qacc, X, ldx, y, tile * MT_ROWS, B, lane);
    }
    for (; k < n_iter; ++k) {                  // the next window is kept (or is the empty one after the last)
        tile += stride;
        mfma_tile_step<0, PK>(t, tl, rb, wreg, acc, qacc, X, ldx, y,
                                  k + 1 < n_iter ? tile * MT_ROWS : B, B, lane);
    }

    // block reduction through LDS, fixed order over waves (same slab layout as above)
    __syncthreads();
    float* ep = lds + wave * SLAB_STRIDE;
#pragma unroll
    for (int s = 0; s < SG; ++s)
        *reinterpret_cast<float4*>(ep + s * GCOLS + 4 * lane) = acc[s];
    float qv = live ? qacc : 0.f;          // lane (i16 < 8, kq): rows 4 kq .. of sample i16
    qv += __shfl_xor(qv, 16);
    qv += __shfl_xor(qv, 32);
    if (lane < SG) ep[SLAB_G + lane] = qv;
    __syncthreads();
    float* out = slab + (int64_t)blockIdx.x * SLAB_STRIDE;
    for (int i = tid; i < SLAB_STRIDE; i += PASS_BLOCK) {
        const int src = i < SLAB_G ? (i & 7) * GCOLS + (i >> 3) : i;
        float v = lds[src];
#pragma unroll
        for (int k = 1; k < PASS_WAVES; ++k) v += lds[k * SLAB_STRIDE + src];
        out[i] = v;
    }
}

// ---- the same pass with the tile brought by LDS-DMA: the default since the end of round 3 (BSC_BLR_DMA=0: the kernel
// above).  166 -> 161 us per 1M x 256 pass in alternating runs on one box (tools/ab_blr_dma.sh) -----------------------
// buffer_load ... lds writes a row's 1 KiB straight into the wave's LDS tile: no 64 staging registers, no sixteen
// ds_write_b128 a tile, and on this part a pure read by LDS-DMA reaches 6.9 TB/s where loads into registers reach 6.4
// (bsc_hbm_read_probe).  One tile buffer per wave as before: the backward reads ALL sixteen rows into registers first
// (the registers the prefetched tile used to occupy), which frees the buffer, issues the next tile's sixteen DMAs, and
// only then does its rank-1 updates from the registers -- so the DMAs have the whole backward and the other wave's turn
// to land, and no LDS read of the tile follows a DMA in flight (the compiler answers such a read with vmcnt(0): the
// residuals, which ARE read during the backward, come by inline asm with their own lgkmcnt waits).
// VROW: the row's offset rides in the VECTOR offset (sixteen loop-invariant registers) instead of the scalar one -- for
// the stealing loop of blr_pass_q_kernel, where the compiler otherwise parks the fifteen scalar products in vector
// registers and wraps every DMA in a waterfall loop.
template <int AUX, bool VROW = false>
__device__ __forceinline__ void dma_mtile(float* __restrict__ tl, const float* __restrict__ X, int64_t ldx,
                                          const float* __restrict__ y, int64_t row0, int64_t B, int lane, float4& yv) {
    const int64_t rem = row0 < 0 ? 0 : B - row0;
    uint64_t xbytes = 0, ybytes = 0;
    if (rem > 0) {
        xbytes = ((uint64_t)(rem - 1) * (uint64_t)ldx + (uint64_t)GCOLS) * 4u;
        ybytes = (uint64_t)rem * 4u;
    }
    const unsigned xrec = xbytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)xbytes;
    const unsigned yrec = ybytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)ybytes;
    const int64_t safe0 = rem > 0 ? row0 : 0;
    auto xs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + safe0 * ldx), 0, xrec, 0x00020000);
    auto ys = __builtin_amdgcn_make_buffer_rsrc((void*)(y + safe0), 0, yrec, 0x00020000);
    const int row_bytes = (int)(ldx * 4);
#pragma unroll
    for (int r = 0; r < MT_ROWS; ++r) {
        if (VROW) __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (bsc_lds_ptr)(tl + r * MT_RS), 16, 16 * lane + r * row_bytes, 0, 0, AUX);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (bsc_lds_ptr)(tl + r * MT_RS), 16, 16 * lane, r * row_bytes, 0, AUX);
    }
    auto v = __builtin_amdgcn_raw_buffer_load_b128(ys, 16 * (lane >> 4), 0, 0);
    yv = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

template <int AUX>
__device__ __forceinline__ void dma_tile_step(float4& yv_cur, float* __restrict__ tl, float* __restrict__ rb,
                                              const float (&wreg)[GCOLS / 4], float4 (&acc)[SG], float& qacc,
                                              const float* __restrict__ X, int64_t ldx, const float* __restrict__ y,
                                              int64_t next_row0, int64_t B, int lane) {
    const int i16 = lane & 15, kq = lane >> 4;
    const bool live = i16 < SG;
    // the tile's DMAs (issued a step ago) and its y have landed
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    asm volatile("" ::: "memory");
    const float4 yv = yv_cur;
    mfma_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    const float* arow = tl + i16 * MT_RS + 64 * kq;
#pragma unroll
    for (int j = 0; j < GCOLS / 16; j += 2) {
        const float4 a0 = *reinterpret_cast<const float4*>(arow + 4 * j);
        const float4 a1 = *reinterpret_cast<const float4*>(arow + 4 * j + 4);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wreg[4 * j + 0], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wreg[4 * j + 4], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wreg[4 * j + 1], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wreg[4 * j + 5], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wreg[4 * j + 2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wreg[4 * j + 6], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wreg[4 * j + 3], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wreg[4 * j + 7], d1, 0, 0, 0);
    }
    if (live) {
        const float r0 = yv.x - (d0[0] + d1[0]), r1 = yv.y - (d0[1] + d1[1]);
        const float r2 = yv.z - (d0[2] + d1[2]), r3 = yv.w - (d0[3] + d1[3]);
        qacc = fmaf(r0, r0, qacc); qacc = fmaf(r1, r1, qacc);
        qacc = fmaf(r2, r2, qacc); qacc = fmaf(r3, r3, qacc);
        float* dst = rb + (4 * kq) * SG + i16;
        dst[0] = r0; dst[SG] = r1; dst[2 * SG] = r2; dst[3 * SG] = r3;
    }
    // every row into registers, then the buffer belongs to the next tile
    float4 x4[MT_ROWS];
#pragma unroll
    for (int r = 0; r < MT_ROWS; ++r) x4[r] = *reinterpret_cast<const float4*>(tl + r * MT_RS + 4 * lane);
    wave_lds_sync();            // (the residual writes before the asm reads below; the row reads before the DMAs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dma_mtile<AUX>(tl, X, ldx, y, next_row0, B, lane, yv_cur);
    // backward from the registers; residuals by LDS broadcast, read by asm (see the header)
    const unsigned rb_addr = (unsigned)(uintptr_t)(bsc_lds_ptr)rb;
#define BSC_BLR_GROUP(G4)                                                                                          \
    {                                                                                                             \
        mfma_f32x4 c0[4], c1[4];                                                                                  \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c0[0]) : "v"(rb_addr), "n"((4 * G4 + 0) * 32) : "memory");      \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c1[0]) : "v"(rb_addr), "n"((4 * G4 + 0) * 32 + 16) : "memory"); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c0[1]) : "v"(rb_addr), "n"((4 * G4 + 1) * 32) : "memory");      \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c1[1]) : "v"(rb_addr), "n"((4 * G4 + 1) * 32 + 16) : "memory"); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c0[2]) : "v"(rb_addr), "n"((4 * G4 + 2) * 32) : "memory");      \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c1[2]) : "v"(rb_addr), "n"((4 * G4 + 2) * 32 + 16) : "memory"); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c0[3]) : "v"(rb_addr), "n"((4 * G4 + 3) * 32) : "memory");      \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c1[3]) : "v"(rb_addr), "n"((4 * G4 + 3) * 32 + 16) : "memory"); \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                       \
                     : "+v"(c0[0]), "+v"(c1[0]), "+v"(c0[1]), "+v"(c1[1]), "+v"(c0[2]), "+v"(c1[2]), "+v"(c0[3]), "+v"(c1[3])); \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                           \
            const float4 xr = x4[4 * G4 + k];                                                                     \
            axpy4_pk(acc[0], c0[k][0], xr); axpy4_pk(acc[1], c0[k][1], xr);                                       \
            axpy4_pk(acc[2], c0[k][2], xr); axpy4_pk(acc[3], c0[k][3], xr);                                       \
            axpy4_pk(acc[4], c1[k][0], xr); axpy4_pk(acc[5], c1[k][1], xr);                                       \
            axpy4_pk(acc[6], c1[k][2], xr); axpy4_pk(acc[7], c1[k][3], xr);                                       \
        }                                                                                                         \
    }
    BSC_BLR_GROUP(0) BSC_BLR_GROUP(1) BSC_BLR_GROUP(2) BSC_BLR_GROUP(3)
#undef BSC_BLR_GROUP
    // (the asm reads of the residuals are done: the next step's forward may overwrite rb after its own barrier)
    wave_lds_sync();
}

template <bool NT>
__global__ __launch_bounds__(PASS_BLOCK, 2) void blr_pass_dma_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y, int64_t B,
    const float* __restrict__ W, int S, float* __restrict__ slab, int n_iter, int rev, int keep) {
    constexpr int LDS_FLOATS = PASS_WAVES * (MT_WAVE_LDS > SLAB_STRIDE ? MT_WAVE_LDS : SLAB_STRIDE);
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    float* tl = lds + wave * MT_WAVE_LDS;
    float* rb = tl + MT_ROWS * MT_RS;
    float wreg[GCOLS / 4];
#pragma unroll
    for (int j = 0; j < GCOLS / 16; ++j) {
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i16 < S) w4 = *reinterpret_cast<const float4*>(W + (int64_t)i16 * GCOLS + 64 * kq + 4 * j);
        wreg[4 * j + 0] = w4.x; wreg[4 * j + 1] = w4.y;
        wreg[4 * j + 2] = w4.z; wreg[4 * j + 3] = w4.w;
    }
    float4 acc[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    float qacc = 0.f;
    const bool live = i16 < SG;
    const int64_t stride0 = (int64_t)gridDim.x * PASS_WAVES;
    const int64_t stride = rev ? -stride0 : stride0;
    int64_t tile = (int64_t)blockIdx.x * PASS_WAVES + wave + (rev ? (int64_t)(n_iter - 1) * stride0 : 0);
    if (!NT) keep = n_iter;
    const int n_stream = n_iter - keep > 0 ? n_iter - keep : 0;
    float4 yv;
    if (n_stream > 0) dma_mtile<2>(tl, X, ldx, y, n_iter > 0 ? tile * MT_ROWS : B, B, lane, yv);
    else dma_mtile<0>(tl, X, ldx, y, n_iter > 0 ? tile * MT_ROWS : B, B, lane, yv);
    int k = 0;
    for (; k + 1 < n_stream; ++k) {
        tile += stride;
        dma_tile_step<2>(yv, tl, rb, wreg, acc, qacc, X, ldx, y, tile * MT_ROWS, B, lane);
    }
    for (; k < n_iter; ++k) {
        tile += stride;
        dma_tile_step<0>(yv, tl, rb, wreg, acc, qacc, X, ldx, y, k + 1 < n_iter ? tile * MT_ROWS : B, B, lane);
    }
    // no LDS-DMA of this wave may still be in flight when the tile region is reused for the block reduction
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    __syncthreads();
    float* ep = lds + wave * SLAB_STRIDE;
#pragma unroll
    for (int s = 0; s < SG; ++s)
        *reinterpret_cast<float4*>(ep + s * GCOLS + 4 * lane) = acc[s];
    float qv = live ? qacc : 0.f;
    qv += __shfl_xor(qv, 16);
    qv += __shfl_xor(qv, 32);
    if (lane < SG) ep[SLAB_G + lane] = qv;
    __syncthreads();
    float* out = slab + (int64_t)blockIdx.x * SLAB_STRIDE;
    for (int i = tid; i < SLAB_STRIDE; i += PASS_BLOCK) {
        const int src = i < SLAB_G ? (i & 7) * GCOLS + (i >> 3) : i;
        float v = lds[src];
#pragma unroll
        for (int kk = 1; kk < PASS_WAVES; ++kk) v += lds[kk * SLAB_STRIDE + src];
        out[i] = v;
    }
}

// The finish kernel's arguments (blr_fused_update_kernel further down; defined here because the round-4 pass kernel can
// carry them: FoldArgs).
struct FusedArgs {
    const float* slab;     // block partials of the pass kernel, or nullptr
    int n_slab;            // slab rows
    const double* stats;   // [Q (S) | G (S*D)] when slab == nullptr
    const double* lam_in;
    double* lam_out;
    double* m1;
    double* m2;
    const double* eps;
    const float* W;
    const double* xi;
    double* eps_next;      // nullptr: no next draw
    int eps_next_ready;    // 1: eps_next already holds the noise of next_step (bsc_blr_noise)
    float* W_next;
    double* xi_next;
    double* elbo;
    double* grad;
    int D, S;
    // the log-joint per draw as a member of the family (bsc_blr_fused_update_general):
    //   f(w, xi; Q) = c0 + c_xi xi + e^{-xi} (-s_q Q / 2 - k_w |w|^2 / 2 - beta)
    double c0, c_xi, s_q, k_w, beta;
    double lr, beta1, beta2, adam_eps, corr1, corr2;
    uint64_t seed;
    uint32_t next_step;
};

// ---- the finish folded into the pass's tail (round 4, option blr_fold) ---------------------------------------------------
// VERDICT r2 #6 / r3 #2(b).  After its slab row a workgroup takes an arrival ticket; the LAST `n_roles` arrivals do
// what the finish kernel's workgroups do (mode 1: role r = that kernel's block r -- slab -> float64 -> ELBO, gradient,
// Adam, next draws; mode 2: the float64 statistics [Q | G] for the all-reduce of the N > 1 structure), once every row
// has arrived.  One launch per update instead of two: the pass -> finish boundary (~1.7 us) and the finish kernel's
// own launch and small-operand round trips leave the critical path.
//
// Visibility across CUs and XCDs (MI355X_MICROARCH.md, "inter-workgroup visibility"): the slab row is stored
// write-through (sc1: relaxed agent-scope atomic stores), every storing wave drains (s_waitcnt vmcnt(0)), the
// workgroup's barrier, then ONE lane adds to the arrival counter (agent scope).  A role workgroup polls that counter
// with sc1 loads (one lane, s_sleep, bounded), barrier, and reads the slab with sc1 loads only (L1 bypassed; no XCD's L2
// can hold a slab line of this launch before the rows are complete: nothing reads the slab earlier, and L1 / L2 start
// a launch invalidated).  The arithmetic of a role is fixed, whichever workgroup performs it: reproducible.  The last
// role to finish zeroes the counters for the next launch.  The role workgroups spin only for workgroups that are
// resident (the grid never exceeds two workgroups per CU).
struct FoldArgs {
    FusedArgs a;           // mode 1
    unsigned* counters;    // [0] arrivals, [1] roles done (zero between launches)
    double* Q;             // mode 2: Q[s_base + s], G[(s_base + s) * D + d]
    double* G;
    int mode;              // 0 = no fold (the slab is the kernel's result)
    int s_base, S_total;
};
// `wait()` is called by every thread of the workgroup right before the first access to the slab: the folded finish
// waits there for the last partial (everything a role can do without the slab happens before it).
template <int BLOCK, bool COH, typename Wait>
__device__ void fused_update_role(const FusedArgs& a, int role, Wait wait);
template <int BLOCK, bool COH, typename Wait>
__device__ void slab_stats_role(const float* __restrict__ slab, int n_blocks, int D, int S, int s_base,
                                double* __restrict__ Q, double* __restrict__ G, int role, Wait wait);

// ---- round 4: BOTH contractions on v_mfma_f32_4x4x1_16B_f32, the tile by LDS-DMA (D == 256, S <= 8) -----------------
//
// The counters of blr_pass_dma_kernel (profiles/r03_pmc_blr_pass_dma.txt) say what holds it at 0.80 of the peak: a
// 16-row tile costs a SIMD 64 v_mfma_f32_16x16x4 of 32 cycles (half of each idle: eight draws in sixteen columns) plus
// ~310 vector instructions at ~4.6 matrix-pipe cycles apiece (fp32 MFMA and VALU do not overlap on a SIMD:
// profiles/r01_ubench_mfma_valu_mix.txt) -- ~3 500 issue cycles a tile, ~0.75 of the time HBM takes to deliver it.
// v_mfma_f32_4x4x1 is sixteen independent 4 x 4 outer products (K = 1) in 8 cycles and has no idle half:
//
//   forward   block b = lane / 4 = (k8 = lane >> 3, sb = (lane >> 2) & 1) contracts the 32 columns
//             {128 h + 16 k8 + c : h < 2, c < 16} for the draws 4 sb .. 4 sb + 3: A = x[4 g + lane % 4][col] (the
//             rows of row group g, read from the LDS tile: eight ds_read_b128 a group, conflict-free because the four
//             k8 of a ds_read_b128 lane group start 16 floats apart), B = w[4 sb + lane % 4][col] (32 registers, loaded
//             once), and register i of accumulator g is the partial dot product (row 4 g + i, draw lane % 8) over
//             the lane's columns.  The eight column blocks are folded by two halving swaps (v_permlane32_swap,
//             v_permlane16_swap: each also halves the values a lane carries, 16 -> 8 -> 4) and one DPP rotation:
//             28 vector instructions a tile, after which lane (kq = lane >> 4, draw = lane % 8) holds the dot
//             products of rows 4 kq .. 4 kq + 3 -- the rows whose y it loaded.  128 MFMAs of 8 cycles.
//   backward  as blr_pass_mx_kernel: A = r[n][4 sb + lane % 4] (LDS broadcast), B = x[n][4 lane + q], register i of
//             accumulator (sb, q) is G[4 sb + i][4 lane + q].  128 MFMAs of 8 cycles.
//
// ~2 050 matrix-pipe cycles and ~45 vector instructions a tile.  The feed is blr_pass_dma_kernel's: one 16-row buffer
// per wave; every operand of the backward (sixteen rows, the residuals) goes to registers before the next tile's
// sixteen DMAs are issued, so those have the backward and the other wave's turn to land.
constexpr int QW = 32;   // forward B operand: registers per lane

// DBG (profiling only, WRONG results; BSC_BLR_Q_DBG + BSC_PROFILING_BUILDS): 1 = no arithmetic at all (the feed's own
// ceiling: DMAs, waits, LDS reads), 2 = forward only, 3 = backward only
// `next_row0()` is called once, right before the next tile's DMAs are issued, `after_dma()` right behind them.
template <int AUX, int DBG, int PRIO, bool VROW = false, typename NextRow, typename AfterDma>
__device__ __forceinline__ void q_tile_step(float4& yv_cur, float* __restrict__ tl, float* __restrict__ rb,
                                            const float (&wreg)[QW], mfma_f32x4 (&acc)[2][4], float& qacc,
                                            const float* __restrict__ X, int64_t ldx, const float* __restrict__ y,
                                            NextRow next_row0, AfterDma after_dma, int64_t B, int lane) {
    const int kq = lane >> 4;
    // the tile's DMAs (issued a step ago) and its y have landed
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    asm volatile("" ::: "memory");
    // PRIO 1: the stretch from "my tile has landed" to "my next tile's DMAs are out" runs at raised priority -- it is
    // the latency chain that sets how fast a wave can draw on HBM; the backward, which only has to finish before
    // the next tile lands, yields to the partner wave's forward.  (PRIO 2: the other way round, for the A/B.)
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    const float4 yv = yv_cur;
    if (DBG == 0 || DBG == 2) {
        mfma_f32x4 d[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) d[g] = mfma_f32x4{0.f, 0.f, 0.f, 0.f};
        const float* abase = tl + (lane & 3) * MT_RS + 16 * (lane >> 3);
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            float4 a[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                a[g] = *reinterpret_cast<const float4*>(abase + 4 * g * MT_RS + 128 * (c8 >> 2) + 4 * (c8 & 3));
#pragma unroll
            for (int g = 0; g < 4; ++g) d[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g].x, wreg[4 * c8 + 0], d[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) d[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g].y, wreg[4 * c8 + 1], d[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) d[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g].z, wreg[4 * c8 + 2], d[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) d[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g].w, wreg[4 * c8 + 3], d[g], 0, 0, 0);
        }
        // fold the eight column blocks (lane bits 5, 4, 3); lanes l and l ^ 8 end with the same four values
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t0 = swap_add32(d[0][i], d[2][i]);    // lanes 0..31: row group 0, lanes 32..63: row group 2
            const float t1 = swap_add32(d[1][i], d[3][i]);    //              row group 1               row group 3
            float v = swap_add16(t0, t1);                     // DPP rows 0, 2 keep t0, rows 1, 3 t1: row group kq
            v += dpp_f32<DPP_ROW_ROR8>(v);
            u[i] = v;
        }
        const float r0 = yv.x - u[0], r1 = yv.y - u[1], r2 = yv.z - u[2], r3 = yv.w - u[3];
        qacc = fmaf(r0, r0, qacc); qacc = fmaf(r1, r1, qacc);
        qacc = fmaf(r2, r2, qacc); qacc = fmaf(r3, r3, qacc);
        // rb[draw][row]; both lanes of a pair (l, l ^ 8) store the same sixteen bytes
        *reinterpret_cast<float4*>(rb + (lane & 7) * MT_ROWS + 4 * kq) = make_float4(r0, r1, r2, r3);
    }
    wave_lds_sync();
    // every operand of the backward into registers, then the buffer belongs to the next tile
    float4 x4[MT_ROWS];
#pragma unroll
    for (int r = 0; r < MT_ROWS; ++r) x4[r] = *reinterpret_cast<const float4*>(tl + r * MT_RS + 4 * lane);
    float4 ra[2][4];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            ra[sb][g] = *reinterpret_cast<const float4*>(rb + (4 * sb + (lane & 3)) * MT_ROWS + 4 * g);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wave_lds_sync();
    dma_mtile<AUX, VROW>(tl, X, ldx, y, next_row0(), B, lane, yv_cur);
    after_dma();
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
    if (DBG == 0 || DBG == 3) {
#pragma unroll
        for (int r = 0; r < MT_ROWS; ++r) {
            const float4 xr = x4[r];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const float4 rr = ra[sb][r >> 2];
                const float a = (r & 3) == 0 ? rr.x : (r & 3) == 1 ? rr.y : (r & 3) == 2 ? rr.z : rr.w;
                acc[sb][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, xr.x, acc[sb][0], 0, 0, 0);
                acc[sb][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, xr.y, acc[sb][1], 0, 0, 0);
                acc[sb][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, xr.z, acc[sb][2], 0, 0, 0);
                acc[sb][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, xr.w, acc[sb][3], 0, 0, 0);
            }
        }
    } else {
        // keep the loads of a deletion build alive
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < MT_ROWS; ++r) s += x4[r].x;
        if (s == 12345.678f) qacc += ra[0][0].x + ra[1][3].w;
    }
}

// ---- the queued tail of blr_pass_q_kernel (option blr_steal) ---------------------------------------------------------
//
// The static schedule below ends when the slowest workgroup does, and the workgroups do not run at one speed: the
// mean end of a workgroup differs by 8-9 us of ~145 between the XCDs of one part (and WHICH XCDs are slow differs from
// part to part: tools/stamps_structure.py), with another ~5 us of launch-to-launch noise on top.  So every wave takes
// only (1000 - blr_steal) per mille of an even share statically and draws the remaining tiles, one at a time, from
// one of up to 64 queues.  Queue q holds a contiguous run of `per_q` tiles and serves the workgroups with
// (blockIdx / 8) % nq = q -- workgroups are dealt to the XCDs round-robin, so a queue is shared by one workgroup of
// every XCD (twice over at 512 workgroups) and the fast XCDs take the tiles the slow ones do not get to.  A wave leaves
// when its queue is empty; nothing moves between queues.
//   * The request goes out BEFORE the wait for the current tile's DMAs and is looked at only when the next tile's DMAs
//     are issued, so its latency is behind a wait the wave has anyway.
//   * A cache line serves one returning atomic per ~15 ns whatever the address in it (tools/ubench_atomic_queue.hip):
//     hence 64 heads 256 bytes apart, not one counter -- and no device-wide "who is last" counter either: the
//     workgroups of a queue count themselves out on the queue's own line and the last one zeroes it.
//   * (Measured and dropped: waves of an empty queue moving on to other queues.  Finding one takes a look at all 64 heads;
//     2 048 waves doing that within microseconds of each other, and then falling on the few queues with a tile left,
//     cost 20-40 us a launch: profiles/r04_ab_pass_q_steal.txt.)
// Which wave adds which tile into its partial sums now depends on timing: the LAST BITS of the f32 sums differ from
// launch to launch (blr_steal = 0 keeps the reproducible static schedule).
constexpr int STEAL_Q = 64, STEAL_STRIDE = 64;      // queues; words between their heads (word 1 of a line: workgroups done)
struct StealArgs {
    unsigned* heads;          // nullptr: static schedule only
    int per_q;                // tiles per queue (the last non-empty one may hold fewer)
    int nq;                   // queues in use: min(64, whole groups of 8 workgroups)
    long long t0, n_dyn;      // first queued tile, queued tiles
};
__device__ __forceinline__ int steal_len(const StealArgs& s, int q) {
    const long long r = s.n_dyn - (long long)q * s.per_q;
    return r <= 0 ? 0 : r < s.per_q ? (int)r : s.per_q;
}
// (inline asm: the compiler's atomic optimizer rewrites a one-lane __hip_atomic_fetch_add and waits for the result -- vmcnt(0),
// so for every DMA in flight as well -- right behind it.  The result is valid after the caller's next vmcnt(0).)
__device__ __forceinline__ unsigned steal_request(const StealArgs& s, int q, int lane) {
    unsigned t = 1u;
    const unsigned off = (unsigned)q * (STEAL_STRIDE * 4u);
    if (lane == 0) asm volatile("global_atomic_add %0, %1, %0, %2 sc0" : "+v"(t) : "v"(off), "s"(s.heads) : "memory");
    return t;
}
// row0 of the tile the request `pend` on queue q drew, or B: the queue is empty
__device__ __forceinline__ int64_t steal_resolve(const StealArgs& s, unsigned pend, int q, int64_t B) {
    // (the value is looked at HERE, not where the request was made; wave-uniform, and said so: or every DMA built on
    // it sits in a waterfall loop)
    asm volatile("" : "+v"(pend));
    const unsigned t = __builtin_amdgcn_readfirstlane(pend);
    return (long long)t < (long long)steal_len(s, q) ? (s.t0 + (int64_t)q * s.per_q + (int64_t)t) * MT_ROWS : B;
}

// Which tiles a wave of blr_pass_q_kernel reads.
//
// STATIC (reproducible: a wave's tiles and their order are a function of the launch geometry alone).  Window w < n_all
// is one contiguous run of (workgroups x 4) tiles, tile = w * w_all + slot, as in every pass kernel of this file.  The
// workgroups with an EVEN blockIdx then take `n_a - n_all` further windows of (even workgroups x 4) tiles among
// themselves: on this part the XCDs 0, 2, 4, 6 stream 7-15 % faster than 1, 3, 5, 7 (tools/ubench_dma_inflight: the
// same 3 GiB split evenly ends after 390 us on the even XCDs and 450 us on the odd ones), workgroups are dealt to the XCDs
// round-robin, and a pass that gives every workgroup the same share ends when the slow half does.  The placement is an
// observation used for speed only: any placement computes the same sums.
//
// (Measured and dropped: tiles popped from eight atomic queues, one per XCD, with stealing -- every workgroup then ends
// within 0.2 us of every other, but 62 500 returning atomics on eight words take 226 us where the static schedule takes
// 160, profiles/r04_ab_pass_q_schedules.txt; and the sums would no longer be reproducible in the last bits.)
struct QSched {
    int n_all, n_mine, rev, keep;
    int64_t w_all, w_a, slot, slot_a, B;
    __device__ __forceinline__ int64_t row0(int p) const {        // p-th tile of this wave's walk; p >= n_mine: the empty tile
        if (p >= n_mine) return B;
        const int w = rev ? n_mine - 1 - p : p;
        const int64_t tile = w < n_all ? (int64_t)w * w_all + slot
                                       : (int64_t)n_all * w_all + (int64_t)(w - n_all) * w_a + slot_a;
        return tile * MT_ROWS;
    }
};

template <bool NT, int DBG, int PRIO, bool STEAL = false>
__global__ __launch_bounds__(PASS_BLOCK, 2) void blr_pass_q_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y, int64_t B,
    const float* __restrict__ W, int S, float* __restrict__ slab, int n_all, int n_a, int rev, int keep,
    unsigned long long* __restrict__ stamps, FoldArgs fold, StealArgs steal) {
    constexpr int LDS_FLOATS = PASS_WAVES * (MT_WAVE_LDS > SLAB_STRIDE ? MT_WAVE_LDS : SLAB_STRIDE);
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long t_start = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    float* tl = lds + wave * MT_WAVE_LDS;
    float* rb = tl + MT_ROWS * MT_RS;          // residuals [draw][row]
    mfma_f32x4 acc[2][4];                      // [draw group][column component]: register i = draw 4 sb + i
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[sb][q] = mfma_f32x4{0.f, 0.f, 0.f, 0.f};
    float qacc = 0.f;
    float4 yv;
    float wreg[QW];
    // forward B operand: w[draw lane % 8][128 h + 16 k8 + 4 m + c], register 4 (4 h + m) + c; draws >= S are zero
    auto load_w = [&]() {
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((lane & 7) < S)
                w4 = *reinterpret_cast<const float4*>(W + (int64_t)(lane & 7) * GCOLS + 128 * (c8 >> 2) + 16 * (lane >> 3) +
                                                      4 * (c8 & 3));
            wreg[4 * c8 + 0] = w4.x; wreg[4 * c8 + 1] = w4.y;
            wreg[4 * c8 + 2] = w4.z; wreg[4 * c8 + 3] = w4.w;
        }
    };
    {
        QSched sc;
        sc.n_all = n_all; sc.rev = rev; sc.B = B;
        sc.n_mine = (blockIdx.x & 1) ? n_all : n_a;
        sc.w_all = (int64_t)gridDim.x * PASS_WAVES;
        sc.w_a = (int64_t)((gridDim.x + 1) >> 1) * PASS_WAVES;
        sc.slot = (int64_t)blockIdx.x * PASS_WAVES + wave;
        sc.slot_a = (int64_t)(blockIdx.x >> 1) * PASS_WAVES + wave;
        const int n_mine = sc.n_mine;
        if (!NT) keep = n_mine;
        const int n_stream = n_mine - keep > 0 ? n_mine - keep : 0;   // tiles read non-temporal; the last `keep` allocate
        // the first tile's DMAs go out before anything else: W (L2 hits) loads behind them
        if (n_stream > 0) dma_mtile<2>(tl, X, ldx, y, sc.row0(0), B, lane, yv);
        else dma_mtile<0>(tl, X, ldx, y, sc.row0(0), B, lane, yv);
        load_w();
        int k = 0;
        if constexpr (STEAL) {
            // static share (the host: n_mine >= 1 for every wave, keep = 0), then one queued tile per step until none is left
            for (; k + 1 < n_mine; ++k)
                q_tile_step<2, DBG, PRIO, true>(yv, tl, rb, wreg, acc, qacc, X, ldx, y, [&] { return sc.row0(k + 1); }, [] {}, B, lane);
            const int q = (blockIdx.x >> 3) % steal.nq;
            if (stamps && tid == 0) stamps[8 * (int64_t)blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();     // wave 0: static share read
            unsigned long long taken = 0;
            for (;;) {
                const unsigned pend = steal_request(steal, q, lane);
                int64_t next = 0;
                q_tile_step<2, DBG, PRIO, true>(yv, tl, rb, wreg, acc, qacc, X, ldx, y,
                                                [&] { next = steal_resolve(steal, pend, q, B); return next; }, [] {}, B, lane);
                if (next >= B) break;
                ++taken;
            }
            if (stamps && tid == 0) {
                stamps[8 * (int64_t)blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();     // wave 0: left the queue
                stamps[8 * (int64_t)blockIdx.x + 6] = taken;                                // ... with this many queued tiles
            }
        } else {
            for (; k + 1 < n_stream; ++k)
                q_tile_step<2, DBG, PRIO>(yv, tl, rb, wreg, acc, qacc, X, ldx, y, [&] { return sc.row0(k + 1); }, [] {}, B, lane);
            for (; k < n_mine; ++k)
                q_tile_step<0, DBG, PRIO>(yv, tl, rb, wreg, acc, qacc, X, ldx, y, [&] { return sc.row0(k + 1); }, [] {}, B, lane);
        }
    }
    // no LDS-DMA of this wave may still be in flight when the tile region is reused for the block reduction
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    __syncthreads();
    __shared__ unsigned steal_last_s;
    if (STEAL && tid == 0) {
        // every wave of this workgroup has left its queue: count the workgroup out on the queue's line; the last of the
        // queue's workgroups (groups of 8 with group % nq = q; the last group of the grid may be short) zeroes the line
        const int q = (blockIdx.x >> 3) % steal.nq;
        int homes = 0;
        for (int j = q; 8 * j < (int)gridDim.x; j += steal.nq) homes += (int)gridDim.x - 8 * j < 8 ? (int)gridDim.x - 8 * j : 8;
        steal_last_s = __hip_atomic_fetch_add(steal.heads + q * STEAL_STRIDE + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                       (unsigned)homes - 1u;
    }
    float* ep = lds + wave * SLAB_STRIDE;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(ep + (4 * sb + i) * GCOLS + 4 * lane) =
                make_float4(acc[sb][0][i], acc[sb][1][i], acc[sb][2][i], acc[sb][3][i]);
    float qv = (lane & 8) ? 0.f : qacc;        // lane (draw lane % 8, kq): rows 4 kq .. of that draw; lanes l ^ 8 repeat them
    qv += __shfl_xor(qv, 16);
    qv += __shfl_xor(qv, 32);
    if (lane < SG) ep[SLAB_G + lane] = qv;
    __syncthreads();
    if (STEAL && steal_last_s && tid < 2)      // the queue's last workgroup: head and count back to zero
        __hip_atomic_store(steal.heads + ((blockIdx.x >> 3) % steal.nq) * STEAL_STRIDE + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* out = slab + (int64_t)blockIdx.x * SLAB_STRIDE;
    for (int i = tid; i < SLAB_STRIDE; i += PASS_BLOCK) {
        const int src = i < SLAB_G ? (i & 7) * GCOLS + (i >> 3) : i;
        float v = lds[src];
#pragma unroll
        for (int kk = 1; kk < PASS_WAVES; ++kk) v += lds[kk * SLAB_STRIDE + src];
        // (folded finish: write-through, so that a workgroup on another XCD reads the row from memory)
        if (!STEAL && fold.mode) __hip_atomic_store(out + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[i] = v;
    }
    if (!STEAL && fold.mode) {
        // ---- the finish in the pass's tail (FoldArgs) ----
        __shared__ unsigned ticket_s;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's slab stores have been written through
        __syncthreads();                                        // ... and every other wave's of the workgroup
        if (tid == 0) {
            if (stamps) stamps[8 * (int64_t)blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();     // slab row written through
            ticket_s = __hip_atomic_fetch_add(fold.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (stamps) { stamps[8 * (int64_t)blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime(); stamps[8 * (int64_t)blockIdx.x + 7] = ticket_s; }
        }
        __syncthreads();
        const int n_roles = fold.mode == 1 ? (fold.a.D + 7) / 8 + 1 : (SLAB_STRIDE + BSC_WAVE - 1) / BSC_WAVE;
        // the LAST arrival takes role 0, the earliest of the last n_roles the highest role (mode 1: the scalar role,
        // which has the most to do before it needs the slab -- and the longest wait for the last partial)
        const int role = (int)gridDim.x - 1 - (int)ticket_s;
        if (role < n_roles) {                                   // one of the last n_roles arrivals
            auto all_arrived = [&] {
                if (tid == 0) {
                    // every row has arrived?  (sc1 poll by one lane; the rows still missing belong to workgroups that
                    // are computing on resident slots.  Bounded: ~2 s of the 100 MHz clock, then carry on.)
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    while (__hip_atomic_load(fold.counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                        __builtin_amdgcn_s_sleep(4);
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;
                    }
                    if (stamps) stamps[8 * (int64_t)blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();
                }
                __syncthreads();
            };
            if (fold.mode == 1) fused_update_role<PASS_BLOCK, true>(fold.a, role, all_arrived);
            else slab_stats_role<PASS_BLOCK, true>(slab, (int)gridDim.x, GCOLS, fold.S_total, fold.s_base, fold.Q, fold.G, role, all_arrived);
            __syncthreads();
            if (tid == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned done = __hip_atomic_fetch_add(fold.counters + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (done == (unsigned)n_roles - 1u) {           // the last role: zero the counters for the next launch
                    __hip_atomic_store(fold.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(fold.counters + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    if (stamps) {
        // measurement aid (option blr_stamps): when did this workgroup start and end, and on which XCD did it run
        __syncthreads();
        if (tid == 0) {
            unsigned long long* st = stamps + 8 * (int64_t)blockIdx.x;
            st[0] = t_start;
            st[1] = __builtin_amdgcn_s_memrealtime();
            st[2] = __builtin_amdgcn_s_getreg((3 << 11) | 20);     // HW_REG_XCC_ID[3:0]
            st[3] = __builtin_amdgcn_s_getreg((15 << 11) | 4);     // HW_REG_HW_ID[15:0]
        }
    }
}

// ---- both contractions on the MFMA pipe (D == 256) ------------------------------------------
//
// With every load served from the caches blr_pass_mfma_kernel still takes ~150 us per 1M x 256
// pass (tools/ab_pass.py floor): 64 v_mfma_f32_16x16x4 (half of each idle: S = 8 of 16 columns)
// plus 256 v_pk_fma_f32 per 16-row tile keep a SIMD busy about as long as HBM takes to deliver the
// tile, so a faster memory schedule alone cannot show.  Here the backward rank-1 updates
// G[s, :] += r[n, s] x[n, :] run on v_mfma_f32_4x4x1_16B_f32 -- sixteen independent 4 x 4 outer
// products per instruction, no idle half: block b = lane / 4 owns columns 16 b .. 16 b + 15, the A
// operand is the residual r[n][4 sb + lane % 4] (the same in every block), the B operand component
// q of the row as the lane holds it (column 4 lane + q), and the result register i of accumulator
// (sb, q) is G[4 sb + i][4 lane + q] -- the accumulators the VALU variant keeps, so the epilogue and
// the slab are unchanged.  128 MFMAs of 8 cycles replace 256 packed FMAs of 8 cycles and leave the
// VALU with the sixteen residuals.
//
// Schedule (`Sched`): the windows (one per iteration: gridDim.x * 4 tiles) are walked forward,
// backward, or ROTATED: workgroup group g = blockIdx >> rot_shift starts at window g % n_iter and
// wraps around.  Windows [0, keep) -- the head of the mini-batch -- are read with the allocating
// policy and stay in the 256 MiB Infinity Cache from pass to pass, all others non-temporal; with the
// rotation a fraction keep / n_iter of the workgroups is in the cached zone at any moment, so cache
// hits and HBM reads overlap for the whole pass instead of taking turns.
struct Sched {
    int n_iter, mode, keep, phi;
    int64_t stride0, slot, B;
    __device__ __forceinline__ int window(int p) const {
        if (mode == 2) { const int w = p + phi; return w >= n_iter ? w - n_iter : w; }
        if (mode == 4) return 0;          // measurement only: every position re-reads window 0
        return mode == 1 ? n_iter - 1 - p : p;
    }
    // allocating policy for the window at position p?  (p = n_iter: the empty prefetch after the last)
    __device__ __forceinline__ bool kept(int p) const {
        if (p >= n_iter) return true;
        return mode == 2 ? window(p) < keep : p >= n_iter - keep;
    }
    __device__ __forceinline__ int64_t row0(int p) const {
        return p < n_iter ? ((int64_t)window(p) * stride0 + slot) * MT_ROWS : B;
    }
};

constexpr int MX_WAVE_LDS = MT_ROWS * MT_RS + MT_ROWS * 16;   // tile + residuals of up to 16 draws

// NSB = sample groups of four per pass: 2 (S <= 8) or 4 (S <= 16: the forward MFMA's sixteen columns
// all carry a draw -- the second eight cost the forward nothing --, the backward doubles; one pass
// over X for 16 draws instead of two, bsc_blr_data_pass with S > 8).  The block partials of draws
// 8 .. 15 go to a second slab behind the first (slab + gridDim.x rows).
template <bool NT, int NSB>
__global__ __launch_bounds__(PASS_BLOCK, 2) void blr_pass_mx_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ y, int64_t B,
    const float* __restrict__ W, int S, float* __restrict__ slab, int n_iter, int mode, int keep,
    int rot_shift) {
    constexpr int NS = 4 * NSB;                 // draws per pass
    constexpr int NH = NSB / 2;                 // slabs (one per eight draws)
    constexpr int LDS_FLOATS = PASS_WAVES * (MX_WAVE_LDS > NH * SLAB_STRIDE ? MX_WAVE_LDS : NH * SLAB_STRIDE);
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    float* tl = lds + wave * MX_WAVE_LDS;      // this wave's tile
    float* rb = tl + MT_ROWS * MT_RS;          // residuals [sample][row]

    float wreg[GCOLS / 4];                     // forward B operand, as in blr_pass_mfma_kernel
#pragma unroll
    for (int j = 0; j < GCOLS / 16; ++j) {
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i16 < S) w4 = *reinterpret_cast<const float4*>(W + (int64_t)i16 * GCOLS + 64 * kq + 4 * j);
        wreg[4 * j + 0] = w4.x; wreg[4 * j + 1] = w4.y;
        wreg[4 * j + 2] = w4.z; wreg[4 * j + 3] = w4.w;
    }
    mfma_f32x4 acc[NSB][4];                    // [sample group][column component]: register i = sample 4 sb + i
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[sb][q] = mfma_f32x4{0.f, 0.f, 0.f, 0.f};
    float qacc = 0.f;
    const bool live = i16 < NS;                // lanes whose forward MFMA column is a sample

    Sched sc;
    sc.n_iter = n_iter; sc.mode = mode; sc.keep = NT ? keep : n_iter;
    sc.stride0 = (int64_t)gridDim.x * PASS_WAVES;
    sc.slot = (int64_t)blockIdx.x * PASS_WAVES + wave;
    sc.B = B;
    sc.phi = (mode == 2 && n_iter > 0) ? (int)((blockIdx.x >> rot_shift) % (unsigned)n_iter) : 0;

    MTile t;
    if (sc.kept(0)) load_mtile_policy<0>(t, X, ldx, y, sc.row0(0), B, lane);
    else load_mtile_policy<2>(t, X, ldx, y, sc.row0(0), B, lane);
    for (int p = 0; p < n_iter; ++p) {
        // the tile to LDS, its registers take the next window's tile (tiles outside the mini-batch
        // read zeros without touching memory)
#pragma unroll
        for (int r = 0; r < MT_ROWS; ++r)
            *reinterpret_cast<float4*>(tl + r * MT_RS + 4 * lane) = t.x[r];
        const float4 yv = t.yv;
        const int64_t next0 = sc.row0(p + 1);
        if (sc.kept(p + 1)) load_mtile_policy<0>(t, X, ldx, y, next0, B, lane);
        else load_mtile_policy<2>(t, X, ldx, y, next0, B, lane);
        wave_lds_sync();

        // forward on v_mfma_f32_16x16x4_f32 (two accumulators, no MFMA waits on its predecessor)
        mfma_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        const float* arow = tl + i16 * MT_RS + 64 * kq;
#pragma unroll
        for (int j = 0; j < GCOLS / 16; j += 2) {
            const float4 a0 = *reinterpret_cast<const float4*>(arow + 4 * j);
            const float4 a1 = *reinterpret_cast<const float4*>(arow + 4 * j + 4);
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wreg[4 * j + 0], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wreg[4 * j + 4], d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wreg[4 * j + 1], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wreg[4 * j + 5], d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wreg[4 * j + 2], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wreg[4 * j + 6], d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wreg[4 * j + 3], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wreg[4 * j + 7], d1, 0, 0, 0);
        }
        // result register reg of lane (i16, kq) = dot(row 4 kq + reg, sample i16); the residuals go
        // to rb[sample][row]: one 16-byte store per live lane
        if (live) {
            const float r0 = yv.x - (d0[0] + d1[0]), r1 = yv.y - (d0[1] + d1[1]);
            const float r2 = yv.z - (d0[2] + d1[2]), r3 = yv.w - (d0[3] + d1[3]);
            qacc = fmaf(r0, r0, qacc); qacc = fmaf(r1, r1, qacc);
            qacc = fmaf(r2, r2, qacc); qacc = fmaf(r3, r3, qacc);
            *reinterpret_cast<float4*>(rb + i16 * MT_ROWS + 4 * kq) = make_float4(r0, r1, r2, r3);
        }
        wave_lds_sync();

        // backward on v_mfma_f32_4x4x1_16B_f32: per row 2 sample groups x 4 column components
#pragma unroll
        for (int g = 0; g < MT_ROWS / 4; ++g) {
            if (g) asm volatile("" ::: "memory");   // four rows of reads in flight
            // A operands of rows 4 g .. 4 g + 3: r[row][4 sb + lane % 4] (broadcast reads)
            float4 ra[NSB];
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
                ra[sb] = *reinterpret_cast<const float4*>(rb + (4 * sb + (lane & 3)) * MT_ROWS + 4 * g);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float4 x4 = *reinterpret_cast<const float4*>(tl + (4 * g + rr) * MT_RS + 4 * lane);
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) {
                    const float a = rr == 0 ? ra[sb].x : rr == 1 ? ra[sb].y : rr == 2 ? ra[sb].z : ra[sb].w;
                    acc[sb][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x4.x, acc[sb][0], 0, 0, 0);
                    acc[sb][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x4.y, acc[sb][1], 0, 0, 0);
                    acc[sb][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x4.z, acc[sb][2], 0, 0, 0);
                    acc[sb][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x4.w, acc[sb][3], 0, 0, 0);
                }
            }
        }
        wave_lds_sync();   // the next iteration overwrites the tile
    }

    // block reduction through LDS, fixed order over waves (slab layout of the other pass kernels,
    // one slab per eight draws)
    __syncthreads();
    float* ep = lds + wave * (NH * SLAB_STRIDE);
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(ep + (sb >> 1) * SLAB_STRIDE + (4 * (sb & 1) + i) * GCOLS + 4 * lane) =
                make_float4(acc[sb][0][i], acc[sb][1][i], acc[sb][2][i], acc[sb][3][i]);
    float qv = live ? qacc : 0.f;              // lane (i16, kq): rows 4 kq .. of draw i16
    qv += __shfl_xor(qv, 16);
    qv += __shfl_xor(qv, 32);
    if (lane < NS) ep[(lane >> 3) * SLAB_STRIDE + SLAB_G + (lane & 7)] = qv;
    __syncthreads();
    for (int i = tid; i < NH * SLAB_STRIDE; i += PASS_BLOCK) {
        const int h = i / SLAB_STRIDE, ii = i - h * SLAB_STRIDE;
        const int src = h * SLAB_STRIDE + (ii < SLAB_G ? (ii & 7) * GCOLS + (ii >> 3) : ii);
        float v = lds[src];
#pragma unroll
        for (int k = 1; k < PASS_WAVES; ++k) v += lds[k * (NH * SLAB_STRIDE) + src];
        slab[((int64_t)h * gridDim.x + blockIdx.x) * SLAB_STRIDE + ii] = v;
    }
}

// float64 sum of p[b * SLAB_STRIDE] over slab rows b = first, first+step, ...
// Loads are issued in batches of 16 before any add: the partials were written by
// another kernel, so every load is a MALL/HBM round trip (~0.4 us) and a
// load-add-load-add chain would serialise them.
// COH: the rows were written by other workgroups of THIS launch (the folded finish): sc1 loads.
template <int BATCH = 16, bool COH = false>
__device__ __forceinline__ double slab_column_sum(const float* __restrict__ p, int first,
                                                  int step, int n_rows) {
    double sum = 0.0;
    for (int b0 = first; b0 < n_rows; b0 += step * BATCH) {
        float v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int b = b0 + j * step;
            if (COH) v[j] = b < n_rows ? __hip_atomic_load(p + (int64_t)b * SLAB_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            else v[j] = b < n_rows ? p[(int64_t)b * SLAB_STRIDE] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) sum += (double)v[j];
    }
    return sum;
}

// Float64 sums of a 64-column run of the slab (columns col0 .. col0+63) over the rows
// wave + n_waves * k that this wave owns: 16-byte buffer loads covering four rows apiece
// (lane = (row group lane>>4, column chunk lane&15)); rows past n_slab read as zero through
// the descriptor.  On return lanes 0-15 hold, in s4[0..3], the sums of columns
// col0 + 4*lane .. +3.  `between` runs after the first batch of loads has been issued and
// before it is consumed (work that does not depend on the slab).
// Kept deliberately compact: these finishing kernels start instruction-cache cold behind the
// 165-us data pass, and straight-line code is fetched at ~0.5 us per 64 bytes -- the 32
// guarded scalar loads this replaces (1.5 KB of code) cost 10 us before the first load had
// even been issued (cycle counters, round 1).
// JJ: loads in flight per lane and trip (8: the finish kernels' 16 waves cover 512 rows in one trip; 32: four waves do).
// AUX: cache policy of the loads (16 = sc1, for rows written by other workgroups of this launch).
template <int N_WAVES, int JJ = 8, int AUX = 0, typename F>
__device__ __forceinline__ void slab_run_sum(const float* __restrict__ slab, int n_slab, int col0,
                                             int wave, int lane, double (&s4)[4], F between) {
    const uint64_t slab_bytes = (uint64_t)n_slab * SLAB_STRIDE * 4u;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)slab, 0, slab_bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)slab_bytes, 0x00020000);
    const int q4 = lane >> 4, c16 = lane & 15;
    const int voff = ((wave + N_WAVES * q4) * SLAB_STRIDE + col0 + 4 * c16) * 4;
    constexpr int BATCH_BYTES = 4 * N_WAVES * SLAB_STRIDE * 4;      // 4 * N_WAVES rows per load
    s4[0] = s4[1] = s4[2] = s4[3] = 0.0;
    for (int base = 0; base < n_slab; base += 4 * JJ * N_WAVES) {   // one trip up to 4 JJ N_WAVES partials
        float4 v8[JJ];
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            auto v = __builtin_amdgcn_raw_buffer_load_b128(
                rs, voff, base * (SLAB_STRIDE * 4) + jj * BATCH_BYTES, AUX);
            v8[jj] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                                 __uint_as_float(v[3]));
        }
        if (base == 0) between();
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            s4[0] += (double)v8[jj].x; s4[1] += (double)v8[jj].y;
            s4[2] += (double)v8[jj].z; s4[3] += (double)v8[jj].w;
        }
    }
    // fold the four row groups (lane bits 4, 5)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s4[i] += __shfl_xor(s4[i], 16);
        s4[i] += __shfl_xor(s4[i], 32);
    }
}

// Sum block partials in float64, fixed order.  One output per lane; the 16
// waves of a block split the slab rows, then combine through LDS in wave order.
constexpr int RED_BLOCK = 1024;
constexpr int RED_WAVES = RED_BLOCK / BSC_WAVE;

template <int BLOCK, bool COH, typename Wait>
__device__ void slab_stats_role(const float* __restrict__ slab, int n_blocks, int D, int S, int s_base,
                                double* __restrict__ Q, double* __restrict__ G, int role, Wait wait) {
    constexpr int WAVES = BLOCK / BSC_WAVE;
    __shared__ double part[WAVES][BSC_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = role * BSC_WAVE + lane;
    double s4[4];
    wait();
    slab_run_sum<WAVES, (WAVES >= 16 ? 8 : 32), (COH ? 16 : 0)>(slab, n_blocks, role * BSC_WAVE, wave, lane, s4, [] {});
    if (lane < 16) {   // (columns past SLAB_STRIDE in the last role are read but never written out)
#pragma unroll
        for (int k = 0; k < 4; ++k) part[wave][4 * lane + k] = s4[k];
    }
    __syncthreads();
    if (wave == 0 && i < SLAB_STRIDE) {
        double tot = part[0][lane];
#pragma unroll
        for (int k = 1; k < WAVES; ++k) tot += part[k][lane];
        if (i < SLAB_G) {
            int s = i & 7, d = i >> 3;
            if (s_base + s < S && d < D) G[(int64_t)(s_base + s) * D + d] = tot;
        } else {
            int s = i - SLAB_G;
            if (s_base + s < S) Q[s_base + s] = tot;
        }
    }
}

__global__ __launch_bounds__(RED_BLOCK) void blr_slab_reduce_kernel(
    const float* __restrict__ slab, int n_blocks, int D, int S, int s_base,
    double* __restrict__ Q, double* __restrict__ G) {
    slab_stats_role<RED_BLOCK, false>(slab, n_blocks, D, S, s_base, Q, G, (int)blockIdx.x, [] {});
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

// eps for `n_steps` consecutive Philox steps in one launch:
// eps[(k*S + s)*(D+1) + d], d < D from stream 0, d == D from stream 1 (the layout
// bsc_blr_sample writes for one step).  The noise does not depend on the
// parameters, so it is produced ahead of time and kept out of the update's
// latency chain.
__global__ void blr_noise_kernel(int D, int S, uint64_t seed, uint32_t step0, int n_steps,
                                 double* __restrict__ eps) {
    const int n_blocks = (D + 3) / 4;
    const int per_step = S * n_blocks + S;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)per_step * n_steps) return;
    const int k = (int)(idx / per_step), r = (int)(idx % per_step);
    double* out = eps + (int64_t)k * S * (D + 1);
    double z[4];
    if (r < S * n_blocks) {
        const int s = r / n_blocks, b = r % n_blocks;
        philox_normal4(seed, (uint32_t)b, (uint32_t)s, 0u, step0 + (uint32_t)k, z);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * b + j < D) out[(int64_t)s * (D + 1) + 4 * b + j] = z[j];
    } else {
        const int s = r - S * n_blocks;
        philox_normal4(seed, 0u, (uint32_t)s, 1u, step0 + (uint32_t)k, z);
        out[(int64_t)s * (D + 1) + D] = z[0];
    }
}

// Draws for Philox block `pb` (columns 4pb..4pb+3) of sample s, given m, rho
// indexed by absolute column.
__device__ __forceinline__ void blr_draw_block(const double* m, const double* rho, int D, int s,
                                               int pb, uint64_t seed, uint32_t step,
                                               double* __restrict__ eps, float* __restrict__ W) {
    double z[4];
    philox_normal4(seed, (uint32_t)pb, (uint32_t)s, 0u, step, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = 4 * pb + j;
        if (d < D) {
            eps[(int64_t)s * (D + 1) + d] = z[j];
            double sd = exp(rho[d]);
            double wv = m[d] + sd * z[j];
            W[(int64_t)s * D + d] = (float)wv;
        }
    }
}

__device__ __forceinline__ void blr_draw_scale(double a, double b, int D, int s, uint64_t seed,
                                               uint32_t step, double* __restrict__ eps,
                                               double* __restrict__ xi) {
    double z[4];
    philox_normal4(seed, 0u, (uint32_t)s, 1u, step, z);
    eps[(int64_t)s * (D + 1) + D] = z[0];
    double sd = exp(b);
    xi[s] = a + sd * z[0];
}

__global__ void blr_sample_kernel(const double* __restrict__ lam, int D, int S, uint64_t seed,
                                  uint32_t step, double* __restrict__ eps,
                                  float* __restrict__ W, double* __restrict__ xi) {
    const int n_blocks = (D + 3) / 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < S * n_blocks) {
        blr_draw_block(lam, lam + D, D, idx / n_blocks, idx % n_blocks, seed, step, eps, W);
    } else if (idx < S * n_blocks + S) {
        blr_draw_scale(lam[2 * D], lam[2 * D + 1], D, idx - S * n_blocks, seed, step, eps, xi);
    }
}

// One workgroup.  Thread d owns column d (strided when D > blockDim).
constexpr int FIN_BLOCK = 256;
constexpr int FIN_WAVES = FIN_BLOCK / BSC_WAVE;
constexpr int FIN_MAX_S = 64;
constexpr double LOG_2PI = 1.8378770664093454835606594728112;

__global__ __launch_bounds__(FIN_BLOCK) void blr_elbo_grad_kernel(
    const double* __restrict__ lam, const double* __restrict__ eps,
    const float* __restrict__ W, const double* __restrict__ xi, const double* __restrict__ Q,
    const double* __restrict__ G, int D, int S, double batch_rows, double scale, double alpha0,
    double beta0, double* __restrict__ elbo, double* __restrict__ grad) {
    __shared__ double red[FIN_WAVES][FIN_MAX_S + 1];
    __shared__ double wsq[FIN_MAX_S];
    __shared__ double e_inv[FIN_MAX_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

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
        for (int k = 0; k < FIN_WAVES; ++k) t += red[k][tid];
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
        for (int k = 0; k < FIN_WAVES; ++k) sum_rho += red[k][S];
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

// ---- fused finish: slab reduce + ELBO/gradient + Adam + next draw ----------
//
// Grid = ceil(D/8) column workgroups + 1 scalar workgroup.  A column workgroup
// owns 8 columns x S<=8 samples (one 256-B run per slab row); the scalar
// workgroup owns Q, |w_s|^2, the entropy term, (a, b) and the ELBO.  State is
// double-buffered by the caller (lam_in/lam_out, cur/next draws) so no
// workgroup reads what another one writes.

__device__ __forceinline__ double adam_ascent_one(double lam, double g, double& m1, double& m2,
                                                  const FusedArgs& a) {
    const double na = a.beta1 * m1 + (1.0 - a.beta1) * g;
    const double nb = a.beta2 * m2 + (1.0 - a.beta2) * g * g;
    m1 = na;
    m2 = nb;
    const double mhat = na / a.corr1;
    const double vhat = nb / a.corr2;
    return lam + a.lr * mhat / (sqrt(vhat) + a.adam_eps);
}

// BLOCK threads per workgroup (1024 = 16 waves: 32 slab rows per wave at 512 partials; fewer waves
// launch sooner -- BSC_BLR_FINISH_BLOCK, A/B in tools/ab_pass.py)
template <int FUSED_BLOCK, bool COH, typename Wait>
__device__ void fused_update_role(const FusedArgs& a, int role, Wait wait) {
    constexpr int FUSED_WAVES = FUSED_BLOCK / BSC_WAVE;
    __shared__ double red[FUSED_WAVES][BSC_WAVE];
    __shared__ double sh[2 * FIN_MAX_S + 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, S = a.S;
    const int n_chunks = (D + 7) / 8;
    const double inv_S = 1.0 / (double)S;

    if (role < n_chunks) {
        // ---------------- column workgroup: columns d0 .. d0+7 ----------------
        const int d0 = 8 * role;
        const int dl = lane >> 3, sl = lane & 7;  // slab order within the run is [d][s]
        const int d = d0 + dl;
        double gm = 0.0, gr = 0.0;
        // every small operand of wave 0 is requested before the slab so that the whole
        // workgroup pays one memory round trip, not one per dependent stage
        double p_m = 0.0, p_rho = 0.0, p_m1 = 0.0, p_m2 = 0.0, p_r1 = 0.0, p_r2 = 0.0;
        double e_next = 0.0, e_rho = 0.0;   // e_rho = exp(rho), set before it is used on either path
        const bool pre_noise = a.eps_next && a.eps_next_ready;
        if (wave == 0) {
            if (sl == 0 && d < D) {
                p_m = a.lam_in[d]; p_rho = a.lam_in[D + d];
                p_m1 = a.m1[d]; p_m2 = a.m2[d];
                p_r1 = a.m1[D + d]; p_r2 = a.m2[D + d];
            }
            if (pre_noise && sl < S && d < D) e_next = a.eps_next[(int64_t)sl * (D + 1) + d];
        }
        if (a.slab) {  // S <= 8: lane <-> (column dl, sample sl) of one 256-B slab run
            const bool live = sl < S && d < D;
            double wv = 0.0, xs = 0.0, ev = 0.0;
            if (wave == 0 && live) {
                wv = (double)a.W[(int64_t)sl * D + d];
                xs = a.xi[sl];
                ev = a.eps[(int64_t)sl * (D + 1) + d];
            }
            // this wave's share of the slab (rows wave + 16 k, the workgroup's 64-float run); the
            // float64 exponentials that do not depend on it run while it is in flight
            double s4[4];
            double e_mxs = 0.0;
            if (COH) {              // (the operands above are on their way; the exponentials do not need the slab either)
                if (wave == 0) {
                    e_mxs = exp(-xs);
                    e_rho = exp(p_rho);
                }
                wait();
            }
            slab_run_sum<FUSED_WAVES, (FUSED_WAVES >= 16 ? 8 : 32), (COH ? 16 : 0)>(a.slab, a.n_slab, 64 * role, wave, lane, s4, [&] {
                if (COH) return;
                if (wave == 0) {
                    e_mxs = exp(-xs);
                    e_rho = exp(p_rho);
                }
            });
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) red[wave][4 * lane + i] = s4[i];
            }
            __syncthreads();
            if (wave != 0) return;
            double g = red[0][lane];
#pragma unroll
            for (int k = 1; k < FUSED_WAVES; ++k) g += red[k][lane];
            if (live) {
                const double dw = e_mxs * (a.s_q * g - a.k_w * wv);
                gm = dw;
                gr = dw * ev;
            }
        } else {
            if (wave != 0) return;
            e_rho = exp(p_rho);
            for (int s = sl; s < S; s += 8) {
                if (d < D) {
                    const double g = a.stats[S + (int64_t)s * D + d];
                    const double wv = (double)a.W[(int64_t)s * D + d];
                    const double dw = exp(-a.xi[s]) * (a.s_q * g - a.k_w * wv);
                    gm += dw;
                    gr += dw * a.eps[(int64_t)s * (D + 1) + d];
                }
            }
        }
        // fold the 8 sample lanes (lane bits 0-2)
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            gm += __shfl_xor(gm, off);
            gr += __shfl_xor(gr, off);
        }
        double* new_m = sh;        // [8]
        double* new_rho = sh + 8;  // [8]
        if (sl == 0 && d < D) {
            const double rho = p_rho;
            const double g_m = gm * inv_S;
            const double g_r = gr * inv_S * e_rho + 1.0;
            a.grad[d] = g_m;
            a.grad[D + d] = g_r;
            double m1 = p_m1, m2 = p_m2;
            const double nm = adam_ascent_one(p_m, g_m, m1, m2, a);
            a.m1[d] = m1; a.m2[d] = m2;
            double r1 = p_r1, r2 = p_r2;
            const double nr = adam_ascent_one(rho, g_r, r1, r2, a);
            a.m1[D + d] = r1; a.m2[D + d] = r2;
            a.lam_out[d] = nm;
            a.lam_out[D + d] = nr;
            new_m[dl] = nm;
            new_rho[dl] = nr;
        }
        wave_lds_sync();
        if (pre_noise) {
            // noise precomputed: w = m + e^rho * eps for this lane's (column, sample)
            for (int s = sl; s < S; s += 8)
                if (d < D) {
                    const double sd = exp(new_rho[dl]);
                    const double en = s == sl ? e_next : a.eps_next[(int64_t)s * (D + 1) + d];
                    a.W_next[(int64_t)s * D + d] = (float)(new_m[dl] + sd * en);
                }
        } else if (a.eps_next) {
            // two Philox blocks per sample cover the 8 columns
            for (int i = lane; i < 2 * S; i += BSC_WAVE) {
                const int s = i >> 1, pb = 2 * role + (i & 1);
                if (4 * pb < D)
                    blr_draw_block(new_m - d0, new_rho - d0, D, s, pb, a.seed, a.next_step,
                                   a.eps_next, a.W_next);
            }
        }
        return;
    }

    // ---------------------------- scalar workgroup ----------------------------
    double* Qs = sh;                    // [S]
    double* wsq = sh + FIN_MAX_S;       // [S]
    double* misc = sh + 2 * FIN_MAX_S;  // [1]=new a [2]=new b
    __shared__ double terms[3 * FIN_MAX_S];
    const bool pre_noise = a.eps_next && a.eps_next_ready;
    // request every small operand first (one memory round trip for the workgroup)
    double x_s = 0.0, e_sD = 0.0, e_next = 0.0;           // thread s < S
    if (tid < S) {
        x_s = a.xi[tid];
        e_sD = a.eps[(int64_t)tid * (D + 1) + D];
        if (pre_noise) e_next = a.eps_next[(int64_t)tid * (D + 1) + D];
    }
    double av = 0.0, bv = 0.0, am1 = 0.0, am2 = 0.0, bm1 = 0.0, bm2 = 0.0;  // thread 0
    if (tid == 0) {
        av = a.lam_in[2 * D]; bv = a.lam_in[2 * D + 1];
        am1 = a.m1[2 * D]; am2 = a.m2[2 * D];
        bm1 = a.m1[2 * D + 1]; bm2 = a.m2[2 * D + 1];
    }
    double rho_part = 0.0;
    for (int d = tid; d < D; d += FUSED_BLOCK) rho_part += a.lam_in[D + d];
    // |w_s|^2: wave-per-sample, fixed order (first 4 loads of each lane issued here)
    float wpre[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave < S) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = lane + BSC_WAVE * j;
            if (d < D) wpre[j] = a.W[(int64_t)wave * D + d];
        }
    }
    // folded finish: everything that does not need the slab first -- this role belongs to the EARLIEST of the role
    // workgroups, which has the longest wait for the last partial (a second draw per wave prefetched too: four
    // waves here, sixteen in the finish kernel)
    float wpre2[4] = {0.f, 0.f, 0.f, 0.f};
    bool sums_done = false;
    if (COH) {
        if (FUSED_WAVES < FIN_MAX_S && wave + FUSED_WAVES < S) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = lane + BSC_WAVE * j;
                if (d < D) wpre2[j] = a.W[(int64_t)(wave + FUSED_WAVES) * D + d];
            }
        }
        for (int s = wave; s < S; s += FUSED_WAVES) {
            double part = 0.0;
            int d = lane;
            if (s == wave || s == wave + FUSED_WAVES) {   // the prefetched columns, same ascending order as the loop below
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float wj = s == wave ? wpre[j] : wpre2[j];
                    if (d < D) part += (double)wj * (double)wj;
                    d += BSC_WAVE;
                }
            }
            for (; d < D; d += BSC_WAVE) {
                const double wv = (double)a.W[(int64_t)s * D + d];
                part += wv * wv;
            }
            part = wave_allsum_f64(part);
            if (lane == 0) wsq[s] = part;
        }
        rho_part = wave_allsum_f64(rho_part);
        if (lane == 0) red[wave][32] = rho_part;
        sums_done = true;
        wait();
    }
    if (a.slab) {  // S <= 8: thread -> (sample tid&7, slab-row group tid>>3)
        double part = slab_column_sum<(FUSED_BLOCK >= 1024 ? 8 : 16), COH>(a.slab + SLAB_G + (tid & 7), tid >> 3, FUSED_BLOCK / 8,
                                         a.n_slab);
        // fold the 8 row groups of this wave (lane bits 3-5), then the 16 waves
        part += __shfl_xor(part, 8);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (lane < 8) red[wave][lane] = part;
    } else {
        for (int s = tid; s < S; s += FUSED_BLOCK) Qs[s] = a.stats[s];
    }
    for (int s = wave; s < S && !sums_done; s += FUSED_WAVES) {
        double part = 0.0;
        int d = lane;
        if (s == wave) {  // the prefetched columns, same ascending order as the loop below
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (d < D) part += (double)wpre[j] * (double)wpre[j];
                d += BSC_WAVE;
            }
        }
        for (; d < D; d += BSC_WAVE) {
            const double wv = (double)a.W[(int64_t)s * D + d];
            part += wv * wv;
        }
        part = wave_allsum_f64(part);
        if (lane == 0) wsq[s] = part;
    }
    if (!sums_done) {
        rho_part = wave_allsum_f64(rho_part);
        if (lane == 0) red[wave][32] = rho_part;  // column 32: clear of the Q staging columns
    }
    __syncthreads();
    if (a.slab && tid < 8) {
        double t = 0.0;
        for (int k = 0; k < FUSED_WAVES; ++k) t += red[k][tid];
        Qs[tid] = t;
    }
    // per-sample terms in parallel (one thread per sample; thread s wrote Qs[s] itself),
    // then a fixed-order sum
    double* t_dxi = terms;              // [S]
    double* t_dxe = t_dxi + FIN_MAX_S;  // [S]
    double* t_f = t_dxe + FIN_MAX_S;    // [S]
    if (tid < S) {
        const int s = tid;
        const double x = x_s, e = exp(-x);
        const double inner = 0.5 * a.s_q * Qs[s] + 0.5 * a.k_w * wsq[s] + a.beta;
        const double dxi = a.c_xi + e * inner;
        t_dxi[s] = dxi;
        t_dxe[s] = dxi * e_sD;
        t_f[s] = a.c0 + a.c_xi * x - e * inner;
    }
    __syncthreads();
    if (tid == 0) {
        double sum_rho = 0.0;
        for (int k = 0; k < FUSED_WAVES; ++k) sum_rho += red[k][32];
        double fa = 0.0, fb = 0.0, fsum = 0.0;
        for (int s = 0; s < S; ++s) {
            fa += t_dxi[s];
            fb += t_dxe[s];
            fsum += t_f[s];
        }
        const double g_a = fa * inv_S;
        const double g_b = fb * inv_S * exp(bv) + 1.0;
        a.grad[2 * D] = g_a;
        a.grad[2 * D + 1] = g_b;
        a.elbo[0] = fsum * inv_S + sum_rho + bv + 0.5 * (double)(D + 1) * (1.0 + LOG_2PI);
        const double na = adam_ascent_one(av, g_a, am1, am2, a);
        a.m1[2 * D] = am1; a.m2[2 * D] = am2;
        const double nb = adam_ascent_one(bv, g_b, bm1, bm2, a);
        a.m1[2 * D + 1] = bm1; a.m2[2 * D + 1] = bm2;
        a.lam_out[2 * D] = na;
        a.lam_out[2 * D + 1] = nb;
        misc[1] = na;
        misc[2] = nb;
    }
    __syncthreads();
    if (pre_noise) {
        if (tid < S) a.xi_next[tid] = misc[1] + exp(misc[2]) * e_next;
    } else if (a.eps_next) {
        for (int s = tid; s < S; s += FUSED_BLOCK)
            blr_draw_scale(misc[1], misc[2], D, s, a.seed, a.next_step, a.eps_next, a.xi_next);
    }
}

template <int FUSED_BLOCK>
__global__ __launch_bounds__(FUSED_BLOCK) void blr_fused_update_kernel(FusedArgs a) {
    fused_update_role<FUSED_BLOCK, false>(a, (int)blockIdx.x, [] {});
}

// Grid and per-wave trip count: fill the resident wave slots, then balance so
// that every wave runs the same number of (almost all real) tiles.
struct PassGrid {
    int n_blocks;
    int n_iter;
};

// Tile height of the variant that will run: 16 = forward on the MFMA pipe (needs the full
// 256-column layout and 16-byte aligned y), else the VALU kernel with 8- or 4-row tiles.
int pass_rows(const bsc_ctx* ctx, int D, const float* y) {
    if (ctx->blr_tile_rows == 16)
        return (D == GCOLS && (((uintptr_t)y) & 15) == 0) ? 16 : 8;
    return ctx->blr_tile_rows;
}

PassGrid pass_grid(bsc_ctx* ctx, int64_t B, int rows) {
    const int64_t n_tiles = (B + rows - 1) / rows;
    int occ = rows == 4 ? Geo<4>::OCC : 2;
    if (ctx->blr_waves_per_simd > 0 && ctx->blr_waves_per_simd < occ) occ = ctx->blr_waves_per_simd;
    const int64_t max_waves = (int64_t)occ * 4 * ctx->cu_count;
    PassGrid g;
    if (n_tiles <= 0) {
        g.n_blocks = 1;
        g.n_iter = 0;
        return g;
    }
    const int64_t n_iter = (n_tiles + max_waves - 1) / max_waves;
    const int64_t waves = (n_tiles + n_iter - 1) / n_iter;
    g.n_blocks = (int)((waves + PASS_WAVES - 1) / PASS_WAVES);
    g.n_iter = (int)n_iter;
    return g;
}

constexpr int MAX_SLAB_ROWS = 4 * 256 + 64;  // workspace sizing hint for callers

int check_pass_args(const float* X, int64_t ldx, const float* y, int64_t B, int32_t D,
                    const float* W, int32_t S, int max_s) {
    BSC_REQUIRE(B >= 0, "bsc_blr_data_pass: B=%lld", (long long)B);
    BSC_REQUIRE(((X && y) || B == 0) && W, "bsc_blr_data_pass: null pointer");
    BSC_REQUIRE(D > 0 && D <= GCOLS && D % 4 == 0,
                "bsc_blr_data_pass: D=%d must be a multiple of 4 in [4,%d]", D, GCOLS);
    BSC_REQUIRE(S >= 1 && S <= max_s, "bsc_blr_data_pass: S=%d must be in [1,%d]", S, max_s);
    BSC_REQUIRE(ldx >= D && ldx % 4 == 0 && ldx < ((int64_t)1 << 26),
                "bsc_blr_data_pass: ldx=%lld must be >= D, %% 4 == 0 and < 2^26", (long long)ldx);
    BSC_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0,
                "bsc_blr_data_pass: X and W must be 16-byte aligned");
    return BSC_OK;
}

template <int ROWS, bool NT>
void launch_pass_rows(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B,
                      int D, const float* W, int sg, PassGrid g, float* slab) {
    if (D == GCOLS)
        hipLaunchKernelGGL((blr_pass_kernel<true, ROWS, NT>), dim3(g.n_blocks), dim3(PASS_BLOCK), 0,
                           ctx->stream, X, ldx, y, B, D, W, sg, slab, g.n_iter);
    else
        hipLaunchKernelGGL((blr_pass_kernel<false, ROWS, NT>), dim3(g.n_blocks), dim3(PASS_BLOCK),
                           0, ctx->stream, X, ldx, y, B, D, W, sg, slab, g.n_iter);
}

// Windows (one per iteration: n_blocks * 4 tiles of 16 rows) that fit the Infinity Cache.
int keep_windows(const bsc_ctx* ctx, int64_t ldx, PassGrid g) {
    if (ctx->blr_keep >= 0) return ctx->blr_keep < g.n_iter ? ctx->blr_keep : g.n_iter;
    const double window = (double)g.n_blocks * PASS_WAVES * MT_ROWS * (double)ldx * 4.0;
    // MI355X_MICROARCH.md: Infinity Cache 256 MiB; y, the slab and the draws live there too, and
    // tools/ab_pass.py sweep measures 6-8 windows of 33 MB best, 9-10 worse: aim at 7/8 of it
    const double cache = 224.0 * 1024.0 * 1024.0;
    int k = window > 0 ? (int)(cache / window + 0.5) : 0;
    return k < g.n_iter ? k : g.n_iter;
}

// sweep: BSC_SWEEP_STREAM (0) forward, every load non-temporal; BSC_SWEEP_FORWARD_KEEP (1) forward,
// BSC_SWEEP_BACKWARD_KEEP (2) backward, the windows read last left in the Infinity Cache.  The 4-
// and 8-row kernels (D != 256) always stream forward.
// Can this pass carry its finish (FoldArgs)?  Only blr_pass_q_kernel does, and only when the grid leaves the role
// workgroups something to wait for.
bool pass_can_fold(const bsc_ctx* ctx, int D, const float* y, int sg, PassGrid g) {
    return ctx->blr_fold && ctx->fold_counters && pass_rows(ctx, D, y) == 16 && ctx->blr_q && !ctx->blr_mx &&
           !ctx->blr_q_dbg && ctx->blr_nt_loads && sg <= SG && g.n_iter > 0 && g.n_blocks >= 2 * ((SLAB_STRIDE + 63) / 64);
}

void launch_pass(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B, int D,
                 const float* W, int sg, PassGrid g, float* slab, int sweep, bool wide = false,
                 const FoldArgs* fold_in = nullptr) {
    const bool nt = ctx->blr_nt_loads != 0;
    const int rows = pass_rows(ctx, D, y);
    bsc_prof_scope prof(ctx);  // times the pass kernel alone
    if (rows == 16) {
        const int rev = sweep == BSC_SWEEP_BACKWARD_KEEP ? 1 : 0;
        const int keep = sweep == BSC_SWEEP_STREAM ? 0 : keep_windows(ctx, ldx, g);
#define BSC_PASS_MFMA(NT_, PK_)                                                                    \
    hipLaunchKernelGGL((blr_pass_mfma_kernel<NT_, PK_>), dim3(g.n_blocks), dim3(PASS_BLOCK), 0,   \
                       ctx->stream, X, ldx, y, B, W, sg, slab, g.n_iter, rev, keep)
        if (ctx->blr_mx || wide) {
            // both contractions on the MFMA pipe; BSC_BLR_MX = 1 turns a keeping sweep into the rotated
            // cached-zone schedule, 4 re-reads window 0 at every position (the compute floor)
            int mode = rev, rot = 0;
            if (sweep != BSC_SWEEP_STREAM && ctx->blr_mx == 1) { mode = 2; rot = ctx->blr_rot; }
            if (ctx->blr_mx == 4) mode = 4;
#define BSC_PASS_MX(NT_, NSB_)                                                                     \
    hipLaunchKernelGGL((blr_pass_mx_kernel<NT_, NSB_>), dim3(g.n_blocks), dim3(PASS_BLOCK), 0,    \
                       ctx->stream, X, ldx, y, B, W, sg, slab, g.n_iter, mode, keep, rot)
            if (wide && nt) BSC_PASS_MX(true, 4);
            else if (wide) BSC_PASS_MX(false, 4);
            else if (nt) BSC_PASS_MX(true, 2);
            else BSC_PASS_MX(false, 2);
#undef BSC_PASS_MX
        } else if (ctx->blr_q) {
            // both contractions on v_mfma_f32_4x4x1, the tile by LDS-DMA (round 4; option blr_q = 0: the kernels below)
            unsigned long long* stamps = nullptr;
            if (ctx->blr_stamps && !ctx->capturing) {
                if (!ctx->stamps && hipMalloc(&ctx->stamps, (size_t)MAX_SLAB_ROWS * 64) != hipSuccess) ctx->stamps = nullptr;
                stamps = (unsigned long long*)ctx->stamps;
                ctx->stamp_rows = g.n_blocks <= MAX_SLAB_ROWS ? g.n_blocks : 0;
                if (!ctx->stamp_rows) stamps = nullptr;
            }
            // windows of every workgroup / of the even ones (QSched): `blr_q_bias` per mille more for the even ones
            const int64_t n_tiles = (B + MT_ROWS - 1) / MT_ROWS;
            const int64_t w_all = (int64_t)g.n_blocks * PASS_WAVES, w_a = (int64_t)((g.n_blocks + 1) / 2) * PASS_WAVES;
            int extra = (int)((int64_t)g.n_iter * ctx->blr_q_bias + 500) / 1000;
            if (g.n_blocks < 2 || sweep != BSC_SWEEP_STREAM) extra = 0;
            int64_t rest = n_tiles - (int64_t)extra * w_a;
            if (rest < 0) { rest = n_tiles; extra = 0; }
            const int n_all = extra ? (int)((rest + w_all - 1) / w_all) : g.n_iter;
            const int n_a = n_all + extra;
            FoldArgs fold{};
            if (fold_in) fold = *fold_in;
            // the stealing tail (StealArgs): every wave `n_st` tiles of its own, the rest in the queues
            StealArgs steal{};
            int n_st = (int)((n_tiles * (1000 - ctx->blr_steal)) / (1000 * w_all));
            if (ctx->blr_steal > 0 && ctx->steal_heads && nt && sweep == BSC_SWEEP_STREAM && n_st >= 1 && !ctx->blr_q_dbg && !fold.mode) {
                steal.heads = ctx->steal_heads;
                steal.t0 = (long long)n_st * w_all;
                steal.n_dyn = n_tiles - steal.t0;
                // whole groups of 8 workgroups only: a short last group joins queue (groups - 1) % nq -- a queue of its own
                // would hold a full share of tiles for an eighth of the waves
                steal.nq = g.n_blocks / 8 < 1 ? 1 : g.n_blocks / 8 < STEAL_Q ? g.n_blocks / 8 : STEAL_Q;
                steal.per_q = (int)((steal.n_dyn + steal.nq - 1) / steal.nq);
            }
            const int q_all = steal.heads ? n_st : n_all, q_a = steal.heads ? n_st : n_a;
#define BSC_PASS_Q(NT_, DBG_, PRIO_)                                                               \
    hipLaunchKernelGGL((blr_pass_q_kernel<NT_, DBG_, PRIO_>), dim3(g.n_blocks), dim3(PASS_BLOCK), 0, \
                       ctx->stream, X, ldx, y, B, W, sg, slab, q_all, q_a, rev, keep, stamps, fold, steal)
            if (ctx->blr_q_dbg == 1) BSC_PASS_Q(true, 1, 0);
            else if (ctx->blr_q_dbg == 2) BSC_PASS_Q(true, 2, 0);
            else if (ctx->blr_q_dbg == 3) BSC_PASS_Q(true, 3, 0);
            else if (steal.heads && ctx->blr_q_prio == 1)
                hipLaunchKernelGGL((blr_pass_q_kernel<true, 0, 1, true>), dim3(g.n_blocks), dim3(PASS_BLOCK), 0, ctx->stream, X, ldx, y,
                                   B, W, sg, slab, q_all, q_a, rev, keep, stamps, fold, steal);
            else if (steal.heads)
                hipLaunchKernelGGL((blr_pass_q_kernel<true, 0, 0, true>), dim3(g.n_blocks), dim3(PASS_BLOCK), 0, ctx->stream, X, ldx, y,
                                   B, W, sg, slab, q_all, q_a, rev, keep, stamps, fold, steal);
            else if (nt && ctx->blr_q_prio == 1) BSC_PASS_Q(true, 0, 1);
            else if (nt && ctx->blr_q_prio == 2) BSC_PASS_Q(true, 0, 2);
            else if (nt) BSC_PASS_Q(true, 0, 0);
            else BSC_PASS_Q(false, 0, 0);
#undef BSC_PASS_Q
        } else if (ctx->blr_dma && ctx->blr_pk) {
            if (nt) hipLaunchKernelGGL((blr_pass_dma_kernel<true>), dim3(g.n_blocks), dim3(PASS_BLOCK), 0, ctx->stream, X, ldx,
                                       y, B, W, sg, slab, g.n_iter, rev, keep);
            else hipLaunchKernelGGL((blr_pass_dma_kernel<false>), dim3(g.n_blocks), dim3(PASS_BLOCK), 0, ctx->stream, X, ldx,
                                    y, B, W, sg, slab, g.n_iter, rev, keep);
        } else if (nt && ctx->blr_pk) BSC_PASS_MFMA(true, true);
        else if (nt) BSC_PASS_MFMA(true, false);
        else if (ctx->blr_pk) BSC_PASS_MFMA(false, true);
        else BSC_PASS_MFMA(false, false);
#undef BSC_PASS_MFMA
    } else if (rows == 8) {
        if (nt) launch_pass_rows<8, true>(ctx, X, ldx, y, B, D, W, sg, g, slab);
        else launch_pass_rows<8, false>(ctx, X, ldx, y, B, D, W, sg, g, slab);
    } else {
        if (nt) launch_pass_rows<4, true>(ctx, X, ldx, y, B, D, W, sg, g, slab);
        else launch_pass_rows<4, false>(ctx, X, ldx, y, B, D, W, sg, g, slab);
    }
}

int data_pass_impl(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B, int32_t D,
                   const float* W, int32_t S, double* Q, double* G, int sweep) {
    int rc = check_pass_args(X, ldx, y, B, D, W, S, FIN_MAX_S);
    if (rc != BSC_OK) return rc;
    BSC_REQUIRE(Q && G, "bsc_blr_data_pass: null output");
    BSC_REQUIRE(sweep >= 0 && sweep <= 2, "bsc_blr_data_pass: sweep=%d (0, 1 or 2)", sweep);
    const PassGrid g = pass_grid(ctx, B, pass_rows(ctx, D, y));
    void* ws = nullptr;
    rc = bsc_workspace(ctx, (size_t)2 * g.n_blocks * SLAB_STRIDE * sizeof(float), &ws);
    if (rc != BSC_OK) return rc;
    float* slab = (float*)ws;
    ctx->slab_rows = 0;  // the slab is consumed here
    const dim3 rgrid((SLAB_STRIDE + BSC_WAVE - 1) / BSC_WAVE);
    const bool wide_ok = pass_rows(ctx, D, y) == 16 && ctx->blr_wide;
    for (int s0 = 0; s0 < S;) {
        // nine or more draws left: sixteen per pass (blr_pass_mx_kernel<., 4>), two slabs
        const bool wide = wide_ok && S - s0 > SG;
        const int cap = wide ? 2 * SG : SG;
        const int sg = (S - s0 < cap) ? (S - s0) : cap;
        if (!wide && pass_can_fold(ctx, (int)D, y, sg, g)) {
            // the float64 statistics come out of the pass's own tail (FoldArgs mode 2): no reduce launch
            FoldArgs fold{};
            fold.mode = 2; fold.counters = ctx->fold_counters; fold.Q = Q; fold.G = G; fold.s_base = s0; fold.S_total = (int)S;
            launch_pass(ctx, X, ldx, y, B, (int)D, W + (int64_t)s0 * D, sg, g, slab, sweep, false, &fold);
            BSC_LAUNCH_CHECK();
            s0 += sg;
            if (sweep != BSC_SWEEP_STREAM) sweep = 3 - sweep;
            continue;
        }
        launch_pass(ctx, X, ldx, y, B, (int)D, W + (int64_t)s0 * D, sg, g, slab, sweep, wide);
        BSC_LAUNCH_CHECK();
        hipLaunchKernelGGL(blr_slab_reduce_kernel, rgrid, dim3(RED_BLOCK), 0, ctx->stream, slab,
                           g.n_blocks, (int)D, (int)S, s0, Q, G);
        BSC_LAUNCH_CHECK();
        if (sg > SG) {
            hipLaunchKernelGGL(blr_slab_reduce_kernel, rgrid, dim3(RED_BLOCK), 0, ctx->stream,
                               slab + (int64_t)g.n_blocks * SLAB_STRIDE, g.n_blocks, (int)D, (int)S,
                               s0 + SG, Q, G);
            BSC_LAUNCH_CHECK();
        }
        s0 += sg;
        if (sweep != BSC_SWEEP_STREAM) sweep = 3 - sweep;   // the next sample group walks back
    }
    return BSC_OK;
}

int data_pass_partial_impl(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B,
                           int32_t D, const float* W, int32_t S, int sweep) {
    int rc = check_pass_args(X, ldx, y, B, D, W, S, SG);
    if (rc != BSC_OK) return rc;
    BSC_REQUIRE(sweep >= 0 && sweep <= 2, "bsc_blr_data_pass_partial: sweep=%d (0, 1 or 2)", sweep);
    const PassGrid g = pass_grid(ctx, B, pass_rows(ctx, D, y));
    void* ws = nullptr;
    rc = bsc_workspace(ctx, (size_t)2 * g.n_blocks * SLAB_STRIDE * sizeof(float), &ws);
    if (rc != BSC_OK) return rc;
    launch_pass(ctx, X, ldx, y, B, (int)D, W, (int)S, g, (float*)ws, sweep);
    BSC_LAUNCH_CHECK();
    ctx->slab_rows = g.n_blocks;
    return BSC_OK;
}

}  // namespace

extern "C" {

int bsc_blr_pass_count(bsc_ctx* ctx, const float* y, int32_t D, int32_t S, int32_t* count) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(count && S >= 1 && D >= 1, "bsc_blr_pass_count: bad arguments");
    const bool wide_ok = pass_rows(ctx, D, y) == 16 && ctx->blr_wide;
    int n = 0;
    for (int s0 = 0; s0 < S; ++n) {
        const int cap = (wide_ok && S - s0 > SG) ? 2 * SG : SG;
        s0 += (S - s0 < cap) ? (S - s0) : cap;
    }
    *count = n;
    return BSC_OK;
}

int bsc_blr_read_stamps(bsc_ctx* ctx, uint64_t* host_stamps, int32_t capacity_rows, int32_t* host_rows) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(host_stamps && host_rows && capacity_rows >= 0, "bsc_blr_read_stamps: bad arguments");
    BSC_HIP(hipStreamSynchronize(ctx->stream));
    const int n = ctx->stamps ? (ctx->stamp_rows < capacity_rows ? ctx->stamp_rows : capacity_rows) : 0;
    if (n > 0) BSC_HIP(hipMemcpy(host_stamps, ctx->stamps, (size_t)n * 64, hipMemcpyDeviceToHost));
    *host_rows = n;
    return BSC_OK;
}

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

int bsc_blr_noise(bsc_ctx* ctx, int32_t D, int32_t S, uint64_t seed, uint32_t step0,
                  int32_t n_steps, double* eps) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(eps && D > 0 && S > 0 && n_steps > 0, "bsc_blr_noise: bad arguments");
    const int64_t n = (int64_t)(S * ((D + 3) / 4) + S) * n_steps;
    hipLaunchKernelGGL(blr_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, (int)D, (int)S, seed, step0, (int)n_steps, eps);
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

int bsc_blr_data_pass(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B,
                      int32_t D, const float* W, int32_t S, double* Q, double* G) {
    BSC_CHECK_CTX(ctx);
    return data_pass_impl(ctx, X, ldx, y, B, D, W, S, Q, G, BSC_SWEEP_STREAM);
}

int bsc_blr_data_pass_sweep(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B,
                            int32_t D, const float* W, int32_t S, double* Q, double* G,
                            int32_t sweep) {
    BSC_CHECK_CTX(ctx);
    return data_pass_impl(ctx, X, ldx, y, B, D, W, S, Q, G, sweep);
}

int bsc_blr_data_pass_partial(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y,
                              int64_t B, int32_t D, const float* W, int32_t S) {
    BSC_CHECK_CTX(ctx);
    return data_pass_partial_impl(ctx, X, ldx, y, B, D, W, S, BSC_SWEEP_STREAM);
}

int bsc_blr_data_pass_partial_sweep(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y,
                                    int64_t B, int32_t D, const float* W, int32_t S,
                                    int32_t sweep) {
    BSC_CHECK_CTX(ctx);
    return data_pass_partial_impl(ctx, X, ldx, y, B, D, W, S, sweep);
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

namespace {
int fused_update_impl(bsc_ctx* ctx, const char* who, const double* stats, const double* lam_in, double* lam_out,
                      double* m1, double* m2, const double* eps, const float* W, const double* xi, int32_t D,
                      int32_t S, double c0, double c_xi, double s_q, double k_w, double beta, int64_t t, double lr,
                      double beta1, double beta2, double adam_eps, uint64_t seed, uint32_t next_step,
                      double* eps_next, int32_t eps_next_ready, float* W_next, double* xi_next, double* elbo,
                      double* grad, FusedArgs* out_args = nullptr) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(lam_in && lam_out && m1 && m2 && eps && W && xi && elbo && grad, "%s: null pointer", who);
    BSC_REQUIRE(lam_in != lam_out, "%s: lam_in and lam_out must differ", who);
    BSC_REQUIRE(D > 0 && S >= 1 && S <= FIN_MAX_S, "%s: D=%d S=%d (S<=%d)", who, D, S, FIN_MAX_S);
    BSC_REQUIRE(t >= 1, "%s: the Adam step count starts at 1", who);
    BSC_REQUIRE((eps_next && W_next && xi_next) || (!eps_next && !W_next && !xi_next),
                "%s: next-draw buffers must be all set or all null", who);
    BSC_REQUIRE(!eps_next || (eps_next != eps && W_next != W && xi_next != xi),
                "%s: next-draw buffers must not alias the current draws", who);
    FusedArgs a;
    a.stats = stats;
    a.slab = nullptr;
    a.n_slab = 0;
    if (!stats && !out_args) {
        BSC_REQUIRE(ctx->slab_rows > 0 && ctx->workspace,
                    "%s: stats is null and no bsc_blr_data_pass_partial slab is pending", who);
        BSC_REQUIRE(S <= SG && D <= GCOLS, "%s: slab input needs S<=8, D<=256", who);
        a.slab = (const float*)ctx->workspace;
        a.n_slab = ctx->slab_rows;
    }
    a.lam_in = lam_in; a.lam_out = lam_out; a.m1 = m1; a.m2 = m2;
    a.eps = eps; a.W = W; a.xi = xi;
    a.eps_next = eps_next; a.W_next = W_next; a.xi_next = xi_next;
    a.eps_next_ready = eps_next_ready != 0;
    a.elbo = elbo; a.grad = grad;
    a.D = D; a.S = S;
    a.c0 = c0; a.c_xi = c_xi; a.s_q = s_q; a.k_w = k_w; a.beta = beta;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.adam_eps = adam_eps;
    a.corr1 = 1.0 - pow(beta1, (double)t);
    a.corr2 = 1.0 - pow(beta2, (double)t);
    a.seed = seed;
    a.next_step = next_step;
    if (out_args) {             // bsc_blr_pass_update: the pass launches with these (FoldArgs mode 1)
        *out_args = a;
        return BSC_OK;
    }
    {
        bsc_prof_scope prof(ctx, /*slot=*/2);  // the finish kernel, timed apart from the pass
        const dim3 fgrid((D + 7) / 8 + 1);
        if (ctx->blr_finish_block == 256)
            hipLaunchKernelGGL(blr_fused_update_kernel<256>, fgrid, dim3(256), 0, ctx->stream, a);
        else if (ctx->blr_finish_block == 512)
            hipLaunchKernelGGL(blr_fused_update_kernel<512>, fgrid, dim3(512), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL(blr_fused_update_kernel<1024>, fgrid, dim3(1024), 0, ctx->stream, a);
    }
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}
}  // namespace

int bsc_blr_fused_update(bsc_ctx* ctx, const double* stats, const double* lam_in, double* lam_out,
                         double* m1, double* m2, const double* eps, const float* W,
                         const double* xi, int32_t D, int32_t S, double batch_rows, double scale,
                         double alpha0, double beta0, int64_t t, double lr, double beta1,
                         double beta2, double adam_eps, uint64_t seed, uint32_t next_step,
                         double* eps_next, int32_t eps_next_ready, float* W_next, double* xi_next,
                         double* elbo, double* grad) {
    BSC_REQUIRE(alpha0 > 0 && beta0 > 0, "bsc_blr_fused_update: bad hyper-parameters");
    // config 2 (oracle.svi.blr_log_joint) as a member of the family:
    //   scale [-B/2 (log 2 pi + xi) - e Q / 2] - D/2 (log 2 pi + xi) - e |w|^2 / 2 + alpha0 log beta0 - lnGamma(alpha0) - alpha0 xi - beta0 e
    const double half = 0.5 * (scale * batch_rows + (double)D);
    return fused_update_impl(ctx, "bsc_blr_fused_update", stats, lam_in, lam_out, m1, m2, eps, W, xi, D, S,
                             -half * LOG_2PI + alpha0 * log(beta0) - lgamma(alpha0), -half - alpha0, scale, 1.0, beta0,
                             t, lr, beta1, beta2, adam_eps, seed, next_step, eps_next, eps_next_ready, W_next,
                             xi_next, elbo, grad);
}

namespace {
// One update = the pass with its finish folded into its tail when the shape allows it (blr_pass_q_kernel: D = 256,
// S <= 8, a grid of at least 66 workgroups, option blr_fold), else the two launches bsc_blr_data_pass_partial_sweep +
// bsc_blr_fused_update[_general] make.  Same results either way up to the order of the float64 slab sum.
int pass_update_impl(bsc_ctx* ctx, const char* who, const float* X, int64_t ldx, const float* y, int64_t B, int32_t D,
                     int32_t sweep, const double* lam_in, double* lam_out, double* m1, double* m2, const double* eps,
                     const float* W, const double* xi, int32_t S, double c0, double c_xi, double s_q, double k_w,
                     double beta, int64_t t, double lr, double beta1, double beta2, double adam_eps, uint64_t seed,
                     uint32_t next_step, double* eps_next, int32_t eps_next_ready, float* W_next, double* xi_next,
                     double* elbo, double* grad) {
    BSC_CHECK_CTX(ctx);
    int rc = check_pass_args(X, ldx, y, B, D, W, S, SG);
    if (rc != BSC_OK) return rc;
    BSC_REQUIRE(sweep >= 0 && sweep <= 2, "%s: sweep=%d (0, 1 or 2)", who, sweep);
    const PassGrid g = pass_grid(ctx, B, pass_rows(ctx, D, y));
    if (!pass_can_fold(ctx, (int)D, y, (int)S, g)) {
        rc = data_pass_partial_impl(ctx, X, ldx, y, B, D, W, S, sweep);
        if (rc != BSC_OK) return rc;
        return fused_update_impl(ctx, who, nullptr, lam_in, lam_out, m1, m2, eps, W, xi, D, S, c0, c_xi, s_q, k_w, beta, t,
                                 lr, beta1, beta2, adam_eps, seed, next_step, eps_next, eps_next_ready, W_next, xi_next,
                                 elbo, grad);
    }
    FoldArgs fold{};
    rc = fused_update_impl(ctx, who, nullptr, lam_in, lam_out, m1, m2, eps, W, xi, D, S, c0, c_xi, s_q, k_w, beta, t, lr,
                           beta1, beta2, adam_eps, seed, next_step, eps_next, eps_next_ready, W_next, xi_next, elbo, grad,
                           &fold.a);
    if (rc != BSC_OK) return rc;
    void* ws = nullptr;
    rc = bsc_workspace(ctx, (size_t)2 * g.n_blocks * SLAB_STRIDE * sizeof(float), &ws);
    if (rc != BSC_OK) return rc;
    fold.a.slab = (const float*)ws;
    fold.a.n_slab = g.n_blocks;
    fold.mode = 1;
    fold.counters = ctx->fold_counters;
    launch_pass(ctx, X, ldx, y, B, (int)D, W, (int)S, g, (float*)ws, sweep, false, &fold);
    BSC_LAUNCH_CHECK();
    ctx->slab_rows = 0;      // consumed inside the launch
    return BSC_OK;
}
}  // namespace

int bsc_blr_pass_update(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B, int32_t D, int32_t sweep,
                        const double* lam_in, double* lam_out, double* m1, double* m2, const double* eps, const float* W,
                        const double* xi, int32_t S, double batch_rows, double scale, double alpha0, double beta0,
                        int64_t t, double lr, double beta1, double beta2, double adam_eps, uint64_t seed,
                        uint32_t next_step, double* eps_next, int32_t eps_next_ready, float* W_next, double* xi_next,
                        double* elbo, double* grad) {
    BSC_REQUIRE(alpha0 > 0 && beta0 > 0, "bsc_blr_pass_update: bad hyper-parameters");
    const double half = 0.5 * (scale * batch_rows + (double)D);       // (config 2 as a member of the family: bsc_blr_fused_update)
    return pass_update_impl(ctx, "bsc_blr_pass_update", X, ldx, y, B, D, sweep, lam_in, lam_out, m1, m2, eps, W, xi, S,
                            -half * LOG_2PI + alpha0 * log(beta0) - lgamma(alpha0), -half - alpha0, scale, 1.0, beta0, t, lr,
                            beta1, beta2, adam_eps, seed, next_step, eps_next, eps_next_ready, W_next, xi_next, elbo, grad);
}

int bsc_blr_pass_update_general(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B, int32_t D,
                                int32_t sweep, const double* lam_in, double* lam_out, double* m1, double* m2,
                                const double* eps, const float* W, const double* xi, int32_t S, double c0, double c_xi,
                                double s_q, double k_w, double beta, int64_t t, double lr, double beta1, double beta2,
                                double adam_eps, uint64_t seed, uint32_t next_step, double* eps_next,
                                int32_t eps_next_ready, float* W_next, double* xi_next, double* elbo, double* grad) {
    BSC_REQUIRE(s_q >= 0.0 && k_w >= 0.0, "bsc_blr_pass_update_general: s_q=%g k_w=%g must not be negative", s_q, k_w);
    return pass_update_impl(ctx, "bsc_blr_pass_update_general", X, ldx, y, B, D, sweep, lam_in, lam_out, m1, m2, eps, W, xi,
                            S, c0, c_xi, s_q, k_w, beta, t, lr, beta1, beta2, adam_eps, seed, next_step, eps_next,
                            eps_next_ready, W_next, xi_next, elbo, grad);
}

int bsc_blr_fused_update_general(bsc_ctx* ctx, const double* stats, const double* lam_in, double* lam_out,
                                 double* m1, double* m2, const double* eps, const float* W, const double* xi,
                                 int32_t D, int32_t S, double c0, double c_xi, double s_q, double k_w, double beta,
                                 int64_t t, double lr, double beta1, double beta2, double adam_eps, uint64_t seed,
                                 uint32_t next_step, double* eps_next, int32_t eps_next_ready, float* W_next,
                                 double* xi_next, double* elbo, double* grad) {
    BSC_REQUIRE(s_q >= 0.0 && k_w >= 0.0, "bsc_blr_fused_update_general: s_q=%g k_w=%g must not be negative", s_q, k_w);
    return fused_update_impl(ctx, "bsc_blr_fused_update_general", stats, lam_in, lam_out, m1, m2, eps, W, xi, D, S,
                             c0, c_xi, s_q, k_w, beta, t, lr, beta1, beta2, adam_eps, seed, next_step, eps_next,
                             eps_next_ready, W_next, xi_next, elbo, grad);
}

}  // extern "C"
