// _tensordot on the device: C[b,m,n] = sum_k A[b,m,k] B[b,k,n] with free strides.
//
// Reference: _tensordot._apply_to_parents, bayesic/algebra.py:1347-1383
// (T.tensordot / T.batched_dot).  The batched branches there are broken; batched
// contraction follows the einsum semantics (:334-338).
//
// float32: v_mfma_f32_32x32x2_f32 (exact f32, bit-for-bit a k-ordered fmaf chain).
// Block tile 128x128, four waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64
// accumulator registers), K step 16 through LDS.  LDS tiles are k-major
// (As[k][m], Bs[k][n]) so an MFMA operand read is 32 consecutive floats per lane
// half -- conflict-free for ds_read_b32 -- and the global->LDS mapping is chosen
// per operand by which stride is 1, so global loads are coalesced for both
// "m contiguous" (X^T) and "k contiguous" (X) operands.
//
// Split-K: tall-skinny products (X^T X: M=N=256, K=1e6) have 4 output tiles for
// 256 CUs, so K is cut into `splits` ranges, partial tiles go to the workspace
// and a second kernel adds them in split order (deterministic; float atomics
// would be order-dependent and cap at ~1.3 TB/s).
#include "bsc_common.h"
#include "bsc_stream.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDA = BM + 4;  // LDS row stride: 528 B keeps 16-byte rows aligned and staggers banks
constexpr int GEMM_BLOCK = 256;
constexpr int TILE = BK * LDA;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;          // or the partial slab when splits > 1
    int64_t M, N, K;
    int64_t sa_b, sa_m, sa_k, sb_b, sb_k, sb_n, sc_b, sc_m, sc_n;
    int tiles_m, tiles_n, splits;
    int64_t k_chunk;   // multiple of BK
    int vec_a, vec_b;  // 16-byte loads allowed along the operand's contiguous axis
    int fast;          // interior tiles may use scalar-base + 32-bit-lane-offset loads (no per-lane address arithmetic)
    // epilogue (bsc_gemm_epilogue): C = epi_scale * acc^epi_pow * E, epi_pow = 1 | -1, E may be NULL
    // (the _mul consumer / the division by a contraction folded into the store)
    const float* E;
    int64_t se_b, se_m, se_n;
    float epi_scale;
    int epi_pow;
    // gemm_f32_stream_kernel: workgroup w takes the whole tiles w, w + n_wg, ... of `rounds` rounds,
    // then its share of the k-tile units of the `tail_tiles` tiles that are left: units
    // [w q + min(w, r), ...) of their tile-major unit list when `sk_stream`, else the tile
    // rounds n_wg + w whole
    int n_kt, sk_q, sk_r, sk_stream, n_wg, rounds, tail_tiles, tiles_pb, group, dbg;
    // tile index -> (batch, strip, row): divisions by tiles_pb, by group * tiles_m and by the last
    // strip's width as multiply-and-shift on the scalar unit (q = t m >> sh, exact for t < 2^31)
    unsigned mg_pb, mg_strip, mg_last;
    int sh_pb, sh_strip, sh_last, group_log2;
    // sym: C = A B with B = A^T (the Gram statistic X^T X): only the tiles on and above the diagonal are
    // computed (tiles_pb = T (T + 1) / 2 of them), each off-diagonal one stored twice
    int sym;
    // ksplit (EDGE instantiation, one tile, extents <= 64): the four waves share the tile's ONE 64 x 64
    // sub-tile and take one 8-deep k-group of every k-tile each -- C[64 x 16] = R^T X over 10M rows is a
    // quarter of one wave's work per k-tile instead of all of it; each wave's partial goes to the slab
    // as a piece of its own (slot 4 (2 w + which) + wave)
    int ksplit;
    int fix_lanes;     // stream_fixup_kernel: piece lists per element quad (4 or 64)
    float* slab;       // [2 n_wg][128 n][128 m] partial tiles
    // prologue (bsc_gemm_fused, stream kernel only): an element-wise producer applied to the operand's fragments between
    // their LDS read and the MFMAs -- dot(exp(X), Y), dot(X * X, A.T) without the intermediate: 0 none, 1 square,
    // 2 exp, 3 abs.  Operand bytes that lie outside the matrix arrive as zeros; f(0) != 0 (exp) is allowed on ONE
    // side only, the other side's zeros then still cancel the padded products.
    int pre_a, pre_b;
    int nt_c;          // whole-tile stores of C non-temporal (a result far larger than the caches, written once)
};

__device__ __forceinline__ float gemm_epilogue(const GemmArgs& g, float v, int64_t b, int64_t row, int64_t col) {
    if (g.epi_pow < 0) v = 1.0f / v;
    if (g.E) v *= g.E[b * g.se_b + row * g.se_m + col * g.se_n];
    return v * g.epi_scale;
}

// A [BK x 128] operand tile travels global -> registers -> LDS (k-major rows of
// 128).  MN_CONTIG: the operand's m (or n) axis has stride 1, a thread takes 4
// consecutive m of one k (one ds_write_b128); otherwise its k axis has stride 1
// (or nothing has), a thread takes 4 consecutive k of one m (four ds_write_b32).
struct Staged {
    float4 v[4];
};

template <bool MN_CONTIG>
__device__ __forceinline__ void stage_load(Staged& st, const float* __restrict__ src, int64_t s_m,
                                           int64_t s_k, int64_t m0, int64_t M, int64_t k0,
                                           int64_t k_end, int vec, int tid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int64_t gm, gk;
        if (MN_CONTIG) {
            gm = m0 + 4 * (tid & 31);
            gk = k0 + (tid >> 5) + 8 * p;
        } else {
            gk = k0 + 4 * (tid & 7);
            gm = m0 + (tid >> 3) + 32 * p;
        }
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* ptr = src + gm * s_m + gk * s_k;
        const int64_t step = MN_CONTIG ? s_m : s_k;
        const bool inside = MN_CONTIG ? (gm + 3 < M && gk < k_end) : (gk + 3 < k_end && gm < M);
        if (inside && vec) {
            f = *reinterpret_cast<const float4*>(ptr);
        } else {
            const int64_t lim = MN_CONTIG ? M - gm : k_end - gk;
            const bool other = MN_CONTIG ? gk < k_end : gm < M;
            if (other) {
                if (lim > 0) f.x = ptr[0];
                if (lim > 1) f.y = ptr[step];
                if (lim > 2) f.z = ptr[2 * step];
                if (lim > 3) f.w = ptr[3 * step];
            }
        }
        st.v[p] = f;
    }
}

// Interior tiles: the tile base is uniform (SGPRs) and the lane's offset inside the tile
// fits 32 bits, so a load is `global_load_dwordx4 v, v_off, s[base]` with no per-lane
// 64-bit multiplies or bounds tests (those cost the MFMA pipe ~4.6 cycles apiece:
// profiles/r01_ubench_mfma_valu_mix.txt; the general loader spends ~290 per 64-MFMA k-tile).
template <bool MN_CONTIG>
__device__ __forceinline__ void stage_load_fast(Staged& st, const float* __restrict__ tile_base,
                                                int64_t s_m, int64_t s_k, unsigned lane_off) {
    // MN_CONTIG: part p is k + 8p;  else: part p is m + 32p
    const int64_t pstride = (MN_CONTIG ? 8 * s_k : 32 * s_m) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const char* bp = reinterpret_cast<const char*>(tile_base) + p * pstride;
        st.v[p] = *reinterpret_cast<const float4*>(bp + lane_off);
    }
}

template <bool MN_CONTIG>
__device__ __forceinline__ unsigned stage_lane_off(int64_t s_m, int64_t s_k, int tid) {
    if (MN_CONTIG) return (unsigned)((4 * (tid & 31)) * s_m + (tid >> 5) * s_k) * 4u;
    return (unsigned)((tid >> 3) * s_m + (4 * (tid & 7)) * s_k) * 4u;
}

template <bool MN_CONTIG>
__device__ __forceinline__ void stage_store(const Staged& st, float* lds, int tid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (MN_CONTIG) {
            const int m = 4 * (tid & 31), k = (tid >> 5) + 8 * p;
            *reinterpret_cast<float4*>(lds + k * LDA + m) = st.v[p];
        } else {
            const int k = 4 * (tid & 7), m = (tid >> 3) + 32 * p;
            lds[(k + 0) * LDA + m] = st.v[p].x;
            lds[(k + 1) * LDA + m] = st.v[p].y;
            lds[(k + 2) * LDA + m] = st.v[p].z;
            lds[(k + 3) * LDA + m] = st.v[p].w;
        }
    }
}

// The accumulators of one 128 x 128 tile (four waves as 2 x 2, two by two 32 x 32 MFMA tiles each)
// go to C, or to the split's partial slab.  `lds` is the workgroup's operand buffer (>= 128 * LDL
// floats), free once the barrier inside has been passed.
template <int LDL>
__device__ __forceinline__ void gemm_store_tile(const GemmArgs& g, f32x16 (&acc)[2][2], float* lds, int64_t b,
                                                int split, int64_t m0, int64_t n0, int wm, int wn, int lane,
                                                int tid) {
    // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* C;
    int64_t sm, sn;
    if (g.splits > 1) {  // partial slab [batch][split][M][N], dense
        C = g.C + ((b * g.splits + split) * g.M) * g.N;
        sm = g.N;
        sn = 1;
    } else {
        C = g.C + b * g.sc_b;
        sm = g.sc_m;
        sn = g.sc_n;
    }
    const bool epi = g.splits == 1 && g.epi_pow;
    // Whole tiles of an n-contiguous C leave through LDS: the accumulator layout gives a wave-level
    // store 128-byte segments (2 rows x 32 columns); staged row-major in the operand buffers
    // (128 rows of LDL floats) the workgroup writes eight 512-byte rows per instruction, 16 B per lane, and
    // the epilogue factor E is read the same way.  What a short-K product is made of: config 4's
    // dot(Th, Bt) with K = 128 writes 2.5 GB and, with C / . folded in, reads another 2.5 GB.
    const bool vec_e = !g.E || ((g.se_n == 1 && g.se_m % 4 == 0 && (((uintptr_t)(g.E + b * g.se_b)) & 15) == 0) ||
                                g.se_n == 0);
    if (sn == 1 && sm % 4 == 0 && (((uintptr_t)C) & 15) == 0 && m0 + BM <= g.M && n0 + BN <= g.N &&
        (!epi || vec_e)) {
        __syncthreads();                               // the last k-tile's operands are still being read
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    lds[row * LDL + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
                }
        __syncthreads();
        const int c4 = 4 * (tid & 31);
#pragma unroll 4
        for (int p = 0; p < 16; ++p) {
            const int row = 8 * p + (tid >> 5);
            float4 v = *reinterpret_cast<const float4*>(lds + row * LDL + c4);
            if (epi) {
                if (g.epi_pow < 0) { v.x = 1.0f / v.x; v.y = 1.0f / v.y; v.z = 1.0f / v.z; v.w = 1.0f / v.w; }
                float4 ev = make_float4(1.f, 1.f, 1.f, 1.f);
                if (g.E) {
                    const float* ep = g.E + b * g.se_b + (m0 + row) * g.se_m;
                    if (g.se_n == 1) ev = *reinterpret_cast<const float4*>(ep + n0 + c4);
                    else { const float s0 = ep[0]; ev = make_float4(s0, s0, s0, s0); }
                }
                const float sc = g.epi_scale;
                v.x *= ev.x * sc; v.y *= ev.y * sc; v.z *= ev.z * sc; v.w *= ev.w * sc;
            }
            if (g.nt_c) {
                typedef float st_f32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(st_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<st_f32x4*>(C + (m0 + row) * sm + n0 + c4));
            } else {
                *reinterpret_cast<float4*>(C + (m0 + row) * sm + n0 + c4) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (epi) {
                // epilogue: all sixteen E values of this 32x32 block are requested before the first
                // is used (one load, then its store, sixteen times over ran at 1.3 TB/s)
                float e[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const bool live = row < g.M && col < g.N;
                    e[r] = (g.E && live) ? g.E[b * g.se_b + row * g.se_m + col * g.se_n] : 1.0f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[i][j][r];
                    if (g.epi_pow < 0) v = 1.0f / v;
                    if (row < g.M && col < g.N) C[row * sm + col * sn] = v * e[r] * g.epi_scale;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < g.M && col < g.N) C[row * sm + col * sn] = acc[i][j][r];
                }
            }
        }
}

template <bool A_M_CONTIG, bool B_N_CONTIG, bool PIPE>
__global__ __launch_bounds__(GEMM_BLOCK, 2) void gemm_f32_mfma_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE];   // As[2], Bs[2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves
    const int tile = blockIdx.x;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int split = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t k_begin = (int64_t)split * g.k_chunk;
    const int64_t k_end = (k_begin + g.k_chunk < g.K) ? k_begin + g.k_chunk : g.K;
    const float* A = g.A + b * g.sa_b;
    const float* B = g.B + b * g.sb_b;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fk = lane >> 5;  // operand fragment: row/col fr, k offset fk
    const bool fast = g.fast && m0 + BM <= g.M && n0 + BN <= g.N;   // uniform
    const unsigned a_off = stage_lane_off<A_M_CONTIG>(g.sa_m, g.sa_k, tid);
    const unsigned b_off = stage_lane_off<B_N_CONTIG>(g.sb_n, g.sb_k, tid);
    const float* a_tile = A + m0 * g.sa_m;
    const float* b_tile = B + n0 * g.sb_n;
    Staged sa, sb;
    stage_load<A_M_CONTIG>(sa, A, g.sa_m, g.sa_k, m0, g.M, k_begin, k_end, g.vec_a, tid);
    stage_load<B_N_CONTIG>(sb, B, g.sb_n, g.sb_k, n0, g.N, k_begin, k_end, g.vec_b, tid);
    stage_store<A_M_CONTIG>(sa, lds, tid);
    stage_store<B_N_CONTIG>(sb, lds + 2 * TILE, tid);
    __syncthreads();
    int cur = 0;
    // the MFMAs of one K step on LDS buffer `buf`
    auto compute = [&](int buf) {
        const float* As = lds + buf * TILE;
        const float* Bs = lds + 2 * TILE + buf * TILE;
        if (PIPE) {
            // operands one k-pair ahead in registers: the 4 MFMAs of pair kk (256 cycles)
            // cover the LDS latency of pair kk+2 (sched_group_barrier pins that order --
            // left alone the compiler reads, waits, then issues the MFMAs)
            const float* ap = As + fk * LDA + wm * 64 + fr;
            const float* bp = Bs + fk * LDA + wn * 64 + fr;
            float a_n[2] = {ap[0], ap[32]}, b_n[2] = {bp[0], bp[32]};
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a[2] = {a_n[0], a_n[1]}, bq[2] = {b_n[0], b_n[1]};
                if (kk + 2 < BK) {
                    a_n[0] = ap[(kk + 2) * LDA];
                    a_n[1] = ap[(kk + 2) * LDA + 32];
                    b_n[0] = bp[(kk + 2) * LDA];
                    b_n[1] = bp[(kk + 2) * LDA + 32];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float a[2], bq[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = As[(kk + fk) * LDA + wm * 64 + i * 32 + fr];
#pragma unroll
                for (int j = 0; j < 2; ++j) bq[j] = Bs[(kk + fk) * LDA + wn * 64 + j * 32 + fr];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // History (in-process A/B at steady state).  Earlier bounds-free staging variants that kept
    // the per-lane 64-bit address arithmetic were 2-30 % SLOWER than the branchy general loader
    // (profiles/r01_ubench_clock_mfma_peak.txt).  What costs is not the branches but the VALU
    // instructions: ~290 per 64-MFMA k-tile in the general loader, each taking ~4.6 cycles
    // from the MFMA pipe.  stage_load_fast (uniform tile base + one 32-bit lane offset) issues
    // ~10 and is worth +21-31 % (4096^3: 101 -> 126 TF, 8192^3: 101 -> 128 TF, X^T X: 92 -> 121 TF).
    {
        for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
            const bool more = k0 + BK < k_end;
            if (more) {   // next tile's global loads fly during this tile's MFMAs
                if (fast && k0 + 2 * BK <= k_end) {
                    stage_load_fast<A_M_CONTIG>(sa, a_tile + (k0 + BK) * g.sa_k, g.sa_m, g.sa_k, a_off);
                    stage_load_fast<B_N_CONTIG>(sb, b_tile + (k0 + BK) * g.sb_k, g.sb_n, g.sb_k, b_off);
                } else {
                    stage_load<A_M_CONTIG>(sa, A, g.sa_m, g.sa_k, m0, g.M, k0 + BK, k_end, g.vec_a, tid);
                    stage_load<B_N_CONTIG>(sb, B, g.sb_n, g.sb_k, n0, g.N, k0 + BK, k_end, g.vec_b, tid);
                }
            }
            compute(cur);
            if (more) {   // the other buffer was last read one step ago, behind a barrier
                stage_store<A_M_CONTIG>(sa, lds + (cur ^ 1) * TILE, tid);
                stage_store<B_N_CONTIG>(sb, lds + 2 * TILE + (cur ^ 1) * TILE, tid);
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    gemm_store_tile<LDA>(g, acc, lds, b, split, m0, n0, wm, wn, lane, tid);
}

// ---- operands by LDS-DMA -------------------------------------------------------------------
// The same 128 x 128 x 32 tile and 2 x 2 waves, but the operand tiles go global -> LDS with
// `buffer_load_dwordx4 ... lds` (1 KiB per wave instruction): no staging registers, no ds_write
// pass, and -- what pays on this part -- an LDS-DMA takes ~45 cycles from the matrix pipe where
// a VGPR-returning load takes ~115 whatever its width (profiles/r02_ubench_mfma_vmem.txt); the
// register-staged kernel above spends 8 of those per wave and k-tile, ~0.2 of its MFMA time.
//
// A DMA writes lane l's 16 bytes at (wave-uniform LDS base) + 16 l, so an LDS image is shaped by
// which GLOBAL address each lane asks for:
//   operand contiguous along m (or n):  image [k][128] floats; instruction q brings k rows 2q, 2q+1
//     (lane l: row 2q + l/32, columns 4 (l%32) .. +3); fragments are ds_read_b32, a lane's 32
//     neighbours read 32 consecutive floats;
//   operand contiguous along k:  image of 16-byte chunks; instruction q brings the 8 rows x 8 chunks
//     of rows 8q .. 8q+7 into one KiB = four 256-byte bank lines of sixteen 16-byte slots.  A
//     fragment is ONE ds_read_b128 = four k of one row, all lanes of a half-wave at the same chunk
//     index c; the hardware serves a b128 read in the 16-lane groups {0-3,12-15,20-27} and
//     {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS), i.e. one half (rows 0-3 or 4-7) of each of four
//     consecutive 8-row blocks.  Chunk (m, c) therefore sits in line 2 (m/4 % 2) + c/4 of its
//     block at slot 4 ((m/8 % 4) ^ (c % 4)) + ((m % 4) ^ (c % 4)): the four blocks of a group land
//     in four different slot quarters, the four rows of a half in four different slots -- no
//     conflict (the plain 8 m + (c ^ (m & 7)) swizzle is two-way: SQ_LDS_BANK_CONFLICT = half of
//     SQ_LDS_IDX_ACTIVE in profiles/r02_pmc_gemm_stream.txt's first pass).
// Both operands agree on the k a lane holds: in the 8-deep group G, lane half h = lane / 32 feeds
// MFMA t (0..3) with k = 8 G + 4 h + t.
//
// Edges: a lane whose row / column / k lies outside the operand asks for offset 2^31 of a
// descriptor 2^31 bytes long and receives zeros; the masks are per tile (rows, columns) and per
// the last k-tile only.  Host side: the contiguous extent must be a multiple of 4 (a 16-byte
// piece is wholly inside or outside) -- otherwise the register-staged kernel runs.
//
// Schedule per k-tile (four 8-deep groups of 16 MFMAs): fragments one group ahead in registers;
// in the last group, once this wave holds its last fragments of the buffer: wait for the own
// DMAs of the next buffer, barrier (everybody's have landed, nobody still reads this buffer),
// read the next buffer's first fragments, and only then refill this buffer two k-tiles ahead --
// one barrier per k-tile, the DMA a whole k-tile in flight.
constexpr int DMA_STAGE = 32768;            // bytes: A image 16 KiB, B image 16 KiB
constexpr unsigned DMA_OUTSIDE = 0x80000000u;

#define GEMM_LDS_B128(DST, ADDR, OFF) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define GEMM_LDS_B32(DST, ADDR, OFF) \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")

typedef float gemm_f32x4 __attribute__((ext_vector_type(4)));

// byte offset, inside the operand's [128 x 32] tile, that lane `lane` asks for in DMA instruction q
template <bool MN_CONTIG>
__device__ __forceinline__ unsigned dma_lane_offset(int q, int lane, int64_t s_mn, int64_t s_k, int64_t mn_left) {
    if (MN_CONTIG) {
        const int krow = 2 * q + (lane >> 5), mn = 4 * (lane & 31);
        return mn < mn_left ? (unsigned)(krow * s_k + mn) * 4u : DMA_OUTSIDE;
    }
    // lane = position in the block's KiB: line lane / 16, slot quarter lane / 4 % 4, slot lane % 4
    const int c = 4 * ((lane >> 4) & 1) + (((lane >> 2) & 3) ^ (q & 3));
    const int mn = 8 * q + 4 * (lane >> 5) + ((lane & 3) ^ (c & 3));
    return mn < mn_left ? (unsigned)(mn * s_mn + 4 * c) * 4u : DMA_OUTSIDE;
}
// ... and whether its k lies in the `k_left` (< 32) that remain
template <bool MN_CONTIG>
__device__ __forceinline__ bool dma_lane_k_inside(int q, int lane, int k_left) {
    if (MN_CONTIG) return 2 * q + (lane >> 5) < k_left;
    return 4 * (4 * ((lane >> 4) & 1) + (((lane >> 2) & 3) ^ (q & 3))) < k_left;
}

// byte offset of chunk c of row `fr` (0..31) in a k-contiguous operand's image
__device__ __forceinline__ unsigned dma_chunk_offset(int fr, int c) {
    const int line = 2 * ((fr >> 2) & 1) + (c >> 2), slot = 4 * (((fr >> 3) & 3) ^ (c & 3)) + ((fr & 3) ^ (c & 3));
    return (unsigned)((fr >> 3) * 1024 + (line * 16 + slot) * 16);
}

template <bool MN_CONTIG>
__device__ __forceinline__ void dma_read_fragments(float (&f)[2][4], unsigned addr) {
    if (MN_CONTIG) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) GEMM_LDS_B32(f[i][t], addr, t * 512 + i * 128);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            gemm_f32x4 v;
            GEMM_LDS_B128(v, addr, i * 4096);
#pragma unroll
            for (int t = 0; t < 4; ++t) f[i][t] = v[t];
        }
    }
}

template <bool A_M_CONTIG, bool B_N_CONTIG>
__global__ __launch_bounds__(GEMM_BLOCK, 2) void gemm_f32_dma_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * DMA_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int split = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t k_begin = (int64_t)split * g.k_chunk;
    const int64_t k_end = (k_begin + g.k_chunk < g.K) ? k_begin + g.k_chunk : g.K;
    const int64_t n_kt = (k_end - k_begin + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // what each of this wave's 4 + 4 DMA instructions asks for
    unsigned va[4], vb[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        va[jj] = dma_lane_offset<A_M_CONTIG>(4 * wave + jj, lane, g.sa_m, g.sa_k, g.M - m0);
        vb[jj] = dma_lane_offset<B_N_CONTIG>(4 * wave + jj, lane, g.sb_n, g.sb_k, g.N - n0);
    }
    const float* a_tile = g.A + b * g.sa_b + m0 * g.sa_m;
    const float* b_tile = g.B + b * g.sb_b + n0 * g.sb_n;
    auto issue = [&](int buf, int64_t k0) {
        const int64_t k_left = k_end - k0;
        const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a_tile + k0 * g.sa_k), 0, DMA_OUTSIDE, 0x00020000);
        const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_tile + k0 * g.sb_k), 0, DMA_OUTSIDE, 0x00020000);
        char* const dst = lds + buf * DMA_STAGE + wave * 4096;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            unsigned v = va[jj];
            if (k_left < BK && !dma_lane_k_inside<A_M_CONTIG>(4 * wave + jj, lane, k_left)) v = DMA_OUTSIDE;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (bsc_lds_ptr)(dst + jj * 1024), 16, v, 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            unsigned v = vb[jj];
            if (k_left < BK && !dma_lane_k_inside<B_N_CONTIG>(4 * wave + jj, lane, k_left)) v = DMA_OUTSIDE;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (bsc_lds_ptr)(dst + 16384 + jj * 1024), 16, v, 0, 0, 0);
        }
    };

    // fragment addresses of group G in buffer 0 (the other buffer: ^ DMA_STAGE)
    const int fr = lane & 31, fk = lane >> 5;
    const unsigned lbase = (unsigned)(uintptr_t)(bsc_lds_ptr)lds;
    unsigned fa[4], fb[4];
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        fa[G] = lbase + (A_M_CONTIG ? (unsigned)((8 * G + 4 * fk) * 512 + (wm * 64 + fr) * 4)
                                    : (unsigned)(wm * 64 * 128) + dma_chunk_offset(fr, 2 * G + fk));
        fb[G] = lbase + 16384 + (B_N_CONTIG ? (unsigned)((8 * G + 4 * fk) * 512 + (wn * 64 + fr) * 4)
                                            : (unsigned)(wn * 64 * 128) + dma_chunk_offset(fr, 2 * G + fk));
    }

    float oa[2][2][4], ob[2][2][4];     // [register set][32-row block][t]
    auto mfmas = [&](int set) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[set][i][t], ob[set][j][t], acc[i][j], 0, 0, 0);
    };

    issue(0, k_begin);
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    dma_read_fragments<A_M_CONTIG>(oa[0], fa[0]);
    dma_read_fragments<B_N_CONTIG>(ob[0], fb[0]);
    if (n_kt > 1) issue(1, k_begin + BK);
    unsigned tog = 0;
    for (int64_t kt = 0; kt < n_kt; ++kt) {
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            const int set = G & 1;
            __builtin_amdgcn_s_waitcnt(BSC_LGKMCNT0);          // group G's fragments have arrived
            __builtin_amdgcn_sched_barrier(0);
            if (G < 3) {
                dma_read_fragments<A_M_CONTIG>(oa[set ^ 1], fa[G + 1] ^ tog);
                dma_read_fragments<B_N_CONTIG>(ob[set ^ 1], fb[G + 1] ^ tog);
            } else if (kt + 1 < n_kt) {
                __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));  // own DMAs of the next buffer
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                dma_read_fragments<A_M_CONTIG>(oa[set ^ 1], fa[0] ^ tog ^ DMA_STAGE);
                dma_read_fragments<B_N_CONTIG>(ob[set ^ 1], fb[0] ^ tog ^ DMA_STAGE);
                if (kt + 2 < n_kt) issue((int)(kt & 1), k_begin + (kt + 2) * BK);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(set);
            __builtin_amdgcn_sched_barrier(0);
        }
        tog ^= DMA_STAGE;
    }
    gemm_store_tile<BN>(g, acc, reinterpret_cast<float*>(lds), b, split, m0, n0, wm, wn, lane, tid);
}

// ---- persistent stream-K on the LDS-DMA pipeline ---------------------------------------------
// One launch of (at most) two workgroups per CU.  The product's k-tile units -- tile-major,
// `n_kt` per 128 x 128 tile -- are dealt to the workgroups in equal contiguous runs, so the
// grid ends together whatever the tile count (33 x 33 tiles on 512 slots used to cost three
// rounds for 2.1 rounds of work) and the DMA pipeline of a workgroup never drains: the first
// k-tiles of its next tile are in flight while the current tile's last MFMAs issue and its
// accumulators are stored.  A run that covers a tile wholly stores it (with the epilogue);
// the at most two tiles a run shares with its neighbours go to the workgroup's two slab
// slots and `stream_fixup_kernel` adds the pieces of each such tile in k order --
// deterministic, no atomics, no spinning on other workgroups.
//
// Orientation: the MFMA's D layout gives a lane four CONSECUTIVE rows (A-operand index) of one
// column, so the host orients the product with its A operand along C's contiguous axis
// (C = A B is computed as C^T = B^T A^T for a row-major C): stores, epilogue-factor loads and
// slab traffic are all 16 bytes per lane straight from / into the accumulator registers, no
// LDS staging (the ring stays busy with the next tile's operands).  The epilogue factor of a
// whole tile is requested when the tile's first k-tile starts and is long there when the
// last one ends.
//
// Tile order: column strips `group` tiles wide, rows fastest inside a strip, and workgroup
// slots permuted so that each XCD's 64 slots are neighbours in that order: at any moment an
// XCD works on a compact block of tiles and its L2 serves every operand panel to several of them.
__device__ __forceinline__ unsigned stream_magic_div(unsigned t, unsigned m, int sh) {
    return (unsigned)(((uint64_t)t * m) >> sh);
}

template <class GP>
__device__ __forceinline__ void stream_decode_tile(GP g, int t, int64_t& b, int64_t& m0, int64_t& n0) {
    const unsigned tiles_pb = (unsigned)g->tiles_pb, group = (unsigned)g->group;
    const unsigned ub = stream_magic_div((unsigned)t, g->mg_pb, g->sh_pb);             // tiles x batch < 2^31 (host)
    const unsigned tt = (unsigned)t - ub * tiles_pb;
    if (g->sym) {                                   // row-major over the upper triangle of T x T tiles
        unsigned i = 0, rem = tt, len = (unsigned)g->tiles_m;
        while (rem >= len) {
            rem -= len;
            --len;
            ++i;
        }
        b = ub;
        m0 = (int64_t)i * BM;
        n0 = (int64_t)(i + rem) * BN;
        return;
    }
    const unsigned strip = group * (unsigned)g->tiles_m;
    const unsigned s = stream_magic_div(tt, g->mg_strip, g->sh_strip), within = tt - s * strip;
    const unsigned left = (unsigned)g->tiles_n - s * group;
    unsigned tm, gw;
    if (left >= group) {
        gw = group;
        tm = within >> g->group_log2;
    } else {
        gw = left;
        tm = stream_magic_div(within, g->mg_last, g->sh_last);
    }
    b = ub;
    m0 = (int64_t)tm * BM;
    n0 = (int64_t)(s * group + (within - tm * gw)) * BN;
}

typedef stream_args_cptr<GemmArgs> gemm_args_cptr;

// Fragments of one 8-deep k-group for the stream kernel.  A k-contiguous operand: one ds_read_b128
// per 32-row block (rows block * 32 + fr).  An m- or n-contiguous operand: ONE ds_read_b64 per k
// for BOTH blocks -- lane fr takes the two neighbours 2 fr, 2 fr + 1 of the wave's 64, i.e. block i
// holds the rows 2 fr + i (half the LDS instructions of one ds_read_b32 per block and k; the
// accumulator-to-matrix maps below follow: stream_row / stream_col).
#define GEMM_LDS_B64(DST, ADDR, OFF) \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
typedef float gemm_f32x2 __attribute__((ext_vector_type(2)));

template <bool MN_CONTIG>
__device__ __forceinline__ void stream_read_fragments(float (&f)[2][4], unsigned addr) {
    if (MN_CONTIG) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            gemm_f32x2 v;
            GEMM_LDS_B64(v, addr, t * 512);
            f[0][t] = v[0];
            f[1][t] = v[1];
        }
    } else {
        dma_read_fragments<false>(f, addr);
    }
}

// EDGE: the instantiation for products most of whose tiles are partial (an extent below 128, or a
// contraction shorter than a k-tile): MFMA blocks that lie wholly outside the matrix and 8-deep
// k-groups past the end of K are skipped (uniform branches in the k-loop, which the instantiation
// for large matrices does without -- its few edge tiles multiply their zero padding).
__device__ __forceinline__ void gemm_prologue(int op, float (&f)[2][4]) {
    if (op == 0) return;           // (wave-uniform)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float v = f[i][t];
            f[i][t] = op == 1 ? v * v : op == 2 ? __builtin_amdgcn_exp2f(v * 1.4426950408889634f) : __builtin_fabsf(v);
        }
}

template <bool A_M_CONTIG, bool B_N_CONTIG, bool EDGE, bool PRE = false>
__global__ __launch_bounds__(GEMM_BLOCK, 2) void gemm_f32_stream_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * DMA_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool ksplit = EDGE && g.ksplit;
    const int wm = ksplit ? 0 : wave >> 1, wn = ksplit ? 0 : wave & 1;     // (ksplit: every wave on sub-tile (0, 0))
    int w = blockIdx.x;
    int n_units, tail_u0, tail_cnt;
    {
        const gemm_args_cptr gc = stream_cold_args<GemmArgs>();
        if ((gc->n_wg & 7) == 0) w = (w & 7) * (gc->n_wg >> 3) + (w >> 3);     // an XCD's slots are neighbours
        tail_u0 = stream_first_unit(gc, w);
        tail_cnt = stream_first_unit(gc, w + 1) - tail_u0;
        n_units = gc->rounds * gc->n_kt + tail_cnt;            // < 2^31 (host)
    }
    if (n_units == 0) return;
    // what the k-loop itself needs of the arguments
    const int64_t step_a = (int64_t)BK * g.sa_k, step_b = (int64_t)BK * g.sb_k;
    const int last_kt = g.n_kt - 1, k_tail = (int)(g.K - (int64_t)last_kt * BK);   // extent of a tile's last k-tile, 1..32
    const int pre_a = PRE ? g.pre_a : 0, pre_b = PRE ? g.pre_b : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- the issuing side: the unit under `ic` is the next to be requested
    // (always_inline: an out-of-line lambda takes its captures -- cursors, descriptors, the kernel
    // arguments -- through a closure in scratch memory, and everything read back from there is per-lane)
    StreamCursor ic;
    ic.begin(stream_cold_args<GemmArgs>(), w, tail_u0, tail_cnt);
    int issued = 0;
    unsigned va[4], vb[4];
    const float *a_tile, *b_tile;
    auto issue_tile = [&]() __attribute__((always_inline)) {
        const gemm_args_cptr gc = stream_cold_args<GemmArgs>();
        int64_t b, m0, n0;
        stream_decode_tile(gc, ic.t, b, m0, n0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            va[jj] = dma_lane_offset<A_M_CONTIG>(4 * wave + jj, lane, gc->sa_m, gc->sa_k, gc->M - m0);
            vb[jj] = dma_lane_offset<B_N_CONTIG>(4 * wave + jj, lane, gc->sb_n, gc->sb_k, gc->N - n0);
        }
        a_tile = gc->A + b * gc->sa_b + m0 * gc->sa_m + (int64_t)ic.kt * BK * gc->sa_k;
        b_tile = gc->B + b * gc->sb_b + n0 * gc->sb_n + (int64_t)ic.kt * BK * gc->sb_k;
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {
        // a_tile / b_tile run along k with the units of their tile.  (Wave-uniform by construction;
        // said explicitly, or a pointer the compiler chose to keep in vector registers turns every
        // DMA into a readfirstlane loop behind a vmcnt(0).)
        const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)stream_uniform_ptr(a_tile), 0, DMA_OUTSIDE, 0x00020000);
        const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)stream_uniform_ptr(b_tile), 0, DMA_OUTSIDE, 0x00020000);
        char* const dst = lds + buf * DMA_STAGE + wave * 4096;
        if (__builtin_expect(ic.kt != last_kt || k_tail == BK, 1)) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (bsc_lds_ptr)(dst + jj * 1024), 16, va[jj], 0, 0, 0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (bsc_lds_ptr)(dst + 16384 + jj * 1024), 16, vb[jj], 0, 0, 0);
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const unsigned v = dma_lane_k_inside<A_M_CONTIG>(4 * wave + jj, lane, k_tail) ? va[jj] : DMA_OUTSIDE;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (bsc_lds_ptr)(dst + jj * 1024), 16, v, 0, 0, 0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const unsigned v = dma_lane_k_inside<B_N_CONTIG>(4 * wave + jj, lane, k_tail) ? vb[jj] : DMA_OUTSIDE;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (bsc_lds_ptr)(dst + 16384 + jj * 1024), 16, v, 0, 0, 0);
            }
        }
        ++issued;
        a_tile += step_a;
        b_tile += step_b;
        if (__builtin_expect(ic.step(), 0) && issued < n_units) {
            ++ic.round;
            ic.segment(stream_cold_args<GemmArgs>(), w, tail_u0);
            issue_tile();
        }
    };

    const int fr = lane & 31, fk = lane >> 5;
    const unsigned lbase = (unsigned)(uintptr_t)(bsc_lds_ptr)lds;
    unsigned fa[4], fb[4];
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        fa[G] = lbase + (A_M_CONTIG ? (unsigned)((8 * G + 4 * fk) * 512 + (wm * 64 + 2 * fr) * 4)
                                    : (unsigned)(wm * 64 * 128) + dma_chunk_offset(fr, 2 * G + fk));
        fb[G] = lbase + 16384 + (B_N_CONTIG ? (unsigned)((8 * G + 4 * fk) * 512 + (wn * 64 + 2 * fr) * 4)
                                            : (unsigned)(wn * 64 * 128) + dma_chunk_offset(fr, 2 * G + fk));
    }
    float oa[2][2][4], ob[2][2][4];
    auto mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[set][i][t], ob[set][j][t], acc[i][j], 0, 0, 0);
    };

    // (a tile at the edge of the matrix, or a matrix with an extent below 128: the blocks that lie wholly
    // outside -- zero operands -- are skipped; C[10M x 64] = X[10M x 16] T[16 x 64] is half outside)
    auto mfmas_masked = [&](int set, int mask) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (mask & (1 << (2 * i + j))) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[set][i][t], ob[set][j][t], acc[i][j], 0, 0, 0);
                }
    };

    // ---- the computing side
    StreamCursor cc;
    cc.begin(stream_cold_args<GemmArgs>(), w, tail_u0, tail_cnt);
    bool tile_start = true, whole = false, fast = false, interior = false;
    int64_t cb = 0, cm0 = 0, cn0 = 0;                     // the tile under `cc`
    int m_left = 0, n_left = 0;                           // rows / columns of this wave's 64 x 64 sub-tile inside the matrix
    int blk_mask = 15;                                    // bit 2 i + j: MFMA block (i, j) of the sub-tile holds anything
    // Where the accumulators live in the wave's 64 x 64 sub-tile.  acc[i][j][r] is MFMA row
    // rho = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31, of block (i, j); an m-contiguous A
    // interleaves its two row blocks (row 2 rho + i), a k-contiguous one stacks them (32 i + rho);
    // the same for B and the columns.  Either way a lane owns, per column, eight runs of four
    // consecutive rows -- the 16-byte pieces of the stores, the slab and the epilogue factor:
    //   piece p of column block j: rows lane_m + piece_m(p) .. + 3, from acc[piece_i(p, e)][j][piece_r(p, e)]
    auto piece_m = [](int p) { return A_M_CONTIG ? 16 * (p >> 1) + 4 * (p & 1) : 32 * (p >> 2) + 8 * (p & 3); };
    auto piece_i = [](int p, int e) { return A_M_CONTIG ? (e & 1) : (p >> 2); };
    auto piece_r = [](int p, int e) { return A_M_CONTIG ? 4 * (p >> 1) + 2 * (p & 1) + (e >> 1) : 4 * (p & 3) + e; };
    const int lane_m = (A_M_CONTIG ? 8 : 4) * (lane >> 5);
    const int lane_n = (B_N_CONTIG ? 2 : 1) * (lane & 31);
    constexpr int BLK_N = B_N_CONTIG ? 1 : 32;               // column offset of block j = 1
    // the epilogue factor's registers: one value chain from here on (the asm loads below update them
    // in place), so that no copy of a not-yet-arrived register is ever made where control flow joins
    gemm_f32x4 ev[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) ev[j][p8] = gemm_f32x4{1.f, 1.f, 1.f, 1.f};

    issue_tile();
    issue(0);
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stream_read_fragments<A_M_CONTIG>(oa[0], fa[0]);
    stream_read_fragments<B_N_CONTIG>(ob[0], fb[0]);
    if (n_units > 1) issue(1);
    unsigned tog = 0;
    // vector-memory operations younger than the DMAs the next wait is for: the stores of the tile
    // that just ended and the epilogue-factor loads of the tile that just began.  Completion is in
    // issue order, so vmcnt(young) retires those DMAs without waiting for a store's acknowledgement.
    int young = 0;
    for (int u = 0; u < n_units; ++u) {
        if (__builtin_expect(tile_start, 0)) {
            tile_start = false;
            const gemm_args_cptr gc = stream_cold_args<GemmArgs>();
            stream_decode_tile(gc, cc.t, cb, cm0, cn0);
            whole = cc.kt == 0 && cc.left == gc->n_kt && !ksplit;      // (ksplit: every wave holds a partial)
            const float* E = gc->E;
            const bool c_vec = gc->sc_m == 1 && gc->sc_n % 4 == 0 && gc->sc_b % 4 == 0 && (((uintptr_t)gc->C) & 15) == 0 && gc->M % 4 == 0;
            const bool e_vec = !E || (gc->se_m == 1 && gc->se_n % 4 == 0 && gc->se_b % 4 == 0 && (((uintptr_t)E) & 15) == 0);
            interior = cm0 + BM <= gc->M && cn0 + BN <= gc->N;
            // 16-byte pieces straight from the accumulators: whole tiles; at the edge of the matrix each
            // piece behind its own in-range test (M % 4 = 0: a piece is wholly inside or outside) -- but
            // the epilogue factor is prefetched for interior tiles only
            fast = whole && c_vec && e_vec && (interior || !(gc->epi_pow != 0 && E));
            {
                const int64_t ml = gc->M - cm0 - wm * 64, nl = gc->N - cn0 - wn * 64;
                m_left = (int)(ml < 0 ? 0 : ml > 64 ? 64 : ml);
                n_left = (int)(nl < 0 ? 0 : nl > 64 ? 64 : nl);
                blk_mask = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if ((A_M_CONTIG ? i : 32 * i) < m_left && (B_N_CONTIG ? j : 32 * j) < n_left) blk_mask |= 1 << (2 * i + j);
            }
            if (fast && gc->epi_pow != 0 && E) {
                // the epilogue factor of this tile, 16 x 16 bytes per lane, by loads the compiler does
                // not track (it would drain the DMA ring at their use): they are older than the DMAs
                // waited for in the tile's second k-tile (or than the vmcnt(0) before its stores)
                const int se_n = (int)gc->se_n;
                const unsigned e_lane = (unsigned)((wn * 64 + lane_n) * se_n + wm * 64 + lane_m) * 4u;
                const float* e_tile = E + cb * gc->se_b + cn0 * se_n + cm0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float* base = e_tile + (int64_t)(j * BLK_N) * se_n;
#pragma unroll
                    for (int p8 = 0; p8 < 8; ++p8)
                        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"
                                     : "+v"(ev[j][p8]) : "v"(e_lane), "s"(base), "n"(piece_m(p8) * 4) : "memory");
                }
                young += 16;
            }
        }
        // the 8-deep groups of this k-tile that hold anything (a tile's last k-tile may be short)
        const int k_groups = cc.kt == last_kt ? (k_tail + 7) >> 3 : 4;
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            const int set = G & 1;
            __builtin_amdgcn_s_waitcnt(BSC_LGKMCNT0);
            __builtin_amdgcn_sched_barrier(0);
            if (G < 3) {
                stream_read_fragments<A_M_CONTIG>(oa[set ^ 1], fa[G + 1] ^ tog);
                stream_read_fragments<B_N_CONTIG>(ob[set ^ 1], fb[G + 1] ^ tog);
            } else if (u + 1 < n_units) {
                if (young == 0) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
                else if (young == 16) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(16));
                else __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(32));
                young = 0;
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                stream_read_fragments<A_M_CONTIG>(oa[set ^ 1], fa[0] ^ tog ^ DMA_STAGE);
                stream_read_fragments<B_N_CONTIG>(ob[set ^ 1], fb[0] ^ tog ^ DMA_STAGE);
                if (u + 2 < n_units) issue(u & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PRE) {     // (fragment set `set` landed before this group's wait; the other set is in flight)
                gemm_prologue(pre_a, oa[set]);
                gemm_prologue(pre_b, ob[set]);
            }
            if (!EDGE) mfmas(set);
            else if (G < k_groups && (!ksplit || G == wave)) mfmas_masked(set, blk_mask);
            __builtin_amdgcn_sched_barrier(0);
        }
        tog ^= DMA_STAGE;
        if (__builtin_expect(cc.step(), 0)) {
            // ---- this run's share of tile cc.t is complete
            const gemm_args_cptr gc = stream_cold_args<GemmArgs>();
            // the epilogue factor is older than the DMAs waited for in this tile's second k-tile; a tile
            // one k-tile long, or the run's last, waits here
            if (gc->n_kt == 1 || u + 1 == n_units) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
            const bool mirror = gc->sym && cm0 != cn0;           // an off-diagonal tile of a symmetric product: stored twice
            if (!whole || (fast && interior && !mirror)) young += 16;    // 16 stores per lane below (edge / mirrored tiles: another count, wait for all)
            else young = 0;
            if (!whole) {
                const int64_t piece = (int64_t)2 * w + (cc.round > gc->rounds ? 1 : 0);
                float* slot = gc->slab + (ksplit ? 4 * piece + wave : piece) * (BM * BN);
                float* p = slot + (wn * 64 + lane_n) * BM + wm * 64 + lane_m;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p8 = 0; p8 < 8; ++p8) {
                        gemm_f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[piece_i(p8, e)][j][piece_r(p8, e)];
                        *reinterpret_cast<gemm_f32x4*>(p + j * BLK_N * BM + piece_m(p8)) = v;
                    }
            } else if (fast) {
                const int sc_n = (int)gc->sc_n, epi_pow = gc->epi_pow, dbg = gc->dbg;
                const float epi_scale = gc->epi_scale;
                const bool has_e = gc->E != nullptr, nt_c = gc->nt_c != 0;
                float* c_tile = gc->C + cb * gc->sc_b + cn0 * sc_n + cm0;
                const unsigned c_lane = (unsigned)((wn * 64 + lane_n) * sc_n + wm * 64 + lane_m) * 4u;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p8 = 0; p8 < 8; ++p8) {
                        gemm_f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[piece_i(p8, e)][j][piece_r(p8, e)];
                        if (epi_pow) {
                            if (epi_pow < 0) {
                                // v_rcp_f32 and one Newton step (3 instructions; the IEEE division
                                // sequence is ~10, 640 per tile -- a fifth of a K = 128 tile's MFMA time)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float x = v[e], r = __builtin_amdgcn_rcpf(x);
                                    v[e] = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
                                }
                            }
                            if (has_e) {
                                // (a use the scheduler cannot lift above the vmcnt wait that makes
                                // the asm-loaded registers valid)
                                asm volatile("" : "+v"(ev[j][p8]));
                                v *= ev[j][p8];
                            }
                            v *= epi_scale;
                        }
                        char* base = reinterpret_cast<char*>(c_tile + (int64_t)(j * BLK_N) * sc_n + piece_m(p8));
                        if (!(dbg & 1) && (interior || (lane_m + piece_m(p8) < m_left && lane_n + j * BLK_N < n_left))) {
                            if (nt_c) __builtin_nontemporal_store(v, reinterpret_cast<gemm_f32x4*>(base + c_lane));
                            else *reinterpret_cast<gemm_f32x4*>(base + c_lane) = v;
                            if (mirror) {                // C[n][m] = C[m][n]: the quad's rows become columns
                                float* mt = gc->C + cb * gc->sc_b + (cm0 + wm * 64 + lane_m + piece_m(p8)) * sc_n +
                                            (cn0 + wn * 64 + lane_n + j * BLK_N);
#pragma unroll
                                for (int e = 0; e < 4; ++e) mt[(int64_t)e * sc_n] = v[e];
                            }
                        }
                    }
            } else {
                const int64_t M = gc->M, N = gc->N, sc_m = gc->sc_m, sc_n = gc->sc_n, se_m = gc->se_m, se_n = gc->se_n;
                const int epi_pow = gc->epi_pow;
                const float epi_scale = gc->epi_scale;
                const float* E = gc->E ? gc->E + cb * gc->se_b : nullptr;
                float* C = gc->C + cb * gc->sc_b;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int64_t col = cn0 + wn * 64 + (B_N_CONTIG ? 2 * (lane & 31) + j : 32 * j + (lane & 31));
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if ((r & 3) == 0) asm volatile("" ::: "memory");      // four loads in flight, not sixty-four (registers)
                            const int rho = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            const int64_t row = cm0 + wm * 64 + (A_M_CONTIG ? 2 * rho + i : 32 * i + rho);
                            if (row < M && col < N) {
                                float v = acc[i][j][r];
                                if (epi_pow) {
                                    if (epi_pow < 0) v = 1.0f / v;
                                    if (E) v *= E[row * se_m + col * se_n];
                                    v *= epi_scale;
                                }
                                C[row * sc_m + col * sc_n] = v;
                                if (mirror) C[col * sc_m + row * sc_n] = v;
                            }
                        }
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            ++cc.round;
            cc.segment(gc, w, tail_u0);
            tile_start = true;
        }
    }
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
}

// The tiles that two or more runs share.  One block column per TAIL tile: the tile is split when its
// first and last unit belong to different workgroups; its pieces -- workgroups owner(first unit) ..
// owner(last unit), in that order -- are added, the epilogue applied, the result stored.  Slab slots
// are [n][m] like the accumulators: 16 bytes per lane along m.
__device__ __forceinline__ int stream_owner(const GemmArgs& g, int u) {          // (sk_stream only)
    const int big = g.sk_r * (g.sk_q + 1);                   // units held by the sk_r workgroups with one more
    return u < big ? u / (g.sk_q + 1) : g.sk_r + (u - big) / g.sk_q;
}

__global__ __launch_bounds__(256) void stream_fixup_kernel(GemmArgs g) {
    // PL piece lists per element quad (4, or 64 when a tile was cut into hundreds of pieces): list
    // position p goes to lane p % PL, each lane adds its pieces in order with four loads in flight, the
    // lanes' sums are added in lane order -- a fixed association, run to run
    __shared__ gemm_f32x4 part[256];
    const int PL = g.fix_lanes, QB = 256 / PL;                   // quads per block
    const int t = (int)blockIdx.x, t_begin = t * g.n_kt, t_end = t_begin + g.n_kt;    // tail tile t
    const int x_first = stream_owner(g, t_begin), x_last = stream_owner(g, t_end - 1);
    if (x_first == x_last) return;                               // one workgroup had all of it: stored there
    const int quad = threadIdx.x % QB, lane4 = threadIdx.x / QB;
    const int e = (blockIdx.y * QB + quad) * 4;                   // element of the [128 n][128 m] tile
    const int n = e >> 7, m = e & 127;
    int64_t b, m0, n0;
    stream_decode_tile(&g, g.rounds * g.n_wg + t, b, m0, n0);
    const int64_t col = n0 + n;
    const bool inside = col < g.N && m0 + m < g.M;              // (elements outside the matrix are not even read)
    gemm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (inside) {
        // workgroup x's piece of this tile: its first tail tile -> slot 0, a later one -> slot 1
        const int subs = g.ksplit ? 4 : 1;                        // (ksplit: a workgroup left four pieces, one per wave)
        const int n_pieces = (x_last - x_first + 1) * subs;
        const float* piece[4];
        int n_piece = 0;
        for (int p = lane4; p < n_pieces; p += PL) {
            const int x = x_first + p / subs, sub = p - (p / subs) * subs;
            const int64_t base = (int64_t)2 * x + (stream_first_unit(&g, x) / g.n_kt != t ? 1 : 0);
            piece[n_piece++] = g.slab + (g.ksplit ? 4 * base + sub : base) * (BM * BN) + e;
            if (n_piece == 4) {
                gemm_f32x4 q4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) q4[i] = *reinterpret_cast<const gemm_f32x4*>(piece[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) v += q4[i];
                n_piece = 0;
            }
        }
        for (int i = 0; i < n_piece; ++i) v += *reinterpret_cast<const gemm_f32x4*>(piece[i]);
    }
    part[lane4 * QB + quad] = v;
    __syncthreads();
    if (lane4 > 0 || !inside) return;
    for (int l = 1; l < PL; ++l) v += part[l * QB + quad];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t row = m0 + m + q;
        if (row < g.M) {
            float c = v[q];
            if (g.epi_pow) c = gemm_epilogue(g, c, b, row, col);
            g.C[b * g.sc_b + row * g.sc_m + col * g.sc_n] = c;
            if (g.sym && m0 != n0) g.C[b * g.sc_b + col * g.sc_m + row * g.sc_n] = c;
        }
    }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int64_t M,
                                     int64_t N, float* __restrict__ C, int64_t sc_b, int64_t sc_m,
                                     int64_t sc_n, GemmArgs g) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (idx >= M * N) return;
    const float* p = slab + (b * splits) * M * N + idx;
    float v = p[0];
    for (int s = 1; s < splits; ++s) v += p[(int64_t)s * M * N];
    if (g.epi_pow) v = gemm_epilogue(g, v, b, idx / N, idx % N);
    C[b * sc_b + (idx / N) * sc_m + (idx % N) * sc_n] = v;
}

// Matrix-vector shapes (N == 1 after orienting): y[m] = sum_k A[m,k] x[k].  Pure
// streaming: the matrix is read once.  Two access patterns, chosen by which stride
// of A is 1, both with float64 accumulation and a fixed-order two-stage finish:
//   K_CONTIG  (A[m, :] contiguous, e.g. X w):   a wave per row group, lanes along k;
//   M_CONTIG  (A[:, k] contiguous, e.g. X^T y): lane <-> output m, waves split k.
struct GemvArgs {
    const float* A;
    const float* x;
    float* y;
    double* partial;   // [splits][M]
    int64_t M, K, sa_m, sa_k, sx, sy;
    int splits;
};

__global__ __launch_bounds__(256) void gemv_kcontig_kernel(GemvArgs g) {
    // one wave per output row; K is contiguous: 16-byte loads when aligned
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= g.M) return;
    const float* row = g.A + m * g.sa_m;
    double acc = 0.0;
    const bool vec = g.sa_k == 1 && g.sx == 1 && (((uintptr_t)row | (uintptr_t)g.x) & 15) == 0;
    int64_t k0 = 0;
    if (vec) {
        const int64_t k4 = g.K / 4;
        for (int64_t i = lane; i < k4; i += 64) {
            const float4 a = reinterpret_cast<const float4*>(row)[i];
            const float4 b = reinterpret_cast<const float4*>(g.x)[i];
            acc += (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z + (double)a.w * b.w;
        }
        k0 = k4 * 4;
    }
    for (int64_t k = k0 + lane; k < g.K; k += 64) acc += (double)row[k * g.sa_k] * g.x[k * g.sx];
    acc = wave_allsum_f64(acc);
    if (lane == 0) g.y[m * g.sy] = (float)acc;
}

__global__ __launch_bounds__(256) void gemv_mcontig_kernel(GemvArgs g) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t groups = (g.M + 63) / 64;
    const int64_t grp = blockIdx.x % groups;
    const int split = (int)(blockIdx.x / groups);
    const int64_t m = grp * 64 + lane;
    const int64_t chunk = (g.K + g.splits - 1) / g.splits;
    const int64_t k0 = split * chunk, k1 = (k0 + chunk < g.K) ? k0 + chunk : g.K;
    double acc = 0.0;
    if (m < g.M) {
        const float* col = g.A + m * g.sa_m;
        constexpr int U = 8;   // loads in flight per lane
        int64_t k = k0 + wave;
        for (; k + 4 * (U - 1) < k1; k += 4 * U) {
            float a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a[u] = col[(k + 4 * u) * g.sa_k];
                b[u] = g.x[(k + 4 * u) * g.sx];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += (double)a[u] * b[u];
        }
        for (; k < k1; k += 4) acc += (double)col[k * g.sa_k] * g.x[k * g.sx];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && m < g.M) {
        const double tot = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        if (g.splits > 1) g.partial[(int64_t)split * g.M + m] = tot;
        else g.y[m * g.sy] = (float)tot;
    }
}

// m contiguous, 16-byte loads: lane <-> 4 consecutive outputs (a wave covers 256 of them per
// matrix row), the four waves of a block and the splits interleave over the rows so that the
// whole grid reads one moving window of memory; x[k] is one broadcast load per row.
typedef float gemv_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gemv_mcontig4_kernel(GemvArgs g) {
    __shared__ double red[4][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t groups = (g.M + 255) / 256;
    const int64_t grp = blockIdx.x % groups;
    const int split = (int)(blockIdx.x / groups);
    const int64_t m = grp * 256 + 4 * lane;           // M % 4 == 0 (host)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (m < g.M) {
        constexpr int U = 4;                           // rows in flight per lane
        const float* base = g.A + m;
        for (int64_t kb = (int64_t)split * 4 * U + wave; kb < g.K; kb += (int64_t)g.splits * 4 * U) {
            gemv_f32x4 a[U];
            float xk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = kb + 4 * u < g.K ? kb + 4 * u : g.K - 1;   // clamped, dropped below
                a[u] = __builtin_nontemporal_load(reinterpret_cast<const gemv_f32x4*>(base + k * g.sa_k));
                xk[u] = g.x[k * g.sx];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (kb + 4 * u < g.K) {
                    const double xv = (double)xk[u];
                    acc[0] += (double)a[u].x * xv; acc[1] += (double)a[u].y * xv;
                    acc[2] += (double)a[u].z * xv; acc[3] += (double)a[u].w * xv;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave][lane][c] = acc[c];
    __syncthreads();
    if (wave == 0 && m < g.M) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double tot = ((red[0][lane][c] + red[1][lane][c]) + red[2][lane][c]) + red[3][lane][c];
            if (g.splits > 1) g.partial[(int64_t)split * g.M + m + c] = tot;
            else g.y[(m + c) * g.sy] = (float)tot;
        }
    }
}

// one wave per output: lanes stride over the split partials (a serial loop per output took
// longer than the streaming kernel itself when there are hundreds of splits)
__global__ __launch_bounds__(256) void gemv_finish_wave_kernel(const double* partial, int splits,
                                                               int64_t M, float* y, int64_t sy) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    double tot = 0.0;
    int s = lane;
    for (; s + 192 < splits; s += 256) {
        double v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = partial[(int64_t)(s + 64 * j) * M + m];
#pragma unroll
        for (int j = 0; j < 4; ++j) tot += v[j];
    }
    for (; s < splits; s += 64) tot += partial[(int64_t)s * M + m];
    tot = wave_allsum_f64(tot);
    if (lane == 0) y[m * sy] = (float)tot;
}

__global__ void gemv_finish_kernel(const double* partial, int splits, int64_t M, float* y,
                                   int64_t sy) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    double tot = 0.0;
    for (int s = 0; s < splits; ++s) tot += partial[(int64_t)s * M + m];
    y[m * sy] = (float)tot;
}

// float64 (and the degenerate shapes): one thread per output, k-ordered fma chain.
template <typename T>
__global__ void gemm_naive_kernel(int64_t M, int64_t N, int64_t K, const T* A, int64_t sa_b,
                                  int64_t sa_m, int64_t sa_k, const T* B, int64_t sb_b,
                                  int64_t sb_k, int64_t sb_n, T* C, int64_t sc_b, int64_t sc_m,
                                  int64_t sc_n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (idx >= M * N) return;
    const int64_t m = idx / N, n = idx % N;
    const T* a = A + b * sa_b + m * sa_m;
    const T* bb = B + b * sb_b + n * sb_n;
    T acc = 0;
    for (int64_t k = 0; k < K; ++k) acc = fma(a[k * sa_k], bb[k * sb_k], acc);
    C[b * sc_b + m * sc_m + n * sc_n] = acc;
}

}  // namespace

int bsc_gram_split(bsc_ctx* ctx, const float* X, int64_t ldx, int64_t N, int64_t D, float* C, int64_t sc_m, int64_t sc_n,
                   float scale, int* handled);      // csrc/bsc_gram.hip

struct Epilogue {
    int pow = 0;            // 0: none
    float scale = 1.f;
    const float* E = nullptr;
    int64_t se_b = 0, se_m = 0, se_n = 0;
    int pre_a = 0, pre_b = 0;     // operand prologues (bsc_gemm_fused); *handled = 0 when the shape takes a path without them
    int* handled = nullptr;
};

static int gemm_impl(bsc_ctx* ctx, int dtype, int64_t batch, int64_t M, int64_t N,
              int64_t K, const void* A, int64_t sa_b, int64_t sa_m, int64_t sa_k,
              const void* B, int64_t sb_b, int64_t sb_k, int64_t sb_n, void* C,
              int64_t sc_b, int64_t sc_m, int64_t sc_n, const Epilogue& epi) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(dtype == BSC_F32 || dtype == BSC_F64, "bsc_gemm_strided_batched: unknown dtype %d",
                dtype);
    BSC_REQUIRE(batch >= 0 && M >= 0 && N >= 0 && K >= 0, "bsc_gemm_strided_batched: negative extent");
    if (batch == 0 || M == 0 || N == 0) return BSC_OK;
    BSC_REQUIRE(C && ((A && B) || K == 0), "bsc_gemm_strided_batched: null pointer");
    if (batch > 65535) {  // the batch index is a grid dimension: run 65535 batches per launch
        const size_t es = dtype == BSC_F64 ? 8 : 4;
        for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
            const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
            Epilogue e2 = epi;
            if (e2.E) e2.E += b0 * e2.se_b;
            int rc = gemm_impl(
                ctx, dtype, nb, M, N, K, A ? (const char*)A + (size_t)(b0 * sa_b) * es : nullptr, sa_b,
                sa_m, sa_k, B ? (const char*)B + (size_t)(b0 * sb_b) * es : nullptr, sb_b, sb_k, sb_n,
                (char*)C + (size_t)(b0 * sc_b) * es, sc_b, sc_m, sc_n, e2);
            if (rc != BSC_OK) return rc;
        }
        return BSC_OK;
    }
    const bool pre = epi.pre_a != 0 || epi.pre_b != 0;
    if (epi.handled) *epi.handled = 0;
    if (pre && (dtype == BSC_F64 || K == 0 || batch > 65535)) return BSC_OK;       // (not handled)
    if (epi.pow && (dtype == BSC_F64 || K == 0))
        return bsc_fail(BSC_ERR_UNSUPPORTED, "bsc_gemm_epilogue: float32 products with K > 0 only");
    if (dtype == BSC_F64 || K == 0) {
        const dim3 grid((unsigned)((M * N + 255) / 256), (unsigned)batch);
        if (dtype == BSC_F64)
            hipLaunchKernelGGL(gemm_naive_kernel<double>, grid, dim3(256), 0, ctx->stream, M, N, K,
                               (const double*)A, sa_b, sa_m, sa_k, (const double*)B, sb_b, sb_k,
                               sb_n, (double*)C, sc_b, sc_m, sc_n);
        else
            hipLaunchKernelGGL(gemm_naive_kernel<float>, grid, dim3(256), 0, ctx->stream, M, N, K,
                               (const float*)A, sa_b, sa_m, sa_k, (const float*)B, sb_b, sb_k, sb_n,
                               (float*)C, sc_b, sc_m, sc_n);
        BSC_LAUNCH_CHECK();
        return BSC_OK;
    }
    if (batch == 1 && (N == 1 || M == 1) && K >= 64 && !epi.pow && !pre) {
        // matrix-vector: orient so that the matrix is "A[m,k]" and the vector "x[k]"
        GemvArgs v;
        if (N == 1) {
            v.A = (const float*)A; v.M = M; v.sa_m = sa_m; v.sa_k = sa_k;
            v.x = (const float*)B; v.sx = sb_k; v.sy = sc_m;
        } else {
            v.A = (const float*)B; v.M = N; v.sa_m = sb_n; v.sa_k = sb_k;
            v.x = (const float*)A; v.sx = sa_k; v.sy = sc_n;
        }
        v.K = K;
        v.y = (float*)C;
        v.partial = nullptr;
        v.splits = 1;
        bsc_prof_scope prof(ctx);
        if (v.sa_k == 1 || (v.sa_m != 1 && v.sa_k < v.sa_m)) {
            hipLaunchKernelGGL(gemv_kcontig_kernel, dim3((unsigned)((v.M + 3) / 4)), dim3(256), 0,
                               ctx->stream, v);
            BSC_LAUNCH_CHECK();
            return BSC_OK;
        }
        const bool vec4 = v.sa_m == 1 && v.M % 4 == 0 && v.sa_k % 4 == 0 && (((uintptr_t)v.A) & 15) == 0;
        const int64_t groups = vec4 ? (v.M + 255) / 256 : (v.M + 63) / 64;
        int64_t splits = (4 * (int64_t)ctx->cu_count) / groups;
        if (splits > K / 256) splits = K / 256;
        if (splits < 1) splits = 1;
        if (splits > 2048) splits = 2048;
        v.splits = (int)splits;
        if (splits > 1) {
            void* ws = nullptr;
            int rc = bsc_workspace(ctx, (size_t)splits * v.M * sizeof(double), &ws);
            if (rc != BSC_OK) return rc;
            v.partial = (double*)ws;
            ctx->slab_rows = 0;
        }
        if (vec4)
            hipLaunchKernelGGL(gemv_mcontig4_kernel, dim3((unsigned)(groups * splits)), dim3(256), 0,
                               ctx->stream, v);
        else
            hipLaunchKernelGGL(gemv_mcontig_kernel, dim3((unsigned)(groups * splits)), dim3(256), 0,
                               ctx->stream, v);
        BSC_LAUNCH_CHECK();
        if (splits > 1) {
            if (v.M < 4096)
                hipLaunchKernelGGL(gemv_finish_wave_kernel, dim3((unsigned)((v.M + 3) / 4)), dim3(256), 0,
                                   ctx->stream, (const double*)v.partial, v.splits, v.M, v.y, v.sy);
            else
                hipLaunchKernelGGL(gemv_finish_kernel, dim3((unsigned)((v.M + 255) / 256)), dim3(256), 0,
                                   ctx->stream, (const double*)v.partial, v.splits, v.M, v.y, v.sy);
            BSC_LAUNCH_CHECK();
        }
        return BSC_OK;
    }
    if (batch == 1 && !epi.pow && !pre) {
        // one tiny extent, the large operand streamed once by LDS-DMA (csrc/bsc_skinny.hip)
        int handled = 0;
        int rc = bsc_gemm_skinny(ctx, M, N, K, (const float*)A, sa_m, sa_k, (const float*)B, sb_k, sb_n,
                                 (float*)C, sc_m, sc_n, &handled);
        if (rc != BSC_OK || handled) return rc;
    }
    if (ctx->mfma_split == 2 && batch == 1 && !pre && A == B && M == N && sa_m == 1 && sb_n == 1 && sa_k == sb_k &&
        !(epi.pow && epi.E) && epi.pow >= 0) {
        // X^T X of a row-major X with the operands as two bf16 terms (csrc/bsc_gram.hip): bound by the read of X
        int handled = 0;
        const int rc = bsc_gram_split(ctx, (const float*)A, sa_k, K, M, (float*)C, sc_m, sc_n, epi.pow ? epi.scale : 1.0f, &handled);
        if (rc != BSC_OK || handled) return rc;
    }
    if (ctx->gemm_dma >= 2) {
        // persistent stream-K on the LDS-DMA pipeline; the A operand goes along C's contiguous axis
        const bool swap = sc_n == 1 && sc_m != 1;
        GemmArgs s;
        s.A = (const float*)(swap ? B : A); s.B = (const float*)(swap ? A : B);
        s.M = swap ? N : M; s.N = swap ? M : N; s.K = K;
        s.sa_b = swap ? sb_b : sa_b; s.sa_m = swap ? sb_n : sa_m; s.sa_k = swap ? sb_k : sa_k;
        s.sb_b = swap ? sa_b : sb_b; s.sb_n = swap ? sa_m : sb_n; s.sb_k = swap ? sa_k : sb_k;
        s.C = (float*)C; s.sc_b = sc_b; s.sc_m = swap ? sc_n : sc_m; s.sc_n = swap ? sc_m : sc_n;
        s.E = epi.E; s.se_b = epi.se_b; s.se_m = swap ? epi.se_n : epi.se_m; s.se_n = swap ? epi.se_m : epi.se_n;
        s.epi_scale = epi.scale; s.epi_pow = epi.pow;
        s.pre_a = swap ? epi.pre_b : epi.pre_a; s.pre_b = swap ? epi.pre_a : epi.pre_b;
        s.nt_c = ctx->gemm_nt_c && (double)M * (double)N * (double)batch * 4.0 >= 128.0 * 1024.0 * 1024.0;
        s.splits = 1; s.k_chunk = 0; s.vec_a = s.vec_b = s.fast = 0;
        const bool a_m = s.sa_m == 1, b_n = s.sb_n == 1;
        auto dma_ok = [&](const void* p, bool mn, int64_t ext_mn, int64_t s_mn, int64_t s_k, int64_t s_b) {
            if (((uintptr_t)p & 15) != 0 || s_b % 4 != 0 || s_mn < 0 || s_k < 0) return false;
            if (mn) return ext_mn % 4 == 0 && s_k % 4 == 0 && (31 * s_k + 128) * 4 < ((int64_t)1 << 31);
            return s_k == 1 && K % 4 == 0 && s_mn % 4 == 0 && (127 * s_mn + 32) * 4 < ((int64_t)1 << 31);
        };
        const int64_t c_span = 127 * (s.sc_n < 0 ? -s.sc_n : s.sc_n), e_span = 127 * (s.se_n < 0 ? -s.se_n : s.se_n);
        if (dma_ok(s.A, a_m, s.M, s.sa_m, s.sa_k, s.sa_b) && dma_ok(s.B, b_n, s.N, s.sb_n, s.sb_k, s.sb_b) &&
            (c_span + 128) * 4 < ((int64_t)1 << 31) && (e_span + 128) * 4 < ((int64_t)1 << 31) && s.sc_n >= 0 && s.se_n >= 0 &&
            ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch < ((int64_t)1 << 31) &&
            ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch / (2 * (int64_t)ctx->cu_count) * ((K + BK - 1) / BK) + 2 * (int64_t)ctx->cu_count * ((K + BK - 1) / BK) < ((int64_t)1 << 31)) {
            s.tiles_m = (int)((s.M + BM - 1) / BM);
            s.tiles_n = (int)((s.N + BN - 1) / BN);
            s.tiles_pb = s.tiles_m * s.tiles_n;
            // A contraction two to six k-tiles long over many full tiles, nothing folded into the store: what a
            // tile costs is its 64 KiB of stores, and one tile per workgroup (the hardware's dispatcher dealing
            // them, LDS-staged 512-byte rows) does that 7 % better than the persistent schedule -- config 4's
            // dot(Th, Bt), K = 128: 1.45 against 1.57 ms (profiles/r02_ab_gemm_stream_vs_tile_b64.txt)
            const bool short_k_plain = !pre && K > 32 && K <= 6 * BK && !(epi.pow && epi.E) &&
                                       (int64_t)s.tiles_pb * batch >= 8 * (int64_t)ctx->cu_count && s.M >= 2 * BM && s.N >= 2 * BN;
            if (!short_k_plain) {
            // X^T X: the same matrix on both sides, transposed -- half the tiles (plus the diagonal)
            s.ksplit = 0;
            s.fix_lanes = 4;
            s.sym = ctx->gemm_sym && s.A == s.B && s.M == s.N && s.sa_m == s.sb_n && s.sa_k == s.sb_k && s.sa_b == s.sb_b &&
                    !(epi.pow && epi.E) && s.tiles_m > 1 && s.pre_a == s.pre_b;
            if (s.sym) s.tiles_pb = s.tiles_m * (s.tiles_m + 1) / 2;
            s.group = 8;
            s.group_log2 = 3;
            auto magic = [](int64_t d, unsigned& m, int& sh) {       // q = t m >> sh for 0 <= t < 2^31, 1 <= d < 2^31
                int l = 0;
                while (((int64_t)1 << l) < d) ++l;
                sh = 31 + l;
                m = (unsigned)((((unsigned __int128)1) << sh) / (unsigned __int128)d + 1);
            };
            magic(s.tiles_pb, s.mg_pb, s.sh_pb);
            magic((int64_t)s.group * s.tiles_m, s.mg_strip, s.sh_strip);
            magic(s.tiles_n % s.group ? s.tiles_n % s.group : s.group, s.mg_last, s.sh_last);
            s.n_kt = (int)((K + BK - 1) / BK);
            s.dbg = ctx->gemm_dbg;
            stream_plan(s, (int64_t)s.tiles_pb * batch, s.n_kt, 2 * (int64_t)ctx->cu_count);
            // one tile with both extents within a wave's 64 x 64 and a long contraction, split along it: the
            // four waves of a workgroup split the k-groups (EDGE instantiation)
            s.ksplit = s.sk_stream && s.tiles_pb * batch == 1 && s.M <= 64 && s.N <= 64 && s.n_wg > 1 && s.rounds == 0 &&
                       K % 32 == 0;
            void* ws = nullptr;
            int rc = bsc_workspace(ctx, (size_t)(s.ksplit ? 8 : 2) * s.n_wg * BM * BN * sizeof(float), &ws);
            if (rc != BSC_OK) return rc;
            s.slab = (float*)ws;
            ctx->slab_rows = 0;
            {
                bsc_prof_scope prof(ctx);
                // mostly partial tiles?
                const bool edge = (s.tiles_m == 1 && s.M <= 96) || (s.tiles_n == 1 && s.N <= 96) || K <= 24;
#define BSC_GEMM_STREAM(AM, BN_)                                                                                          \
    do {                                                                                                                  \
        if (pre && edge)                                                                                                  \
            hipLaunchKernelGGL((gemm_f32_stream_kernel<AM, BN_, true, true>), dim3((unsigned)s.n_wg), dim3(GEMM_BLOCK), 0, \
                               ctx->stream, s);                                                                           \
        else if (pre)                                                                                                     \
            hipLaunchKernelGGL((gemm_f32_stream_kernel<AM, BN_, false, true>), dim3((unsigned)s.n_wg), dim3(GEMM_BLOCK), 0, \
                               ctx->stream, s);                                                                           \
        else if (edge)                                                                                                    \
            hipLaunchKernelGGL((gemm_f32_stream_kernel<AM, BN_, true>), dim3((unsigned)s.n_wg), dim3(GEMM_BLOCK), 0,      \
                               ctx->stream, s);                                                                           \
        else                                                                                                              \
            hipLaunchKernelGGL((gemm_f32_stream_kernel<AM, BN_, false>), dim3((unsigned)s.n_wg), dim3(GEMM_BLOCK), 0,     \
                               ctx->stream, s);                                                                           \
    } while (0)
                if (a_m && b_n) BSC_GEMM_STREAM(true, true);
                else if (a_m) BSC_GEMM_STREAM(true, false);
                else if (b_n) BSC_GEMM_STREAM(false, true);
                else BSC_GEMM_STREAM(false, false);
#undef BSC_GEMM_STREAM
            }
            BSC_LAUNCH_CHECK();
            if (stream_has_pieces(s)) {
                // pieces per split tile: about n_wg / tail_tiles (x 4 with ksplit)
                s.fix_lanes = (int64_t)s.n_wg * (s.ksplit ? 4 : 1) >= 64 * (int64_t)s.tail_tiles ? 64 : 4;
                hipLaunchKernelGGL(stream_fixup_kernel, dim3((unsigned)s.tail_tiles, BM * BN / 4 / (256 / s.fix_lanes)), dim3(256), 0,
                                   ctx->stream, s);
                BSC_LAUNCH_CHECK();
            }
            if (pre && epi.handled) *epi.handled = 1;
            return BSC_OK;
            }   // !short_k_plain
        }
    }
    if (pre) return BSC_OK;      // (not handled: the kernels below take their operands as they are)
    GemmArgs g;
    g.A = (const float*)A; g.B = (const float*)B;
    g.M = M; g.N = N; g.K = K;
    g.sa_b = sa_b; g.sa_m = sa_m; g.sa_k = sa_k;
    g.sb_b = sb_b; g.sb_k = sb_k; g.sb_n = sb_n;
    g.sc_b = sc_b; g.sc_m = sc_m; g.sc_n = sc_n;
    g.E = epi.E; g.se_b = epi.se_b; g.se_m = epi.se_m; g.se_n = epi.se_n;
    g.epi_scale = epi.scale; g.epi_pow = epi.pow;
    g.n_kt = 0; g.sk_q = 0; g.sk_r = 0; g.sk_stream = 0; g.n_wg = 0; g.rounds = 0; g.tail_tiles = 0; g.tiles_pb = 0;
    g.group = 1; g.dbg = 0; g.slab = nullptr;
    g.mg_pb = g.mg_strip = g.mg_last = 0; g.sh_pb = g.sh_strip = g.sh_last = g.group_log2 = 0; g.sym = 0; g.ksplit = 0; g.fix_lanes = 4;
    g.pre_a = g.pre_b = 0;
    g.nt_c = ctx->gemm_nt_c && (double)M * (double)N * (double)batch * 4.0 >= 128.0 * 1024.0 * 1024.0;
    g.tiles_m = (int)((M + BM - 1) / BM);
    g.tiles_n = (int)((N + BN - 1) / BN);
    const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n * batch;
    // split K until the grid has about two workgroups per CU, keeping >= 512 of K per split
    int64_t splits = 1;
    const int64_t want = 2 * (int64_t)ctx->cu_count;
    if (tiles < want) {
        splits = want / tiles;
        const int64_t max_splits = K / 512;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        if (splits > 1024) splits = 1024;
    } else if (tiles % want != 0 && tiles < 8 * want) {
        // A grid of a few rounds wastes the tail of its last round (33 x 33 tiles on 512 slots:
        // 2.1 rounds cost 3).  Splitting K makes the rounds shorter; take the split with the
        // best slot efficiency when that buys more than the extra pass over the partial
        // results costs (taken as 4 % per split: an M x N float slab written and read back).
        auto eff = [&](int64_t sp) {
            const int64_t wgs = tiles * sp;
            return (double)wgs / (double)(want * ((wgs + want - 1) / want));
        };
        double best = eff(1);
        for (int64_t sp = 2; sp <= 6 && K / sp >= 1024; ++sp) {
            const double e = eff(sp) - 0.04 * (double)(sp - 1);
            if (e > best + 0.03) {
                best = e;
                splits = sp;
            }
        }
    }
    int64_t chunk = (K + splits - 1) / splits;
    chunk = (chunk + BK - 1) / BK * BK;
    splits = (K + chunk - 1) / chunk;
    g.splits = (int)splits;
    g.k_chunk = chunk;
    g.C = (float*)C;
    float* slab = nullptr;
    if (splits > 1) {
        void* ws = nullptr;
        int rc = bsc_workspace(ctx, (size_t)batch * splits * M * N * sizeof(float), &ws);
        if (rc != BSC_OK) return rc;
        slab = (float*)ws;
        g.C = slab;
        ctx->slab_rows = 0;
    }
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)splits, (unsigned)batch);
    const bool a_m = (sa_m == 1) || (sa_k != 1 && M >= K);   // which axis consecutive threads walk
    const bool b_n = (sb_n == 1) || (sb_k != 1 && N >= K);
    // 16-byte loads along the contiguous axis need that stride to be 1, the other
    // strides multiples of 4 elements and the base 16-byte aligned
    auto vec_ok = [](const void* p, bool mn, int64_t s_mn, int64_t s_k, int64_t s_b) {
        const int64_t unit = mn ? s_mn : s_k, other = mn ? s_k : s_mn;
        return (int)(unit == 1 && other % 4 == 0 && s_b % 4 == 0 && ((uintptr_t)p & 15) == 0);
    };
    g.vec_a = vec_ok(A, a_m, sa_m, sa_k, sa_b);
    g.vec_b = vec_ok(B, b_n, sb_n, sb_k, sb_b);
    // lane offsets inside a [128 x 32] operand tile must fit 32 bits
    auto span_ok = [](int64_t s_mn, int64_t s_k) {
        const int64_t span = (127 * (s_mn < 0 ? -s_mn : s_mn) + 31 * (s_k < 0 ? -s_k : s_k) + 4) * 4;
        return s_mn >= 0 && s_k >= 0 && span < ((int64_t)1 << 31);
    };
    g.fast = ctx->gemm_fast && g.vec_a && g.vec_b && span_ok(sa_m, sa_k) && span_ok(sb_n, sb_k);
    // LDS-DMA staging: 16-byte pieces wholly inside or outside the operand, lane offsets below 2^31
    auto dma_ok = [&](const void* p, bool mn, int64_t ext_mn, int64_t s_mn, int64_t s_k, int64_t s_b) {
        if (((uintptr_t)p & 15) != 0 || s_b % 4 != 0 || s_mn < 0 || s_k < 0) return false;
        if (mn) return s_mn == 1 && ext_mn % 4 == 0 && s_k % 4 == 0 && (31 * s_k + 128) * 4 < ((int64_t)1 << 31);
        return s_k == 1 && K % 4 == 0 && s_mn % 4 == 0 && (127 * s_mn + 32) * 4 < ((int64_t)1 << 31);
    };
    const bool dma = ctx->gemm_dma && dma_ok(A, a_m, M, sa_m, sa_k, sa_b) && dma_ok(B, b_n, N, sb_n, sb_k, sb_b);
    {
        bsc_prof_scope prof(ctx);
#define BSC_GEMM_DMA(AM, BN_) \
    hipLaunchKernelGGL((gemm_f32_dma_kernel<AM, BN_>), grid, dim3(GEMM_BLOCK), 0, ctx->stream, g)
        if (dma) {
            if (a_m && b_n) BSC_GEMM_DMA(true, true);
            else if (a_m) BSC_GEMM_DMA(true, false);
            else if (b_n) BSC_GEMM_DMA(false, true);
            else BSC_GEMM_DMA(false, false);
        } else {
#undef BSC_GEMM_DMA
#define BSC_GEMM(AM, BN_)                                                                       \
    do {                                                                                        \
        if (ctx->gemm_pipe)                                                                     \
            hipLaunchKernelGGL((gemm_f32_mfma_kernel<AM, BN_, true>), grid, dim3(GEMM_BLOCK), 0, \
                               ctx->stream, g);                                                 \
        else                                                                                    \
            hipLaunchKernelGGL((gemm_f32_mfma_kernel<AM, BN_, false>), grid, dim3(GEMM_BLOCK), 0, \
                               ctx->stream, g);                                                 \
    } while (0)
        if (a_m && b_n) BSC_GEMM(true, true);
        else if (a_m) BSC_GEMM(true, false);
        else if (b_n) BSC_GEMM(false, true);
        else BSC_GEMM(false, false);
#undef BSC_GEMM
        }
    }
    BSC_LAUNCH_CHECK();
    if (splits > 1) {
        const dim3 rgrid((unsigned)((M * N + 255) / 256), (unsigned)batch);
        hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, ctx->stream, slab, g.splits, M,
                           N, (float*)C, sc_b, sc_m, sc_n, g);
        BSC_LAUNCH_CHECK();
    }
    return BSC_OK;
}

extern "C" {

int bsc_gemm_strided_batched(bsc_ctx* ctx, int dtype, int64_t batch, int64_t M, int64_t N,
                             int64_t K, const void* A, int64_t sa_b, int64_t sa_m, int64_t sa_k,
                             const void* B, int64_t sb_b, int64_t sb_k, int64_t sb_n, void* C,
                             int64_t sc_b, int64_t sc_m, int64_t sc_n) {
    return gemm_impl(ctx, dtype, batch, M, N, K, A, sa_b, sa_m, sa_k, B, sb_b, sb_k, sb_n, C, sc_b, sc_m, sc_n,
                     Epilogue{});
}

/* The persistent kernels' schedule, host side only (no device needed): how `tiles` tiles of `n_kt` units
 * are dealt to `slots` resident workgroups.  out = {n_wg, rounds, tail_tiles, sk_stream, sk_q, sk_r}.
 * Exposed so that the partition can be property-tested without a GPU (tests/test_stream_plan.py). */
int bsc_stream_plan(int64_t tiles, int32_t n_kt, int64_t slots, int32_t out[6]) {
    BSC_REQUIRE(out != nullptr && tiles >= 1 && n_kt >= 1 && slots >= 1, "bsc_stream_plan: bad arguments");
    GemmArgs s{};
    stream_plan(s, tiles, n_kt, slots);
    out[0] = s.n_wg; out[1] = s.rounds; out[2] = s.tail_tiles; out[3] = s.sk_stream; out[4] = s.sk_q; out[5] = s.sk_r;
    return BSC_OK;
}

int bsc_gemm_epilogue(bsc_ctx* ctx, int dtype, int64_t batch, int64_t M, int64_t N, int64_t K, const void* A,
                      int64_t sa_b, int64_t sa_m, int64_t sa_k, const void* B, int64_t sb_b, int64_t sb_k,
                      int64_t sb_n, void* C, int64_t sc_b, int64_t sc_m, int64_t sc_n, int power, double scale,
                      const void* E, int64_t se_b, int64_t se_m, int64_t se_n) {
    if (power != 1 && power != -1)
        return bsc_fail(BSC_ERR_INVALID, "bsc_gemm_epilogue: power must be 1 or -1 (got %d)", power);
    Epilogue e;
    e.pow = power;
    e.scale = (float)scale;
    e.E = (const float*)E;
    e.se_b = se_b; e.se_m = se_m; e.se_n = se_n;
    return gemm_impl(ctx, dtype, batch, M, N, K, A, sa_b, sa_m, sa_k, B, sb_b, sb_k, sb_n, C, sc_b, sc_m, sc_n, e);
}

int bsc_gemm_fused(bsc_ctx* ctx, int dtype, int64_t batch, int64_t M, int64_t N, int64_t K, const void* A, int64_t sa_b,
                   int64_t sa_m, int64_t sa_k, int pre_a, const void* B, int64_t sb_b, int64_t sb_k, int64_t sb_n, int pre_b,
                   void* C, int64_t sc_b, int64_t sc_m, int64_t sc_n, int power, double scale, const void* E, int64_t se_b,
                   int64_t se_m, int64_t se_n, int32_t* handled) {
    BSC_REQUIRE(handled, "bsc_gemm_fused: handled is null");
    *handled = 0;
    BSC_REQUIRE(pre_a >= 0 && pre_a <= 3 && pre_b >= 0 && pre_b <= 3, "bsc_gemm_fused: prologue %d / %d (0..3)", pre_a, pre_b);
    BSC_REQUIRE(power == 0 || power == 1 || power == -1, "bsc_gemm_fused: power must be 0, 1 or -1 (got %d)", power);
    if ((pre_a == 0 && pre_b == 0) || (pre_a == 2 && pre_b == 2)) return BSC_OK;   // nothing to fuse / exp(0) on both sides
    Epilogue e;
    e.pow = power;
    e.scale = (float)scale;
    e.E = (const float*)E;
    e.se_b = se_b; e.se_m = se_m; e.se_n = se_n;
    if (power == 0 && scale != 1.0) e.pow = 1;
    e.pre_a = pre_a; e.pre_b = pre_b;
    int h = 0;
    e.handled = &h;
    const int rc = gemm_impl(ctx, dtype, batch, M, N, K, A, sa_b, sa_m, sa_k, B, sb_b, sb_k, sb_n, C, sc_b, sc_m, sc_n, e);
    *handled = h;
    return rc;
}

}  // extern "C"
