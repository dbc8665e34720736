// `_tensordot` with one tiny extent: the skinny products of the general VI engines
// (bayesic/algebra.py:1347-1351 un-batched tensordot; SURVEY.md 8(a) A7: cfg 5's
// dot(W, X.T) -> _tensordot(W, _dimshuffle(X,1,0), [1],[0]) and its transpose in the pathwise
// gradient, dot(R, X) -> _tensordot(R, X, [1],[0])).
//
// The 128 x 128-tile GEMM of csrc/bsc_gemm.hip runs an 8 x 1M x 256 product with 94 % of every
// tile empty (0.63-0.78 ms).  Both kernels here stream the large operand ONCE, HBM-bound, with the
// machinery of config 5's pass (csrc/bsc_bbvi.hip): each wave owns 16-row tiles of the large
// matrix, brings them in by LDS-DMA (buffer_load ... lds: no VGPR round trip, ~45 instead of ~115
// cycles of issue per KiB -- tools/ubench_mfma_vmem.hip) into a ring of its own, reads them back as
// v_mfma_f32_16x16x4_f32 operands with conflict-free ds_read_b128, and keeps the small operand in
// registers.  LDS-DMA completes in issue order and is covered only by the issuing wave's vmcnt:
// waits are counted by hand and every read of DMA'd bytes is inline asm (see bsc_bbvi.hip).
//
//   NT  C[m, n] = sum_k A[m, k] B[n, k]    M <= 32, K <= 256, N large; B rows k-contiguous
//   TN  C[m, d] = sum_n A[m, n] B[n, d]    M <= 16, D <= 256, K = n large; B rows d-contiguous,
//                                          A rows n-contiguous; deterministic float64 finish
#include "bsc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SW = 8;          // waves per workgroup of the TN kernel (one workgroup per CU)
constexpr int NTW = 4;         // ... of the NT kernel: one wave per SIMD streams faster (200 vs 236 us at 8 x 1M x 256)
constexpr int ST = 16;         // rows of the large operand per wave tile
constexpr int SR = 16;         // ring slots (1 KiB each) per wave: a whole tile ahead

#define BSC_LDS_B128(DST, ADDR, OFF) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define BSC_LDS_B32(DST, ADDR, OFF) \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")

// ---- NT: the large operand is B[n, k] (rows = n), 16-row tiles --------------------------------
// v_mfma_f32_16x16x4_f32 with the SMALL matrix as the A operand: lane (i16, kq) holds A[m = 16 sb + i16][16 j + 4 kq + r]
// (16 NSB registers per strip j of 16 columns) and, as the B operand, B[row i16 of the tile][the same column]; the
// result acc[sb][r] = C[m = 16 sb + 4 kq + r][n = row0 + i16] has the LONG axis on adjacent lanes, so a store
// instruction writes 64-byte runs (with the operands the other way round a lane held four consecutive n of one m and a
// store instruction was 64 separate 16-byte pieces: 54 of 226 us at 8 x 1M x 256, tools/ab_skinny_nt.py).
//
// The tile arrives by LDS-DMA in WHOLE 128-byte lines: piece s = 2 c + sp is rows 8 sp .. + 7 x columns 32 c .. + 31
// (8 x 128 bytes; pieces of 16 rows x 64 bytes read 23 % slower, same tool), lane l of the DMA = (row l >> 3, 16-byte
// position l & 7).  The 16-byte chunks of a row are XOR-permuted -- on the GLOBAL side, the fill stays lane-linear --
// by f = (row >> 1) | (sp << 2), which makes the operand read (ds_read_b128, lane (i16, kq) <- row i16, chunk kq + 4 t
// of column block c) free of bank conflicts in each of the instruction's four 16-lane groups.
struct NtArgs {
    const float* A; int64_t sa_m, sa_k;     // small [M, K]
    const float* B; int64_t ldb;            // large [N, K], k-contiguous
    float* C; int64_t sc_m, sc_n;
    int64_t N;
    int M, K;
};

// DBG (timing only, results wrong; BSC_SKINNY_NT_DBG behind BSC_PROFILING_BUILDS): bit 0 no MFMAs, bit 2 no stores
template <int NSB, int KS, int DBG = 0>
__global__ __launch_bounds__(64 * NTW, 2) void gemm_skinny_nt_kernel(NtArgs a) {
    static_assert(KS % 2 == 0 && KS <= SR, "two strips per column block, one ring slot per strip");
    __shared__ __attribute__((aligned(16))) char ring[NTW * SR * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int K = a.K;
    constexpr int NCB = KS / 2;             // column blocks of 32

    f32x4 wreg[NSB][KS];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int m = 16 * sb + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 16 * j + 4 * kq + r;
                wreg[sb][j][r] = (m < a.M && col < K) ? a.A[(int64_t)m * a.sa_m + (int64_t)col * a.sa_k] : 0.f;
            }
        }

    const int64_t n_waves = (int64_t)gridDim.x * NTW;
    int64_t tile = (int64_t)blockIdx.x * NTW + wave;
    const int64_t n_tiles = (a.N + ST - 1) / ST;
    char* const my = ring + wave * SR * 1024;
    // the DMA's side: row ra of the piece, position rp; the chunk fetched there is rp ^ f
    const int ra = lane >> 3, rp = lane & 7;
    const int row_bytes = (int)(a.ldb * 4);
    const int voff0 = ra * row_bytes + 16 * (rp ^ (ra >> 1));
    const int voff1 = ra * row_bytes + 16 * (rp ^ ((ra >> 1) | 4));
    auto dma = [&](decltype(bsc_rows_rsrc(a.B, a.ldb, K, a.N, 0)) rs, int s) {      // piece s -> slot s
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bsc_lds_ptr)(my + s * 1024), 16, (s & 1) ? voff1 : voff0,
                                                 (s & 1) * 8 * row_bytes + 128 * (s >> 1), 0, 2);
    };
    // the reader's side: row i16 = (sp, a), chunk kq + 4 t: strip j = 2 c + t is at addr_t + 2048 c
    const int sp = i16 >> 3, ar = i16 & 7;
    const unsigned addr0 = (unsigned)(uintptr_t)(bsc_lds_ptr)my + 1024u * sp + 128u * ar + 16u * (kq ^ ((ar >> 1) | (sp << 2)));
    const unsigned addr1 = addr0 ^ 64u;

    if (tile < n_tiles) {
        const auto rs = bsc_rows_rsrc(a.B, a.ldb, K, a.N, tile * ST);
#pragma unroll
        for (int s = 0; s < KS; ++s) dma(rs, s);
    }
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    asm volatile("" ::: "memory");
    f32x4 an;
    BSC_LDS_B128(an, addr0, 0);

    for (; tile < n_tiles; tile += n_waves) {
        const int64_t row0 = tile * ST;
        const auto rs_next = bsc_rows_rsrc(a.B, a.ldb, K, a.N, (tile + n_waves) * ST);
        f32x4 acc[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) acc[sb] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto multiply = [&](const f32x4& x, int j) {
            if (DBG & 1) {
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) acc[sb] += x * wreg[sb][j];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int sb = 0; sb < NSB; ++sb)
                        acc[sb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[sb][j][r], x[r], acc[sb], 0, 0, 0);
            }
        };
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            // strip 2 c (read a strip ago); strip 2 c + 1 sits in the same two slots
            __builtin_amdgcn_s_waitcnt(BSC_LGKMCNT0);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 x = an;
            if (32 * c + 4 * kq >= K) x = f32x4{0.f, 0.f, 0.f, 0.f};     // columns past K (K % 32 != 0, or padded strips)
            BSC_LDS_B128(an, addr1, c * 2048);
            __builtin_amdgcn_sched_barrier(0);
            multiply(x, 2 * c);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(BSC_LGKMCNT0);
            __builtin_amdgcn_sched_barrier(0);
            x = an;
            if (32 * c + 16 + 4 * kq >= K) x = f32x4{0.f, 0.f, 0.f, 0.f};
            // column block c + 1 (block 0 of the next tile after the last) must have landed: every piece of a tile is
            // issued during the previous tile, two per column block, so the loads issued after block c + 1's are the
            // KS - 4 - 2 c pieces behind it and this tile's 2 c -- and the previous tile's stores, which are NOT
            // counted here: a smaller count only waits for a few younger pieces as well, whatever order stores retire in
            __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(KS - 4));
            asm volatile("" ::: "memory");
            BSC_LDS_B128(an, addr0, ((c + 1) % NCB) * 2048);
            __builtin_amdgcn_sched_barrier(0);
            multiply(x, 2 * c + 1);
            dma(rs_next, 2 * c);            // both strips of the block are in registers: its two slots are free
            dma(rs_next, 2 * c + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // C[m = 16 sb + 4 kq + r][row0 + i16]
        const int64_t n = row0 + i16;
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * sb + 4 * kq + r;
                if ((DBG & 4) && acc[sb][r] != 12345.678f) continue;
                if (m < a.M && n < a.N) a.C[(int64_t)m * a.sc_m + n * a.sc_n] = acc[sb][r];
            }
    }
    // no LDS-DMA of this wave may still be in flight when the workgroup's LDS is released
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ---- TN: the large operand is B[n, d] (rows = n, the contracted axis) ---------------------------
// A 16-row tile contracts in four k-steps c: rows 4 c + kq.  B operand of (c, column block q, e):
// lane (i16, kq) <- B[row0 + 4 c + kq][64 q + 4 i16 + e] -- one ds_read_b128 per (c, q) feeds the four
// MFMAs e = 0..3 (unpadded 1-KiB rows are conflict-free for this pattern).  A operand of step c:
// A[m = i16][row0 + 4 c + kq], from the 1-KiB tile of A that one DMA brings (lane (i16, kq) <- 16
// bytes of row i16 at column row0 + 4 kq).  acc[4 q + e][r] = C[m = 4 kq + r][d = 64 q + 4 i16 + e].
struct TnArgs {
    const float* A; int64_t lda;        // small [M, K], k(=n)-contiguous rows
    const float* B; int64_t ldb;        // large [K, D], d-contiguous rows
    double* slab;                        // [blocks][16 * 256] float64 workgroup partials
    int64_t K;
    int M, D;
};

__global__ __launch_bounds__(64 * SW, 2) void gemm_skinny_tn_kernel(TnArgs a) {
    __shared__ __attribute__((aligned(16))) char ring[SW * (SR * 1024 + 2 * 1024)];   // rows | A tile [2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int D = a.D;
    constexpr int WAVE_BYTES = SR * 1024 + 2048, A_OFF = SR * 1024;

    const int64_t n_waves = (int64_t)gridDim.x * SW;
    int64_t tile = (int64_t)blockIdx.x * SW + wave;
    const int64_t n_tiles = (a.K + ST - 1) / ST;
    char* const my = ring + wave * WAVE_BYTES;
    const unsigned my_addr = (unsigned)(uintptr_t)(bsc_lds_ptr)my;
    // row DMA: lane l <- 16 bytes at column 4 l of the row (lanes past D read nothing: offset out of range)
    const int row_voff = 4 * lane < D ? 16 * lane : 0x7FFFFF00;
    auto row_dma = [&](decltype(bsc_rows_rsrc(a.B, a.ldb, D, a.K, 0)) rs, int row) {      // row -> slot `row`
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bsc_lds_ptr)(my + row * 1024), 16, row_voff,
                                                 row * (int)(a.ldb * 4), 0, 2);
    };
    // A tile: lane (i16, kq) <- A[i16][row0 + 4 kq .. +3]; rows m >= M read nothing
    const int a_voff = i16 < a.M ? i16 * (int)(a.lda * 4) + 16 * kq : 0x7FFFFF00;
    auto a_dma = [&](int64_t row0, int par) {
        const int64_t rem = a.K - row0;
        // records: up to the end of the LAST row of A, so that columns past K of earlier rows still
        // read (finite) neighbours; those products meet zero rows of B (its descriptor ends at K)
        uint64_t bytes = rem > 0 ? ((uint64_t)(a.M - 1) * (uint64_t)a.lda + (uint64_t)rem) * 4u : 0;
        const unsigned rec = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (rem > 0 ? row0 : 0)), 0, rec, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bsc_lds_ptr)(my + A_OFF + par * 1024), 16, a_voff, 0, 0, 0);
    };
    const unsigned addr_b = my_addr + 1024u * kq + 16u * i16;               // + c*4096 + q*256
    const unsigned addr_a = my_addr + A_OFF + 256u * 0 + 16u * i16 + 4u * kq;   // + par*1024 + c*256

    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (tile < n_tiles) {
        const auto rs = bsc_rows_rsrc(a.B, a.ldb, D, a.K, tile * ST);
        a_dma(tile * ST, 0);
#pragma unroll
        for (int r = 0; r < ST; ++r) row_dma(rs, r);
    }
    int par = 0;
    for (; tile < n_tiles; tile += n_waves) {
        const int64_t row0 = tile * ST;
        const auto rs_next = bsc_rows_rsrc(a.B, a.ldb, D, a.K, (tile + n_waves) * ST);
        a_dma((tile + n_waves) * ST, par ^ 1);        // FIRST: older than this tile's row DMAs
        const bool tail = row0 + ST > a.K;           // (wave-uniform) the tile that holds row K
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // rows 4c .. 4c+3 of this tile were issued in the previous tile's step c; after them came
            // that tile's 4 (3 - c) rows, this tile's A DMA and this tile's 4 c rows = 13 operations
            __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(13));
            asm volatile("" ::: "memory");
            f32x4 b[4];
            float av;
#pragma unroll
            for (int q = 0; q < 4; ++q) BSC_LDS_B128(b[q], addr_b, c * 4096 + q * 256);
            if (par) BSC_LDS_B32(av, addr_a, 1024 + c * 256);
            else BSC_LDS_B32(av, addr_a, c * 256);
            __builtin_amdgcn_s_waitcnt(BSC_LGKMCNT0);
            __builtin_amdgcn_sched_barrier(0);
            if (tail && row0 + 4 * c + kq >= a.K) av = 0.f;       // A's columns past K are not zeros
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[4 * q + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[q][e], acc[4 * q + e], 0, 0, 0);
            // rows 4c .. 4c+3 are in registers: the same rows of the next tile
#pragma unroll
            for (int r = 0; r < 4; ++r) row_dma(rs_next, 4 * c + r);
            __builtin_amdgcn_sched_barrier(0);
        }
        par ^= 1;
    }
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // wave partials [16 m][256 d] through LDS (each wave's own 16 KiB of ring), then the workgroup's
    // waves in a fixed order, float64
    float* mine = reinterpret_cast<float*>(my);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(4 * kq + r) * 256 + 64 * q + 4 * i16 + e] = acc[4 * q + e][r];
    __syncthreads();
    for (int idx = tid; idx < 16 * 256; idx += 64 * SW) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < SW; ++w) v += (double)reinterpret_cast<const float*>(ring + w * WAVE_BYTES)[idx];
        a.slab[(int64_t)blockIdx.x * 4096 + idx] = v;
    }
}

// C[m, d] = sum over workgroup partials, fixed order.  One thread per output (consecutive threads
// read consecutive slab words); sixteen loads in flight, added in block order.
__global__ __launch_bounds__(64) void skinny_tn_finish_kernel(const double* __restrict__ slab, int n_blocks,
                                                              int M, int D, float* __restrict__ C,
                                                              int64_t sc_m, int64_t sc_n) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= M * 256) return;
    const int m = idx >> 8, d = idx & 255;
    if (d >= D) return;
    double v = 0.0;
    int b = 0;
    for (; b + 16 <= n_blocks; b += 16) {
        double t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = slab[(int64_t)(b + j) * 4096 + idx];
#pragma unroll
        for (int j = 0; j < 16; ++j) v += t[j];
    }
    for (; b < n_blocks; ++b) v += slab[(int64_t)b * 4096 + idx];
    C[(int64_t)m * sc_m + (int64_t)d * sc_n] = (float)v;
}

int launch_nt(bsc_ctx* ctx, const NtArgs& a) {
    const int64_t n_tiles = (a.N + ST - 1) / ST;
    int n_blocks = (int)((n_tiles + NTW - 1) / NTW);
    if (n_blocks > ctx->skinny_nt_wg_per_cu * ctx->cu_count) n_blocks = ctx->skinny_nt_wg_per_cu * ctx->cu_count;
    const int nsb = a.M <= 16 ? 1 : 2;
    const int ks = a.K <= 64 ? 4 : a.K <= 128 ? 8 : a.K <= 192 ? 12 : 16;
    bsc_prof_scope prof(ctx);
    if (ctx->skinny_nt_dbg && nsb == 1 && ks == 16) {
#define BSC_NTD(DBG) case DBG: hipLaunchKernelGGL((gemm_skinny_nt_kernel<1, 16, DBG>), dim3(n_blocks), dim3(64 * NTW), 0, ctx->stream, a); break;
        switch (ctx->skinny_nt_dbg) { BSC_NTD(1) BSC_NTD(4) BSC_NTD(5) default: return bsc_fail(BSC_ERR_INVALID, "BSC_SKINNY_NT_DBG: 1, 4 or 5"); }
#undef BSC_NTD
        BSC_LAUNCH_CHECK();
        return BSC_OK;
    }
#define BSC_NT(NSB, KS) hipLaunchKernelGGL((gemm_skinny_nt_kernel<NSB, KS>), dim3(n_blocks), dim3(64 * NTW), 0, ctx->stream, a)
    if (nsb == 1) {
        if (ks == 4) BSC_NT(1, 4); else if (ks == 8) BSC_NT(1, 8); else if (ks == 12) BSC_NT(1, 12); else BSC_NT(1, 16);
    } else {
        if (ks == 4) BSC_NT(2, 4); else if (ks == 8) BSC_NT(2, 8); else if (ks == 12) BSC_NT(2, 12); else BSC_NT(2, 16);
    }
#undef BSC_NT
    BSC_LAUNCH_CHECK();
    return BSC_OK;
}

}  // namespace

int bsc_gemm_skinny(bsc_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k,
                    const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t sc_m, int64_t sc_n, int* handled) {
    *handled = 0;
    if (!ctx->gemm_skinny) return BSC_OK;
    auto aligned = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    auto small_span = [](int64_t s) { return s >= 0 && s < ((int64_t)1 << 26); };
    // ---- NT: one side has <= 32 rows, the other is long with k-contiguous rows, K <= 256 ----
    if (K >= 4 && K <= 256 && K % 4 == 0) {
        // as given: A small [M, K], B large seen as rows n with stride sb_n, k stride sb_k
        if (M <= 32 && N >= 4096 && sb_k == 1 && sb_n >= K && sb_n % 4 == 0 && small_span(sb_n) && aligned(B)) {
            NtArgs a{A, sa_m, sa_k, B, sb_n, C, sc_m, sc_n, N, (int)M, (int)K};
            *handled = 1;
            return launch_nt(ctx, a);
        }
        // transposed: C^T[n, m] = sum_k B^T[n, k] A^T[k, m]: B small, A large
        if (N <= 32 && M >= 4096 && sa_k == 1 && sa_m >= K && sa_m % 4 == 0 && small_span(sa_m) && aligned(A)) {
            NtArgs a{B, sb_n, sb_k, A, sa_m, C, sc_n, sc_m, M, (int)N, (int)K};
            *handled = 1;
            return launch_nt(ctx, a);
        }
    }
    // ---- TN: both free extents small, the contracted axis long ----
    auto launch_tn = [&](const float* As, int64_t lda, int Ms, const float* Bl, int64_t ldb, int Dl, int64_t sc_ms,
                         int64_t sc_ds) {
        const int64_t n_tiles = (K + ST - 1) / ST;
        int n_blocks = (int)((n_tiles + SW - 1) / SW);
        if (n_blocks > ctx->cu_count) n_blocks = ctx->cu_count;
        void* ws = nullptr;
        int rc = bsc_workspace(ctx, (size_t)n_blocks * 4096 * sizeof(double), &ws);
        if (rc != BSC_OK) return rc;
        ctx->slab_rows = 0;
        TnArgs a{As, lda, Bl, ldb, (double*)ws, K, Ms, Dl};
        {
            bsc_prof_scope prof(ctx);
            hipLaunchKernelGGL(gemm_skinny_tn_kernel, dim3(n_blocks), dim3(64 * SW), 0, ctx->stream, a);
        }
        BSC_LAUNCH_CHECK();
        hipLaunchKernelGGL(skinny_tn_finish_kernel, dim3((Ms * 256 + 63) / 64), dim3(64), 0, ctx->stream,
                           (const double*)ws, n_blocks, Ms, Dl, C, sc_ms, sc_ds);
        BSC_LAUNCH_CHECK();
        return (int)BSC_OK;
    };
    if (K >= 16384) {
        if (M <= 16 && N <= 256 && N % 4 == 0 && sa_k == 1 && sb_n == 1 && sa_m % 4 == 0 && sb_k % 4 == 0 &&
            sb_k >= N && sa_m >= K && small_span(sb_k) && small_span(sa_m) && sa_m * 15 + K < ((int64_t)1 << 29) &&
            aligned(A) && aligned(B)) {
            *handled = 1;
            return launch_tn(A, sa_m, (int)M, B, sb_k, (int)N, sc_m, sc_n);
        }
        // transposed: C^T[n, m] = sum_k B^T[n, k] A^T[k, m]
        if (N <= 16 && M <= 256 && M % 4 == 0 && sb_k == 1 && sa_m == 1 && sb_n % 4 == 0 && sa_k % 4 == 0 &&
            sa_k >= M && sb_n >= K && small_span(sa_k) && small_span(sb_n) && sb_n * 15 + K < ((int64_t)1 << 29) &&
            aligned(A) && aligned(B)) {
            *handled = 1;
            return launch_tn(B, sb_n, (int)N, A, sa_k, (int)M, sc_n, sc_m);
        }
    }
    return BSC_OK;
}
