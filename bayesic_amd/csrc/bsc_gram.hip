// X^T X on the operand-split bf16 route (ctx->mfma_split == 2; csrc/bsc_bf16split.h): the summed second-moment
// statistic of iid rows (bayesic/distribution/base.py:146-172; in bayesic.algebra terms dot(X.T, X),
// bayesic/algebra.py:1347-1383), SURVEY.md 8(d)'s "companion conjugate statistic" of config 2: 2 N D^2 = 131 GF on
// 1 GB at 1M x 256 -- 0.83 ms of f32 MFMA against 0.17 ms of HBM.  With X as two bf16 terms the three cross
// products are 12 us of matrix-pipe time per CU-second of data and the product is bound by the read of X.
//
// A workgroup (4 waves, two per CU) walks 32-row steps of ALL D <= 256 columns, so X is read ONCE:
//   * wave w loads its 64 columns of the step (a lane: 32 consecutive floats of one row = eight 16-byte loads,
//     one step ahead in registers), splits them and writes the terms to LDS as [column block][term][32 rows][32
//     columns] bf16 images with 64-byte rows (8-byte chunks XOR-permuted by 2 ((row >> 1) & 3): conflict-free
//     16-byte writes and transposed reads);
//   * the contraction runs over the ROWS, the lane index of both operands: both come back through
//     ds_read_b64_tr_b16 (lane = column), as in mog_estep_bx_kernel's backward;
//   * the D / 32 (D / 32 + 1) / 2 blocks on and above the diagonal are dealt round-robin to the four waves
//     (9 each at D = 256: 144 accumulator registers);
//   * images double-buffered, one barrier per step; the workgroups' partial blocks go to a slab and are added in
//     float64 in workgroup order by gram_bx_reduce_kernel, which also writes the mirrored triangle.
#include "bsc_common.h"
#include "bsc_bf16split.h"
#include <type_traits>

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int GB_BLOCK = 256;
constexpr int GB_IMG = 2048;                 // [32 rows][32 cols] bf16

// s_waitcnt lgkmcnt(n) alone, n a compile-time constant after inlining (12..20 here; the counter holds 15)
__device__ __forceinline__ void lda_like_wait(int n) {
    switch (n) {
        case 12: __builtin_amdgcn_s_waitcnt(0xCC7F); break;
        case 13: __builtin_amdgcn_s_waitcnt(0xCD7F); break;
        case 14: __builtin_amdgcn_s_waitcnt(0xCE7F); break;
        default: __builtin_amdgcn_s_waitcnt(0xCF7F); break;       // 15 or more: "at most 15 outstanding"
    }
}

constexpr int gram_tri_index(int DB, int db, int eb) { return db * DB - db * (db - 1) / 2 + (eb - db); }

template <int DB>          // column blocks of 32: D = 32 DB
__global__ __launch_bounds__(GB_BLOCK, 2) void gram_bx_kernel(const float* __restrict__ X, int64_t ldx, int64_t N, int D,
                                                              float* __restrict__ slab, int n_steps_total) {
    constexpr int NB = DB * (DB + 1) / 2;            // blocks on and above the diagonal
    constexpr int PER = (NB + 3) / 4;                // per wave
    constexpr int CPW = (DB + 3) / 4;                // column blocks a wave loads and converts
    __shared__ __attribute__((aligned(16))) char lds[2 * DB * 2 * GB_IMG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned lb = (unsigned)(uintptr_t)(bsc_lds_ptr)lds;

    bsc_f32x16 acc[PER];
#pragma unroll
    for (int b = 0; b < PER; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    // writer: this lane's 16 columns (16 half ..) of row l31 of column block cb -> chunks 4 half .. 4 half + 3 of the row
    const int sw = 2 * ((l31 >> 1) & 3);
    const unsigned wr0 = lb + (unsigned)(l31 * 64 + 8 * ((4 * half) ^ sw)), wr1 = lb + (unsigned)(l31 * 64 + 8 * ((4 * half + 2) ^ sw));
    // transposed reads: lane 4 q + p of a 16-lane group points at row r0 + q, chunk 4 (gg & 1) + p; rows 8 (gg >> 1) + 4 e + q
    const int gg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    unsigned rd[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 8 * (gg >> 1) + 4 * e + tq, ch = 4 * (gg & 1) + tp;
        rd[e] = lb + (unsigned)(row * 64 + 8 * (ch ^ (2 * ((row >> 1) & 3))));
    }

    // the step's rows of this wave's column blocks, one step ahead: 16 floats per (lane, column block)
    float4 nx[CPW][4];
    auto load_step = [&](int64_t step) __attribute__((always_inline)) {
        const int64_t row = step * 32 + l31;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int cb = wave + 4 * c;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cb < DB && row < N) {
                    const f32x4_t w = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(X + row * ldx + 32 * cb + 16 * half + 4 * q));
                    v = make_float4(w.x, w.y, w.z, w.w);
                }
                nx[c][q] = v;
            }
        }
    };
    auto write_step = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int cb = wave + 4 * c;
            if (cb >= DB) continue;
            bsc_u32x4 t0[2], t1[2];       // columns 16 half .. + 7 and + 8 .. + 15, terms 0 / 1
            unsigned pk[2];
            bsc_split_pk<2>(nx[c][0].x, nx[c][0].y, pk); t0[0][0] = pk[0]; t0[1][0] = pk[1];
            bsc_split_pk<2>(nx[c][0].z, nx[c][0].w, pk); t0[0][1] = pk[0]; t0[1][1] = pk[1];
            bsc_split_pk<2>(nx[c][1].x, nx[c][1].y, pk); t0[0][2] = pk[0]; t0[1][2] = pk[1];
            bsc_split_pk<2>(nx[c][1].z, nx[c][1].w, pk); t0[0][3] = pk[0]; t0[1][3] = pk[1];
            bsc_split_pk<2>(nx[c][2].x, nx[c][2].y, pk); t1[0][0] = pk[0]; t1[1][0] = pk[1];
            bsc_split_pk<2>(nx[c][2].z, nx[c][2].w, pk); t1[0][1] = pk[0]; t1[1][1] = pk[1];
            bsc_split_pk<2>(nx[c][3].x, nx[c][3].y, pk); t1[0][2] = pk[0]; t1[1][2] = pk[1];
            bsc_split_pk<2>(nx[c][3].z, nx[c][3].w, pk); t1[0][3] = pk[0]; t1[1][3] = pk[1];
            const unsigned off = (unsigned)((buf * DB + cb) * 2 * GB_IMG);
#pragma unroll
            for (int term = 0; term < 2; ++term) {
                asm volatile("ds_write_b128 %0, %1" : : "v"(wr0 + off + term * GB_IMG), "v"(t0[term]) : "memory");
                asm volatile("ds_write_b128 %0, %1" : : "v"(wr1 + off + term * GB_IMG), "v"(t1[term]) : "memory");
            }
        }
    };
    // operand fragment of column block cb, k-step s (rows 16 s ..), both terms
    auto read_frag = [&](int buf, int cb, int s, bsc_u32x4 (&f)[2]) __attribute__((always_inline)) {
        const unsigned off = (unsigned)((buf * DB + cb) * 2 * GB_IMG + s * 1024);
        bsc_u32x2 a0, a1, b0, b1;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a0) : "v"(rd[0] + off) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a1) : "v"(rd[1] + off) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0) : "v"(rd[0] + off + GB_IMG) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1) : "v"(rd[1] + off + GB_IMG) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
        f[0] = bsc_u32x4{a0[0], a0[1], a1[0], a1[1]};
        f[1] = bsc_u32x4{b0[0], b0[1], b1[0], b1[1]};
    };

    const int64_t stride = gridDim.x;
    int64_t step = blockIdx.x;
    if (step < n_steps_total) load_step(step);
    int buf = 0;
    for (; step < n_steps_total; step += stride) {
        write_step(buf);
        if (step + stride < n_steps_total) load_step(step + stride);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        // this wave's blocks: the list (0,0), (0,1), .., (0,DB-1), (1,1), .. dealt round-robin
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            int bi = 0;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                bsc_u32x4 fa[2];
                bool have_a = false;
#pragma unroll
                for (int eb = db; eb < DB; ++eb, ++bi) {
                    if ((bi & 3) != wave) continue;          // (wave-uniform)
                    if (!have_a) {
                        read_frag(buf, db, s, fa);
                        have_a = true;
                    }
                    bsc_u32x4 fb[2];
                    if (eb == db) { fb[0] = fa[0]; fb[1] = fa[1]; }
                    else read_frag(buf, eb, s, fb);
                    acc[bi >> 2] = bsc_mfma_split<2>(fa, fb, acc[bi >> 2]);
                }
            }
        }
        buf ^= 1;
        // (the other buffer is written next; its readers finished before the barrier above of the step before)
    }
    // partial blocks -> slab [workgroup][block][32 x 32]
    float* out = slab + (int64_t)blockIdx.x * NB * 1024;
    {
        int bi = 0;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int eb = db; eb < DB; ++eb, ++bi) {
                if ((bi & 3) != wave) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;       // column of X block db
                    out[bi * 1024 + row * 32 + l31] = acc[bi >> 2][r];
                }
            }
    }
}

// D = 256: ONE workgroup of eight waves per CU (the generic kernel's nine blocks a wave need 144 accumulator
// registers and, dealt round-robin, 14 operand fragments per k-step -- 28 KiB of LDS reads, each waited for on its own).
// Compact regions: the triangle of column blocks 0..3 in two waves, the rectangle rows 0..3 x columns 4..7 in four
// 2 x 2 squares, the triangle 4..7 in two waves: at most five blocks and four fragments a wave; the fragments of a
// k-step are read once, then up to 15 MFMAs issue without a wait.  Wave w loads and converts column block w.
struct GramPlan {
    int n_frag, n_blk;
    int frag[4];           // column blocks
    int blk[5][2];         // indices into frag
};
__device__ constexpr GramPlan GRAM_PLAN8[8] = {
    {4, 5, {0, 1, 2, 3}, {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 1}}},
    {3, 5, {1, 2, 3, 0}, {{0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}}},
    {4, 4, {0, 1, 4, 5}, {{0, 2}, {0, 3}, {1, 2}, {1, 3}, {0, 0}}},
    {4, 4, {0, 1, 6, 7}, {{0, 2}, {0, 3}, {1, 2}, {1, 3}, {0, 0}}},
    {4, 4, {2, 3, 4, 5}, {{0, 2}, {0, 3}, {1, 2}, {1, 3}, {0, 0}}},
    {4, 4, {2, 3, 6, 7}, {{0, 2}, {0, 3}, {1, 2}, {1, 3}, {0, 0}}},
    {4, 5, {4, 5, 6, 7}, {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 1}}},
    {3, 5, {5, 6, 7, 0}, {{0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}}},
};

// The step's 32 x 256 floats arrive by LDS-DMA into a ring of GR_STAGES raw tiles (a row = one 1-KiB instruction of
// wave row / 4; 16-byte chunk c of row r at position c ^ (r & 15), applied on the global side, so that the converting
// lanes -- one row each, the same chunk index -- read conflict-free), GR_STAGES - 1 steps ahead: with the rows held
// one step ahead in registers instead, 32 KiB in flight per CU kept the product at 1.9 TB/s.
constexpr int GR_STAGES = 3;
constexpr int GR_RAW = 32 * 1024;             // one raw tile
constexpr int GR_IMG_SET = 8 * 2 * GB_IMG;    // one set of bf16 images: [column block][term]
constexpr int GR_IMG_BYTES = 2 * GR_IMG_SET;  // two sets: step i + 1 is converted while step i is multiplied (one barrier a step,
                                              // and the conversion's vector instructions run beside the other wave's MFMAs)

// DBG (timing only, results wrong; BSC_GRAM_DBG behind BSC_PROFILING_BUILDS): bit 0 no conversion (raw reads, splits, image
// writes), bit 1 no MFMAs, bit 2 no DMAs after the prologue's
template <int DBG = 0>
__global__ __launch_bounds__(512, 1) void gram256_bx_kernel(const float* __restrict__ X, int64_t ldx, int64_t N,
                                                            float* __restrict__ slab, int n_steps_total) {
    constexpr int DB = 8, NB = 36;
    __shared__ __attribute__((aligned(1024))) char lds[GR_IMG_BYTES + GR_STAGES * GR_RAW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned lb = (unsigned)(uintptr_t)(bsc_lds_ptr)lds;
    bsc_f32x16 acc[5];
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const int sw = 2 * ((l31 >> 1) & 3);
    const unsigned wr0 = lb + (unsigned)(wave * 2 * GB_IMG + l31 * 64 + 8 * ((4 * half) ^ sw));
    const unsigned wr1 = lb + (unsigned)(wave * 2 * GB_IMG + l31 * 64 + 8 * ((4 * half + 2) ^ sw));
    const int gg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    unsigned rd[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 8 * (gg >> 1) + 4 * e + tq, ch = 4 * (gg & 1) + tp;
        rd[e] = lb + (unsigned)(row * 64 + 8 * (ch ^ (2 * ((row >> 1) & 3))));
    }
    // raw-tile reads of the converting lane: row l31, chunks 8 wave + 4 half + q
    unsigned rr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rr[q] = lb + GR_IMG_BYTES + (unsigned)(l31 * 1024 + 16 * ((8 * wave + 4 * half + q) ^ (l31 & 15)));

    const int64_t stride = gridDim.x;
    const int my_steps = (int)((n_steps_total - (int64_t)blockIdx.x + stride - 1) / stride);     // >= 1 (host: grid <= steps)
    // DMA of step index i (this workgroup's i-th step) into stage i % GR_STAGES: wave w brings rows 4 w .. 4 w + 3
    auto issue = [&](int i) __attribute__((always_inline)) {
        const int64_t row0 = ((int64_t)blockIdx.x + (int64_t)i * stride) * 32;
        const int64_t rem = N - row0;                 // > 0
        const uint64_t bytes = ((uint64_t)(rem - 1) * (uint64_t)ldx + 256u) * 4u;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + row0 * ldx), 0, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes, 0x00020000);
        char* const dst = lds + GR_IMG_BYTES + (i % GR_STAGES) * GR_RAW + (4 * wave) * 1024;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int row = 4 * wave + jj;
            // rows past N: the descriptor's range ends with the last row -> zeros
            const unsigned voff = (unsigned)(row * (int)(ldx * 4) + 16 * (lane ^ (row & 15)));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bsc_lds_ptr)(dst + jj * 1024), 16, voff, 0, 0, 2);
        }
    };
    // LDS latency off the critical path: the fragments of k-step 0 are requested BEFORE the conversion of the next step,
    // those of k-step 1 before the MFMAs of k-step 0.
    auto frag_read = [&](auto wc, unsigned set, int s, bsc_u32x2 (&raw)[4][4]) __attribute__((always_inline)) {
        constexpr int W = decltype(wc)::value;
        constexpr GramPlan P = GRAM_PLAN8[W];
        const unsigned r0 = rd[0] + set + (unsigned)(s * 1024), r1 = rd[1] + set + (unsigned)(s * 1024);
#pragma unroll
        for (int f = 0; f < P.n_frag; ++f) {
            const unsigned off = (unsigned)(P.frag[f] * 2 * GB_IMG);
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(raw[f][0]) : "v"(r0), "n"(off) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(raw[f][1]) : "v"(r1), "n"(off) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(raw[f][2]) : "v"(r0), "n"(off + GB_IMG) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(raw[f][3]) : "v"(r1), "n"(off + GB_IMG) : "memory");
        }
    };
    auto frag_mfma = [&](auto wc, bsc_u32x2 (&raw)[4][4]) __attribute__((always_inline)) {
        constexpr int W = decltype(wc)::value;
        constexpr GramPlan P = GRAM_PLAN8[W];
        bsc_u32x4 fr[4][2];
#pragma unroll
        for (int f = 0; f < P.n_frag; ++f) {
            fr[f][0] = bsc_u32x4{raw[f][0][0], raw[f][0][1], raw[f][1][0], raw[f][1][1]};
            fr[f][1] = bsc_u32x4{raw[f][2][0], raw[f][2][1], raw[f][3][0], raw[f][3][1]};
        }
        if (DBG & 2) {
#pragma unroll
            for (int f = 0; f < P.n_frag; ++f) acc[0][f] += __uint_as_float(fr[f][0][0] ^ fr[f][1][3]);
            return;
        }
#pragma unroll
        for (int b = 0; b < P.n_blk; ++b) acc[b] = bsc_mfma_split<2>(fr[P.blk[b][0]], fr[P.blk[b][1]], acc[b]);
    };
    // convert step i: this lane's 16 floats of row l31 -> two bf16 terms -> this wave's column-block images of set i & 1
    auto convert = [&](int i) __attribute__((always_inline)) {
        if (DBG & 1) return;
        f32x4_t nx[4];
        const unsigned soff = (unsigned)((i % GR_STAGES) * GR_RAW);
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(nx[q]) : "v"(rr[q] + soff) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]));
        bsc_u32x4 t0[2], t1[2];
        unsigned pk[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bsc_split_pk<2>(nx[q][0], nx[q][1], pk); t0[0][2 * q] = pk[0]; t0[1][2 * q] = pk[1];
            bsc_split_pk<2>(nx[q][2], nx[q][3], pk); t0[0][2 * q + 1] = pk[0]; t0[1][2 * q + 1] = pk[1];
            bsc_split_pk<2>(nx[2 + q][0], nx[2 + q][1], pk); t1[0][2 * q] = pk[0]; t1[1][2 * q] = pk[1];
            bsc_split_pk<2>(nx[2 + q][2], nx[2 + q][3], pk); t1[0][2 * q + 1] = pk[0]; t1[1][2 * q + 1] = pk[1];
        }
        const unsigned set = (unsigned)((i & 1) * GR_IMG_SET);
#pragma unroll
        for (int term = 0; term < 2; ++term) {
            asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(wr0 + set), "v"(t0[term]), "n"(term * GB_IMG) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(wr1 + set), "v"(t1[term]), "n"(term * GB_IMG) : "memory");
        }
    };
#pragma unroll
    for (int i = 0; i < GR_STAGES; ++i)
        if (i < my_steps) issue(i);
    // step 0 has landed when at most the DMAs of steps 1, 2 are outstanding
    if (my_steps >= 3) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(8));
    else if (my_steps == 2) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(4));
    else __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    __syncthreads();
    asm volatile("" ::: "memory");
    convert(0);
    for (int i = 0; i < my_steps; ++i) {
        // step i + 1 has landed (own rows) when at most step i + 2's DMAs are outstanding
        if (i + 2 < my_steps) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(4));
        else __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own image writes of step i
        // every wave: its image of step i is written, its rows of step i + 1 have landed, it is done reading the images of
        // step i - 1 and the raw tile of step i
        __syncthreads();
        asm volatile("" ::: "memory");
        if (!(DBG & 4) && i + GR_STAGES < my_steps) issue(i + GR_STAGES);      // into the raw stage of step i
        const unsigned set = (unsigned)((i & 1) * GR_IMG_SET);
        auto work = [&](auto wc) __attribute__((always_inline)) {
            bsc_u32x2 f0[4][4], f1[4][4];
            frag_read(wc, set, 0, f0);
            if (i + 1 < my_steps) convert(i + 1);                 // (waits for its own raw reads: the fragments above land too)
            frag_read(wc, set, 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            // the 4 n_frag reads of k-step 1 (and the conversion's four image writes, when there was no conversion's wait
            // to cover the reads of k-step 0) may still be in flight: k-step 0's have landed when at most those remain
            if (i + 1 < my_steps) lda_like_wait(4 * GRAM_PLAN8[decltype(wc)::value].n_frag + 4);
            else lda_like_wait(4 * GRAM_PLAN8[decltype(wc)::value].n_frag);
            __builtin_amdgcn_sched_barrier(0);
            frag_mfma(wc, f0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            frag_mfma(wc, f1);
        };
        switch (wave) {
            case 0: work(std::integral_constant<int, 0>{}); break;
            case 1: work(std::integral_constant<int, 1>{}); break;
            case 2: work(std::integral_constant<int, 2>{}); break;
            case 3: work(std::integral_constant<int, 3>{}); break;
            case 4: work(std::integral_constant<int, 4>{}); break;
            case 5: work(std::integral_constant<int, 5>{}); break;
            case 6: work(std::integral_constant<int, 6>{}); break;
            default: work(std::integral_constant<int, 7>{}); break;
        }
    }
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    float* out = slab + (int64_t)blockIdx.x * NB * 1024;
    auto store = [&](auto wc) __attribute__((always_inline)) {
        constexpr int W = decltype(wc)::value;
        constexpr GramPlan P = GRAM_PLAN8[W];
#pragma unroll
        for (int b = 0; b < P.n_blk; ++b) {
            const int bi = gram_tri_index(8, P.frag[P.blk[b][0]], P.frag[P.blk[b][1]]);
#pragma unroll
            for (int r = 0; r < 16; ++r) out[bi * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[b][r];
        }
    };
    switch (wave) {
        case 0: store(std::integral_constant<int, 0>{}); break;
        case 1: store(std::integral_constant<int, 1>{}); break;
        case 2: store(std::integral_constant<int, 2>{}); break;
        case 3: store(std::integral_constant<int, 3>{}); break;
        case 4: store(std::integral_constant<int, 4>{}); break;
        case 5: store(std::integral_constant<int, 5>{}); break;
        case 6: store(std::integral_constant<int, 6>{}); break;
        default: store(std::integral_constant<int, 7>{}); break;
    }
}

// ---- round 4: the same product with the two waves of a SIMD taking TURNS on the matrix pipe (option gram_pp = 1; OFF) ----
//
// (MI355X_MICROARCH.md, "Two waves per SIMD": one wave in a matrix segment beside its partner in an LDS / DMA segment, a
// barrier between segments.)  gram256_bx_kernel's one barrier a step puts all eight waves into the same phase --
// fragment reads and conversion, then MFMAs: the matrix pipe idles while everybody reads, the LDS pipe while everybody
// multiplies.  Here a step is two half-steps with a barrier after each:
//     half-step 2 i    : waves 0-3 multiply step i                | waves 4-7 read the fragments of step i, convert step i + 1
//     half-step 2 i + 1: waves 0-3 read step i + 1, convert i + 2 | waves 4-7 multiply step i
// Images of step j are written in half-steps 2 j - 3 (waves 0-3) and 2 j - 2 (waves 4-7) and first read in 2 j - 1: two
// image sets as before.  The raw tile of step j + 1 is last read in half-step 2 j, so a wave sends its four DMAs of the
// step that reuses the stage at the START OF ITS NEXT MATRIX SEGMENT (a DMA costs ~60 cycles beside bare MFMAs, 100-185
// in a segment full of LDS reads).  The images and the raw ring are SEPARATE __shared__ arrays: the fragment reads are
// compiler-visible ds_read_b64_tr_b16 builtins (two reads land in adjacent registers: no copies into the MFMA operand;
// the waits are the compiler's), and they do not draw a vmcnt(0) for the DMAs in flight into the other array.
//
// MEASURED (tools/ab_gram_pp.py, profiles/r04_ab_gram_pp.txt): same bits as gram256_bx_kernel, 267 us of kernel against its
// 261 (a first form inside gram256_bx_kernel, asm reads and register copies kept: 284-288).  The deletion builds say why
// taking turns buys nothing here: fragment reads + barriers alone 75 us, + conversion 135, MFMAs + reads 175, all
// but the DMAs 220, everything 290 -- the parts ADD UP in either loop.  The transposed reads alone run the LDS pipe at
// ~100 of its 128 bytes a clock for those 75 us; with the conversion's reads and writes and the DMAs' writes a step
// moves ~210 KiB through LDS, and what the matrix pipe waits for is that pipe, whoever's turn it is.  The lever is
// fewer LDS bytes per MFMA (larger register blocks per wave), not the order of the segments.  Kept as an option, not used.
template <int DBG = 0>
__global__ __launch_bounds__(512, 1) void gram256_pp_kernel(const float* __restrict__ X, int64_t ldx, int64_t N,
                                                            float* __restrict__ slab, int n_steps_total) {
    constexpr int NB = 36;
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef short v8s __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) v4s* tr_ptr;
    __shared__ __attribute__((aligned(1024))) char img[GR_IMG_BYTES];
    __shared__ __attribute__((aligned(1024))) char raw[GR_STAGES * GR_RAW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // (not __syncthreads(): its fence waits for every DMA in flight -- vmcnt(0) -- now that the kernel has compiler-visible
    // LDS accesses.  What a barrier here has to order is this wave's image writes, and the DMAs it waited for by hand.)
    auto wg_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    bsc_f32x16 acc[5];
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    // image writes (this lane's 16 columns of row l31 of column block `wave`) and transposed reads: byte offsets in `img`
    const int sw = 2 * ((l31 >> 1) & 3);
    const int wr0 = wave * 2 * GB_IMG + l31 * 64 + 8 * ((4 * half) ^ sw);
    const int wr1 = wave * 2 * GB_IMG + l31 * 64 + 8 * ((4 * half + 2) ^ sw);
    const int gg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    int rd[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 8 * (gg >> 1) + 4 * e + tq, ch = 4 * (gg & 1) + tp;
        rd[e] = row * 64 + 8 * (ch ^ (2 * ((row >> 1) & 3)));
    }
    // raw-tile reads of the converting lane (inline asm: DMA-written memory, waited for by hand): row l31, chunks 8 wave + 4 half + q
    const unsigned raw_base = (unsigned)(uintptr_t)(bsc_lds_ptr)raw;
    unsigned rr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rr[q] = raw_base + (unsigned)(l31 * 1024 + 16 * ((8 * wave + 4 * half + q) ^ (l31 & 15)));

    const int64_t stride = gridDim.x;
    const int my_steps = (int)((n_steps_total - (int64_t)blockIdx.x + stride - 1) / stride);     // >= 1 (host: grid <= steps)
    auto issue = [&](int i) __attribute__((always_inline)) {
        const int64_t row0 = ((int64_t)blockIdx.x + (int64_t)i * stride) * 32;
        const int64_t rem = N - row0;                 // > 0
        const uint64_t bytes = ((uint64_t)(rem - 1) * (uint64_t)ldx + 256u) * 4u;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + row0 * ldx), 0, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes, 0x00020000);
        char* const dst = raw + (i % GR_STAGES) * GR_RAW + (4 * wave) * 1024;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int row = 4 * wave + jj;
            const unsigned voff = (unsigned)(row * (int)(ldx * 4) + 16 * (lane ^ (row & 15)));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bsc_lds_ptr)(dst + jj * 1024), 16, voff, 0, 0, 2);
        }
    };
    auto tr8 = [&](int off) __attribute__((always_inline)) {           // rows r0 .. r0 + 7 of this lane's column: one operand term
        const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(img + rd[0] + off));
        const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(img + rd[1] + off));
        return __builtin_bit_cast(bsc_u32x4, (v8s)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto frag_read = [&](auto wc, int set, int s, bsc_u32x4 (&fr)[4][2]) __attribute__((always_inline)) {
        constexpr GramPlan P = GRAM_PLAN8[decltype(wc)::value];
#pragma unroll
        for (int f = 0; f < P.n_frag; ++f) {
            const int off = set + P.frag[f] * 2 * GB_IMG + s * 1024;
            fr[f][0] = tr8(off);
            fr[f][1] = tr8(off + GB_IMG);
        }
    };
    auto frag_mfma = [&](auto wc, bsc_u32x4 (&fr)[4][2]) __attribute__((always_inline)) {
        constexpr GramPlan P = GRAM_PLAN8[decltype(wc)::value];
        if (DBG & 2) {
#pragma unroll
            for (int f = 0; f < P.n_frag; ++f) acc[0][f] += __uint_as_float(fr[f][0][0] ^ fr[f][1][3]);
            return;
        }
#pragma unroll
        for (int b = 0; b < P.n_blk; ++b) acc[b] = bsc_mfma_split<2>(fr[P.blk[b][0]], fr[P.blk[b][1]], acc[b]);
    };
    auto convert = [&](int i) __attribute__((always_inline)) {
        if (DBG & 1) return;
        f32x4_t nx[4];
        const unsigned soff = (unsigned)((i % GR_STAGES) * GR_RAW);
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(nx[q]) : "v"(rr[q] + soff) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]));
        bsc_u32x4 t0[2], t1[2];
        unsigned pk[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bsc_split_pk<2>(nx[q][0], nx[q][1], pk); t0[0][2 * q] = pk[0]; t0[1][2 * q] = pk[1];
            bsc_split_pk<2>(nx[q][2], nx[q][3], pk); t0[0][2 * q + 1] = pk[0]; t0[1][2 * q + 1] = pk[1];
            bsc_split_pk<2>(nx[2 + q][0], nx[2 + q][1], pk); t1[0][2 * q] = pk[0]; t1[1][2 * q] = pk[1];
            bsc_split_pk<2>(nx[2 + q][2], nx[2 + q][3], pk); t1[0][2 * q + 1] = pk[0]; t1[1][2 * q + 1] = pk[1];
        }
        const int set = (i & 1) * GR_IMG_SET;
#pragma unroll
        for (int term = 0; term < 2; ++term) {
            *reinterpret_cast<bsc_u32x4*>(img + wr0 + set + term * GB_IMG) = t0[term];
            *reinterpret_cast<bsc_u32x4*>(img + wr1 + set + term * GB_IMG) = t1[term];
        }
    };
#pragma unroll
    for (int i = 0; i < GR_STAGES; ++i)
        if (i < my_steps) issue(i);
    // step 0 has landed when at most the DMAs of steps 1, 2 are outstanding
    if (my_steps >= 3) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(8));
    else if (my_steps == 2) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(4));
    else __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    wg_barrier();
    convert(0);
    // before half-step -1: everybody's image of step 0 is written and its rows of step 1 have landed (step 2's may be out)
    if (my_steps >= 3) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(4));
    else __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    wg_barrier();
    // one copy of the half-step loop per wave (its plan and its group are compile-time constants in it); all copies pass
    // the same 2 my_steps + 1 barriers
    auto run = [&](auto wc) __attribute__((always_inline)) {
        constexpr int GRP = decltype(wc)::value >> 2;
        bsc_u32x4 f0[4][2], f1[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) { f0[a][c] = bsc_u32x4{0u, 0u, 0u, 0u}; f1[a][c] = bsc_u32x4{0u, 0u, 0u, 0u}; }
        auto load_phase = [&](int j) __attribute__((always_inline)) {
            const int set = (j & 1) * GR_IMG_SET;
            frag_read(wc, set, 0, f0);
            if (j + 1 < my_steps) convert(j + 1);
            frag_read(wc, set, 1, f1);
        };
        for (int h = -1; h < 2 * my_steps; ++h) {
            if (GRP == 0) {
                if (h & 1) {
                    const int j = (h + 1) >> 1;
                    if (j < my_steps) load_phase(j);
                } else {
                    const int j4 = (h >> 1) + 3;                // the stage of step h / 2: last read two half-steps ago
                    if (!(DBG & 4) && j4 < my_steps) issue(j4);
                    frag_mfma(wc, f0);
                    frag_mfma(wc, f1);
                }
            } else {
                if (h & 1) {
                    const int j4 = (h - 1) / 2 + 4;             // the stage of step (h + 1) / 2: last read a half-step ago
                    if (!(DBG & 4) && j4 < my_steps) issue(j4);
                    if (h > 0) {
                        frag_mfma(wc, f0);
                        frag_mfma(wc, f1);
                    }
                } else {
                    load_phase(h >> 1);
                }
            }
            if (!(h & 1)) {
                // the rows of step h / 2 + 2 are converted from the next half-step on: own DMAs landed, step h / 2 + 3's may be out
                if ((h >> 1) + 3 < my_steps) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(4));
                else __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
            }
            wg_barrier();
        }
    };
    switch (wave) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 5: run(std::integral_constant<int, 5>{}); break;
        case 6: run(std::integral_constant<int, 6>{}); break;
        default: run(std::integral_constant<int, 7>{}); break;
    }
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    float* out = slab + (int64_t)blockIdx.x * NB * 1024;
    auto store = [&](auto wc) __attribute__((always_inline)) {
        constexpr GramPlan P = GRAM_PLAN8[decltype(wc)::value];
#pragma unroll
        for (int b = 0; b < P.n_blk; ++b) {
            const int bi = gram_tri_index(8, P.frag[P.blk[b][0]], P.frag[P.blk[b][1]]);
#pragma unroll
            for (int r = 0; r < 16; ++r) out[bi * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[b][r];
        }
    };
    switch (wave) {
        case 0: store(std::integral_constant<int, 0>{}); break;
        case 1: store(std::integral_constant<int, 1>{}); break;
        case 2: store(std::integral_constant<int, 2>{}); break;
        case 3: store(std::integral_constant<int, 3>{}); break;
        case 4: store(std::integral_constant<int, 4>{}); break;
        case 5: store(std::integral_constant<int, 5>{}); break;
        case 6: store(std::integral_constant<int, 6>{}); break;
        default: store(std::integral_constant<int, 7>{}); break;
    }
}

// C[d][e] = scale * sum over workgroups of their partials, float64, in workgroup order -- in two levels (one thread per
// element walking all 256 partials took ~100 us of a 360-us call): chunk sums of GR_CHUNK workgroups' partials (every CU
// busy), then the chunks in order; both triangles written.
constexpr int GR_CHUNK = 16;

__global__ __launch_bounds__(256) void gram_bx_reduce1_kernel(const float* __restrict__ slab, int n_wg, int n_elem,
                                                              double* __restrict__ part) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_elem) return;
    const int w0 = blockIdx.y * GR_CHUNK;
    float v[GR_CHUNK];
#pragma unroll
    for (int j = 0; j < GR_CHUNK; ++j) v[j] = w0 + j < n_wg ? slab[(int64_t)(w0 + j) * n_elem + idx] : 0.f;
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < GR_CHUNK; ++j) sum += (double)v[j];
    part[(int64_t)blockIdx.y * n_elem + idx] = sum;
}

__global__ __launch_bounds__(256) void gram_bx_reduce_kernel(const double* __restrict__ part, int n_chunks, int DB, int D,
                                                             float* __restrict__ C, int64_t sc_m, int64_t sc_n, float scale) {
    const int NB = DB * (DB + 1) / 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;          // element of the block list
    if (idx >= NB * 1024) return;
    const int bi = idx >> 10, within = idx & 1023, row = within >> 5, col = within & 31;
    int db = 0, rem = bi, len = DB;
    while (rem >= len) { rem -= len; --len; ++db; }
    const int eb = db + rem;
    double sum = 0.0;
    for (int c = 0; c < n_chunks; ++c) sum += part[(int64_t)c * NB * 1024 + idx];
    const int d = 32 * db + row, e = 32 * eb + col;
    // (a diagonal block holds both of its triangles, but its (d, e) and (e, d) add the cross terms h_d m_e, m_d h_e in
    // opposite orders: the upper one is written to both places, so that the result is symmetric to the bit)
    if (d < D && e < D && (db != eb || col >= row)) {
        const float val = (float)sum * scale;
        C[d * sc_m + e * sc_n] = val;
        if (d != e) C[e * sc_m + d * sc_n] = val;
    }
}

}  // namespace

// X [N x D] row-major (ldx), D a multiple of 4 up to 256: C = scale * X^T X.  *handled = 0 when the shape is not
// taken (the caller then runs the f32 product).
int bsc_gram_split(bsc_ctx* ctx, const float* X, int64_t ldx, int64_t N, int64_t D, float* C, int64_t sc_m, int64_t sc_n,
                   float scale, int* handled) {
    *handled = 0;
    // (the kernels address a step's rows through 32-bit byte offsets, row * ldx * 4 with row < 32: a leading dimension of
    // 2^25 floats or more would wrap them -- such a view is declined here and takes the general product)
    if (D < 32 || D > 256 || D % 32 != 0 || N < 4096 || ldx % 4 != 0 || ldx < D || ldx >= ((int64_t)1 << 25) ||
        (((uintptr_t)X) & 15) != 0)
        return BSC_OK;
    const int DB = (int)(D / 32), NB = DB * (DB + 1) / 2;
    const int64_t steps = (N + 31) / 32;
    int64_t n_wg = (DB == 8 ? 1 : 2) * (int64_t)ctx->cu_count;
    if (n_wg > steps) n_wg = steps;
    void* ws = nullptr;
    const int n_chunks = (int)((n_wg + GR_CHUNK - 1) / GR_CHUNK);
    const size_t slab_bytes = (size_t)n_wg * NB * 1024 * sizeof(float);
    int rc = bsc_workspace(ctx, slab_bytes + (size_t)n_chunks * NB * 1024 * sizeof(double), &ws);
    if (rc != BSC_OK) return rc;
    double* part = (double*)((char*)ws + slab_bytes);
    ctx->slab_rows = 0;
    {
        bsc_prof_scope prof(ctx);
#define BSC_GRAM(DB_) case DB_: hipLaunchKernelGGL(gram_bx_kernel<DB_>, dim3((unsigned)n_wg), dim3(GB_BLOCK), 0, ctx->stream, X, ldx, N, (int)D, (float*)ws, (int)steps); break;
        if (DB == 8 && ctx->gram_dbg) {
#define BSC_GRAMD(DBG) case DBG: hipLaunchKernelGGL(gram256_bx_kernel<DBG>, dim3((unsigned)n_wg), dim3(512), 0, ctx->stream, X, ldx, N, (float*)ws, (int)steps); break;
            switch (ctx->gram_dbg) { BSC_GRAMD(1) BSC_GRAMD(2) BSC_GRAMD(3) BSC_GRAMD(4) BSC_GRAMD(5) BSC_GRAMD(6) BSC_GRAMD(7) default: break; }
#undef BSC_GRAMD
        } else if (DB == 8 && ctx->gram_pp) hipLaunchKernelGGL((gram256_pp_kernel<0>), dim3((unsigned)n_wg), dim3(512), 0, ctx->stream, X, ldx, N, (float*)ws, (int)steps);
        else if (DB == 8) hipLaunchKernelGGL(gram256_bx_kernel<0>, dim3((unsigned)n_wg), dim3(512), 0, ctx->stream, X, ldx, N, (float*)ws, (int)steps);
        else switch (DB) {
            BSC_GRAM(1) BSC_GRAM(2) BSC_GRAM(3) BSC_GRAM(4) BSC_GRAM(5) BSC_GRAM(6) BSC_GRAM(7)
        }
#undef BSC_GRAM
    }
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_bx_reduce1_kernel, dim3((unsigned)((NB * 1024 + 255) / 256), (unsigned)n_chunks), dim3(256), 0,
                       ctx->stream, (const float*)ws, (int)n_wg, NB * 1024, part);
    BSC_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_bx_reduce_kernel, dim3((unsigned)((NB * 1024 + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double*)part, n_chunks, DB, (int)D, C, sc_m, sc_n, scale);
    BSC_LAUNCH_CHECK();
    *handled = 1;
    return BSC_OK;
}
