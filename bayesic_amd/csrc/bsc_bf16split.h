// Operand-split bf16 MFMA (ctx->mfma_split, bsc_set_mfma_split): pieces shared by the kernels that offer it.
//
// An f32 value x is written as a sum of bf16 terms by repeated round-to-nearest: h = bf16(x),
// m = bf16(x - h), l = bf16(x - h - m); |x - h| <= 2^-9 |x|, |x - h - m| <= 2^-18 |x|, and three terms
// carry all 24 bits.  A product of two such sums is evaluated on v_mfma_f32_32x32x16_bf16 (16 x the
// f32 MFMA rate, products exact, f32 accumulation) as
//   SPLIT = 2:  a_h b_h + a_h b_m + a_m b_h                       3 products, relative error <= ~2^-17 per term
//   SPLIT = 3:  ... + a_m b_m + a_h b_l + a_l b_h                 6 products, <= ~2^-23: the f32 class
// (the terms left out are 2^-18 resp. 2^-27 of the product).  The headline dtype of every config stays
// f32; this is an opt-in per context.
//
// Lane maps (gfx950, cdna_hip_programming.md "Fragment layout"): lane l = (r = l & 31, h = l >> 5) holds
// A[row r][k = 8 h + j] and B[k = 8 h + j][col r] in element j = 0..7 of its fragment; the result has
// its column on the lane and row (reg & 3) + 8 (reg >> 2) + 4 h in register reg.  A result tile is the
// next product's operand without lane movement when that product sums over the tile's ROW index:
// registers 8 s .. 8 s + 7, converted pairwise, are the fragment of k-step s, and its element j of lane
// half h is row 16 s + 8 (j >> 2) + 4 h + (j & 3) -- the other operand must follow that order.
#pragma once
#include "bsc_common.h"

typedef __bf16 bsc_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bsc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bsc_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned bsc_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bsc_u32x2 __attribute__((ext_vector_type(2)));
typedef float bsc_f32x16 __attribute__((ext_vector_type(16)));

// two f32 -> packed bf16 (round to nearest even; `lo` in bits 0..15): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned bsc_pk_bf16(float lo, float hi) {
    const bsc_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bsc_bf16x2));
}
__device__ __forceinline__ float bsc_bf16_lo(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bsc_bf16_hi(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

// (a, b) -> the packed terms of SPLIT-term sums; t[0] = leading terms
template <int SPLIT>
__device__ __forceinline__ void bsc_split_pk(float a, float b, unsigned (&t)[SPLIT]) {
    bsc_f32x2 ab = {a, b};
#pragma unroll
    for (int c = 0; c < SPLIT; ++c) {
        t[c] = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, bsc_bf16x2));
        if (c + 1 < SPLIT)      // exact: the difference of an f32 and its own leading bits -- both of the pair in ONE v_pk_add_f32
            ab -= bsc_f32x2{bsc_bf16_lo(t[c]), bsc_bf16_hi(t[c])};
    }
}

__device__ __forceinline__ bsc_f32x16 bsc_mfma_bf16(bsc_u32x4 a, bsc_u32x4 b, bsc_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bsc_bf16x8, a), __builtin_bit_cast(bsc_bf16x8, b),
                                                   c, 0, 0, 0);
}

// 16x16x32: lane l = (r = l & 15, g = l >> 4) holds A[row r][k = 8 g + j], B[k = 8 g + j][col r]; result register i of
// lane l is row 4 g + i, column r (the layout of v_mfma_f32_16x16x4_f32).
typedef float bsc_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bsc_f32x4 bsc_mfma16_bf16(bsc_u32x4 a, bsc_u32x4 b, bsc_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bsc_bf16x8, a), __builtin_bit_cast(bsc_bf16x8, b),
                                                   c, 0, 0, 0);
}

// The products of a SPLIT-term A with a SPLIT-term B, smallest terms first.
template <int SPLIT>
__device__ __forceinline__ bsc_f32x16 bsc_mfma_split(const bsc_u32x4 (&a)[SPLIT], const bsc_u32x4 (&b)[SPLIT],
                                                     bsc_f32x16 c) {
    if (SPLIT == 3) {
        c = bsc_mfma_bf16(a[0], b[2], c);
        c = bsc_mfma_bf16(a[2], b[0], c);
        c = bsc_mfma_bf16(a[1], b[1], c);
    }
    if (SPLIT >= 2) {
        c = bsc_mfma_bf16(a[0], b[1], c);
        c = bsc_mfma_bf16(a[1], b[0], c);
    }
    return bsc_mfma_bf16(a[0], b[0], c);
}

// LDS image of a [rows][128 x 16-bit] tile with plain 256-byte rows whose 16-byte chunks are XOR-permuted
// so that BOTH the row read of a 32x32x16 operand (ds_read_b128: lane = row, one chunk index per read) and
// the transposed read (ds_read_b64_tr_b16: lane = column) are free of bank conflicts
// (cdna_hip_programming.md T10, image (b)).  Byte offset of chunk `ch` (0..15) of row `row`:
__device__ __forceinline__ unsigned bsc_img256_off(int row, int ch) {
    return 256u * (unsigned)row + 16u * (unsigned)(ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// Address a lane gives ds_read_b64_tr_b16 to receive, in elements q = 0..3, rows r0 .. r0 + 3 of column
// c0 + (lane & 15) of such an image, c0 a multiple of 16: lane 4 q + p of each 16-lane group points at
// row r0 + q, columns c0 + 4 p .. + 3.  (r0 and c0 may differ between the four groups of a wave.)
__device__ __forceinline__ unsigned bsc_img256_tr_addr(int lane, int r0, int c0) {
    const int q = (lane >> 2) & 3, p = lane & 3;
    return bsc_img256_off(r0 + q, (c0 >> 3) + (p >> 1)) + 8u * (unsigned)(p & 1);
}

#define BSC_LDS_TR_B64(DST, ADDR, OFF) \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define BSC_LDS_B128(DST, ADDR, OFF) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
