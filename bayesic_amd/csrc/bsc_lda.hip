// Fixed-gamma local step of the LDA-style Dirichlet-Multinomial model (BASELINE
// config 4; SURVEY.md 8(b) names the entry bsc_lda_sstats):
//
//   sstats[k,v] = Bt[k,v] * sum_d Th[d,k] * C[d,v] / (sum_k' Th[d,k'] Bt[k',v])
//
// i.e. the algebra expression  Bt * dot(Th.T, C / dot(Th, Bt))  whose lowered
// form (bayesic/algebra.py:553-765) is two _tensordot GEMMs around an
// element-wise division.  Run through the generic executor that costs two
// docs x V intermediates (2.5 GB each per GPU at config 4); here both
// contractions and the division happen in ONE pass over C, and neither
// phinorm nor the ratio ever leaves the registers.
//
// Bound: fp32 MFMA (4*docs*V*K flops on 4*docs*V bytes = 128 flop/B at K=128).
//
// Workgroup = 4 waves = 128 vocabulary columns (32 per wave) x a range of
// documents, walked 32 at a time:
//   phase 1  P[32 d, 32 v] = Th[32 d, K] . Bt[K, 32 v]     (Bt column block in
//            registers for the whole kernel, Th tile from LDS)
//   ratio    R = C / P   in the MFMA result layout (C tile staged through LDS)
//   phase 2  S[K, 32 v] += Th[32 d, K]^T . R[32 d, 32 v]
// The result layout of v_mfma_f32_32x32x2_f32 (lane = column, 16 rows per lane)
// is exactly its B-operand layout when the contraction index is taken to be
// those rows, so R feeds phase 2 straight from registers.
#include "bsc_common.h"
#include "bsc_stream.h"
#include "bsc_bf16split.h"

namespace {

constexpr int LDA_BLOCK = 256;
constexpr int DT = 32;        // documents per step
constexpr int VT = 128;       // vocabulary columns per workgroup
constexpr int TH_LD = 132;    // Th tile row stride in LDS: 16-byte rows, conflict-free b128 reads
constexpr int C_LD = VT + 4;  // C tile row stride

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LdaArgs {
    const float* C;
    const float* Th;
    const float* Bt;
    float* out;        // sstats [K, V] (ld = ldo) when splits == 1, else partial [splits][K][V]
    int64_t ldc, ldth, ldb, ldo;
    int64_t docs, V;
    int64_t docs_per_split;   // multiple of DT
    int splits;
    int vec_c, vec_th;        // 16-byte global loads allowed
    int fast;                 // interior steps may use uniform-base + 32-bit-lane-offset loads
    float* ll_slab;           // BOUND: per-wave partial of sum_dv C log2(phinorm), [gridDim.y][gridDim.x][4]
};

// The words' part of the evidence lower bound (README.md:30-37; Hoffman, Blei, Bach 2010 eq. 7 with the
// per-word assignments at their optimum): sum_dv C_dv log(phinorm_dv).  phinorm exists only in the registers
// of these kernels, so the sum is taken there -- one v_log_f32 and one fma per element next to the v_rcp_f32
// of the ratio, per-lane float32 partials, one float per wave at the end, added in float64 in a fixed order
// (lda_ll_reduce_kernel).  Padded documents / columns carry zero counts and contribute 0 * log2(1e-30) = -0.
__global__ __launch_bounds__(256) void lda_ll_reduce_kernel(const float* __restrict__ slab, int64_t n, double mult,
                                                            double* __restrict__ out) {
    // 256 lanes x 8 independent loads in flight (a single wave walking the slab paid one memory round trip
    // per element: 12 us for 2 048 floats); lane partials in index order, lanes by the fixed butterfly, waves in order
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i0 = 0; i0 < n; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = i0 + u * 256 + threadIdx.x;
            v[u] = i < n ? slab[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    acc = wave_allsum_f64(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = mult * ((red[0] + red[1]) + (red[2] + red[3]));
}

// global -> registers for one step: Th tile [32][K] and C tile [32][128]
template <int KT>
struct Staged {
    float4 th[KT];   // 32 * 32KT / 4 float4 over 256 threads
    float4 c[4];
};

template <int KT>
__device__ __forceinline__ void stage_load(Staged<KT>& st, const LdaArgs& a, int64_t d0, int64_t d_end,
                                           int64_t v_base, int tid) {
    constexpr int K = 32 * KT;
#pragma unroll
    for (int p = 0; p < KT; ++p) {
        const int idx = tid + LDA_BLOCK * p;
        const int row = idx / (K / 4), c4 = idx % (K / 4);
        const int64_t d = d0 + row;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d < d_end) {
            const float* src = a.Th + d * a.ldth + 4 * c4;
            if (a.vec_th) f = *reinterpret_cast<const float4*>(src);
            else f = make_float4(src[0], src[1], src[2], src[3]);
        }
        st.th[p] = f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = tid + LDA_BLOCK * p;
        const int row = idx >> 5, c4 = idx & 31;
        const int64_t d = d0 + row, v = v_base + 4 * c4;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d < d_end) {
            const float* src = a.C + d * a.ldc + v;
            if (a.vec_c && v + 3 < a.V) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
                f = make_float4(t.x, t.y, t.z, t.w);
            } else {
                if (v < a.V) f.x = src[0];
                if (v + 1 < a.V) f.y = src[1];
                if (v + 2 < a.V) f.z = src[2];
                if (v + 3 < a.V) f.w = src[3];
            }
        }
        st.c[p] = f;
    }
}

// Interior steps (all 32 documents and all 128 columns inside, 16-byte loads allowed): the tile
// bases are uniform and a lane's offset inside the tile fits 32 bits, so the loads need no
// per-lane 64-bit multiplies, bounds tests or zero fills -- VALU instructions that would
// each take ~4.6 cycles from the MFMA pipe (profiles/r01_ubench_mfma_valu_mix.txt).
template <int KT>
__device__ __forceinline__ void stage_load_fast(Staged<KT>& st, const LdaArgs& a, int64_t d0,
                                                int64_t v_base, int tid) {
    static_assert(KT == 1 || KT == 2 || KT == 4, "row of part p must be row0 + (256 / (8 KT)) p");
    // lane offsets inside the step's Th [32 x K] and C [32 x 128] tiles, recomputed per step
    // (six VALU instructions) rather than kept in two of the kernel's last free registers
    asm volatile("" : "+v"(tid));
    const unsigned th_off = (unsigned)((tid / (8 * KT)) * (int)a.ldth + 4 * (tid % (8 * KT))) * 4u;
    const unsigned c_off = (unsigned)((tid >> 5) * (int)a.ldc + 4 * (tid & 31)) * 4u;
    const char* th_base = reinterpret_cast<const char*>(a.Th + d0 * a.ldth);
    const char* c_base = reinterpret_cast<const char*>(a.C + d0 * a.ldc + v_base);
    const int64_t th_part = (int64_t)(LDA_BLOCK / (8 * KT)) * a.ldth * 4, c_part = 8 * a.ldc * 4;
#pragma unroll
    for (int p = 0; p < KT; ++p)
        st.th[p] = *reinterpret_cast<const float4*>(th_base + p * th_part + th_off);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(c_base + p * c_part + c_off));
        st.c[p] = make_float4(t.x, t.y, t.z, t.w);
    }
}

template <int KT>
__device__ __forceinline__ void stage_store(const Staged<KT>& st, float* th, float* ct, int tid) {
    constexpr int K = 32 * KT;
#pragma unroll
    for (int p = 0; p < KT; ++p) {
        const int idx = tid + LDA_BLOCK * p;
        const int row = idx / (K / 4), c4 = idx % (K / 4);
        *reinterpret_cast<float4*>(th + row * TH_LD + 4 * c4) = st.th[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = tid + LDA_BLOCK * p;
        const int row = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<float4*>(ct + row * C_LD + 4 * c4) = st.c[p];
    }
}

// K = 32 * KT topics.
//
// MFMA operand maps (v_mfma_f32_32x32x2_f32, D = A.B + C, lane l, h = l >> 5, j = l & 31):
//   A[i = j][kk = h],  B[kk = h][col = j],  D[row = (r&3) + 8(r>>2) + 4h][col = j] in register r.
// phase 1: i = document, col = v; the K contraction is split between the lane halves
//   (half h owns k = (K/2) h + t) so a lane's A values are contiguous in LDS (b128 reads).
// phase 2: col = v, contraction over the 32 documents two at a time: step r pairs the
//   documents row_r + 4h -- exactly what register r of P holds -- and i = topic with the
//   interleaved map k = KT * i + kt, so one b128 read of Th[d][KT*i ..] feeds the KT tiles.
template <int KT, bool BOUND>
__global__ __launch_bounds__(LDA_BLOCK, 2) void lda_sstats_kernel(LdaArgs a) {
    constexpr int K = 32 * KT;
    __shared__ __attribute__((aligned(16))) float th_s[2][DT * TH_LD];
    __shared__ __attribute__((aligned(16))) float c_s[2][DT * C_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int64_t v_base = (int64_t)blockIdx.x * VT;
    const int64_t v = v_base + 32 * wave + j;
    const bool v_ok = v < a.V;
    const int split = blockIdx.y;
    const int64_t d_begin = (int64_t)split * a.docs_per_split;
    const int64_t d_end = d_begin + a.docs_per_split < a.docs ? d_begin + a.docs_per_split : a.docs;

    // Bt column block of this lane: k = (K/2) h + t
    float bt[K / 2];
#pragma unroll
    for (int t = 0; t < K / 2; ++t)
        bt[t] = v_ok ? a.Bt[(int64_t)((K / 2) * h + t) * a.ldb + v] : 0.f;

    f32x16 S[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
    float ll = 0.f;

    const bool fast = (KT == 1 || KT == 2 || KT == 4) && a.fast && v_base + VT <= a.V;   // uniform
    Staged<KT> st;
    if (d_begin < d_end) {
        stage_load<KT>(st, a, d_begin, d_end, v_base, tid);
        stage_store<KT>(st, th_s[0], c_s[0], tid);
    }
    __syncthreads();
    int cur = 0;
    for (int64_t d0 = d_begin; d0 < d_end; d0 += DT) {
        const bool more = d0 + DT < d_end;
        if (more) {
            if constexpr (KT == 1 || KT == 2 || KT == 4) {
                if (fast && d0 + 2 * DT <= d_end) stage_load_fast<KT>(st, a, d0 + DT, v_base, tid);
                else stage_load<KT>(st, a, d0 + DT, d_end, v_base, tid);
            } else {
                stage_load<KT>(st, a, d0 + DT, d_end, v_base, tid);
            }
        }
        const float* th = th_s[cur];
        const float* ct = c_s[cur];

        // ---- phase 1: P = Th . Bt --------------------------------------------------
        f32x16 P;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* arow = th + j * TH_LD + (K / 2) * h;
#pragma unroll
        for (int t4 = 0; t4 < K / 8; ++t4) {
            const float4 av = *reinterpret_cast<const float4*>(arow + 4 * t4);
            // the first MFMA takes the constant 0 as its C operand: no 16 moves to clear P
            P = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bt[4 * t4 + 0], t4 == 0 ? zero16 : P, 0, 0, 0);
            P = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bt[4 * t4 + 1], P, 0, 0, 0);
            P = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bt[4 * t4 + 2], P, 0, 0, 0);
            P = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bt[4 * t4 + 3], P, 0, 0, 0);
        }

        // ---- ratio in the result layout, then phase 2: S += Th^T . R -------------------
        // Four result registers at a time: their counts come with one LDS round trip, the
        // division is v_rcp_f32 (1 ulp; the statistic's tolerance is 3e-5), and their 4 x KT
        // MFMAs run while the next four are fetched and divided -- the MFMA pipe would
        // otherwise idle through the whole ratio stretch.  Padded rows / columns have count 0.
        float c_n[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c_n[i] = ct[(i + 4 * h) * C_LD + 32 * wave + j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = c_n[i];
            if (q + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    c_n[i] = ct[(i + 8 * (q + 1) + 4 * h) * C_LD + 32 * wave + j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * q + i;
                // padded documents / columns were staged as zero counts; their P is 0 (Th or Bt
                // is 0 there), so the clamp alone keeps 0 * rcp(0) from becoming NaN -- one
                // v_max instead of a compare and a select per element (real P are > 0)
                const float pc = fmaxf(P[r], 1.0e-30f);
                if constexpr (BOUND) ll = __builtin_fmaf(c[i], __builtin_amdgcn_logf(pc), ll);
                P[r] = c[i] * __builtin_amdgcn_rcpf(pc);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * q + i;
                const int row = i + 8 * q + 4 * h;
                const float* src = th + row * TH_LD + KT * j;
                float av[KT];
                if constexpr (KT == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(src);
                    av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
                } else if constexpr (KT == 2) {
                    const float2 t = *reinterpret_cast<const float2*>(src);
                    av[0] = t.x; av[1] = t.y;
                } else {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) av[kt] = src[kt];
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kt], P[r], S[kt], 0, 0, 0);
            }
        }

        if (more) stage_store<KT>(st, th_s[cur ^ 1], c_s[cur ^ 1], tid);
        __syncthreads();
        cur ^= 1;
    }

    if constexpr (BOUND) {
        const float t = wave_allsum(ll);
        if (lane == 0) a.ll_slab[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave] = t;
    }
    // ---- epilogue: S[kt] register r is topic k = KT * row_r + kt, column v --------------
    if (!v_ok) return;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int64_t k = KT * row + kt;
            if (a.splits == 1)
                a.out[k * a.ldo + v] = S[kt][r] * a.Bt[k * a.ldb + v];
            else
                a.out[((int64_t)split * K + k) * a.V + v] = S[kt][r];
        }
}

// sstats = Bt * (sum of the split partials), fixed order
__global__ __launch_bounds__(256) void lda_reduce_kernel(const float* __restrict__ partial, int splits,
                                                         int64_t K, int64_t V,
                                                         const float* __restrict__ Bt, int64_t ldb,
                                                         float* __restrict__ out, int64_t ldo) {
    const int64_t n = K * V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float tot = partial[i];
        for (int s = 1; s < splits; ++s) tot += partial[(int64_t)s * n + i];
        const int64_t k = i / V, v = i - k * V;
        out[k * ldo + v] = tot * Bt[k * ldb + v];
    }
}

// ---- K = 128: persistent, operands by LDS-DMA ---------------------------------------------------
// The same arithmetic as lda_sstats_kernel<4> (bit for bit: same operand maps, same order), with
// what csrc/bsc_gemm.hip's stream kernel learned about this part:
//   * the Th [32 x 128] and C [32 x 128] tiles of a step arrive by `buffer_load_dwordx4 ... lds`
//     (4 + 4 per wave; ~45 cycles of matrix-pipe time apiece where a VGPR load takes ~115 and
//     needs a ds_write pass besides).  The Th image keeps rows of 512 B with chunk c of row d at
//     position c ^ (d % 16): phase 1 (lane = document, one chunk index per read) and phase 2
//     (lane = chunk, one document per half-wave) both read it with conflict-free ds_read_b128;
//     the C tile is a plain [32][128].  Documents past the end and columns past V are zeros by
//     out-of-range descriptor offsets;
//   * persistent: the (column block, 32-document step) units are dealt as in bsc_stream.h -- whole
//     rounds of column blocks, then the blocks left over split along the documents among all
//     workgroups, their partial statistics added in order by lda_stream_fixup_kernel.  (The
//     one-block-per-workgroup kernel splits ALL blocks five ways at config 4's size: 256 MB of
//     partials and a last round 4.5 % empty);
//   * LDS fragments two reads ahead in a ring of registers, hand-counted lgkmcnt; the step body
//     exists once per LDS buffer so that the buffer is an immediate offset.
constexpr int LS_STAGE = 32768;        // bytes: Th image 16 KiB, C tile 16 KiB
constexpr unsigned LS_OUTSIDE = 0x80000000u;

struct LdaStreamArgs {
    const float* C;
    const float* Th;
    const float* Bt;
    float* out;
    float* slab;       // [2 n_wg][128 k][128 v] partial statistics
    int64_t ldc, ldth, ldb, ldo, docs, V;
    int n_kt, sk_q, sk_r, sk_stream, n_wg, rounds, tail_tiles;   // tiles = 128-column blocks, units = 32-document steps
    int d_tail;        // documents in a block's last step, 1..32
    int K;             // 32, 64 or 128
    float* ll_slab;    // BOUND: [n_wg][4] per-wave partials of sum_dv C log2(phinorm)
};
typedef stream_args_cptr<LdaStreamArgs> lda_args_cptr;

#define LDA_LDS_B128(DST, ADDR, OFF) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define LDA_LDS_B64(DST, ADDR, OFF) \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define LDA_LDS_B32(DST, ADDR, OFF) \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")

typedef float lda_f32x4 __attribute__((ext_vector_type(4)));
typedef float lda_f32x2 __attribute__((ext_vector_type(2)));

// s_waitcnt lgkmcnt(n) alone (n a compile-time constant after unrolling)
__device__ __forceinline__ void lda_wait_lgkm(int n) {
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0xC07F); break;
        case 1: __builtin_amdgcn_s_waitcnt(0xC17F); break;
        case 2: __builtin_amdgcn_s_waitcnt(0xC27F); break;
        case 3: __builtin_amdgcn_s_waitcnt(0xC37F); break;
        case 4: __builtin_amdgcn_s_waitcnt(0xC47F); break;
        case 5: __builtin_amdgcn_s_waitcnt(0xC57F); break;
        case 6: __builtin_amdgcn_s_waitcnt(0xC67F); break;
        case 7: __builtin_amdgcn_s_waitcnt(0xC77F); break;
        default: __builtin_amdgcn_s_waitcnt(0xC87F); break;
    }
}

// KT = K / 32 in {1, 2, 4}: a Th row is RB = 128 KT bytes = CPR = 8 KT chunks; 8 / KT rows per DMA
// instruction, KT instructions per wave and step; chunk c of row d at position c ^ (d % min(CPR, 16)).
template <int KT, bool BOUND>
__device__ __forceinline__ void lda_sstats_stream_body(const LdaStreamArgs& a, char* lds) {
    constexpr int K = 32 * KT, RB = 128 * KT, CPR = 8 * KT, SWZ = (CPR < 16 ? CPR : 16) - 1, RPI = 8 / KT;
    constexpr int N1 = 4 * KT;                       // phase-1 reads (and 4-MFMA groups) per step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    int w = blockIdx.x;
    int n_units, tail_u0, tail_cnt;
    {
        const lda_args_cptr gc = stream_cold_args<LdaStreamArgs>();
        if ((gc->n_wg & 7) == 0) w = (w & 7) * (gc->n_wg >> 3) + (w >> 3);
        tail_u0 = stream_first_unit(gc, w);
        tail_cnt = stream_first_unit(gc, w + 1) - tail_u0;
        n_units = gc->rounds * gc->n_kt + tail_cnt;
    }
    if (n_units == 0) {
        if (BOUND && lane == 0) a.ll_slab[blockIdx.x * 4 + wave] = 0.f;
        return;
    }
    float ll = 0.f;
    const int64_t step_th = 32 * a.ldth, step_c = 32 * a.ldc;
    const int last_kt = a.n_kt - 1, d_tail = a.d_tail;

    // ---- issuing side
    StreamCursor ic;
    ic.begin(stream_cold_args<LdaStreamArgs>(), w, tail_u0, tail_cnt);
    int issued = 0;
    unsigned vth[KT], vc[4];
#pragma unroll
    for (int jj = 0; jj < KT; ++jj) {
        const int row = RPI * (KT * wave + jj) + lane / CPR;
        vth[jj] = (unsigned)(row * (int)a.ldth + 4 * ((lane % CPR) ^ (row & SWZ))) * 4u;
    }
    const float *th_ptr, *c_ptr;
    auto issue_tile = [&]() __attribute__((always_inline)) {
        const lda_args_cptr gc = stream_cold_args<LdaStreamArgs>();
        const int64_t v_base = (int64_t)ic.t * VT, v_left = gc->V - v_base;
        const int ldc = (int)gc->ldc;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int row = 2 * (4 * wave + jj) + (lane >> 5), col = 4 * (lane & 31);
            vc[jj] = col < v_left ? (unsigned)(row * ldc + col) * 4u : LS_OUTSIDE;
        }
        th_ptr = gc->Th + (int64_t)ic.kt * 32 * gc->ldth;
        c_ptr = gc->C + (int64_t)ic.kt * 32 * gc->ldc + v_base;
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {
        const auto rth = __builtin_amdgcn_make_buffer_rsrc((void*)stream_uniform_ptr(th_ptr), 0, LS_OUTSIDE, 0x00020000);
        const auto rc = __builtin_amdgcn_make_buffer_rsrc((void*)stream_uniform_ptr(c_ptr), 0, LS_OUTSIDE, 0x00020000);
        char* const dst = lds + buf * LS_STAGE + wave * 4096;            // C tile: 4 KiB per wave at + 16 KiB
        char* const dth = lds + buf * LS_STAGE + wave * (KT * 1024);      // Th image: KT KiB per wave
        if (__builtin_expect(ic.kt != last_kt || d_tail == DT, 1)) {
#pragma unroll
            for (int jj = 0; jj < KT; ++jj)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rth, (bsc_lds_ptr)(dth + jj * 1024), 16, vth[jj], 0, 0, 0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (bsc_lds_ptr)(dst + 16384 + jj * 1024), 16, vc[jj], 0, 0, 2);
        } else {
#pragma unroll
            for (int jj = 0; jj < KT; ++jj) {
                const bool in = RPI * (KT * wave + jj) + lane / CPR < d_tail;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rth, (bsc_lds_ptr)(dth + jj * 1024), 16, in ? vth[jj] : LS_OUTSIDE, 0, 0, 0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bool in = 2 * (4 * wave + jj) + (lane >> 5) < d_tail;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (bsc_lds_ptr)(dst + 16384 + jj * 1024), 16, in ? vc[jj] : LS_OUTSIDE, 0, 0, 2);
            }
        }
        ++issued;
        th_ptr += step_th;
        c_ptr += step_c;
        if (__builtin_expect(ic.step(), 0) && issued < n_units) {
            ++ic.round;
            ic.segment(stream_cold_args<LdaStreamArgs>(), w, tail_u0);
            issue_tile();
        }
    };

    // ---- fragment addresses in buffer 0 (buffer 1: + LS_STAGE as an immediate)
    const unsigned lb = (unsigned)(uintptr_t)(bsc_lds_ptr)lds;
    unsigned p1[N1], p2[8];
#pragma unroll
    for (int t4 = 0; t4 < N1; ++t4)        // phase 1: document j, chunk N1 h + t4
        p1[t4] = lb + (unsigned)(j * RB + (((N1 * h + t4) ^ (j & SWZ)) * 16));
    // phase 2: the KT topics KT j .. of document i + 8 q + 4 h (chunk KT j / 4, byte (4 KT j) % 16 of it),
    // c2 = i + 4 (q & 1)
#pragma unroll
    for (int c2 = 0; c2 < 8; ++c2)
        p2[c2] = lb + (unsigned)(4 * h * RB + ((((KT * j) >> 2) ^ ((4 * h) & SWZ) ^ (((c2 & 3) | ((c2 >> 2) << 3)) & SWZ)) * 16) +
                                 ((4 * KT * j) & 15));
    const unsigned pc = lb + 16384 + (unsigned)(4 * h * 512 + (32 * wave + j) * 4);

    float bt[K / 2];      // Bt[(K / 2) h + t][v]; at the end of a block: the factor Bt[KT row_r + kt][v] in bt[16 kt + r]
#pragma unroll
    for (int t = 0; t < K / 2; ++t) bt[t] = 0.f;
    f32x16 S[KT];
    f32x16 P;
    float ring[5][4];

    // ---- computing side
    StreamCursor cc;
    cc.begin(stream_cold_args<LdaStreamArgs>(), w, tail_u0, tail_cnt);
    bool tile_start = true, whole = false, v_ok = false;
    unsigned v_lane = 0;          // byte offset of this lane's column in a row of Bt / out (clamped inside V)

    auto step_body = [&](auto buf_c, int u) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr int OFF = BUF * LS_STAGE;
        if (__builtin_expect(tile_start, 0)) {
            tile_start = false;
            const lda_args_cptr gc = stream_cold_args<LdaStreamArgs>();
            const int64_t v = (int64_t)cc.t * VT + 32 * wave + j;
            v_ok = v < gc->V;
            v_lane = (unsigned)(v_ok ? v : gc->V - 1) * 4u;
            whole = cc.kt == 0 && cc.left == gc->n_kt;
            const unsigned off = v_lane + (unsigned)((K / 2) * h) * (unsigned)gc->ldb * 4u;
            const float* base = gc->Bt;
            const int64_t ldb = gc->ldb;
#pragma unroll
            for (int t = 0; t < K / 2; ++t) {
                asm volatile("global_load_dword %0, %1, %2" : "+v"(bt[t]) : "v"(off), "s"(base) : "memory");
                base += ldb;
            }
            __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));      // (also the DMAs in flight: once per block)
#pragma unroll
            for (int t = 0; t < K / 2; ++t) {
                asm volatile("" : "+v"(bt[t]));
                bt[t] = v_ok ? bt[t] : 0.f;
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
        }
        // slots: 0..15 phase-1 reads (b128); then per q: the four counts (4 x b32), four Th reads (b128)
        constexpr int NS = N1 + 20;                  // slots per step
        auto slot_size = [](int s) { return s < N1 ? 1 : ((s - N1) % 5 == 0 ? 4 : 1); };
        auto issue_slot = [&](int s) __attribute__((always_inline)) {
            float (&dst)[4] = ring[s % 5];
            if (s < N1) {
                lda_f32x4 v4;
                LDA_LDS_B128(v4, p1[s], OFF);
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = v4[e];
            } else {
                const int q = (s - N1) / 5, r5 = (s - N1) % 5;
                if (r5 == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) LDA_LDS_B32(dst[i], pc, OFF + (i + 8 * q) * 512);
                } else {
                    const int i = r5 - 1;
                    if (KT == 4) {
                        lda_f32x4 v4;
                        LDA_LDS_B128(v4, p2[i + 4 * (q & 1)], OFF + (i + 8 * q) * RB);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[e] = v4[e];
                    } else if (KT == 2) {
                        lda_f32x2 v2;
                        LDA_LDS_B64(v2, p2[i + 4 * (q & 1)], OFF + (i + 8 * q) * RB);
                        dst[0] = v2[0];
                        dst[1] = v2[1];
                    } else {
                        LDA_LDS_B32(dst[0], p2[i + 4 * (q & 1)], OFF + (i + 8 * q) * RB);
                    }
                }
            }
        };
        issue_slot(0);
        issue_slot(1);
        issue_slot(2);
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int after = (s + 1 < NS ? slot_size(s + 1) : 0) + (s + 2 < NS ? slot_size(s + 2) : 0);
            lda_wait_lgkm(after);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 3 < NS) issue_slot(s + 3);
            __builtin_amdgcn_sched_barrier(0);
            const float (&f)[4] = ring[s % 5];
            if (s < N1) {
                // the first MFMA takes the constant 0 as its C operand: no 16 moves to clear P
                P = __builtin_amdgcn_mfma_f32_32x32x2f32(f[0], bt[4 * s + 0], s == 0 ? zero16 : P, 0, 0, 0);
                P = __builtin_amdgcn_mfma_f32_32x32x2f32(f[1], bt[4 * s + 1], P, 0, 0, 0);
                P = __builtin_amdgcn_mfma_f32_32x32x2f32(f[2], bt[4 * s + 2], P, 0, 0, 0);
                P = __builtin_amdgcn_mfma_f32_32x32x2f32(f[3], bt[4 * s + 3], P, 0, 0, 0);
            } else {
                const int q = (s - N1) / 5, r5 = (s - N1) % 5;
                if (r5 == 0) {
                    // ratio in the result layout; padded documents / columns carry zero counts and P = 0:
                    // the clamp keeps 0 * rcp(0) from becoming NaN (real P are > 0)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                    {
                        // max as integers: one v_max_i32 (a float max quiets its input first -- a second
                        // instruction); the same result for every non-NaN P
                        const int pi = __float_as_int(P[4 * q + i]), lo = __float_as_int(1.0e-30f);
                        const float pc = __int_as_float(pi > lo ? pi : lo);
                        if constexpr (BOUND) ll = __builtin_fmaf(f[i], __builtin_amdgcn_logf(pc), ll);
                        P[4 * q + i] = f[i] * __builtin_amdgcn_rcpf(pc);
                    }
                } else {
                    const int i = r5 - 1;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
                        S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[kt], P[4 * q + i], S[kt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (__builtin_expect(cc.step(), 0)) {
            // ---- this run's share of the block is complete
            const lda_args_cptr gc = stream_cold_args<LdaStreamArgs>();
            if (whole) {
                // sstats = Bt * S: the factors Bt[4 row_r + kt][v] into the (now free) bt registers
                const int64_t ldb = gc->ldb, ldo = gc->ldo;
                const unsigned off = v_lane + (unsigned)(4 * KT * h) * (unsigned)ldb * 4u;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float* base = gc->Bt + (int64_t)(KT * ((r & 3) + 8 * (r >> 2)) + kt) * ldb;
                        asm volatile("global_load_dword %0, %1, %2" : "+v"(bt[16 * kt + r]) : "v"(off), "s"(base) : "memory");
                    }
                __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
                float* out = gc->out;
                if (v_ok) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            asm volatile("" : "+v"(bt[16 * kt + r]));
                            const int64_t k = KT * ((r & 3) + 8 * (r >> 2) + 4 * h) + kt;
                            *reinterpret_cast<float*>(reinterpret_cast<char*>(out + k * ldo) + v_lane) = S[kt][r] * bt[16 * kt + r];
                        }
                }
            } else {
                float* slot = gc->slab + ((int64_t)2 * w + (cc.round > gc->rounds ? 1 : 0)) * (K * VT) + 32 * wave + j;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = KT * ((r & 3) + 8 * (r >> 2) + 4 * h) + kt;
                        slot[k * VT] = S[kt][r];
                    }
            }
            ++cc.round;
            cc.segment(gc, w, tail_u0);
            tile_start = true;
        }
        if (u + 1 < n_units) {
            __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));      // own DMAs of the next step (and a finished block's stores)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (u + 2 < n_units) issue(BUF);
        }
    };

    issue_tile();
    issue(0);
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (n_units > 1) issue(1);
    for (int u = 0; u < n_units; u += 2) {
        step_body(std::integral_constant<int, 0>{}, u);
        if (u + 1 < n_units) step_body(std::integral_constant<int, 1>{}, u + 1);
    }
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    if constexpr (BOUND) {
        const float t = wave_allsum(ll);
        if (lane == 0) stream_cold_args<LdaStreamArgs>()->ll_slab[blockIdx.x * 4 + wave] = t;
    }
}

// (three plain kernels around the one body: hipcc left the host stub of a `template <int KT> __global__`
// version of this kernel undefined)
#define LDA_STREAM_KERNEL(NAME, KT_, BOUND_)                                             \
    __global__ __launch_bounds__(LDA_BLOCK, 2) void NAME(LdaStreamArgs a) {              \
        __shared__ __attribute__((aligned(1024))) char lds[2 * LS_STAGE];                \
        lda_sstats_stream_body<KT_, BOUND_>(a, lds);                                     \
    }
LDA_STREAM_KERNEL(lda_sstats_stream_kernel, 4, false)          // K = 128
LDA_STREAM_KERNEL(lda_sstats_stream_k64_kernel, 2, false)
LDA_STREAM_KERNEL(lda_sstats_stream_k32_kernel, 1, false)
LDA_STREAM_KERNEL(lda_sstats_stream_bound_kernel, 4, true)    // ... + sum_dv C log2(phinorm)
LDA_STREAM_KERNEL(lda_sstats_stream_k64_bound_kernel, 2, true)
LDA_STREAM_KERNEL(lda_sstats_stream_k32_bound_kernel, 1, true)
#undef LDA_STREAM_KERNEL

// The column blocks that two or more runs share (see stream_fixup_kernel in csrc/bsc_gemm.hip):
// sstats = Bt * (the pieces in document order).
__global__ __launch_bounds__(64) void lda_stream_fixup_kernel(LdaStreamArgs g) {
    const int w = (int)blockIdx.x + 1;
    const int b0 = stream_first_unit(&g, w);
    const int t = b0 / g.n_kt, t_begin = t * g.n_kt, t_end = t_begin + g.n_kt;        // tail block t
    if (b0 == t_begin || stream_first_unit(&g, w - 1) > t_begin) return;
    const int e = (blockIdx.y * 64 + threadIdx.x) * 4;           // element of the [128 k][128 v] block
    lda_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int x = w - 1; x < g.n_wg; ++x) {
        const int x0 = stream_first_unit(&g, x);
        if (x0 >= t_end) break;
        if (stream_first_unit(&g, x + 1) == x0) continue;
        v += *reinterpret_cast<const lda_f32x4*>(g.slab + ((int64_t)2 * x + (x0 / g.n_kt != t ? 1 : 0)) * ((int64_t)g.K * VT) + e);
    }
    const int k = e >> 7;
    const int64_t col = (int64_t)(g.rounds * g.n_wg + t) * VT + (e & 127);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (col + q < g.V) g.out[k * g.ldo + col + q] = v[q] * g.Bt[k * g.ldb + col + q];
}

// ---- K = 128 on the operand-split bf16 route (ctx->mfma_split = 2 or 3; csrc/bsc_bf16split.h) ----
//
// The schedule, the C tile, the slab and the fix-up of lda_sstats_stream_kernel; both contractions on
// v_mfma_f32_32x32x16_bf16 with Th and Bt written as sums of SPLIT bf16 terms by two parameter-sized
// kernels in front (lda_split_th_kernel: [term][document][128]; lda_split_bt_kernel: transposed,
// [term][word][128], so that a lane's eight contraction values are one 16-byte load) and the ratio split
// in registers, where it is born:
//   phase 1  P = Th Bt       A = a 16-byte row read of the Th image (lane = document), B = the Bt terms
//                            held in registers for the whole column block (32 SPLIT VGPRs);
//   ratio    as before, then registers 8 s .. 8 s + 7 converted pairwise are the B fragment of document
//            k-step s of phase 2 -- in the permuted document order of a result tile;
//   phase 2  S[32 kb ..] += Th^T R:  A = ds_read_b64_tr_b16 of THE SAME Th image (lane = topic) following
//            that document order: no second, transposed copy of Th anywhere.
// Per 32-document step and wave 16 SPLIT (SPLIT + 1) / 2 ... = 48 (SPLIT 2) or 96 (SPLIT 3) MFMAs of 32 cycles against
// 128 of 64 cycles on the f32 route.
struct LdaBxArgs {
    LdaStreamArgs s;
    const unsigned short* ThS;   // [SPLIT][docs_pad][128] bf16
    const unsigned short* BtS;   // [SPLIT][V_pad][128] bf16, V_pad = 128 n column blocks
    unsigned th_term_bytes;      // docs_pad * 256
    int64_t bt_term_elems;       // V_pad * 128
    int dbg;                     // profiling only (ctx->lda_dbg)
};
typedef stream_args_cptr<LdaBxArgs> lda_bx_cptr;

__device__ __forceinline__ void lda_wait_lgkm16(int n) {
    switch (n) {
        case 9: __builtin_amdgcn_s_waitcnt(0xC97F); break;
        case 10: __builtin_amdgcn_s_waitcnt(0xCA7F); break;
        case 11: __builtin_amdgcn_s_waitcnt(0xCB7F); break;
        case 12: __builtin_amdgcn_s_waitcnt(0xCC7F); break;
        case 13: __builtin_amdgcn_s_waitcnt(0xCD7F); break;
        case 14: __builtin_amdgcn_s_waitcnt(0xCE7F); break;
        case 15: __builtin_amdgcn_s_waitcnt(0xCF7F); break;
        default: lda_wait_lgkm(n); break;
    }
}

// LDS: two Th images (from the L2: one step ahead is enough) and a ring of NC count tiles (from HBM: NC - 1 steps
// ahead -- with one step ahead the kernel took the SUM of its DMA-only and arithmetic-only times, 0.83 ms of
// 0.41 + 0.56 at config 4's size, tools/bench_lda_split.py under BSC_LDA_DBG).  SPLIT 2: 2 x 16 + 3 x 16 KiB = 80 KiB,
// two workgroups per CU; SPLIT 3: 2 x 24 + 4 x 16 = 112 KiB, one.
template <int SPLIT>
struct LdaBxGeo {
    static constexpr int THB = SPLIT * 8192, NC = SPLIT == 2 ? 3 : 4, CB0 = 2 * THB, LDS_BYTES = CB0 + NC * 16384;
    static constexpr int PERIOD = NC == 3 ? 6 : 4;      // lcm(2, NC): the main loop's unroll
};

template <int SPLIT, bool BOUND>
__device__ __forceinline__ void lda_sstats_bx_body(const LdaBxArgs& a, char* lds) {
    constexpr int K = 128, THB = LdaBxGeo<SPLIT>::THB, NC = LdaBxGeo<SPLIT>::NC, CB0 = LdaBxGeo<SPLIT>::CB0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31, g16 = lane >> 4;
    int w = blockIdx.x;
    int n_units, tail_u0, tail_cnt;
    {
        const lda_bx_cptr gc = stream_cold_args<LdaBxArgs>();
        if ((gc->s.n_wg & 7) == 0) w = (w & 7) * (gc->s.n_wg >> 3) + (w >> 3);
        tail_u0 = stream_first_unit(&gc->s, w);
        tail_cnt = stream_first_unit(&gc->s, w + 1) - tail_u0;
        n_units = gc->s.rounds * gc->s.n_kt + tail_cnt;
    }
    if (n_units == 0) {
        if (BOUND && lane == 0) a.s.ll_slab[blockIdx.x * 4 + wave] = 0.f;
        return;
    }
    float ll = 0.f;
    const int64_t step_c = 32 * a.s.ldc;
    const int last_kt = a.s.n_kt - 1, d_tail = a.s.d_tail;
    const int dbg = a.dbg;

    // ---- issuing side: the wave's share of a step is 2 SPLIT 1-KiB pieces of the Th image and four of the C tile.  Piece
    // (term c, rows 4 i ..) has its chunks permuted by ((row & 3) << 2) | (i & 3) on the global side; wave w takes the
    // pieces with i % 4 == w (i = w, w + 4 of every term), so ONE lane offset serves them all and the rest is a scalar
    // offset.  The same for the C tile: rows 2 (4 w + jj) + lane / 32.
    StreamCursor it, ic;          // next step whose Th image / C tile is to be requested
    it.begin(&stream_cold_args<LdaBxArgs>()->s, w, tail_u0, tail_cnt);
    ic.begin(&stream_cold_args<LdaBxArgs>()->s, w, tail_u0, tail_cnt);
    int issued_t = 0, issued_c = 0;
    const unsigned vth = (unsigned)((lane >> 4) * 256 + 16 * ((lane & 15) ^ ((((lane >> 4) & 3) << 2) | wave)));
    unsigned vc;
    const unsigned short* th_ptr = stream_cold_args<LdaBxArgs>()->ThS + (int64_t)it.kt * (32 * 128);
    const float* c_ptr;
    auto c_tile = [&]() __attribute__((always_inline)) {
        const lda_bx_cptr gc = stream_cold_args<LdaBxArgs>();
        const int64_t v_base = (int64_t)ic.t * VT, v_left = gc->s.V - v_base;
        const int ldc = (int)gc->s.ldc;
        const int row = 8 * wave + (lane >> 5), col = 4 * (lane & 31);
        vc = col < v_left ? (unsigned)(row * ldc + col) * 4u : LS_OUTSIDE;
        c_ptr = gc->s.C + (int64_t)ic.kt * 32 * gc->s.ldc + v_base;
    };
    auto issue_th = [&](int buf) __attribute__((always_inline)) {
        const auto rth = __builtin_amdgcn_make_buffer_rsrc((void*)stream_uniform_ptr((const float*)th_ptr), 0, LS_OUTSIDE, 0x00020000);
        char* const dth = lds + buf * THB + wave * 1024;
        const unsigned term_bytes = stream_cold_args<LdaBxArgs>()->th_term_bytes;
        if (__builtin_expect(!((dbg & 1) && issued_t >= 2), 1)) {
#pragma unroll
            for (int c = 0; c < SPLIT; ++c)
#pragma unroll
                for (int half = 0; half < 2; ++half)      // rows 4 (w + 4 half) .. of term c (documents past the end: zero rows of ThS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rth, (bsc_lds_ptr)(dth + c * 8192 + half * 4096), 16, vth,
                                                             c * term_bytes + (4 * wave + 16 * half) * 256, 0, 0);
        }
        ++issued_t;
        th_ptr += 32 * 128;
        if (__builtin_expect(it.step(), 0) && issued_t < n_units) {
            ++it.round;
            it.segment(&stream_cold_args<LdaBxArgs>()->s, w, tail_u0);
            th_ptr = stream_cold_args<LdaBxArgs>()->ThS + (int64_t)it.kt * (32 * 128);
        }
    };
    auto issue_c = [&](int buf) __attribute__((always_inline)) {
        const auto rc = __builtin_amdgcn_make_buffer_rsrc((void*)stream_uniform_ptr(c_ptr), 0, LS_OUTSIDE, 0x00020000);
        char* const dst = lds + CB0 + buf * 16384 + wave * 4096;
        const int row_bytes = 8 * (int)stream_cold_args<LdaBxArgs>()->s.ldc;       // two rows of C
        if (__builtin_expect((dbg & 1) && issued_c >= NC, 0)) {
        } else if (__builtin_expect(ic.kt != last_kt || d_tail == DT, 1)) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (bsc_lds_ptr)(dst + jj * 1024), 16, vc, jj * row_bytes, 0, 2);
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bool in = 2 * (4 * wave + jj) + (lane >> 5) < d_tail;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (bsc_lds_ptr)(dst + jj * 1024), 16, in ? vc : LS_OUTSIDE, jj * row_bytes, 0, 2);
            }
        }
        ++issued_c;
        c_ptr += step_c;
        if (__builtin_expect(ic.step(), 0) && issued_c < n_units) {
            ++ic.round;
            ic.segment(&stream_cold_args<LdaBxArgs>()->s, w, tail_u0);
            c_tile();
        }
    };

    // ---- fragment addresses in Th image 0 / C tile 0 (buffer, term and k-step as immediates)
    const unsigned lb = (unsigned)(uintptr_t)(bsc_lds_ptr)lds;
    unsigned p1[8], p2[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) p1[ks] = lb + bsc_img256_off(j, 2 * ks + h);            // document j, topics 16 ks + 8 h ..
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 2; ++e)           // topics 32 kb + (lane & 31), documents 8 e + 4 h .. + 3 (+ 16 s: + 4096)
            p2[2 * kb + e] = lb + bsc_img256_tr_addr(lane, 8 * e + 4 * (g16 >> 1), 32 * kb + 16 * (g16 & 1));
    const unsigned pc = lb + CB0 + (unsigned)(4 * h * 512 + (32 * wave + j) * 4);

    bsc_u32x4 bt[8][SPLIT];       // Bt[16 ks + 8 h ..][v], term c
    bsc_f32x16 S[4];
    bsc_f32x16 P;
    unsigned ring[3][SPLIT][4];
    bsc_u32x4 xf[2][SPLIT];

    StreamCursor cc;
    cc.begin(&stream_cold_args<LdaBxArgs>()->s, w, tail_u0, tail_cnt);
    bool tile_start = true, whole = false, v_ok = false;
    unsigned v_lane = 0;

    auto step_body = [&](auto tb_c, auto cb_c, int u) __attribute__((always_inline)) {
        constexpr int TB = decltype(tb_c)::value, CB = decltype(cb_c)::value;
        constexpr int OFF = TB * THB, OFF_C = CB * 16384;
        if (__builtin_expect(tile_start, 0)) {
            tile_start = false;
            const lda_bx_cptr gc = stream_cold_args<LdaBxArgs>();
            const int64_t v = (int64_t)cc.t * VT + 32 * wave + j;
            v_ok = v < gc->s.V;
            v_lane = (unsigned)(v_ok ? v : gc->s.V - 1) * 4u;
            whole = cc.kt == 0 && cc.left == gc->s.n_kt;
            // (v < V_pad: zero columns past V.  Loads the compiler does not see: it would otherwise make every
            // step wait for its vmcnt -- i.e. for the NEXT steps' DMAs -- before the first use of bt)
            const unsigned off = (unsigned)v * 256u + 16u * (unsigned)h;
            const unsigned short* base = gc->BtS;
            const int64_t term = gc->bt_term_elems;
#pragma unroll
            for (int c = 0; c < SPLIT; ++c) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(bt[ks][c]) : "v"(off), "s"(base), "n"(32 * ks) : "memory");
                base += term;
            }
            __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));      // (also the DMAs in flight: once per block)
#pragma unroll
            for (int c = 0; c < SPLIT; ++c)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(bt[ks][c]));
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[kb][r] = 0.f;
        }
        // LDS reads in issue order: G0 G1 | the sixteen counts (8 x ds_read2st64_b32: rows i, i + 1 of result registers
        // 4 q + i) | G2 .. G7 | T0 .. T7, where G ks = the SPLIT row reads of phase-1 k-step ks and T (4 ds + kb) = the
        // 2 SPLIT transposed reads of document k-step ds, topic block kb; each group is requested two groups before it is
        // used, into a ring of three register sets.  The ratio and its split (131 vector instructions a step -- with the
        // MFMAs of the wave's own phases around them they cost it nothing; in a block of their own they cost 500 cycles of
        // 3 500): registers 0..7 between the phases, 8..15 behind the MFMAs of phase 2's first four groups.
        constexpr int NG = 16;                       // G0..G7, T0..T7
        auto group_size = [](int gi) { return gi < 8 ? SPLIT : 2 * SPLIT; };
        float cv[16];
        auto issue_group = [&](int gi) __attribute__((always_inline)) {
            unsigned (&dst)[SPLIT][4] = ring[gi % 3];
            if (gi < 8) {
#pragma unroll
                for (int c = 0; c < SPLIT; ++c) {
                    bsc_u32x4 v4;
                    BSC_LDS_B128(v4, p1[gi], OFF + c * 8192);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[c][e] = v4[e];
                }
            } else {
                const int ds = (gi - 8) >> 2, kb = (gi - 8) & 3;
#pragma unroll
                for (int c = 0; c < SPLIT; ++c)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        bsc_u32x2 v2;
                        BSC_LDS_TR_B64(v2, p2[2 * kb + e], OFF + c * 8192 + ds * 4096);
                        dst[c][2 * e] = v2[0];
                        dst[c][2 * e + 1] = v2[1];
                    }
            }
        };
        auto ratio = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float cnt = cv[4 * q + i];
                // (padded documents / columns carry zero counts and P = 0: the clamp keeps 0 * rcp(0) from becoming NaN)
                const int pi = __float_as_int(P[4 * q + i]), lo = __float_as_int(1.0e-30f);
                const float pcl = __int_as_float(pi > lo ? pi : lo);
                if constexpr (BOUND) ll = __builtin_fmaf(cnt, __builtin_amdgcn_logf(pcl), ll);
                P[4 * q + i] = cnt * __builtin_amdgcn_rcpf(pcl);
            }
        };
        auto split_pairs = [&](int ds, int t0) __attribute__((always_inline)) {
#pragma unroll
            for (int t = t0; t < t0 + 2; ++t) {
                unsigned pk[SPLIT];
                bsc_split_pk<SPLIT>(P[8 * ds + 2 * t], P[8 * ds + 2 * t + 1], pk);
#pragma unroll
                for (int c = 0; c < SPLIT; ++c) xf[ds][c][t] = pk[c];
            }
        };
        const bsc_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (__builtin_expect(!(dbg & 2), 1)) {
            issue_group(0);
            issue_group(1);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    bsc_f32x2 v2;
                    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v2) : "v"(pc),
                                 "n"(64 * CB + 2 * (i + 8 * q)), "n"(64 * CB + 2 * (i + 8 * q) + 2) : "memory");
                    cv[4 * q + i] = v2[0];
                    cv[4 * q + i + 1] = v2[1];
                }
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                // group gi has landed; group gi + 1 (and, for gi = 0, the counts) may be in flight
                lda_wait_lgkm16((gi + 1 < NG ? group_size(gi + 1) : 0) + (gi == 0 ? 8 : 0));
                __builtin_amdgcn_sched_barrier(0);
                if (gi + 2 < NG) issue_group(gi + 2);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned (&f)[SPLIT][4] = ring[gi % 3];
                bsc_u32x4 fa[SPLIT];
#pragma unroll
                for (int c = 0; c < SPLIT; ++c) fa[c] = bsc_u32x4{f[c][0], f[c][1], f[c][2], f[c][3]};
                if (gi < 8) {
                    P = bsc_mfma_split<SPLIT>(fa, bt[gi], gi == 0 ? zero16 : P);
                    if (gi == 7) {
                        __builtin_amdgcn_sched_barrier(0);
                        ratio(0);
                        ratio(1);
                        split_pairs(0, 0);
                        split_pairs(0, 2);
                    }
                } else {
                    const int ds = (gi - 8) >> 2, kb = (gi - 8) & 3;
                    S[kb] = bsc_mfma_split<SPLIT>(fa, xf[ds], S[kb]);
                    if (ds == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (kb == 0) ratio(2);
                        else if (kb == 1) ratio(3);
                        else split_pairs(1, 2 * (kb - 2));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (__builtin_expect(cc.step(), 0)) {
            // ---- this run's share of the block is complete: register r of S[kb] is topic 32 kb + (r & 3) + 8 (r >> 2) + 4 h
            const lda_bx_cptr gc = stream_cold_args<LdaBxArgs>();
            if (whole) {
                // sstats = Bt * S: a scalar row base and one lane offset per matrix (v_lane + 4 h rows), 16 factors in flight
                const int64_t ldb = gc->s.ldb, ldo = gc->s.ldo;
                const unsigned off_b = v_lane + (unsigned)(4 * h) * (unsigned)ldb * 4u;
                const unsigned off_o = v_lane + (unsigned)(4 * h) * (unsigned)ldo * 4u;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    float fac[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float* base = gc->s.Bt + (int64_t)(32 * kb + (r & 3) + 8 * (r >> 2)) * ldb;
                        asm volatile("global_load_dword %0, %1, %2" : "=v"(fac[r]) : "v"(off_b), "s"(base) : "memory");
                    }
                    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
                    if (v_ok) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            asm volatile("" : "+v"(fac[r]));
                            float* base = gc->s.out + (int64_t)(32 * kb + (r & 3) + 8 * (r >> 2)) * ldo;
                            const float val = S[kb][r] * fac[r];
                            asm volatile("global_store_dword %0, %1, %2" : : "v"(off_o), "v"(val), "s"(base) : "memory");
                        }
                    }
                }
            } else {
                float* slot = gc->s.slab + ((int64_t)2 * w + (cc.round > gc->s.rounds ? 1 : 0)) * (K * VT) + 32 * wave + j;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) slot[(32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h) * VT] = S[kb][r];
            }
            ++cc.round;
            cc.segment(&gc->s, w, tail_u0);
            tile_start = true;
        }
        if (u + 1 < n_units) {
            // step u + 1 needs its Th image (requested one step ago, FOLLOWED by the count tile of step u + NC - 1) and
            // its count tile (requested before that): everything but that last request has to have landed
            if (u - 1 + NC < n_units) __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(4));
            else __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (u + 2 < n_units) issue_th(TB);
            if (u + NC < n_units) issue_c(CB);
        }
    };

    c_tile();
    issue_th(0);
    issue_c(0);
    if (n_units > 1) issue_th(1);
#pragma unroll
    for (int b = 1; b < NC; ++b)
        if (b < n_units) issue_c(b);
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int u = 0; u < n_units; u += LdaBxGeo<SPLIT>::PERIOD) {
        step_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, u);
        if (u + 1 < n_units) step_body(std::integral_constant<int, 1>{}, std::integral_constant<int, 1 % NC>{}, u + 1);
        if (u + 2 < n_units) step_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 % NC>{}, u + 2);
        if (u + 3 < n_units) step_body(std::integral_constant<int, 1>{}, std::integral_constant<int, 3 % NC>{}, u + 3);
        if constexpr (LdaBxGeo<SPLIT>::PERIOD == 6) {
            if (u + 4 < n_units) step_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 4 % NC>{}, u + 4);
            if (u + 5 < n_units) step_body(std::integral_constant<int, 1>{}, std::integral_constant<int, 5 % NC>{}, u + 5);
        }
    }
    __builtin_amdgcn_s_waitcnt(bsc_vmcnt_only(0));
    if constexpr (BOUND) {
        const float t = wave_allsum(ll);
        if (lane == 0) stream_cold_args<LdaBxArgs>()->s.ll_slab[blockIdx.x * 4 + wave] = t;
    }
}

#define LDA_BX_KERNEL(NAME, SPLIT_, BOUND_, OCC_)                                               \
    __global__ __launch_bounds__(LDA_BLOCK, OCC_) void NAME(LdaBxArgs a) {                      \
        __shared__ __attribute__((aligned(1024))) char lds[LdaBxGeo<SPLIT_>::LDS_BYTES];        \
        lda_sstats_bx_body<SPLIT_, BOUND_>(a, lds);                                             \
    }
LDA_BX_KERNEL(lda_sstats_bx2_kernel, 2, false, 2)
LDA_BX_KERNEL(lda_sstats_bx2_bound_kernel, 2, true, 2)
LDA_BX_KERNEL(lda_sstats_bx3_kernel, 3, false, 1)
LDA_BX_KERNEL(lda_sstats_bx3_bound_kernel, 3, true, 1)
#undef LDA_BX_KERNEL

// Th [docs][128] f32 -> SPLIT bf16 terms [term][docs_pad][128]; rows past `docs` are zeros.  One thread per pair.
template <int SPLIT>
__global__ __launch_bounds__(256) void lda_split_th_kernel(const float* __restrict__ Th, int64_t ldth, int64_t docs,
                                                           int64_t docs_pad, unsigned* __restrict__ ThS) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // pair index: row i / 64, topics 2 (i % 64) ..
    if (i >= docs_pad * 64) return;
    const int64_t d = i >> 6;
    const int k = 2 * (int)(i & 63);
    float2 v = make_float2(0.f, 0.f);
    if (d < docs) v = *reinterpret_cast<const float2*>(Th + d * ldth + k);
    unsigned pk[SPLIT];
    bsc_split_pk<SPLIT>(v.x, v.y, pk);
#pragma unroll
    for (int c = 0; c < SPLIT; ++c) ThS[c * docs_pad * 64 + i] = pk[c];
}

// Bt [128][V] f32 -> SPLIT bf16 terms, transposed: [term][V_pad][128]; words past V are zeros.  A workgroup
// takes 64 words: the [128][64] tile through LDS, then 16 bytes (eight topics of one word) per store.
template <int SPLIT>
__global__ __launch_bounds__(256) void lda_split_bt_kernel(const float* __restrict__ Bt, int64_t ldb, int64_t V,
                                                           int64_t V_pad, bsc_u32x4* __restrict__ BtS) {
    __shared__ float tile[128][65];
    const int tid = threadIdx.x;
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    for (int e = tid; e < 128 * 64; e += 256) {
        const int k = e >> 6, v = e & 63;
        tile[k][v] = v0 + v < V ? Bt[k * ldb + v0 + v] : 0.f;
    }
    __syncthreads();
    const int v = tid & 63;
    for (int ch = tid >> 6; ch < 16; ch += 4) {          // topics 8 ch .. 8 ch + 7 of word v0 + v
        bsc_u32x4 o[SPLIT];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            unsigned pk[SPLIT];
            bsc_split_pk<SPLIT>(tile[8 * ch + 2 * t][v], tile[8 * ch + 2 * t + 1][v], pk);
#pragma unroll
            for (int c = 0; c < SPLIT; ++c) o[c][t] = pk[c];
        }
#pragma unroll
        for (int c = 0; c < SPLIT; ++c) BtS[(c * V_pad + v0 + v) * 16 + ch] = o[c];
    }
}

// ---- sparse counts (compressed sparse COLUMN): one pass over the nonzeros ----------
//
// Real bag-of-words data is ~1 % dense; the dense kernel above spends its MFMAs on
// zeros.  Here a workgroup owns 64 consecutive words (16 per wave); a word's
// nonzeros (document id, count) are walked a dozen at a time, three or four per 16-lane group,
// each lane holding K/16 consecutive topics of that document's Th row:
//   p = <Th[d,:], Bt[:,v]>  (in-lane partial + 4 DPP adds inside the 16-lane row)
//   acc[k] += Th[d,k] * c / p
// No atomics: a word belongs to one wave, groups combine in a fixed order at the end.
// Bound: L2 -> CU traffic of the Th rows (4K bytes per nonzero; Th itself is
// docs x K and stays L2-resident) and the VALU instructions per nonzero (13.8 on the general
// path, about half that through buffer descriptors -- see the FAST branch).
constexpr int CSC_WORDS = 64;          // words per workgroup
constexpr int CSC_LD = CSC_WORDS + 1;  // Bt / result tile row stride in LDS

struct LdaCscArgs {
    const int64_t* colptr;
    const int32_t* rowidx;
    const float* vals;
    const float* Th;
    const float* Bt;
    float* out;
    int64_t ldth, ldb, ldo, docs, V;
    int vec_th;
    int fast;   // docs * ldth * 4 < 2^31 and 16-byte Th rows: gathers go through buffer descriptors
    float* ll_slab;   // BOUND: [grid][4] per-wave partials of 16 * sum_nz C log2(phinorm) (every lane of a group adds it)
};

template <int KPL, bool FAST, bool BOUND>   // topics per lane; K = 16 * KPL
__global__ __launch_bounds__(LDA_BLOCK) void lda_sstats_csc_kernel(LdaCscArgs a) {
    constexpr int K = 16 * KPL;
    __shared__ float tile[K * CSC_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, gl = lane & 15;
    const int64_t v_base = (int64_t)blockIdx.x * CSC_WORDS;

    for (int i = tid; i < K * CSC_WORDS; i += LDA_BLOCK) {   // Bt[:, v_base .. +63], coalesced rows
        const int k = i / CSC_WORDS, c = i % CSC_WORDS;
        tile[k * CSC_LD + c] = v_base + c < a.V ? a.Bt[(int64_t)k * a.ldb + v_base + c] : 0.f;
    }
    __syncthreads();

    float ll = 0.f;
    for (int w = 0; w < CSC_WORDS / 4; ++w) {
        const int c = 16 * wave + w;
        const int64_t v = v_base + c;
        if (v >= a.V) break;                                  // uniform per wave
        float bt[KPL], acc[KPL];
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            bt[i] = tile[(gl * KPL + i) * CSC_LD + c];
            acc[i] = 0.f;
        }
        const int64_t begin = a.colptr[v], end = a.colptr[v + 1];
        constexpr int U = KPL <= 4 ? 4 : 3;   // nonzeros per 16-lane group in flight (register budget)
        if constexpr (FAST) {
            // The kernel is VALU-bound (13.8 instructions per nonzero before this path: 64-bit
            // index arithmetic, clamps, selects and a true division).  Here the word's
            // (document id, count) run and the Th rows are read through buffer descriptors:
            // 32-bit offsets, and anything out of range -- the tail of the run, a bad document
            // id -- reads as 0, which contributes nothing; the division is v_rcp_f32 as in the
            // dense kernel (tolerance 3e-5).
            const int64_t nnz64 = end - begin;
            const unsigned run_bytes = nnz64 <= 0 ? 0u : (nnz64 > 0x3FFFFFFF ? 0xFFFFFFFCu : (unsigned)nnz64 * 4u);
            const auto ri = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rowidx + begin), 0, run_bytes, 0x00020000);
            const auto va = __builtin_amdgcn_make_buffer_rsrc((void*)(a.vals + begin), 0, run_bytes, 0x00020000);
            const auto thr = __builtin_amdgcn_make_buffer_rsrc((void*)a.Th, 0, (unsigned)(a.docs * a.ldth * 4), 0x00020000);
            const int ldth4 = (int)a.ldth * 4, lane_off = gl * KPL * 4;
            const int nnz = (int)(run_bytes >> 2);
            // (requesting the next step's ids ahead of this step's rows, or four instead of
            // three nonzeros per group in flight at K = 128, changed nothing: 1.02 ms either way)
            for (int j0 = 0; j0 < nnz; j0 += 4 * U) {
                float th[U][KPL], cnt[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int voff = (j0 + 4 * u + g) * 4;
                    const int d = (int)__builtin_amdgcn_raw_buffer_load_b32(ri, voff, 0, 0);
                    cnt[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(va, voff, 0, 0));
                    const int toff = d * ldth4 + lane_off;
                    if constexpr (KPL % 4 == 0) {
#pragma unroll
                        for (int q = 0; q < KPL / 4; ++q) {
                            auto t = __builtin_amdgcn_raw_buffer_load_b128(thr, toff + 16 * q, 0, 0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) th[u][4 * q + e] = __uint_as_float(t[e]);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < KPL; ++i)
                            th[u][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(thr, toff + 4 * i, 0, 0));
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float p = 0.f;
#pragma unroll
                    for (int i = 0; i < KPL; ++i) p += th[u][i] * bt[i];
                    p = row16_allsum(p);
                    const float pc = fmaxf(p, 1.0e-30f);
                    if constexpr (BOUND) ll = __builtin_fmaf(cnt[u], __builtin_amdgcn_logf(pc), ll);
                    const float r = cnt[u] * __builtin_amdgcn_rcpf(pc);
#pragma unroll
                    for (int i = 0; i < KPL; ++i) acc[i] += th[u][i] * r;
                }
            }
        } else {
        for (int64_t j0 = begin; j0 < end; j0 += 4 * U) {
            float th[U][KPL], cnt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = j0 + 4 * u + g;
                const bool ok = j < end;
                const int64_t jj = ok ? j : begin;            // a valid address either way
                int64_t d = a.rowidx[jj];
                d = d < 0 ? 0 : (d >= a.docs ? a.docs - 1 : d);
                cnt[u] = ok ? a.vals[jj] : 0.f;
                const float* row = a.Th + d * a.ldth + gl * KPL;
                if (KPL % 4 == 0 && a.vec_th) {
#pragma unroll
                    for (int q = 0; q < KPL / 4; ++q) {
                        const float4 t = reinterpret_cast<const float4*>(row)[q];
                        th[u][4 * q + 0] = t.x; th[u][4 * q + 1] = t.y;
                        th[u][4 * q + 2] = t.z; th[u][4 * q + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < KPL; ++i) th[u][i] = row[i];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float p = 0.f;
#pragma unroll
                for (int i = 0; i < KPL; ++i) p += th[u][i] * bt[i];
                p = row16_allsum(p);
                const float r = cnt[u] != 0.f ? cnt[u] / p : 0.f;
                if constexpr (BOUND) ll += cnt[u] != 0.f ? cnt[u] * __builtin_amdgcn_logf(p) : 0.f;
#pragma unroll
                for (int i = 0; i < KPL; ++i) acc[i] += th[u][i] * r;
            }
        }
        }
        // the four groups saw disjoint nonzeros of this word: fold them (fixed order)
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            acc[i] += __shfl_xor(acc[i], 16);
            acc[i] += __shfl_xor(acc[i], 32);
        }
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < KPL; ++i) tile[(gl * KPL + i) * CSC_LD + c] = acc[i] * bt[i];
        }
    }
    if constexpr (BOUND) {
        const float t = wave_allsum(ll);
        if (lane == 0) a.ll_slab[(int64_t)blockIdx.x * 4 + wave] = t;
    }
    __syncthreads();
    for (int i = tid; i < K * CSC_WORDS; i += LDA_BLOCK) {
        const int k = i / CSC_WORDS, c = i % CSC_WORDS;
        if (v_base + c < a.V) a.out[(int64_t)k * a.ldo + v_base + c] = tile[k * CSC_LD + c];
    }
}

template <int KT>
void launch_lda(bsc_ctx* ctx, const LdaArgs& a, dim3 grid) {
    if (a.ll_slab) hipLaunchKernelGGL((lda_sstats_kernel<KT, true>), grid, dim3(LDA_BLOCK), 0, ctx->stream, a);
    else hipLaunchKernelGGL((lda_sstats_kernel<KT, false>), grid, dim3(LDA_BLOCK), 0, ctx->stream, a);
}

constexpr double LN2 = 0.69314718055994530941723212145818;

}  // namespace

namespace {

int lda_sstats_impl(bsc_ctx* ctx, const float* C, int64_t ldc, int64_t docs, int64_t V, int32_t K,
                    const float* Th, int64_t ldth, const float* Bt, int64_t ldb, float* sstats,
                    int64_t ldo, double* ll) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(docs >= 0 && V >= 0 && K > 0, "bsc_lda_sstats: bad extents");
    if (K % 32 != 0 || K > 128)
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_lda_sstats: K must be 32, 64, 96 or 128 (got %d); use the algebra "
                        "executor for other topic counts", K);
    if (V == 0) {
        if (ll) BSC_HIP(hipMemsetAsync(ll, 0, sizeof(double), ctx->stream));
        return BSC_OK;
    }
    BSC_REQUIRE(Bt && sstats && (docs == 0 || (C && Th)), "bsc_lda_sstats: null pointer");
    BSC_REQUIRE(ldc >= V && ldth >= K && ldb >= V && ldo >= V, "bsc_lda_sstats: leading dimension");
    LdaArgs a{};
    a.C = C; a.Th = Th; a.Bt = Bt;
    a.ldc = ldc; a.ldth = ldth; a.ldb = ldb; a.ldo = ldo;
    a.docs = docs; a.V = V;
    a.vec_c = (ldc % 4 == 0) && (((uintptr_t)C) & 15) == 0;
    a.vec_th = (ldth % 4 == 0) && (((uintptr_t)Th) & 15) == 0;
    a.fast = a.vec_c && a.vec_th && (32 * ldc + VT) * 4 < ((int64_t)1 << 31) &&
             (32 * ldth + K) * 4 < ((int64_t)1 << 31);
    const int64_t n_vt = (V + VT - 1) / VT;
    if ((K == 128 || K == 64 || K == 32) && ctx->lda_stream && docs > 0 && a.vec_c && a.vec_th && V % 4 == 0 &&
        (31 * ldc + VT) * 4 < ((int64_t)1 << 31) && (31 * ldth + K) * 4 < ((int64_t)1 << 31) &&
        (K * ldb + V) * 4 < ((int64_t)1 << 32) && V * 4 < ((int64_t)1 << 32) && n_vt < ((int64_t)1 << 24) &&
        (docs + DT - 1) / DT < ((int64_t)1 << 22)) {
        LdaStreamArgs g{};
        g.C = C; g.Th = Th; g.Bt = Bt; g.out = sstats;
        g.ldc = ldc; g.ldth = ldth; g.ldb = ldb; g.ldo = ldo; g.docs = docs; g.V = V; g.K = K;
        const int n_kt = (int)((docs + DT - 1) / DT);
        g.d_tail = (int)(docs - (int64_t)(n_kt - 1) * DT);
        const int split = K == 128 && ldth % 2 == 0 && (K * ldo + V) * 4 < ((int64_t)1 << 32) ? ctx->mfma_split : 0;     // operand-split bf16 route (K = 128)
        // (three terms: one workgroup per CU -- 288 registers per lane and 80 KiB of LDS)
        stream_plan(g, n_vt, n_kt, (split == 3 ? 1 : 2) * (int64_t)ctx->cu_count);
        void* ws = nullptr;
        const size_t slab_floats = (size_t)2 * g.n_wg * K * VT;
        const int64_t docs_pad = (int64_t)n_kt * DT, V_pad = n_vt * VT;
        const size_t th_bytes = (size_t)split * docs_pad * 256, bt_bytes = (size_t)split * V_pad * 256;
        int rc = bsc_workspace(ctx, (slab_floats + (size_t)4 * g.n_wg) * sizeof(float) + th_bytes + bt_bytes, &ws);
        if (rc != BSC_OK) return rc;
        ctx->slab_rows = 0;
        g.slab = (float*)ws;
        g.ll_slab = ll ? (float*)ws + slab_floats : nullptr;
        if (split) {
            LdaBxArgs x{};
            x.s = g;
            char* const terms = (char*)((float*)ws + slab_floats + (size_t)4 * g.n_wg);
            x.ThS = (const unsigned short*)terms;
            x.BtS = (const unsigned short*)(terms + th_bytes);
            x.th_term_bytes = (unsigned)(docs_pad * 256);
            x.bt_term_elems = V_pad * 128;
            x.dbg = ctx->lda_dbg;
            const dim3 grid((unsigned)g.n_wg), block(LDA_BLOCK);
            const unsigned th_blocks = (unsigned)((docs_pad * 64 + 255) / 256), bt_blocks = (unsigned)(V_pad / 64);
            if (split == 2) {
                hipLaunchKernelGGL(lda_split_th_kernel<2>, dim3(th_blocks), block, 0, ctx->stream, Th, ldth, docs, docs_pad, (unsigned*)terms);
                hipLaunchKernelGGL(lda_split_bt_kernel<2>, dim3(bt_blocks), block, 0, ctx->stream, Bt, ldb, V, V_pad, (bsc_u32x4*)(terms + th_bytes));
            } else {
                hipLaunchKernelGGL(lda_split_th_kernel<3>, dim3(th_blocks), block, 0, ctx->stream, Th, ldth, docs, docs_pad, (unsigned*)terms);
                hipLaunchKernelGGL(lda_split_bt_kernel<3>, dim3(bt_blocks), block, 0, ctx->stream, Bt, ldb, V, V_pad, (bsc_u32x4*)(terms + th_bytes));
            }
            BSC_LAUNCH_CHECK();
            {
                bsc_prof_scope prof(ctx);
                if (split == 2 && ll) hipLaunchKernelGGL(lda_sstats_bx2_bound_kernel, grid, block, 0, ctx->stream, x);
                else if (split == 2) hipLaunchKernelGGL(lda_sstats_bx2_kernel, grid, block, 0, ctx->stream, x);
                else if (ll) hipLaunchKernelGGL(lda_sstats_bx3_bound_kernel, grid, block, 0, ctx->stream, x);
                else hipLaunchKernelGGL(lda_sstats_bx3_kernel, grid, block, 0, ctx->stream, x);
            }
        } else {
            bsc_prof_scope prof(ctx);
            const dim3 grid((unsigned)g.n_wg), block(LDA_BLOCK);
            if (ll) {
                if (K == 128) hipLaunchKernelGGL(lda_sstats_stream_bound_kernel, grid, block, 0, ctx->stream, g);
                else if (K == 64) hipLaunchKernelGGL(lda_sstats_stream_k64_bound_kernel, grid, block, 0, ctx->stream, g);
                else hipLaunchKernelGGL(lda_sstats_stream_k32_bound_kernel, grid, block, 0, ctx->stream, g);
            } else {
                if (K == 128) hipLaunchKernelGGL(lda_sstats_stream_kernel, grid, block, 0, ctx->stream, g);
                else if (K == 64) hipLaunchKernelGGL(lda_sstats_stream_k64_kernel, grid, block, 0, ctx->stream, g);
                else hipLaunchKernelGGL(lda_sstats_stream_k32_kernel, grid, block, 0, ctx->stream, g);
            }
        }
        BSC_LAUNCH_CHECK();
        if (ll) {
            hipLaunchKernelGGL(lda_ll_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float*)g.ll_slab,
                               (int64_t)4 * g.n_wg, LN2, ll);
            BSC_LAUNCH_CHECK();
        }
        if (stream_has_pieces(g)) {
            hipLaunchKernelGGL(lda_stream_fixup_kernel, dim3((unsigned)(g.n_wg - 1), K * VT / 256), dim3(64), 0,
                               ctx->stream, g);
            BSC_LAUNCH_CHECK();
        }
        return BSC_OK;
    }
    // Split the documents so that the grid fills whole rounds of the 2 x CU resident
    // workgroups (782 column tiles on 512 slots would leave a quarter of the chip idle).
    const int64_t slots = 2 * (int64_t)ctx->cu_count;
    const int64_t steps = (docs + DT - 1) / DT;
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 8; ++s) {
        if (s > 1 && steps / s < 8) break;
        const int64_t wgs = n_vt * s;
        const double eff = (double)wgs / (double)(slots * ((wgs + slots - 1) / slots));
        if (eff > best_eff + 0.02) {
            best_eff = eff;
            best = s;
        }
    }
    a.splits = best;
    a.docs_per_split = ((steps + best - 1) / best) * DT;
    if (a.docs_per_split == 0) a.docs_per_split = DT;
    float* partial = nullptr;
    const size_t partial_floats = best > 1 ? (size_t)best * K * V : 0, ll_floats = ll ? (size_t)4 * n_vt * best : 0;
    if (partial_floats + ll_floats) {
        void* ws = nullptr;
        int rc = bsc_workspace(ctx, (partial_floats + ll_floats) * sizeof(float), &ws);
        if (rc != BSC_OK) return rc;
        ctx->slab_rows = 0;
        if (ll) a.ll_slab = (float*)ws + partial_floats;
        if (best > 1) partial = (float*)ws;
    }
    a.out = best > 1 ? partial : sstats;
    const dim3 grid((unsigned)n_vt, (unsigned)best);
    {
        bsc_prof_scope prof(ctx);  // times the fused kernel alone
        switch (K / 32) {
            case 1: launch_lda<1>(ctx, a, grid); break;
            case 2: launch_lda<2>(ctx, a, grid); break;
            case 3: launch_lda<3>(ctx, a, grid); break;
            default: launch_lda<4>(ctx, a, grid); break;
        }
    }
    BSC_LAUNCH_CHECK();
    if (best > 1) {
        int64_t blocks = ((int64_t)K * V + 255) / 256;
        const int64_t cap = (int64_t)ctx->cu_count * 8;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(lda_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           partial, best, (int64_t)K, V, Bt, ldb, sstats, ldo);
        BSC_LAUNCH_CHECK();
    }
    if (ll) {
        hipLaunchKernelGGL(lda_ll_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float*)a.ll_slab,
                           (int64_t)ll_floats, LN2, ll);
        BSC_LAUNCH_CHECK();
    }
    return BSC_OK;
}

int lda_sstats_csc_impl(bsc_ctx* ctx, const int64_t* colptr, const int32_t* rowidx, const float* vals,
                        int64_t docs, int64_t V, int32_t K, const float* Th, int64_t ldth,
                        const float* Bt, int64_t ldb, float* sstats, int64_t ldo, double* ll);

}  // namespace

extern "C" {

int bsc_lda_sstats_round_columns(bsc_ctx* ctx, int32_t K, int64_t* host_cols) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(host_cols != nullptr, "bsc_lda_sstats_round_columns: null result");
    // one whole round of the persistent kernel: one 128-column block per resident workgroup (lda_sstats_impl's
    // stream_plan: two workgroups a CU; one with three split terms at K = 128)
    const int split = K == 128 ? ctx->mfma_split : 0;
    *host_cols = (int64_t)(split == 3 ? 1 : 2) * ctx->cu_count * VT;
    return BSC_OK;
}

int bsc_lda_sstats(bsc_ctx* ctx, const float* C, int64_t ldc, int64_t docs, int64_t V, int32_t K,
                   const float* Th, int64_t ldth, const float* Bt, int64_t ldb, float* sstats,
                   int64_t ldo) {
    return lda_sstats_impl(ctx, C, ldc, docs, V, K, Th, ldth, Bt, ldb, sstats, ldo, nullptr);
}

int bsc_lda_sstats_bound(bsc_ctx* ctx, const float* C, int64_t ldc, int64_t docs, int64_t V, int32_t K,
                         const float* Th, int64_t ldth, const float* Bt, int64_t ldb, float* sstats,
                         int64_t ldo, double* ll) {
    if (!ll) return bsc_fail(BSC_ERR_INVALID, "bsc_lda_sstats_bound: ll is NULL");
    return lda_sstats_impl(ctx, C, ldc, docs, V, K, Th, ldth, Bt, ldb, sstats, ldo, ll);
}

int bsc_lda_sstats_csc(bsc_ctx* ctx, const int64_t* colptr, const int32_t* rowidx, const float* vals,
                       int64_t docs, int64_t V, int32_t K, const float* Th, int64_t ldth,
                       const float* Bt, int64_t ldb, float* sstats, int64_t ldo) {
    return lda_sstats_csc_impl(ctx, colptr, rowidx, vals, docs, V, K, Th, ldth, Bt, ldb, sstats, ldo, nullptr);
}

int bsc_lda_sstats_csc_bound(bsc_ctx* ctx, const int64_t* colptr, const int32_t* rowidx, const float* vals,
                             int64_t docs, int64_t V, int32_t K, const float* Th, int64_t ldth,
                             const float* Bt, int64_t ldb, float* sstats, int64_t ldo, double* ll) {
    if (!ll) return bsc_fail(BSC_ERR_INVALID, "bsc_lda_sstats_csc_bound: ll is NULL");
    return lda_sstats_csc_impl(ctx, colptr, rowidx, vals, docs, V, K, Th, ldth, Bt, ldb, sstats, ldo, ll);
}

}  // extern "C"

namespace {

int lda_sstats_csc_impl(bsc_ctx* ctx, const int64_t* colptr, const int32_t* rowidx, const float* vals,
                        int64_t docs, int64_t V, int32_t K, const float* Th, int64_t ldth,
                        const float* Bt, int64_t ldb, float* sstats, int64_t ldo, double* ll) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(docs >= 0 && V >= 0 && K > 0, "bsc_lda_sstats_csc: bad extents");
    if (K % 32 != 0 || K > 128)
        return bsc_fail(BSC_ERR_UNSUPPORTED,
                        "bsc_lda_sstats_csc: K must be 32, 64, 96 or 128 (got %d)", K);
    if (V == 0) {
        if (ll) BSC_HIP(hipMemsetAsync(ll, 0, sizeof(double), ctx->stream));
        return BSC_OK;
    }
    BSC_REQUIRE(colptr && Bt && sstats, "bsc_lda_sstats_csc: null pointer");
    BSC_REQUIRE(ldth >= K && ldb >= V && ldo >= V, "bsc_lda_sstats_csc: leading dimension");
    LdaCscArgs a{};
    a.colptr = colptr; a.rowidx = rowidx; a.vals = vals;
    a.Th = Th; a.Bt = Bt; a.out = sstats;
    a.ldth = ldth; a.ldb = ldb; a.ldo = ldo; a.docs = docs; a.V = V;
    a.vec_th = (ldth % 4 == 0) && (((uintptr_t)Th) & 15) == 0;
    a.fast = a.vec_th && docs > 0 && docs * ldth * 4 < ((int64_t)1 << 31) && ctx->csc_fast;
    const dim3 grid((unsigned)((V + CSC_WORDS - 1) / CSC_WORDS));
    if (ll) {
        void* ws = nullptr;
        int rc = bsc_workspace(ctx, (size_t)4 * grid.x * sizeof(float), &ws);
        if (rc != BSC_OK) return rc;
        ctx->slab_rows = 0;
        a.ll_slab = (float*)ws;
    }
    {
        bsc_prof_scope prof(ctx);
        switch (K / 32) {
#define BSC_CSC1(KPL, F)                                                                               \
    do {                                                                                               \
        if (ll) hipLaunchKernelGGL((lda_sstats_csc_kernel<KPL, F, true>), grid, dim3(LDA_BLOCK), 0, ctx->stream, a);   \
        else hipLaunchKernelGGL((lda_sstats_csc_kernel<KPL, F, false>), grid, dim3(LDA_BLOCK), 0, ctx->stream, a);     \
    } while (0)
#define BSC_CSC(KPL)                                                                                   \
    do {                                                                                               \
        if (a.fast) BSC_CSC1(KPL, true);                                                               \
        else BSC_CSC1(KPL, false);                                                                     \
    } while (0)
            case 1: BSC_CSC(2); break;
            case 2: BSC_CSC(4); break;
            case 3: BSC_CSC(6); break;
            default: BSC_CSC(8); break;
#undef BSC_CSC
#undef BSC_CSC1
        }
    }
    BSC_LAUNCH_CHECK();
    if (ll) {
        // every lane of a 16-lane group added the group's term: 1/16 of the wave sums
        hipLaunchKernelGGL(lda_ll_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float*)a.ll_slab,
                           (int64_t)4 * grid.x, LN2 / 16.0, ll);
        BSC_LAUNCH_CHECK();
    }
    return BSC_OK;
}

}  // namespace
