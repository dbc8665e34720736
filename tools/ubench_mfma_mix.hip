// What else can a wave issue before the fp32 MFMA pipe starts to starve?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_mix.hip -o /tmp/ubench_mfma_mix && /tmp/ubench_mfma_mix
// Every wave runs a chain-free v_mfma_f32_32x32x2_f32 loop (4 independent accumulators) and,
// per MFMA, a fixed amount of other work from the SAME wave: independent VALU ops, a VALU op
// that produces the MFMA's B operand, LDS reads, a transcendental.  Reported: MFMA pipe
// cycles per MFMA per SIMD (64 = pipe saturated) at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { M_CHAIN1, M_CHAIN2, M_ONLY, M_VALU1, M_VALU2, M_VALU4, M_VALU8, M_DEP1, M_LDS1, M_LDS2, M_LDS2_DEP, M_EXP1, M_PK2, M_PK4, M_B128_4, M_B128_2, M_B128_1, M_DEP_AHEAD4, M_KINDS };
static const char* NAMES[M_KINDS] = {
    "MFMA only, ONE dependent chain", "MFMA only, two dependent chains", "MFMA only", "+1 independent v_mul / MFMA", "+2 independent v_fma / MFMA", "+4 independent v_fma / MFMA",
    "+8 independent v_fma / MFMA", "+1 v_mul feeding the MFMA's B operand", "+1 ds_read_b32 / MFMA",
    "+2 ds_read_b32 / MFMA", "+2 ds_read_b32 -> v_mul -> B operand (read one MFMA ahead)", "+1 v_exp_f32 / MFMA",
    "+2 v_pk_fma_f32 (= 4 fma) / MFMA", "+4 v_pk_fma_f32 (= 8 fma) / MFMA",
    "+1 ds_read_b128 / 4 MFMA (4 B/cycle/SIMD)", "+1 ds_read_b128 / 2 MFMA (8 B/cycle/SIMD)", "+1 ds_read_b128 / MFMA (16 B/cycle/SIMD)",
    "+2 ds_read_b32 -> v_mul -> B operand, reads issued FOUR MFMAs ahead"};

template <int KIND>
__global__ __launch_bounds__(256) void mix_loop(int iters, float* sink) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = threadIdx.x * 1e-9f;
    float y = 1.0f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
    const unsigned la = (threadIdx.x & 63) * 4u, lb = ((threadIdx.x & 63) + 64) * 4u + 2048u;
    float l0 = 1.f, l1 = 1.f, n0 = 1.f, n1 = 1.f;
    typedef float f32x4q __attribute__((ext_vector_type(4)));
    f32x4q q0 = {0}, q1 = {0}, q2 = {0}, q3 = {0};
    const unsigned lq = (threadIdx.x & 63) * 16u;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 w0 = {v[0], v[1]}, w1 = {v[2], v[3]}, w2 = {v[4], v[5]}, w3 = {1.0000001f, 1e-7f};
    if (KIND == M_LDS2_DEP) {
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(n0), "=v"(n1) : "v"(la), "v"(lb));
    }
    float r0a = 1.f, r0b = 1.f, r1a = 1.f, r1b = 1.f, r2a = 1.f, r2b = 1.f, r3a = 1.f, r3b = 1.f;
    if (KIND == M_DEP_AHEAD4) {
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(r0a), "=v"(r0b) : "v"(la), "v"(lb));
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(r1a), "=v"(r1b) : "v"(la), "v"(lb));
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(r2a), "=v"(r2b) : "v"(la), "v"(lb));
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(r3a), "=v"(r3b) : "v"(la), "v"(lb));
    }
#define EXTRA(ACC)                                                                                  \
    if (KIND == M_VALU1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[0]) : "v"(v[1]));            \
    if (KIND == M_VALU2 || KIND == M_VALU4 || KIND == M_VALU8) {                                    \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(v[7]));                         \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[1]) : "v"(v[7]));                         \
    }                                                                                               \
    if (KIND == M_VALU4 || KIND == M_VALU8) {                                                       \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[2]) : "v"(v[7]));                         \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[3]) : "v"(v[7]));                         \
    }                                                                                               \
    if (KIND == M_VALU8) {                                                                          \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[4]) : "v"(v[7]));                         \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[5]) : "v"(v[7]));                         \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[6]) : "v"(v[7]));                         \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(v[7]));                         \
    }                                                                                               \
    if (KIND == M_DEP1) asm volatile("v_mul_f32 %0, %1, %2\n\ts_nop 1" : "=v"(y) : "v"(v[0]), "v"(v[1])); \
    if (KIND == M_LDS1) asm volatile("ds_read_b32 %0, %1" : "=v"(l0) : "v"(la));                   \
    if (KIND == M_LDS2) asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(l0), "=v"(l1) : "v"(la), "v"(lb)); \
    if (KIND == M_LDS2_DEP) {                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mul_f32 %0, %1, %2" : "=v"(y) : "v"(n0), "v"(n1));  \
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(n0), "=v"(n1) : "v"(la), "v"(lb)); \
    }                                                                                               \
    if (KIND == M_EXP1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));                              \
    if (KIND == M_PK2 || KIND == M_PK4) {                                                           \
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w0) : "v"(w3));                           \
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w1) : "v"(w3));                           \
    }                                                                                               \
    if (KIND == M_PK4) {                                                                            \
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w2) : "v"(w3));                           \
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w0) : "v"(w3));                           \
    }                                                                                               \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, ACC, 0, 0, 0);
    for (int i = 0; i < iters; ++i) {
        if (KIND == M_CHAIN1) {
            EXTRA(a0) EXTRA(a0) EXTRA(a0) EXTRA(a0)
            continue;
        }
        if (KIND == M_CHAIN2) {
            EXTRA(a0) EXTRA(a1) EXTRA(a0) EXTRA(a1)
            continue;
        }
        if (KIND == M_DEP_AHEAD4) {
            // four operand pairs in flight: pair j was requested four MFMAs ago
            float y0, y1, y2, y3;
            asm volatile("s_waitcnt lgkmcnt(6)\n\tv_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(r0a), "v"(r0b));
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(r0a), "=v"(r0b) : "v"(la), "v"(lb));
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y0, a0, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(6)\n\tv_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(r1a), "v"(r1b));
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(r1a), "=v"(r1b) : "v"(la), "v"(lb));
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y1, a1, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(6)\n\tv_mul_f32 %0, %1, %2" : "=v"(y2) : "v"(r2a), "v"(r2b));
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(r2a), "=v"(r2b) : "v"(la), "v"(lb));
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y2, a2, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(6)\n\tv_mul_f32 %0, %1, %2" : "=v"(y3) : "v"(r3a), "v"(r3b));
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=v"(r3a), "=v"(r3b) : "v"(la), "v"(lb));
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y3, a3, 0, 0, 0);
            continue;
        }
        if (KIND == M_B128_4 || KIND == M_B128_2 || KIND == M_B128_1) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"(lq));
            EXTRA(a0)
            if (KIND == M_B128_1) asm volatile("ds_read_b128 %0, %1" : "=v"(q1) : "v"(lq));
            EXTRA(a1)
            if (KIND == M_B128_2 || KIND == M_B128_1) asm volatile("ds_read_b128 %0, %1" : "=v"(q2) : "v"(lq));
            EXTRA(a2)
            if (KIND == M_B128_1) asm volatile("ds_read_b128 %0, %1" : "=v"(q3) : "v"(lq));
            EXTRA(a3)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
        EXTRA(a0)
        EXTRA(a1)
        EXTRA(a2)
        EXTRA(a3)
        if (KIND == M_LDS1 || KIND == M_LDS2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef EXTRA
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = a0[0] + a1[1] + a2[2] + a3[3] + l0 + l1 + n0 + n1 + y;
    for (int j = 0; j < 8; ++j) s += v[j];
    s += w0[0] + w0[1] + w1[0] + w1[1] + w2[0] + w2[1] + q0[0] + q1[1] + q2[2] + q3[3] + r0a + r0b + r1a + r1b + r2a + r2b + r3a + r3b;
    if (s == 12345.f) sink[0] = 1.f;
}

// v_mfma_f32_16x16x4_f32 (8 passes = 32 cycles nominal): 4 independent accumulators, or 2 / 1 chains
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(256) void small_mfma_loop(int iters, float* sink) {
    f32x4v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = threadIdx.x * 1e-9f, y = 1.0f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        if (CHAINS >= 2) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        if (CHAINS >= 4) a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        if (CHAINS >= 4) a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        else if (CHAINS >= 2) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    }
    if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) sink[0] = 1.f;
}

template <int CHAINS>
void run_small(int cus, float* sink) {
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * wps, iters = 80000 / wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(small_mfma_loop<CHAINS>, dim3(blocks), dim3(256), 0, 0, iters, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double cyc = ms * 1e-3 * 2.39e9 / (4.0 * iters * wps);
        printf("16x16x4 f32 MFMA only, %d independent chain(s)                  waves/SIMD=%d  %6.2f ms  %6.1f cycles per MFMA per SIMD  (pipe %.0f %%)\n",
               CHAINS, wps, ms, cyc, 3200.0 / cyc);
    }
}

template <int KIND>
void run(int cus, float* sink) {
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * wps, iters = 40000 / wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {   // the first launches run in the start-up ramp
            hipEventRecord(e0);
            hipLaunchKernelGGL(mix_loop<KIND>, dim3(blocks), dim3(256), 0, 0, iters, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double mfma_per_simd = 4.0 * iters * wps;
        const double cycles = ms * 1e-3 * 2.39e9;
        printf("%-60s waves/SIMD=%d  %6.2f ms  %6.1f cycles per MFMA per SIMD  (pipe %.0f %%)\n", NAMES[KIND],
               wps, ms, cycles / mfma_per_simd, 6400.0 / (cycles / mfma_per_simd));
    }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs; cycles at the 2.39 GHz sustained clock of tools/ubench_clock.hip\n",
           p.gcnArchName, cus);
    float* sink;
    hipMalloc(&sink, 4);
    run_small<4>(cus, sink);
    run_small<2>(cus, sink);
    run_small<1>(cus, sink);
    run<M_CHAIN1>(cus, sink);
    run<M_CHAIN2>(cus, sink);
    run<M_ONLY>(cus, sink);
    run<M_VALU1>(cus, sink);
    run<M_VALU2>(cus, sink);
    run<M_VALU4>(cus, sink);
    run<M_VALU8>(cus, sink);
    run<M_DEP1>(cus, sink);
    run<M_LDS1>(cus, sink);
    run<M_LDS2>(cus, sink);
    run<M_LDS2_DEP>(cus, sink);
    run<M_EXP1>(cus, sink);
    run<M_PK2>(cus, sink);
    run<M_PK4>(cus, sink);
    run<M_B128_4>(cus, sink);
    run<M_B128_2>(cus, sink);
    run<M_B128_1>(cus, sink);
    run<M_DEP_AHEAD4>(cus, sink);
    return 0;
}
