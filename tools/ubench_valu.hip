// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for
// plain v_fma_f32, v_pk_fma_f32, DPP adds, v_readlane and permlane swaps at 1/2/4
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 2000

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,
          x6 = x0 + 6, x7 = x0 + 7;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
        if (KIND == 0) {  // 8 independent plain FMAs
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(a), "v"(b));
        } else if (KIND == 1) {  // 4 packed FMAs (same flops as 8 plain)
            asm volatile(
                "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n"
                "v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                : "v"(pa), "v"(pb));
        } else if (KIND == 2) {  // 8 DPP adds
            asm volatile(
                "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (KIND == 3) {  // 4 readlane + 4 fma using the sgpr
            asm volatile(
                "v_readlane_b32 s20, %0, 1\n v_readlane_b32 s21, %1, 2\n"
                "v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 4\n"
                "v_fma_f32 %4, s20, %4, %4\n v_fma_f32 %5, s21, %5, %5\n"
                "v_fma_f32 %6, s22, %6, %6\n v_fma_f32 %7, s23, %7, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                :: "s20", "s21", "s22", "s23");
        } else if (KIND == 4) {  // 4 permlane32_swap + 4 permlane16_swap
            asm volatile(
                "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n"
                "v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n"
                "v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (KIND == 5) {  // 8 v_cndmask with vcc
            asm volatile(
                "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n"
                "v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n"
                "v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                :: "vcc");
        } else if (KIND == 6) {  // 8 plain fma with SGPR operand
            asm volatile(
                "v_fma_f32 %0, s20, %8, %0\n v_fma_f32 %1, s20, %8, %1\n v_fma_f32 %2, s20, %8, %2\n"
                "v_fma_f32 %3, s20, %8, %3\n v_fma_f32 %4, s20, %8, %4\n v_fma_f32 %5, s20, %8, %5\n"
                "v_fma_f32 %6, s20, %8, %6\n v_fma_f32 %7, s20, %8, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(a) : "s20");
        } else if (KIND == 7) {  // 8 v_fmac_f32 (VOP2)
            asm volatile(
                "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n"
                "v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n"
                "v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(a), "v"(b));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (KIND == 1) { x0 = p0[0] + p0[1]; x1 = p1[0] + p1[1]; x2 = p2[0] + p2[1]; x3 = p3[0] + p3[1]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, int n_instr) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float) * 4);
    hipMalloc(&cyc, 256 * 64 * sizeof(unsigned long long));
    for (int wps : {1, 2, 4, 8}) {       // waves per SIMD: block = 256*wps threads, 1 block/CU
        int threads = 256 * wps > 1024 ? 1024 : 256 * wps;
        int blocks = 256 * (256 * wps / threads);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= h.size();
        // s_memtime ticks at 100 MHz on gfx9? report both raw ticks and wall-derived cycles
        double instr_per_simd = (double)ITERS * n_instr * wps;
        printf("%-28s waves/SIMD=%d  wall=%.1f us  ticks/wave=%.0f  wall-ns per instr per SIMD=%.3f\n",
               name, wps, ms * 1e3, mean, ms * 1e6 / instr_per_simd);
    }
}

int main() {
    run<0>("v_fma_f32 x8", 8);
    run<7>("v_fmac_f32 x8", 8);
    run<1>("v_pk_fma_f32 x4", 4);
    run<6>("v_fma_f32 sgpr x8", 8);
    run<2>("v_add_f32_dpp x8", 8);
    run<3>("readlane x4 + fma x4", 8);
    run<4>("permlane32/16 swap x8", 8);
    run<5>("v_cndmask x8", 8);
    return 0;
}
