// Sustained shader clock and fp32-MFMA issue rate under full-chip load.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_clock.hip -o /tmp/ubench_clock && /tmp/ubench_clock
// Every wave runs a long chain-free MFMA loop (4 independent accumulators) or a VALU FMA
// loop; lane 0 of each wave samples s_memtime (shader clock) and s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(int iters, unsigned long long* out, float* sink) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = threadIdx.x * 1e-9f, y = 1.0f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = r1 - r0;
    }
    if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) sink[0] = 1.f;
}

__global__ __launch_bounds__(256) void valu_loop(int iters, unsigned long long* out, float* sink) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 1.0000001f, 1e-7f);
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = r1 - r0;
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 12345.f) sink[0] = s;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, nominal clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    unsigned long long* out;
    float* sink;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int blocks = cus * waves_per_simd, waves = blocks * 4;
        hipMalloc(&out, waves * 16);
        hipMalloc(&sink, 4);
        for (int kind = 0; kind < 2; ++kind) {
            const int iters = kind == 0 ? 200000 : 2000000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, iters, out, sink);
                else hipLaunchKernelGGL(valu_loop, dim3(blocks), dim3(256), 0, 0, iters, out, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * waves);
            hipMemcpy(h.data(), out, waves * 16, hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0;
            for (int w = 0; w < waves; ++w) { cyc += h[2 * w]; rt += h[2 * w + 1]; }
            cyc /= waves; rt /= waves;
            const double mhz = cyc / (rt / 100.0);       // s_memrealtime ticks at 100 MHz
            if (kind == 0) {
                const double mfma_per_simd = 4.0 * iters * waves_per_simd;
                const double tf = 4.0 * iters * (double)waves * 4096.0 / (ms * 1e-3) / 1e12;
                printf("MFMA 32x32x2 f32, %d wave(s)/SIMD: kernel %.2f ms, shader clock %.0f MHz, "
                       "%.1f cycles per MFMA per SIMD, %.1f TF\n", waves_per_simd, ms, mhz,
                       cyc / mfma_per_simd, tf);
            } else {
                const double tf = 8.0 * iters * (double)waves * 64 * 2.0 / (ms * 1e-3) / 1e12;
                printf("VALU fma f32,      %d wave(s)/SIMD: kernel %.2f ms, shader clock %.0f MHz, %.1f TF\n",
                       waves_per_simd, ms, mhz, tf);
            }
        }
        hipFree(out); hipFree(sink);
    }
    return 0;
}
