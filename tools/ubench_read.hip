// Pure streaming-read ceiling on MI355X: 1 GiB of float4 loads, trivial VALU.
// Variants: grid size (blocks/CU), loads in flight per lane, global vs buffer loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ x, size_t n4, float* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { float4 v = x[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

// row-tile pattern like blr_pass_kernel: a wave reads 8 consecutive 1-KiB rows per step
__global__ __launch_bounds__(256) void tile_read_kernel(const float4* __restrict__ x, size_t rows, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t stride = (size_t)gridDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t tile = (size_t)blockIdx.x * 4 + wave; tile * 8 + 7 < rows; tile += stride) {
        float4 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = x[(tile * 8 + r) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 8; ++r) { acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

template <typename F>
double time_us(F launch, int reps = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const size_t rows = 1000000, n4 = rows * 64;   // 1.024e9 bytes, like X
    float4* x; float* out;
    hipMalloc(&x, n4 * 16); hipMalloc(&out, 4);
    hipMemset(x, 1, n4 * 16);
    const double gb = n4 * 16 / 1e9;
    for (int bpc : {1, 2, 4, 8, 16}) {
        int blocks = 256 * bpc;
        double a = time_us([&] { hipLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
        double b = time_us([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
        double c = time_us([&] { hipLaunchKernelGGL(read_kernel<8>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
        double d = time_us([&] { hipLaunchKernelGGL(tile_read_kernel, dim3(blocks), dim3(256), 0, 0, x, rows, out); });
        printf("blocks/CU=%2d  unroll1 %.1f us (%.0f GB/s)  unroll4 %.1f us (%.0f GB/s)  unroll8 %.1f us (%.0f GB/s)  tile8rows %.1f us (%.0f GB/s)\n",
               bpc, a, gb / a * 1e6, b, gb / b * 1e6, c, gb / c * 1e6, d, gb / d * 1e6);
    }
    return 0;
}
