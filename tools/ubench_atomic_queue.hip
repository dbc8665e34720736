// What does a device-wide tile queue cost on gfx950?  Returning agent-scope atomic adds from one lane of each of
// 2048 resident waves (512 workgroups x 4 waves, the pass kernels' launch) on 1, 8 (one per blockIdx % 8 ~ XCD) or 64
// addresses: throughput with every wave hammering, and the latency one wave sees with `gap` ~us of s_sleep between its
// own requests (the load a work-stealing tail of blr_pass_q_kernel would put on the counters).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_atomic_queue tools/ubench_atomic_queue.hip && tools/ubench_atomic_queue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e = (x);                                                                        \
        if (e != hipSuccess) {                                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                 \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

__global__ __launch_bounds__(256) void grab_kernel(unsigned* counters, int n_addr, int stride_words, int per_wave, int sleep_units,
                                                   unsigned long long* lat_ticks, unsigned* sink) {
    const int lane = threadIdx.x & 63;
    if (lane != 0) return;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned* c = counters + (size_t)(blockIdx.x % n_addr) * stride_words;
    unsigned acc = 0;
    unsigned long long waited = 0;
    for (int i = 0; i < per_wave; ++i) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += t;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        waited += __builtin_amdgcn_s_memrealtime() - t0;
        for (int s = 0; s < sleep_units; ++s) __builtin_amdgcn_s_sleep(127);       // 127 x 64 cycles ~ 3.4 us at 2.4 GHz
    }
    lat_ticks[wave] = waited;
    if (acc == 0xffffffffu) sink[0] = acc;
}

int main() {
    unsigned* counters = nullptr;
    unsigned long long* lat = nullptr;
    unsigned* sink = nullptr;
    CHECK(hipMalloc(&counters, 64 * 1024));
    CHECK(hipMalloc(&lat, 2048 * 8));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(2048);
    const int addrs[] = {1, 8, 64};
    for (int sleep_units : {0, 1}) {
        for (int n_addr : addrs) {
            for (int stride_words : {1, 64}) {               // counters in one 128-byte line or one line each
                if (n_addr == 1 && stride_words != 1) continue;
                const int per_wave = sleep_units ? 8 : 32;
                float best = 1e30f;
                for (int r = 0; r < 5; ++r) {
                    CHECK(hipMemset(counters, 0, 64 * 1024));
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(grab_kernel, dim3(512), dim3(256), 0, 0, counters, n_addr, stride_words, per_wave, sleep_units, lat, sink);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                }
                CHECK(hipMemcpy(h.data(), lat, 2048 * 8, hipMemcpyDeviceToHost));
                std::vector<double> l(2048);
                for (int i = 0; i < 2048; ++i) l[i] = (double)h[i] / per_wave * 10.0;        // 100 MHz ticks -> ns
                std::sort(l.begin(), l.end());
                const double n = 2048.0 * per_wave;
                printf("%-26s addresses=%2d stride=%3d B: %6.1f us for %6.0f atomics = %5.2f ns each;  latency seen by a wave: p10 %6.0f p50 %6.0f p90 %6.0f max %6.0f ns\n",
                       sleep_units ? "3.4 us between requests" : "back to back", n_addr, stride_words * 4, best * 1e3, n, best * 1e6 / n,
                       l[204], l[1024], l[1843], l[2047]);
            }
        }
    }
    return 0;
}
