// How many bytes must a CU keep in flight for a pure LDS-DMA read of HBM to reach its ceiling, and how evenly
// do the workgroups of a statically partitioned stream finish?  (round 4: sizing the feed of blr_pass_q_kernel)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_dma_inflight tools/ubench_dma_inflight.hip
//   tools/ubench_dma_inflight            -> one line per (waves per CU, KiB in flight per wave)
//
// Each wave owns a ring of DEPTH 1-KiB slots and streams 16-KiB tiles (tile = wave index + k * waves, the
// partition of the pass kernels); a slot is re-issued as soon as the DMA that last filled it has landed
// (s_waitcnt vmcnt(DEPTH - 1)), so DEPTH KiB per wave are in flight at all times.  Every workgroup stamps
// s_memrealtime at its start and end: the spread of the end stamps is the imbalance a static partition pays.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int vmcnt_only(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e = (x);                                                                        \
        if (e != hipSuccess) {                                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                 \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

template <int DEPTH, int AUX>
__global__ __launch_bounds__(256) void stream_kernel(const void* x, unsigned bytes, unsigned n_tiles,
                                                     unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) char ring[4 * DEPTH * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, bytes, 0x00020000);
    char* const my = ring + wave * DEPTH * 1024;
    const unsigned n_waves = gridDim.x * 4;
    int slot = 0;
    for (unsigned tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += n_waves) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            __builtin_amdgcn_s_waitcnt(vmcnt_only(DEPTH - 1));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(my + ((slot + j) % DEPTH) * 1024), 16, 16 * lane,
                                                     tile * 16384u + 1024u * j, 0, AUX);
        }
        slot = (slot + 16) % DEPTH;
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[4 * blockIdx.x] = t0;
        stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((3 << 11) | 20);    // HW_REG_XCC_ID[3:0]
        stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((15 << 11) | 4);    // HW_REG_HW_ID[15:0]: cu_id [11:8], sh_id [12], se_id [15:13]
    }
}

template <int DEPTH, int AUX>
int run(const void* buf, size_t bytes, int wg_per_cu, int cus, unsigned long long* stamps_d, bool detail = false) {
    const unsigned n_tiles = (unsigned)(bytes / 16384);
    const int grid = wg_per_cu * cus;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> ms;
    std::vector<unsigned long long> st(4 * grid);
    double spread_us = 0, p50_us = 0;
    for (int r = 0; r < 12; ++r) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((stream_kernel<DEPTH, AUX>), dim3(grid), dim3(256), 0, 0, buf, (unsigned)(n_tiles * 16384ull),
                           n_tiles, stamps_d);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms.push_back(t);
        if (r == 11) {
            CHECK(hipMemcpy(st.data(), stamps_d, st.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long first = ~0ull;
            for (int b = 0; b < grid; ++b) first = std::min(first, st[4 * b]);
            std::vector<double> ends(grid);
            double xs[8] = {0}, xn[8] = {0}, xmin[8], xmax[8] = {0};
            for (int x = 0; x < 8; ++x) xmin[x] = 1e30;
            for (int b = 0; b < grid; ++b) {
                ends[b] = (double)(st[4 * b + 1] - first) / 100.0;   // 100 MHz ticks -> us
                const int x = (int)(st[4 * b + 2] & 7);
                xs[x] += ends[b]; xn[x] += 1; xmin[x] = std::min(xmin[x], ends[b]); xmax[x] = std::max(xmax[x], ends[b]);
            }
            if (detail) {
                printf("    per XCD (workgroups: mean end, min, max us):");
                for (int x = 0; x < 8; ++x) printf("  [%d] %3.0f: %.0f %.0f %.0f", x, xn[x], xn[x] ? xs[x] / xn[x] : 0, xmin[x], xmax[x]);
                printf("\n    block -> (xcd, hw_id>>8, end us), first 24:");
                for (int b = 0; b < 24 && b < grid; ++b) printf(" %d:(%d,%02x,%.0f)", b, (int)(st[4 * b + 2] & 7), (unsigned)(st[4 * b + 3] >> 8) & 0xff, ends[b]);
                printf("\n");
            }
            std::sort(ends.begin(), ends.end());
            spread_us = ends.back() - ends.front();
            p50_us = ends[grid / 2];
            printf("    end stamps (us after the first start): min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", ends.front(),
                   ends[grid / 10], p50_us, ends[grid * 9 / 10], ends.back());
        }
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    printf("waves/CU=%2d  KiB in flight per wave=%2d (per CU %3d)  aux=%d  median %.1f us  %.2f TB/s  (min %.1f us)  end spread %.1f us\n",
           4 * wg_per_cu, DEPTH, 4 * wg_per_cu * DEPTH, AUX, med * 1e3, (double)n_tiles * 16384.0 / (med * 1e-3) / 1e12,
           ms.front() * 1e3, spread_us);
    (void)p50_us;
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t bytes = (size_t)3 << 30;      // 3 GiB: nothing of one sweep is left in the 256 MiB Infinity Cache for the next
    void* buf = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 1, bytes));
    unsigned long long* stamps = nullptr;
    CHECK(hipMalloc(&stamps, 4 * 8 * 4096));
    printf("device %s, %d CUs; 3 GiB streamed per launch by LDS-DMA in 1-KiB pieces\n", prop.gcnArchName, cus);
    for (int wg = 1; wg <= 2; ++wg) {
        if (run<4, 2>(buf, bytes, wg, cus, stamps)) return 1;
        if (run<8, 2>(buf, bytes, wg, cus, stamps)) return 1;
        if (run<12, 2>(buf, bytes, wg, cus, stamps)) return 1;
        if (run<16, 2>(buf, bytes, wg, cus, stamps, true)) return 1;
    }
    if (run<16, 0>(buf, bytes, 2, cus, stamps)) return 1;
    if (run<8, 2>(buf, bytes, 3, cus, stamps)) return 1;
    if (run<8, 2>(buf, bytes, 4, cus, stamps)) return 1;
    // 32 KiB per wave, one workgroup per CU (128 KiB per CU from 4 waves)
    if (run<32, 2>(buf, bytes, 1, cus, stamps)) return 1;
    return 0;
}
