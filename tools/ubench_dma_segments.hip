// LDS-DMA streaming rate of a 1M x 256 float matrix (1 GB) as a function of how a 1-KiB wave instruction is cut into
// row segments: 16 rows x 64 B (the strips of logreg_loglik_dma_*_kernel and gemm_skinny_*), 8 x 128 B, 4 x 256 B,
// 1 x 1 KiB.  No arithmetic: a wave owns 16-row tiles and keeps RING instructions in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dma_segments.hip -o tools/ubench_dma_segments
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int SEG_BYTES, int WAVES, int RING = 16, int KEEP = 15>        // bytes of one row a wave instruction takes; ring slots; outstanding allowed at a wait
__global__ __launch_bounds__(64 * WAVES) void stream_kernel(const float* X, long N, int* sink) {
    __shared__ __attribute__((aligned(1024))) char lds[WAVES * RING * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int LPR = SEG_BYTES / 16;            // lanes per row segment
    constexpr int ROWS = 64 / LPR;                 // rows per instruction
    constexpr int IPT = 16 * 1024 / (ROWS * SEG_BYTES) ;   // instructions per 16-row tile ... = 16 always
    const long n_tiles = N / 16, n_waves = (long)gridDim.x * WAVES;
    char* my = lds + wave * RING * 1024;
    int issued = 0;
    for (long t = (long)blockIdx.x * WAVES + wave; t < n_tiles; t += n_waves) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + t * 16 * 256), 0, 16 * 1024, 0x00020000);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // instruction i of the tile: rows (i % (16 / ROWS)) * ROWS + lane / LPR, byte offset in the row (i / (16 / ROWS)) * SEG_BYTES + 16 (lane % LPR)
            const int rg = i % (16 / ROWS), cg = i / (16 / ROWS);
            const unsigned voff = (unsigned)((rg * ROWS + lane / LPR) * 1024 + cg * SEG_BYTES + 16 * (lane % LPR));
            if (issued >= RING) __builtin_amdgcn_s_waitcnt((KEEP & 0xF) | ((KEEP >> 4) << 14) | (0x7 << 4) | (0xF << 8));   // vmcnt(KEEP)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(my + (i % RING) * 1024), 16, voff, 0, 0, 2);
            ++issued;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (sink && lds[threadIdx.x] == 77 && threadIdx.x == 9999) sink[0] = 1;
}

template <int SEG, int WAVES, int RING = 16, int KEEP = 15>
void run(const float* X, long N, int cus, int wg_per_cu) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<SEG, WAVES, RING, KEEP>), dim3(cus * wg_per_cu), dim3(64 * WAVES), 0, 0, X, N, (int*)nullptr);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<SEG, WAVES, RING, KEEP>), dim3(cus * wg_per_cu), dim3(64 * WAVES), 0, 0, X, N, (int*)nullptr);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("segment %4d B, %d waves x %d workgroups per CU, ring %d, vmcnt(%d): %.1f us  %.2f TB/s\n", SEG, WAVES, wg_per_cu, RING, KEEP, ms / reps * 1e3, N * 1024.0 / (ms / reps) * 1e-9);
}

int main() {
    const long N = 1000000 / 16 * 16;
    float* X;
    CHECK(hipMalloc(&X, N * 1024));
    CHECK(hipMemset(X, 0, N * 1024));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    run<64, 8>(X, N, cus, 1);
    run<128, 8>(X, N, cus, 1);
    run<256, 8>(X, N, cus, 1);
    run<1024, 8>(X, N, cus, 1);
    run<64, 4>(X, N, cus, 2);
    run<1024, 4>(X, N, cus, 2);
    run<64, 8, 8, 7>(X, N, cus, 1);       // the ring of logreg_loglik_dma_*: eight strips
    run<64, 8, 8, 4>(X, N, cus, 1);       // ... with the split kernel's wait (at most four outstanding)
    run<64, 8, 8, 2>(X, N, cus, 1);
    run<64, 8, 4, 3>(X, N, cus, 1);
    return 0;
}
