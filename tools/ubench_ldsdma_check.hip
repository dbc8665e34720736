// Semantics check of buffer_load_dwordx4 ... lds on gfx950: where does lane i's 16 bytes land, and
// what do out-of-range lanes write?   hipcc --offload-arch=gfx950 -O3 ldsdma_check.hip && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* out, int n) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -7.f;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)n * 4u, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, threadIdx.x * 16, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, threadIdx.x * 16, 1024, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    const int n = 400;   // the second load (floats 256..511) runs past the end at 400
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)i;
    float *x, *out;
    hipMalloc(&x, 512 * 4); hipMalloc(&out, 512 * 4);
    hipMemcpy(x, h.data(), 512 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, out, n);
    std::vector<float> o(512);
    hipMemcpy(o.data(), out, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 512; ++i) {
        const float want = i < n ? (float)i : 0.f;
        if (o[i] != want) { if (bad < 8) printf("lds[%d] = %g, expected %g\n", i, o[i], want); ++bad; }
    }
    printf("%s (%d mismatches)\n", bad ? "DIFFERENT" : "lane i writes bytes [16 i, 16 i + 16) from M0; out-of-range reads store 0", bad);
    return 0;
}
