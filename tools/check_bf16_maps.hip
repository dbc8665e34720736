// One wave, one 32-document step of the LDA statistic on the operand-split bf16 route, against the host:
// checks csrc/bsc_bf16split.h's lane maps, the LDS image and its two kinds of read, the LDS-DMA fill, and
// prints the error of 1-, 2- and 3-term splits.  Exact-integer data first (any wrong map shows), then
// random positive data.
//   hipcc --offload-arch=gfx950 -O3 -I bayesic_amd/csrc tools/check_bf16_maps.hip -o tools/check_bf16_maps
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bsc_bf16split.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr;

// Th: SPLIT row-major [32][128] bf16 terms; BtT: SPLIT [32 v][128 k] bf16 terms; C [32][32] f32.
// out: P (16 per lane), S (4 x 16 per lane)
template <int SPLIT, bool INTEGER>
__global__ __launch_bounds__(64) void step_kernel(const unsigned short* Th, const unsigned short* BtT, const float* C,
                                                  float* outP, float* outS) {
    __shared__ __attribute__((aligned(1024))) char lds[SPLIT * 8192];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    // fill by LDS-DMA: instruction i covers LDS bytes 1024 i ..: row 4 i + lane / 16, position lane % 16
    for (int c = 0; c < SPLIT; ++c) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Th + c * 32 * 128), 0, 8192, 0x00020000);
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + (lane >> 4), pos = lane & 15;
            const int ch = pos ^ (((row & 3) << 2) | ((row >> 2) & 3));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + c * 8192 + i * 1024), 16, (unsigned)(row * 256 + ch * 16), 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const unsigned lb = (unsigned)(uintptr_t)(lds_ptr)lds;

    // phase 1: P = Th Bt
    bsc_f32x16 P = {};
    for (int ks = 0; ks < 8; ++ks) {
        bsc_u32x4 a[SPLIT], b[SPLIT];
        for (int c = 0; c < SPLIT; ++c) {
            a[c] = *reinterpret_cast<const bsc_u32x4*>(lds + c * 8192 + bsc_img256_off(r, 2 * ks + h));
            b[c] = *reinterpret_cast<const bsc_u32x4*>(BtT + c * 32 * 128 + r * 128 + 16 * ks + 8 * h);
        }
        P = bsc_mfma_split<SPLIT>(a, b, P);
    }
    for (int i = 0; i < 16; ++i) outP[lane * 16 + i] = P[i];
    // the middle: X = an element-wise function of P in the result layout
    float X[16];
    for (int i = 0; i < 16; ++i) {
        const int doc = (i & 3) + 8 * (i >> 2) + 4 * h;
        if (INTEGER) X[i] = P[i] - 5.f * floorf(P[i] / 5.f) - 2.f;
        else X[i] = C[doc * 32 + r] / P[i];
    }
    bsc_u32x4 xf[2][SPLIT];
    for (int s = 0; s < 2; ++s)
        for (int t = 0; t < 4; ++t) {
            unsigned pk[SPLIT];
            bsc_split_pk<SPLIT>(X[8 * s + 2 * t], X[8 * s + 2 * t + 1], pk);
            for (int c = 0; c < SPLIT; ++c) xf[s][c][t] = pk[c];
        }
    // phase 2: S[kb] = Th^T X
    const int g = lane >> 4;
    for (int kb = 0; kb < 4; ++kb) {
        bsc_f32x16 S = {};
        for (int s = 0; s < 2; ++s) {
            bsc_u32x4 a[SPLIT];
            for (int c = 0; c < SPLIT; ++c) {
                bsc_u32x2 t0, t1;
                const unsigned a0 = lb + c * 8192 + bsc_img256_tr_addr(lane, 16 * s + 4 * (g >> 1), 32 * kb + 16 * (g & 1));
                const unsigned a1 = lb + c * 8192 + bsc_img256_tr_addr(lane, 16 * s + 8 + 4 * (g >> 1), 32 * kb + 16 * (g & 1));
                BSC_LDS_TR_B64(t0, a0, 0);
                BSC_LDS_TR_B64(t1, a1, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1));      // (the values depend on the wait)
                a[c] = bsc_u32x4{t0[0], t0[1], t1[0], t1[1]};
            }
            S = bsc_mfma_split<SPLIT>(a, xf[s], S);
        }
        for (int i = 0; i < 16; ++i) outS[(kb * 64 + lane) * 16 + i] = S[i];
    }
}

static unsigned short to_bf16(float x) {   // round to nearest even
    unsigned u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float from_bf16(unsigned short b) {
    unsigned u = (unsigned)b << 16;
    float x;
    std::memcpy(&x, &u, 4);
    return x;
}

template <int SPLIT, bool INTEGER>
static double run(const std::vector<float>& Th, const std::vector<float>& Bt, const std::vector<float>& C, double* errP) {
    std::vector<unsigned short> ThS(SPLIT * 32 * 128), BtS(SPLIT * 32 * 128);
    for (int i = 0; i < 32 * 128; ++i) {
        float a = Th[i];
        const int k = i / 32, v = i % 32;          // Bt [128][32] -> BtT [32][128]
        float b = Bt[i];
        for (int c = 0; c < SPLIT; ++c) {
            ThS[c * 4096 + i] = to_bf16(a);
            a -= from_bf16(ThS[c * 4096 + i]);
            BtS[c * 4096 + v * 128 + k] = to_bf16(b);
            b -= from_bf16(BtS[c * 4096 + v * 128 + k]);
        }
    }
    unsigned short *dTh, *dBt;
    float *dC, *dP, *dS;
    CHECK(hipMalloc(&dTh, ThS.size() * 2)); CHECK(hipMalloc(&dBt, BtS.size() * 2));
    CHECK(hipMalloc(&dC, 32 * 32 * 4)); CHECK(hipMalloc(&dP, 64 * 16 * 4)); CHECK(hipMalloc(&dS, 4 * 64 * 16 * 4));
    CHECK(hipMemcpy(dTh, ThS.data(), ThS.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dBt, BtS.data(), BtS.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dC, C.data(), 32 * 32 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((step_kernel<SPLIT, INTEGER>), dim3(1), dim3(64), 0, 0, dTh, dBt, dC, dP, dS);
    CHECK(hipDeviceSynchronize());
    std::vector<float> P(64 * 16), S(4 * 64 * 16);
    CHECK(hipMemcpy(P.data(), dP, P.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(S.data(), dS, S.size() * 4, hipMemcpyDeviceToHost));
    // host reference in float64
    std::vector<double> Pr(32 * 32), Xr(32 * 32), Sr(128 * 32);
    for (int d = 0; d < 32; ++d)
        for (int v = 0; v < 32; ++v) {
            double t = 0;
            for (int k = 0; k < 128; ++k) t += (double)Th[d * 128 + k] * (double)Bt[k * 32 + v];
            Pr[d * 32 + v] = t;
            Xr[d * 32 + v] = INTEGER ? t - 5.0 * floor(t / 5.0) - 2.0 : (double)C[d * 32 + v] / t;
        }
    for (int k = 0; k < 128; ++k)
        for (int v = 0; v < 32; ++v) {
            double t = 0;
            for (int d = 0; d < 32; ++d) t += (double)Th[d * 128 + k] * Xr[d * 32 + v];
            Sr[k * 32 + v] = t;
        }
    double eP = 0, eS = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), col = lane & 31;
            const double p = Pr[row * 32 + col];
            eP = fmax(eP, fabs(P[lane * 16 + i] - p) / (INTEGER ? 1.0 : fabs(p)));
            for (int kb = 0; kb < 4; ++kb) {
                const double s = Sr[(32 * kb + row) * 32 + col];
                eS = fmax(eS, fabs(S[(kb * 64 + lane) * 16 + i] - s) / (INTEGER ? 1.0 : fabs(s)));
            }
        }
    *errP = eP;
    CHECK(hipFree(dTh)); CHECK(hipFree(dBt)); CHECK(hipFree(dC)); CHECK(hipFree(dP)); CHECK(hipFree(dS));
    return eS;
}

int main() {
    std::vector<float> Th(32 * 128), Bt(128 * 32), C(32 * 32);
    srand(7);
    for (auto& x : Th) x = (float)(rand() % 7 - 3);
    for (auto& x : Bt) x = (float)(rand() % 5 - 2);
    double eP, eS;
    eS = run<1, true>(Th, Bt, C, &eP);
    printf("integer data, 1 term : max |P - ref| = %g, max |S - ref| = %g  (%s)\n", eP, eS, eP == 0 && eS == 0 ? "maps OK" : "MAPS WRONG");
    int bad = !(eP == 0 && eS == 0);
    eS = run<2, true>(Th, Bt, C, &eP);
    printf("integer data, 2 terms: max |P - ref| = %g, max |S - ref| = %g  (%s)\n", eP, eS, eP == 0 && eS == 0 ? "maps OK" : "MAPS WRONG");
    bad |= !(eP == 0 && eS == 0);
    for (auto& x : Th) x = 0.05f + (float)(rand() % 100000) / 100000.f;
    for (auto& x : Bt) x = 0.05f + (float)(rand() % 100000) / 100000.f;
    for (auto& x : C) x = (float)(rand() % 4);
    eS = run<1, false>(Th, Bt, C, &eP);
    printf("random data, 1 term  (1 product):  max rel error P %.3g, S %.3g\n", eP, eS);
    eS = run<2, false>(Th, Bt, C, &eP);
    printf("random data, 2 terms (3 products): max rel error P %.3g, S %.3g\n", eP, eS);
    eS = run<3, false>(Th, Bt, C, &eP);
    printf("random data, 3 terms (6 products): max rel error P %.3g, S %.3g\n", eP, eS);
    return bad;
}
