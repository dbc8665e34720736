// What does a vector-memory instruction cost the fp32 MFMA pipe, and who pays it?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_vmem.hip -o /tmp/ubench_mfma_vmem && /tmp/ubench_mfma_vmem
// Question behind it (config 5's log-likelihood pass): 16 rows x 256 columns of X feed 256
// v_mfma_f32_16x16x4_f32 (S = 64) and take sixteen 1-KiB loads.  The deletion profile of the
// kernel (tools/ab_bbvi.py --deletion) charges ~65 shader cycles of MFMA time to every vector
// memory instruction.  Is that price paid by the SIMD (then nothing helps but fewer instructions)
// or by the issuing wave (then dedicated loader waves hide it)?
//
// One 768-thread workgroup per CU = 12 waves = 3 per SIMD.  Waves 0-7 ("compute", two per SIMD)
// run 16 MFMAs per step on four accumulator chains with operands taken from memory once (random
// data: the clock the chip holds depends on the operand values).  Per step one 1-KiB load is
// issued, by the compute wave itself or by waves 8-11 ("loaders", one per SIMD, two loads per
// step each), streaming a 1-GiB buffer exactly once per launch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum Mode {
    MFMA_ONLY,          // no memory instructions at all
    SELF_BUF_ROW,       // compute wave: buffer_load_dwordx4, lane l reads bytes [16 l, 16 l + 16) of a 1-KiB run
    SELF_BUF_TILE,      // compute wave: buffer_load_dwordx4, 16 rows x 64 B (row stride 1 KiB) per instruction
    SELF_GLOBAL,        // compute wave: global_load_dwordx4 with an SGPR base
    SELF_LDSDMA,        // compute wave: buffer_load_dwordx4 ... lds (no VGPR destination)
    SELF_GATHER,        // compute wave: buffer_load_dword, four 64-B segments per instruction
    LOADER_BUF,         // loader waves issue the buffer loads (to registers, discarded)
    LOADER_LDSDMA,      // loader waves issue LDS-DMA
    LOADER_BUF_NO_MFMA, // loader waves alone: how long the loads take by themselves
    SELF_BUF_ROW_X2,    // compute wave: two loads per step (twice the bytes): does the price double?
    SELF_LDSDMA_TILE,   // compute wave: LDS-DMA, 16 rows x 64 B per instruction (the A-operand strip of a 16-row tile)
    SELF_LDSDMA_TILE_RD,// ... plus the operand reads of config 5's inner loop: 5 ds_read_b128 per 16 MFMAs (1 A strip + 4 B)
    LOADER_LDSDMA_TILE, // loader waves issue the strip DMAs
    SELF_LDSDMA_TILE_RD_1W, // as SELF_LDSDMA_TILE_RD with ONE compute wave per SIMD (4 compute waves per CU, twice the steps)
    N_MODES
};
static const char* NAMES[N_MODES] = {
    "MFMA only",
    "compute wave loads: buffer_load_dwordx4, 1-KiB run",
    "compute wave loads: buffer_load_dwordx4, 16 rows x 64 B",
    "compute wave loads: global_load_dwordx4 saddr",
    "compute wave loads: buffer_load_dwordx4 ... lds (LDS-DMA)",
    "compute wave loads: buffer_load_dword gather (4 x 64 B)",
    "loader waves (1 per SIMD) load: buffer_load_dwordx4",
    "loader waves (1 per SIMD) load: LDS-DMA",
    "loader waves alone, no MFMA waves",
    "compute wave loads: 2 x buffer_load_dwordx4 per 16 MFMAs",
    "compute wave loads: LDS-DMA, 16 rows x 64 B",
    "compute wave: LDS-DMA strip + 5 ds_read_b128 per 16 MFMAs",
    "loader waves (1 per SIMD) load: LDS-DMA, 16 rows x 64 B",
    "ONE compute wave per SIMD: LDS-DMA strip + 5 ds_read_b128",
};

constexpr int COMPUTE_WAVES = 8, LOADER_WAVES = 4;

template <int MODE>
__global__ __launch_bounds__(768, 3) void kern(const float* __restrict__ X, size_t bytes, int steps,
                                               float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float lds[12 * 8 * 256];   // 8 x 1-KiB DMA slots per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool compute = wave < COMPUTE_WAVES;
    const bool loaders_load = MODE == LOADER_BUF || MODE == LOADER_LDSDMA || MODE == LOADER_BUF_NO_MFMA ||
                              MODE == LOADER_LDSDMA_TILE;
    if (MODE == SELF_LDSDMA_TILE_RD_1W && wave >= 4) return;
    if (MODE == SELF_LDSDMA_TILE_RD || MODE == SELF_LDSDMA_TILE_RD_1W) {
        for (int i = threadIdx.x; i < 12 * 8 * 256; i += 768) lds[i] = X[(size_t)blockIdx.x * 24576 + i];   // random operands
        __syncthreads();
    }
    // the stream a wave owns: compute wave c of block b reads [((b*8 + c) * steps + t) KiB]
    const size_t n_streams = (size_t)gridDim.x * COMPUTE_WAVES;
    f32x4 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float out = 0.f;
    if (compute) {
        if (MODE == LOADER_BUF_NO_MFMA) return;
        // (ONE compute wave per SIMD: 4 streams per block, each twice as long -- the same 1 GiB)
        const size_t stream = MODE == SELF_LDSDMA_TILE_RD_1W ? (size_t)blockIdx.x * 4 + wave
                                                              : (size_t)blockIdx.x * COMPUTE_WAVES + wave;
        const char* base = (const char*)X + stream * (size_t)steps * 1024;
        // operands: random data, loaded once
        const f32x4 av = *(const f32x4*)(base + 16 * lane);
        const f32x4 bv = *(const f32x4*)((const char*)X + ((stream * 7919u) % (n_streams / 2)) * 1024 + 16 * lane);
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)(steps * 1024), 0x00020000);
        const int i16 = lane & 15, kq = lane >> 4;
        const int voff_row = 16 * lane;
        const int voff_tile = i16 * 1024 + 16 * kq;     // 16 rows x 64 B; the step offset walks the 64-B column
        const int voff_gather = ((lane >> 4) * 977 % 61) * 64 + 4 * (lane & 15);
        __attribute__((address_space(3))) void* slot = (__attribute__((address_space(3))) void*)(lds + wave * 8 * 256);
        constexpr bool RD = MODE == SELF_LDSDMA_TILE_RD || MODE == SELF_LDSDMA_TILE_RD_1W;
        const f32x4* lq = (const f32x4*)lds;
        struct Ops { f32x4 a, b0, b1, b2, b3; };
        Ops P = {av, bv, bv, bv, bv}, Q = P;
        // one step: MFMAs on `cur` (read a step ago), operand reads for the next step into `nxt`
        auto step = [&](const Ops& cur, Ops& nxt, int t) {
            const f32x4 a = cur.a, b0 = cur.b0, b1 = cur.b1, b2 = cur.b2, b3 = cur.b3;
            if (RD) {
                // config 5's inner loop: the A strip DMA'd 7 steps ago and four static B blocks, read one
                // step ahead of the MFMAs that consume them
                f32x4 &an = nxt.a, &b0n = nxt.b0, &b1n = nxt.b1, &b2n = nxt.b2, &b3n = nxt.b3;
                int wo = lane;
                asm volatile("" : "+v"(wo));
                an = lq[wo + wave * 512 + ((t + 1) & 7) * 64];
                const int bb = ((wave + 3) % 12) * 512 + (t & 3) * 64;
                b0n = lq[wo + bb]; b1n = lq[wo + bb + 64]; b2n = lq[wo + bb + 128]; b3n = lq[wo + bb + 192];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b0[q], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], RD ? b1[q] : bv[(q + 1) & 3], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], RD ? b2[q] : bv[(q + 2) & 3], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], RD ? b3[q] : bv[(q + 3) & 3], c3, 0, 0, 0);
            }
            const int soff = t * 1024;
            if (MODE == SELF_BUF_ROW || MODE == SELF_BUF_ROW_X2) {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(r[t & 7]) : "v"(voff_row), "s"(rs), "s"(soff));
                if (MODE == SELF_BUF_ROW_X2)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(r[(t + 4) & 7]) : "v"(voff_row), "s"(rs), "s"(soff));
            } else if (MODE == SELF_BUF_TILE) {
                // tile = 16 steps: 16 rows of 1 KiB, step j reads the 64-B column j of every row
                const int so = (t >> 4) * 16384 + (t & 15) * 64;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(r[t & 7]) : "v"(voff_tile), "s"(rs), "s"(so));
            } else if (MODE == SELF_GLOBAL) {
                const char* p = base + (size_t)t * 1024;
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r[t & 7]) : "v"(voff_row), "s"(p));
            } else if (MODE == SELF_LDSDMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)slot + (t & 7) * 1024),
                                                         16, voff_row, soff, 0, 2);
            } else if (MODE == SELF_LDSDMA_TILE || MODE == SELF_LDSDMA_TILE_RD || MODE == SELF_LDSDMA_TILE_RD_1W) {
                const int so = (t >> 4) * 16384 + (t & 15) * 64;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)slot + (t & 7) * 1024),
                                                         16, voff_tile, so, 0, 2);
            } else if (MODE == SELF_GATHER) {
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(r[t & 7][0]) : "v"(voff_gather), "s"(rs), "s"(soff & ~63));
            }
            if (MODE != MFMA_ONLY && !loaders_load) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        };
        for (int t = 0; t < steps; t += 2) {      // steps is even
            step(P, Q, t);
            step(Q, P, t + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        out = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        if (!loaders_load) return;
        // loader l of this block serves the streams of compute waves 2l and 2l+1
        const int l = wave - COMPUTE_WAVES;
        __attribute__((address_space(3))) void* slot = (__attribute__((address_space(3))) void*)(lds + wave * 8 * 256);
        const int voff_row = 16 * lane;
        const size_t s0 = (size_t)blockIdx.x * COMPUTE_WAVES + 2 * l;
        const auto rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)X + s0 * (size_t)steps * 1024), 0,
                                                           (unsigned)(steps * 1024), 0x00020000);
        const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)X + (s0 + 1) * (size_t)steps * 1024), 0,
                                                           (unsigned)(steps * 1024), 0x00020000);
        for (int t = 0; t < steps; ++t) {
            const int soff = t * 1024;
            if (MODE == LOADER_LDSDMA_TILE) {
                const int so = (t >> 4) * 16384 + (t & 15) * 64;
                const int voff_tile = (lane & 15) * 1024 + 16 * (lane >> 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)slot + (t & 3) * 2048),
                                                         16, voff_tile, so, 0, 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)slot + (t & 3) * 2048 + 1024),
                                                         16, voff_tile, so, 0, 2);
            } else if (MODE == LOADER_LDSDMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)slot + (t & 3) * 2048),
                                                         16, voff_row, soff, 0, 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)slot + (t & 3) * 2048 + 1024),
                                                         16, voff_row, soff, 0, 2);
            } else {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(r[(2 * t) & 7]) : "v"(voff_row), "s"(rs0), "s"(soff));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(r[(2 * t + 1) & 7]) : "v"(voff_row), "s"(rs1), "s"(soff));
            }
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) out += r[k][0] + r[k][3];
    if (out == 12345.678f) sink[0] = lds[lane];
}

template <int MODE>
void run(int cus, const float* X, size_t bytes, float* sink) {
    int steps = (int)(bytes / 1024 / ((size_t)cus * COMPUTE_WAVES));
    const int compute_waves = MODE == SELF_LDSDMA_TILE_RD_1W ? 4 : COMPUTE_WAVES;
    if (MODE == SELF_LDSDMA_TILE_RD_1W) steps *= 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern<MODE>, dim3(cus), dim3(768), 0, 0, X, bytes, steps, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    const double mfma = 16.0 * steps * compute_waves * cus;
    const double tf = mfma * 2048.0 / (best * 1e-3) / 1e12;
    const double loads = (double)steps * compute_waves * cus * (MODE == SELF_BUF_ROW_X2 ? 2 : 1);
    printf("%-62s %7.1f us", NAMES[MODE], best * 1e3);
    if (MODE != LOADER_BUF_NO_MFMA) printf("  %6.1f TF (%.3f of 157.3)", tf, tf / 157.3);
    if (MODE != MFMA_ONLY) printf("  %.2f TB/s", loads * 1024 / (best * 1e-3) / 1e12);
    printf("\n");
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const size_t bytes = (size_t)1 << 30;
    float *X, *sink;
    hipMalloc(&X, bytes + (1 << 20));
    hipMalloc(&sink, 4);
    // random operands (the clock the chip holds depends on the data)
    float* h = (float*)malloc(bytes);
    unsigned s = 12345u;
    for (size_t i = 0; i < bytes / 4; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22));
    }
    hipMemcpy(X, h, bytes, hipMemcpyHostToDevice);
    free(h);
    printf("device %s, %d CUs; 16 v_mfma_f32_16x16x4_f32 per step and compute wave, one 1-KiB load per step; 1 GiB streamed per launch\n",
           p.gcnArchName, cus);
    // warm the clocks: ~100 ms of the MFMA loop
    for (int i = 0; i < 300; ++i)
        hipLaunchKernelGGL(kern<MFMA_ONLY>, dim3(cus), dim3(768), 0, 0, X, bytes, (int)(bytes / 1024 / ((size_t)cus * 8)), sink);
    hipDeviceSynchronize();
    run<MFMA_ONLY>(cus, X, bytes, sink);
    run<SELF_BUF_ROW>(cus, X, bytes, sink);
    run<SELF_BUF_TILE>(cus, X, bytes, sink);
    run<SELF_GLOBAL>(cus, X, bytes, sink);
    run<SELF_LDSDMA>(cus, X, bytes, sink);
    run<SELF_GATHER>(cus, X, bytes, sink);
    run<SELF_BUF_ROW_X2>(cus, X, bytes, sink);
    run<LOADER_BUF>(cus, X, bytes, sink);
    run<LOADER_LDSDMA>(cus, X, bytes, sink);
    run<LOADER_BUF_NO_MFMA>(cus, X, bytes, sink);
    run<SELF_LDSDMA_TILE>(cus, X, bytes, sink);
    run<SELF_LDSDMA_TILE_RD>(cus, X, bytes, sink);
    run<LOADER_LDSDMA_TILE>(cus, X, bytes, sink);
    run<SELF_LDSDMA_TILE_RD_1W>(cus, X, bytes, sink);
    run<MFMA_ONLY>(cus, X, bytes, sink);
    return 0;
}
