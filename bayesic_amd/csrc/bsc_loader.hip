// Mini-batch streaming: host memory -> HBM slots on a copy stream of its own, so
// that batch t+1 crosses PCIe while the update of batch t runs on the context's
// stream (README.md:69-79: "stochastic updates applied via subsampled
// minibatches"; SURVEY.md 8(f) rank 4: the step before the path).  The update
// kernels only ever see device-resident batches -- the hot path is unchanged.
//
// Slot life cycle:  submit (H2D queued on the copy stream, ordered after the
// kernels that last read the slot)  ->  acquire (compute stream waits for the
// copy)  ->  release (marks the point on the compute stream after which the slot
// may be overwritten).  The device only ever reads host memory that the HIP runtime allocated page-locked
// (bsc_host_alloc; torch pin_memory tensors) -- never pages that malloc owns: a pageable source is copied by the
// host into the slot's page-locked bounce buffer inside submit.  (Round 2 registered numpy heap arrays in place
// and let the runtime lock pageable sources on the fly; a later, unrelated blocking H2D then faulted on a heap
// address.  Both ways of showing malloc's pages to the device are gone: DESIGN.md section 10.)
#include "bsc_common.h"

#include <cstring>
#include <vector>

struct bsc_loader {
    bsc_ctx* ctx = nullptr;
    hipStream_t copy_stream = nullptr;
    int64_t max_rows = 0;
    int32_t D = 0;
    struct Slot {
        float* X = nullptr;
        float* y = nullptr;
        int64_t rows = 0;
        hipEvent_t copied = nullptr;     // recorded on the copy stream after the H2D
        hipEvent_t consumed = nullptr;   // recorded on the compute stream at release
        bool has_consumed = false;
        // page-locked bounce buffer of this slot for PAGEABLE sources (made on first use).  The runtime
        // would otherwise lock the caller's pages on the fly and let the copy engine read them; a
        // pageable source is instead copied here by the host inside bsc_loader_submit, so the device
        // only ever reads memory this library allocated (or the caller registered), and a pageable
        // source is free again as soon as submit returns.
        float* bounce = nullptr;
    };
    std::vector<Slot> slots;
    int64_t submitted = 0, acquired = 0, released = 0;   // monotone counters; slot = counter % n
};

// page-locked and known to the runtime (hipHostMalloc'ed, or hipHostRegister'ed by the caller)?
static bool loader_is_page_locked(const void* p) {
    hipPointerAttribute_t attr;
    const hipError_t err = hipPointerGetAttributes(&attr, p);
    if (err != hipSuccess) {
        (void)hipGetLastError();          // an ordinary malloc'ed pointer: "invalid value", not an error here
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

extern "C" {

/* page-locked host memory owned by the runtime (hipHostMalloc): the source to stream batches from at the
 * full PCIe rate without registering memory that malloc owns */
int bsc_host_alloc(size_t bytes, void** out) {
    BSC_REQUIRE(out != nullptr && bytes > 0, "bsc_host_alloc: bad arguments");
    BSC_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return BSC_OK;
}

int bsc_host_free(void* host_ptr) {
    if (!host_ptr) return BSC_OK;
    // a copy out of this block may still be queued on some loader's copy stream: nothing may read it after
    // this call returns, so the device is drained first (by construction, not by the caller's discipline)
    BSC_HIP(hipDeviceSynchronize());
    BSC_HIP(hipHostFree(host_ptr));
    return BSC_OK;
}

int bsc_loader_create(bsc_ctx* ctx, int64_t max_rows, int32_t D, int32_t n_slots, bsc_loader** out) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(out != nullptr, "bsc_loader_create: out is null");
    BSC_REQUIRE(max_rows > 0 && D > 0 && n_slots >= 2 && n_slots <= 16,
                "bsc_loader_create: need max_rows > 0, D > 0, 2 <= n_slots <= 16");
    bsc_loader* L = new bsc_loader();
    L->ctx = ctx;
    L->max_rows = max_rows;
    L->D = D;
    L->slots.resize(n_slots);
    hipError_t err = hipStreamCreateWithFlags(&L->copy_stream, hipStreamNonBlocking);
    for (auto& s : L->slots) {
        if (err == hipSuccess) err = hipMalloc((void**)&s.X, (size_t)max_rows * D * sizeof(float));
        if (err == hipSuccess) err = hipMalloc((void**)&s.y, (size_t)max_rows * sizeof(float));
        if (err == hipSuccess) err = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming);
    }
    if (err != hipSuccess) {
        for (auto& s : L->slots) {
            if (s.X) (void)hipFree(s.X);
            if (s.y) (void)hipFree(s.y);
            if (s.copied) (void)hipEventDestroy(s.copied);
            if (s.consumed) (void)hipEventDestroy(s.consumed);
        }
        if (L->copy_stream) (void)hipStreamDestroy(L->copy_stream);
        delete L;
        return bsc_fail(BSC_ERR_HIP, "bsc_loader_create: %s", hipGetErrorString(err));
    }
    *out = L;
    return BSC_OK;
}

int bsc_loader_destroy(bsc_loader* L) {
    if (!L) return BSC_OK;
    (void)hipStreamSynchronize(L->copy_stream);
    (void)hipStreamSynchronize(L->ctx->stream);
    for (auto& s : L->slots) {
        (void)hipFree(s.X);
        (void)hipFree(s.y);
        if (s.bounce) (void)hipHostFree(s.bounce);
        (void)hipEventDestroy(s.copied);
        (void)hipEventDestroy(s.consumed);
    }
    (void)hipStreamDestroy(L->copy_stream);
    delete L;
    return BSC_OK;
}

int bsc_loader_submit(bsc_loader* L, const float* host_X, int64_t ldx, const float* host_y,
                      int64_t rows) {
    BSC_REQUIRE(L != nullptr, "bsc_loader_submit: loader is null");
    BSC_REQUIRE(host_X && host_y && rows > 0 && rows <= L->max_rows && ldx >= L->D,
                "bsc_loader_submit: bad batch (rows=%lld, max_rows=%lld, ldx=%lld, D=%d)",
                (long long)rows, (long long)L->max_rows, (long long)ldx, L->D);
    const int64_t n = (int64_t)L->slots.size();
    BSC_REQUIRE(L->submitted - L->released < n,
                "bsc_loader_submit: all %lld slots are in flight (release one first)", (long long)n);
    auto& s = L->slots[L->submitted % n];
    // The HOST may not run further ahead than the slots: the copy that last targeted this slot
    // (submission number submitted - n) must have left its host buffer before the caller is told,
    // by this call returning, that it may reuse or free that buffer.  acquire/release only enqueue
    // stream waits, so without this the caller could drop a pinned source whose copy has not started.
    if (L->submitted >= n) BSC_HIP(hipEventSynchronize(s.copied));
    // the copy may not overtake the kernels that last read this slot
    if (s.has_consumed) BSC_HIP(hipStreamWaitEvent(L->copy_stream, s.consumed, 0));
    if (!loader_is_page_locked(host_X) || !loader_is_page_locked(host_y)) {
        // pageable: through the slot's bounce buffer ([rows x D] dense, then y).  The previous copy out
        // of this buffer has completed: it is the one `copied` was just waited for above (or none yet).
        if (!s.bounce)
            BSC_HIP(hipHostMalloc((void**)&s.bounce, (size_t)L->max_rows * (L->D + 1) * sizeof(float), hipHostMallocDefault));
        for (int64_t r = 0; r < rows; ++r)
            std::memcpy(s.bounce + r * L->D, host_X + r * ldx, (size_t)L->D * sizeof(float));
        std::memcpy(s.bounce + rows * L->D, host_y, (size_t)rows * sizeof(float));
        host_X = s.bounce;
        host_y = s.bounce + rows * L->D;
        ldx = L->D;
    }
    if (ldx == L->D)
        BSC_HIP(hipMemcpyAsync(s.X, host_X, (size_t)rows * L->D * sizeof(float), hipMemcpyHostToDevice,
                               L->copy_stream));
    else
        BSC_HIP(hipMemcpy2DAsync(s.X, (size_t)L->D * sizeof(float), host_X, (size_t)ldx * sizeof(float),
                                 (size_t)L->D * sizeof(float), (size_t)rows, hipMemcpyHostToDevice,
                                 L->copy_stream));
    BSC_HIP(hipMemcpyAsync(s.y, host_y, (size_t)rows * sizeof(float), hipMemcpyHostToDevice,
                           L->copy_stream));
    BSC_HIP(hipEventRecord(s.copied, L->copy_stream));
    s.rows = rows;
    L->submitted += 1;
    return BSC_OK;
}

int bsc_loader_acquire(bsc_loader* L, const float** dX, const float** dy, int64_t* rows) {
    BSC_REQUIRE(L != nullptr && dX && dy && rows, "bsc_loader_acquire: null argument");
    BSC_REQUIRE(L->acquired < L->submitted, "bsc_loader_acquire: no submitted batch is waiting");
    BSC_REQUIRE(L->acquired == L->released,
                "bsc_loader_acquire: release the batch acquired before taking the next one");
    auto& s = L->slots[L->acquired % (int64_t)L->slots.size()];
    BSC_HIP(hipStreamWaitEvent(L->ctx->stream, s.copied, 0));
    *dX = s.X;
    *dy = s.y;
    *rows = s.rows;
    L->acquired += 1;
    return BSC_OK;
}

int bsc_loader_release(bsc_loader* L) {
    BSC_REQUIRE(L != nullptr, "bsc_loader_release: loader is null");
    BSC_REQUIRE(L->released < L->acquired, "bsc_loader_release: nothing is acquired");
    auto& s = L->slots[L->released % (int64_t)L->slots.size()];
    BSC_HIP(hipEventRecord(s.consumed, L->ctx->stream));
    s.has_consumed = true;
    L->released += 1;
    return BSC_OK;
}

}  // extern "C"
