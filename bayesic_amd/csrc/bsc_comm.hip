// The one exchange step of the data-parallel update: an all-reduce(sum) over RCCL
// (xGMI inside a node), issued on the context's own HIP stream so the data pass, the
// collective and the finish are one in-order queue with no cross-stream hops.
//
// README.md:69-79 (mini-batch SVI) + SURVEY.md 8(e): rows are iid, every per-batch
// quantity is a sum over rows, so ranks hold row blocks and exchange ONE vector per update.
//
// librccl is bound at run time (dlopen) the first time a communicator is asked for: a
// single-GPU process -- and the torch-free binding of INTEGRATION.md section 1 -- never
// loads it, and a process that already has RCCL mapped (torch ships one) reuses that copy
// instead of mapping a second.
#include "bsc_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

namespace {

struct rccl_api {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* (*GetLastError)(ncclComm_t) = nullptr;
};

rccl_api g_rccl;

int load_rccl() {
    if (g_rccl.handle) return BSC_OK;
    void* h = nullptr;
    // a copy that is already mapped wins (same SONAME librccl.so.1 in torch/lib and /opt/rocm/lib)
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!h) return bsc_fail(BSC_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded: %s", dlerror());
    rccl_api a;
    a.handle = h;
#define BSC_SYM(field, name)                                                             \
    *(void**)(&a.field) = dlsym(h, name);                                                \
    if (!a.field) {                                                                      \
        return bsc_fail(BSC_ERR_UNSUPPORTED, "librccl has no symbol %s", name);          \
    }
    BSC_SYM(GetVersion, "ncclGetVersion")
    BSC_SYM(GetUniqueId, "ncclGetUniqueId")
    BSC_SYM(CommInitRank, "ncclCommInitRank")
    BSC_SYM(CommDestroy, "ncclCommDestroy")
    BSC_SYM(AllReduce, "ncclAllReduce")
    BSC_SYM(AllGather, "ncclAllGather")
    BSC_SYM(GetErrorString, "ncclGetErrorString")
    BSC_SYM(GetLastError, "ncclGetLastError")
#undef BSC_SYM
    g_rccl = a;
    return BSC_OK;
}

#define BSC_RCCL(ctx, call)                                                                   \
    do {                                                                                      \
        ncclResult_t r__ = (call);                                                            \
        if (r__ != ncclSuccess)                                                               \
            return bsc_fail(BSC_ERR_HIP, "%s failed: %s (%s)", #call,                         \
                            g_rccl.GetErrorString(r__),                                       \
                            g_rccl.GetLastError((ncclComm_t)(ctx)->comm));                    \
    } while (0)

}  // namespace

extern "C" {

int bsc_comm_unique_id(void* host_id) {
    BSC_REQUIRE(host_id != nullptr, "bsc_comm_unique_id: host_id is null");
    static_assert(sizeof(ncclUniqueId) == BSC_COMM_ID_BYTES, "BSC_COMM_ID_BYTES != ncclUniqueId");
    int rc = load_rccl();
    if (rc != BSC_OK) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess)
        return bsc_fail(BSC_ERR_HIP, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
    memcpy(host_id, &id, sizeof(id));
    return BSC_OK;
}

int bsc_comm_init_rank(bsc_ctx* ctx, const void* host_id, int32_t rank, int32_t world) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(host_id != nullptr, "bsc_comm_init_rank: host_id is null");
    BSC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bsc_comm_init_rank: rank %d of %d", rank,
                world);
    BSC_REQUIRE(ctx->comm == nullptr, "bsc_comm_init_rank: the context already has a communicator");
    int rc = load_rccl();
    if (rc != BSC_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, host_id, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess)
        return bsc_fail(BSC_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank,
                        world, ctx->device, g_rccl.GetErrorString(r));
    ctx->comm = (void*)comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return BSC_OK;
}

int bsc_comm_destroy(bsc_ctx* ctx) {
    if (!ctx) return BSC_OK;
    if (ctx->comm_stream) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->comm_stream);
        for (int i = 0; i < BSC_EXCHANGE_SLOTS; ++i) {
            if (ctx->xch_ready[i]) (void)hipEventDestroy(ctx->xch_ready[i]);
            if (ctx->xch_done[i]) (void)hipEventDestroy(ctx->xch_done[i]);
            ctx->xch_ready[i] = ctx->xch_done[i] = nullptr;
            ctx->xch_pending[i] = 0;
        }
        (void)hipStreamDestroy(ctx->comm_stream);
        ctx->comm_stream = nullptr;
    }
    if (!ctx->comm) return BSC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    if (r != ncclSuccess)
        return bsc_fail(BSC_ERR_HIP, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
    return BSC_OK;
}

int bsc_comm_info(bsc_ctx* ctx, int32_t* host_rank, int32_t* host_world, int32_t* host_rccl_version) {
    BSC_REQUIRE(ctx != nullptr, "null bsc_ctx");
    if (host_rank) *host_rank = ctx->comm_rank;
    if (host_world) *host_world = ctx->comm ? ctx->comm_world : 1;
    if (host_rccl_version) {
        int v = 0;
        if (g_rccl.handle) (void)g_rccl.GetVersion(&v);
        *host_rccl_version = v;
    }
    return BSC_OK;
}

int bsc_allreduce_sum(bsc_ctx* ctx, void* buf, int64_t n, int dtype) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(dtype == BSC_F32 || dtype == BSC_F64, "bsc_allreduce_sum: dtype %d", dtype);
    BSC_REQUIRE(n >= 0 && (buf != nullptr || n == 0), "bsc_allreduce_sum: null buffer");
    if (!ctx->comm) {
        // no communicator = a world of one: the sum over ranks is the buffer itself
        return BSC_OK;
    }
    if (n == 0) return BSC_OK;
    bsc_prof_scope prof(ctx, /*slot=*/1);
    BSC_RCCL(ctx, g_rccl.AllReduce(buf, buf, (size_t)n, dtype == BSC_F64 ? ncclFloat64 : ncclFloat32,
                                   ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return BSC_OK;
}

int bsc_allreduce_sum_begin(bsc_ctx* ctx, void* buf, int64_t n, int dtype, int32_t slot) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(dtype == BSC_F32 || dtype == BSC_F64, "bsc_allreduce_sum_begin: dtype %d", dtype);
    BSC_REQUIRE(n >= 0 && (buf != nullptr || n == 0), "bsc_allreduce_sum_begin: null buffer");
    BSC_REQUIRE(slot >= 0 && slot < BSC_EXCHANGE_SLOTS, "bsc_allreduce_sum_begin: slot %d (0 .. %d)", slot,
                BSC_EXCHANGE_SLOTS - 1);
    BSC_REQUIRE(!ctx->xch_pending[slot], "bsc_allreduce_sum_begin: slot %d has a collective that was not ended", slot);
    if (!ctx->comm || n == 0) return BSC_OK;            // a world of one: nothing to wait for either
    if (ctx->capturing) {
        // inside a graph capture everything stays on the captured stream (no overlap, same result)
        bsc_prof_scope prof(ctx, /*slot=*/1);
        BSC_RCCL(ctx, g_rccl.AllReduce(buf, buf, (size_t)n, dtype == BSC_F64 ? ncclFloat64 : ncclFloat32, ncclSum,
                                       (ncclComm_t)ctx->comm, ctx->stream));
        return BSC_OK;
    }
    if (!ctx->comm_stream) BSC_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    if (!ctx->xch_ready[slot]) {
        BSC_HIP(hipEventCreateWithFlags(&ctx->xch_ready[slot], hipEventDisableTiming));
        BSC_HIP(hipEventCreateWithFlags(&ctx->xch_done[slot], hipEventDisableTiming));
    }
    // everything enqueued on the context's stream so far (the kernels that produced `buf`) precedes the collective ...
    BSC_HIP(hipEventRecord(ctx->xch_ready[slot], ctx->stream));
    BSC_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->xch_ready[slot], 0));
    {
        bsc_prof_scope prof(ctx, /*slot=*/1, ctx->comm_stream);
        BSC_RCCL(ctx, g_rccl.AllReduce(buf, buf, (size_t)n, dtype == BSC_F64 ? ncclFloat64 : ncclFloat32, ncclSum,
                                       (ncclComm_t)ctx->comm, ctx->comm_stream));
    }
    // ... and bsc_allreduce_sum_end makes the context's stream wait for it
    BSC_HIP(hipEventRecord(ctx->xch_done[slot], ctx->comm_stream));
    ctx->xch_pending[slot] = 1;
    return BSC_OK;
}

int bsc_allreduce_sum_end(bsc_ctx* ctx, int32_t slot) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(slot >= 0 && slot < BSC_EXCHANGE_SLOTS, "bsc_allreduce_sum_end: slot %d (0 .. %d)", slot,
                BSC_EXCHANGE_SLOTS - 1);
    if (!ctx->xch_pending[slot]) return BSC_OK;          // a world of one, an empty buffer, or a capture: already in order
    BSC_HIP(hipStreamWaitEvent(ctx->stream, ctx->xch_done[slot], 0));
    ctx->xch_pending[slot] = 0;
    return BSC_OK;
}

int bsc_allreduce_max(bsc_ctx* ctx, void* buf, int64_t n, int dtype) {
    BSC_CHECK_CTX(ctx);
    BSC_REQUIRE(dtype == BSC_F32 || dtype == BSC_F64, "bsc_allreduce_max: dtype %d", dtype);
    BSC_REQUIRE(n >= 0 && (buf != nullptr || n == 0), "bsc_allreduce_max: null buffer");
    if (!ctx->comm || n == 0) return BSC_OK;
    BSC_RCCL(ctx, g_rccl.AllReduce(buf, buf, (size_t)n, dtype == BSC_F64 ? ncclFloat64 : ncclFloat32,
                                   ncclMax, (ncclComm_t)ctx->comm, ctx->stream));
    return BSC_OK;
}

}  // extern "C"
