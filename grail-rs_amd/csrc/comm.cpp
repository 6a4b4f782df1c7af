// comm.cpp — RCCL: the one collective of the path (a broadcast of the voice table, §8e), loaded on first use.
#include "api_internal.hpp"

using namespace grail;
using namespace grail::host;

namespace {

// RCCL is loaded on first use so the library loads (and its symbols can be
// checked) on hosts without a GPU stack that can initialise RCCL.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        // The process environment is the host application's: nothing is set here.  A single-node
        // launcher that wants RCCL to skip the InfiniBand / interface probing (up to 2 minutes on a
        // box without a network) exports NCCL_IB_DISABLE=1 NCCL_SOCKET_IFNAME=lo itself, as bench.py
        // and the tests do.
        // The ROCm installation's RCCL by absolute path first: a bare "librccl.so" would be
        // satisfied by any copy the host process already holds (PyTorch wheels bundle one that
        // is bound to their own private HIP runtime, not to the one this library links).
        const char *env = getenv("GRAIL_RCCL_PATH");
        if (env && *env) x.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) x.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) x.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) x.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) return x;
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.handle, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.handle, "ncclCommInitRank");
        x.CommInitAll = (decltype(x.CommInitAll))dlsym(x.handle, "ncclCommInitAll");
        x.Broadcast = (decltype(x.Broadcast))dlsym(x.handle, "ncclBroadcast");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.handle, "ncclCommDestroy");
        x.CommCount = (decltype(x.CommCount))dlsym(x.handle, "ncclCommCount");
        x.CommUserRank = (decltype(x.CommUserRank))dlsym(x.handle, "ncclCommUserRank");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.handle, "ncclGetErrorString");
        x.ok = x.GetUniqueId && x.CommInitRank && x.Broadcast && x.CommDestroy;
        return x;
    }();
    return r;
}

int rccl_fail(ncclResult_t r, const char *what)
{
    const char *s = rccl().GetErrorString ? rccl().GetErrorString(r) : "?";
    return fail(GRAIL_ERR_RCCL, std::string(what) + ": " + s);
}

}  // namespace

namespace grail {
namespace host {

void comm_release(grail_ctx *ctx)
{
    if (ctx->comm && rccl().ok) rccl().CommDestroy(ctx->comm);
    ctx->comm = nullptr;
}

// One process, n devices (grail_node): ncclCommInitAll forms the n communicators at once from the calling thread; context i
// becomes rank i.  RCCL refuses a device list that names a GPU twice (ncclInvalidUsage) — the caller reports that.
int comm_init_all(grail_ctx *const *ctxs, const int *devices, uint32_t n)
{
    if (!ctxs || !devices || n == 0) return fail(GRAIL_ERR_INVALID_ARG, "comm_init_all: no contexts");
    if (!rccl().ok || !rccl().CommInitAll) return fail(GRAIL_ERR_RCCL, "librccl.so could not be loaded (or lacks ncclCommInitAll)");
    // (checked here rather than left to the library: an RCCL that did not check would meet itself on one GPU and hang)
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < i; ++j)
            if (devices[i] == devices[j])
                return fail(GRAIL_ERR_RCCL, "RCCL cannot form a communicator that names a GPU twice (device " +
                                                std::to_string(devices[i]) + " holds slots " + std::to_string(j) + " and " +
                                                std::to_string(i) + ")");
    for (uint32_t i = 0; i < n; ++i) comm_release(ctxs[i]);
    std::vector<ncclComm_t> comms(n, nullptr);
    ncclResult_t r = rccl().CommInitAll(comms.data(), (int)n, devices);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitAll");
    for (uint32_t i = 0; i < n; ++i) {
        ctxs[i]->comm = comms[i];
        ctxs[i]->comm_rank = i;
        ctxs[i]->comm_world = n;
    }
    return GRAIL_OK;
}

}  // namespace host
}  // namespace grail

extern "C" {

int grail_comm_unique_id(uint8_t id[GRAIL_UNIQUE_ID_BYTES])
{
    if (!id) return fail(GRAIL_ERR_INVALID_ARG, "id is NULL");
    if (!rccl().ok) return fail(GRAIL_ERR_RCCL, "librccl.so could not be loaded");
    ncclUniqueId uid;
    ncclResult_t r = rccl().GetUniqueId(&uid);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    std::memcpy(id, uid.internal, GRAIL_UNIQUE_ID_BYTES);
    return GRAIL_OK;
}

int grail_comm_init(grail_ctx *ctx, const uint8_t id[GRAIL_UNIQUE_ID_BYTES], uint32_t rank,
                    uint32_t world)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!id || world == 0 || rank >= world) return fail(GRAIL_ERR_INVALID_ARG, "bad rank/world/id");
    if (!rccl().ok) return fail(GRAIL_ERR_RCCL, "librccl.so could not be loaded");
    if (ctx->comm) {
        rccl().CommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, GRAIL_UNIQUE_ID_BYTES);
    ncclResult_t r = rccl().CommInitRank(&ctx->comm, (int)world, uid, (int)rank);
    if (r != ncclSuccess) {
        ctx->comm = nullptr;
        return rccl_fail(r, "ncclCommInitRank");
    }
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return GRAIL_OK;
}

int grail_broadcast_voices(grail_ctx *ctx, uint32_t n_voices, uint32_t root)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!ctx->comm) return fail(GRAIL_ERR_RCCL, "call grail_comm_init first");
    if (n_voices == 0 || root >= ctx->comm_world) return fail(GRAIL_ERR_INVALID_ARG, "bad n_voices/root");
    if (ctx->comm_rank == root && ctx->voices.size() != n_voices)
        return fail(GRAIL_ERR_INVALID_ARG, "root's voice table does not hold n_voices voices");
    const size_t bytes = (size_t)n_voices * sizeof(grail_voice);
    void *d_blob = nullptr;
    HIP_TRY(hipMalloc(&d_blob, bytes));
    hipError_t e = hipSuccess;
    if (ctx->comm_rank == root)
        e = hipMemcpyAsync(d_blob, ctx->voices.data(), bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(d_blob);
        return hip_fail(e, "voice blob upload");
    }
    // one ncclBroadcast over xGMI: root's HBM -> every rank's HBM
    ncclResult_t r = rccl().Broadcast(d_blob, d_blob, bytes, ncclUint8, (int)root, ctx->comm, ctx->stream);
    if (r != ncclSuccess) {
        (void)hipFree(d_blob);
        return rccl_fail(r, "ncclBroadcast");
    }
    std::vector<grail_voice> got(n_voices);
    e = hipMemcpyAsync(got.data(), d_blob, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_blob);
    if (e != hipSuccess) return hip_fail(e, "voice blob download");
    if (ctx->comm_rank == root) return GRAIL_OK;  // already installed
    return install_voices(ctx, got.data(), n_voices);
}

int grail_comm_info(grail_ctx *ctx, uint32_t *ranks, uint32_t *rank)
{
    if (!ctx) return fail(GRAIL_ERR_INVALID_ARG, "ctx is NULL");
    if (ranks) *ranks = 0;
    if (rank) *rank = 0;
    if (!ctx->comm) return GRAIL_OK;              // no communicator: 0 ranks
    if (!rccl().ok || !rccl().CommCount || !rccl().CommUserRank)
        return fail(GRAIL_ERR_RCCL, "librccl.so lacks ncclCommCount / ncclCommUserRank");
    int n = 0, r = 0;
    ncclResult_t e = rccl().CommCount(ctx->comm, &n);
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommCount");
    e = rccl().CommUserRank(ctx->comm, &r);
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommUserRank");
    if (ranks) *ranks = (uint32_t)n;
    if (rank) *rank = (uint32_t)r;
    return GRAIL_OK;
}

int grail_comm_destroy(grail_ctx *ctx)
{
    if (!ctx) return fail(GRAIL_ERR_INVALID_ARG, "ctx is NULL");
    if (ctx->comm && rccl().ok) rccl().CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    return GRAIL_OK;
}

}  // extern "C"
