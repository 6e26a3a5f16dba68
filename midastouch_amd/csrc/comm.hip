// comm.hip - a library-owned RCCL communicator for the particle-sharded frame.
//
// The reference has no distributed code; BASELINE.json's north_star asks for the particle shards of one
// node to exchange their per-shard sums "with an RCCL all-reduce ... over xGMI".  Round 2 issued those collectives from Python
// through torch.distributed - nine host-level calls per frame, which bound the sharded frame.  Here the
// library holds its own ncclComm_t, so the whole frame - kernels AND the record all_gather - is enqueued on the context's
// stream by one C call (midas_shard_step, api.hip).
//
// librccl is opened at run time (dlopen) rather than linked: the process normally has torch's copy loaded already, and two
// RCCL instances in one process would each set up their own transports.  The caller passes the path of the copy to use
// (midastouch_amd/dist.py passes torch's); NULL tries the loader's default "librccl.so.1" / "librccl.so".
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "midas_internal.hpp"

struct midas_comm {
    midas_ctx* ctx = nullptr;
    void* lib = nullptr;
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
};

namespace {
void* open_rccl(midas_ctx* ctx, const char* path) {
    const char* names[] = {path, "librccl.so.1", "librccl.so"};
    for (int i = path ? 0 : 1; i < 3; ++i) {
        if (!names[i]) continue;
        if (void* h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL)) return h;
        if (path) break;  // an explicit path that does not load is an error, not a reason to take another copy
    }
    midas_set_error(ctx, MIDAS_ERR_INVALID, "dlopen(librccl)", dlerror());
    return nullptr;
}
template <class F>
bool sym(void* lib, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(lib, name));
    return out != nullptr;
}
}  // namespace

extern "C" {

#define MIDAS_EXPORT __attribute__((visibility("default")))

MIDAS_EXPORT int midas_comm_unique_id(midas_ctx* ctx, const char* rccl_path, void* id128_out) {
    if (!ctx || !id128_out) return MIDAS_ERR_INVALID;
    void* lib = open_rccl(ctx, rccl_path);
    if (!lib) return MIDAS_ERR_INVALID;
    ncclResult_t (*get_id)(ncclUniqueId*) = nullptr;
    if (!sym(lib, "ncclGetUniqueId", get_id)) return midas_set_error(ctx, MIDAS_ERR_INVALID, "dlsym", "ncclGetUniqueId");
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
    const ncclResult_t r = get_id(&id);
    if (r != ncclSuccess) return midas_set_error(ctx, MIDAS_ERR_HIP, "ncclGetUniqueId", "failed");
    memcpy(id128_out, &id, 128);
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_comm_create(midas_ctx* ctx, const char* rccl_path, const void* id128, int32_t world, int32_t rank, midas_comm** out) {
    if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return MIDAS_ERR_INVALID;
    MIDAS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    void* lib = open_rccl(ctx, rccl_path);
    if (!lib) return MIDAS_ERR_INVALID;
    midas_comm* c = new midas_comm();
    c->ctx = ctx; c->lib = lib; c->world = world; c->rank = rank;
    ncclResult_t (*init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    if (!sym(lib, "ncclCommInitRank", init_rank) || !sym(lib, "ncclAllGather", c->all_gather) ||
        !sym(lib, "ncclCommDestroy", c->comm_destroy) || !sym(lib, "ncclGetErrorString", c->error_string)) {
        delete c;
        return midas_set_error(ctx, MIDAS_ERR_INVALID, "dlsym", "librccl lacks ncclCommInitRank / ncclAllGather / ncclCommDestroy");
    }
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    const ncclResult_t r = init_rank(&c->comm, world, id, rank);  // collective: every rank calls it with the same id
    if (r != ncclSuccess) {
        const int rc = midas_set_error(ctx, MIDAS_ERR_HIP, "ncclCommInitRank", c->error_string(r));
        delete c;
        return rc;
    }
    *out = c;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_comm_destroy(midas_comm* c) {
    if (!c) return MIDAS_OK;
    if (c->comm && c->comm_destroy) (void)c->comm_destroy(c->comm);
    delete c;
    return MIDAS_OK;
}

// every rank contributes `bytes` bytes (a multiple of 8) from send_dev; recv_dev gets world x bytes in rank order; on the context's stream
MIDAS_EXPORT int midas_comm_all_gather(midas_comm* c, const void* send_dev, void* recv_dev, int64_t bytes) {
    if (!c || !send_dev || !recv_dev || bytes <= 0 || bytes % 8) return MIDAS_ERR_INVALID;
    const ncclResult_t r = c->all_gather(send_dev, recv_dev, (size_t)(bytes / 8), ncclFloat64, c->comm, c->ctx->stream);
    if (r != ncclSuccess) return midas_set_error(c->ctx, MIDAS_ERR_HIP, "ncclAllGather", c->error_string(r));
    return MIDAS_OK;
}

}  // extern "C"
