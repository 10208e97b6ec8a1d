// comm.cpp -- optional library-owned RCCL communicator for the row-stripe protocol.
// librccl is loaded lazily (dlopen) so single-GPU users never pay for it.  The only
// collective the path needs is all-reduce(sum) over small u64 histogram buffers.
#include <dlfcn.h>
#include <rccl/rccl.h> // types and enums from RCCL's own header (the library itself is dlopen'ed: no link dependency)

#include <cstring>
#include <string>

#include "internal.h"

namespace {
static_assert(sizeof(ncclUniqueId) == 128, "sarpro_hip_comm_unique_id hands out 128 bytes");

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

bool load_rccl(Rccl *r, std::string *err) {
    static Rccl cached;
    if (!cached.lib) {
        // A process that already maps an RCCL (a torch process maps torch/lib/librccl.so) must not get a second instance: two
        // RCCLs in one process each bootstrap their own transport state.  RTLD_NOLOAD returns the mapped one, by either name;
        // ncclAllReduce already resolvable in the global scope means the same.  Only then load the system library.
        void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!lib && dlsym(RTLD_DEFAULT, "ncclAllReduce")) lib = dlopen(nullptr, RTLD_NOW);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { *err = std::string("cannot load librccl: ") + dlerror(); return false; }
        Rccl t;
        t.lib = lib;
        t.GetUniqueId = reinterpret_cast<decltype(t.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        t.CommInitRank = reinterpret_cast<decltype(t.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        t.AllReduce = reinterpret_cast<decltype(t.AllReduce)>(dlsym(lib, "ncclAllReduce"));
        t.CommDestroy = reinterpret_cast<decltype(t.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!t.GetUniqueId || !t.CommInitRank || !t.AllReduce || !t.CommDestroy) {
            *err = "librccl lacks a required symbol";
            dlclose(lib);
            return false;
        }
        cached = t;
    }
    *r = cached;
    return true;
}
} // namespace

extern "C" int sarpro_hip_comm_unique_id(uint8_t uid_out[128]) {
    if (!uid_out) return SARPRO_HIP_ERR_INVALID_ARG;
    Rccl r;
    std::string err;
    if (!load_rccl(&r, &err)) return SARPRO_HIP_ERR_RCCL;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return SARPRO_HIP_ERR_RCCL;
    std::memcpy(uid_out, id.internal, 128);
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_comm_init(sarpro_hip_ctx *ctx, int nranks, int rank, const uint8_t uid[128]) {
    if (!ctx || !uid || nranks <= 0 || rank < 0 || rank >= nranks) return SARPRO_HIP_ERR_INVALID_ARG;
    if (ctx->comm) { ctx->err = "communicator already initialised"; return SARPRO_HIP_ERR_INVALID_ARG; }
    Rccl r;
    if (!load_rccl(&r, &ctx->err)) return SARPRO_HIP_ERR_RCCL;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return SARPRO_HIP_ERR_HIP; }
    ncclUniqueId id;
    std::memcpy(id.internal, uid, 128);
    ncclComm_t comm = nullptr;
    ncclResult_t rc = r.CommInitRank(&comm, nranks, id, rank);
    if (rc != ncclSuccess) {
        ctx->err = std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error");
        return SARPRO_HIP_ERR_RCCL;
    }
    ctx->comm = comm; ctx->comm_nranks = nranks; ctx->comm_rank = rank;
    return SARPRO_HIP_OK;
}

namespace sarpro {
// all-reduce(sum, u64) enqueued on the context's stream, no host synchronisation (the stripe chain keeps
// running on the stream behind it)
int comm_allreduce_sum_u64_async(sarpro_hip_ctx *ctx, uint64_t *d_buf, size_t count) {
    if (!ctx->comm) { ctx->err = "communicator not initialised"; return SARPRO_HIP_ERR_INVALID_ARG; }
    if (!count) return SARPRO_HIP_OK;
    if (!d_buf) return SARPRO_HIP_ERR_INVALID_ARG;
    Rccl r;
    if (!load_rccl(&r, &ctx->err)) return SARPRO_HIP_ERR_RCCL;
    ncclResult_t rc = r.AllReduce(d_buf, d_buf, count, ncclUint64, ncclSum, static_cast<ncclComm_t>(ctx->comm), ctx->stream);
    if (rc != ncclSuccess) {
        ctx->err = std::string("ncclAllReduce: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error");
        return SARPRO_HIP_ERR_RCCL;
    }
    return SARPRO_HIP_OK;
}
} // namespace sarpro

extern "C" int sarpro_hip_comm_allreduce_sum_u64(sarpro_hip_ctx *ctx, uint64_t *d_buf, size_t count) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    int rc = sarpro::comm_allreduce_sum_u64_async(ctx, d_buf, count);
    if (rc) return rc;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "stream sync after all-reduce failed"; return SARPRO_HIP_ERR_HIP; }
    return SARPRO_HIP_OK;
}

extern "C" void sarpro_hip_comm_destroy(sarpro_hip_ctx *ctx) {
    if (!ctx || !ctx->comm) return;
    Rccl r;
    std::string err;
    if (load_rccl(&r, &err)) r.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr;
}
