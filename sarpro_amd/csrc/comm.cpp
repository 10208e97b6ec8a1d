// comm.cpp -- optional library-owned RCCL communicator for the row-stripe protocol.
// librccl is loaded lazily (dlopen) so single-GPU users never pay for it.  The only
// collective the path needs is all-reduce(sum) over small u64 histogram buffers.
#include <dlfcn.h>
#include <rccl/rccl.h> // types and enums from RCCL's own header (the library itself is dlopen'ed: no link dependency)

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

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

// ---------------------------------------------------------------------------------------
// the in-process communicator: contexts of one process, one host thread per rank
// ---------------------------------------------------------------------------------------
struct sarpro_hip_local_group {
    int nranks = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    std::vector<int> devices;           // the device of each rank that has joined (-1: not yet)
    std::vector<const uint64_t *> bufs; // the ranks' device buffers of the collective in flight
    std::vector<size_t> counts;
    bool mismatch = false;
    bool aborted = false; // a rank failed outside a collective (an argument check, an allocation, a HIP error mid-chain): nobody may wait for it
    // false: the group was aborted (before or while this rank waited) -- the collective did not complete
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        const unsigned long long g = generation;
        if (++arrived == nranks) { arrived = 0; ++generation; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != g || aborted; });
        return generation != g; // (released by the last arrival, not by the abort)
    }
    void abort() {
        { std::lock_guard<std::mutex> lk(m); aborted = true; }
        cv.notify_all();
    }
};

extern "C" int sarpro_hip_local_group_create(int nranks, sarpro_hip_local_group **out) {
    if (!out || nranks <= 0 || nranks > 64) return SARPRO_HIP_ERR_INVALID_ARG;
    sarpro_hip_local_group *g = new (std::nothrow) sarpro_hip_local_group();
    if (!g) return SARPRO_HIP_ERR_OOM;
    g->nranks = nranks;
    g->devices.assign((size_t)nranks, -1);
    g->bufs.assign((size_t)nranks, nullptr);
    g->counts.assign((size_t)nranks, 0);
    *out = g;
    return SARPRO_HIP_OK;
}
extern "C" void sarpro_hip_local_group_destroy(sarpro_hip_local_group *group) { delete group; }

extern "C" int sarpro_hip_comm_init_local(sarpro_hip_ctx *ctx, sarpro_hip_local_group *group, int rank) {
    if (!ctx || !group || rank < 0 || rank >= group->nranks) return SARPRO_HIP_ERR_INVALID_ARG;
    if (ctx->comm || ctx->local_group) { ctx->err = "communicator already initialised"; return SARPRO_HIP_ERR_INVALID_ARG; }
    {   // the sum kernel of a rank reads every other rank's buffer: ranks on different devices need peer access both ways (round 4
        // advice: a mixed-device group without it faulted inside the first collective instead of failing here)
        std::lock_guard<std::mutex> lk(group->m);
        for (int r = 0; r < group->nranks; ++r) {
            const int other = group->devices[(size_t)r];
            if (other < 0 || other == ctx->device || r == rank) continue;
            int a = 0, b = 0;
            if (hipDeviceCanAccessPeer(&a, ctx->device, other) != hipSuccess || hipDeviceCanAccessPeer(&b, other, ctx->device) != hipSuccess || !a || !b) {
                ctx->err = "in-process communicator: ranks on devices " + std::to_string(ctx->device) + " and " + std::to_string(other) + " cannot access each other's memory";
                return SARPRO_HIP_ERR_INVALID_ARG;
            }
            (void)hipSetDevice(ctx->device);
            const hipError_t e = hipDeviceEnablePeerAccess(other, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { ctx->err = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); return SARPRO_HIP_ERR_HIP; }
            (void)hipGetLastError();
        }
        group->devices[(size_t)rank] = ctx->device;
    }
    ctx->local_group = group; ctx->comm_nranks = group->nranks; ctx->comm_rank = rank;
    return SARPRO_HIP_OK;
}

namespace sarpro {
hipError_t launch_sum_rank_buffers(const uint64_t *const *d_ptrs, int nranks, uint64_t *out, size_t count, hipStream_t s); // kernels.hip

// The in-process all-reduce: synchronous (the ranks meet at two barriers), and the stream is complete when it returns.
static int local_allreduce_sum_u64(sarpro_hip_ctx *ctx, uint64_t *d_buf, size_t count) {
    sarpro_hip_local_group *g = ctx->local_group;
    // (a rank that fails still walks through both barriers: its peers must not be left waiting)
    const bool stream_ok = hipSetDevice(ctx->device) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    g->bufs[(size_t)ctx->comm_rank] = d_buf;
    g->counts[(size_t)ctx->comm_rank] = stream_ok ? count : (size_t)-1;
    if (!g->barrier()) { ctx->err = "local all-reduce: the group was aborted (another rank failed)"; return SARPRO_HIP_ERR_RCCL; } // every rank's buffer is complete and published
    bool same = true;
    for (int r = 0; r < g->nranks; ++r) same = same && g->counts[(size_t)r] == count;
    int rc = SARPRO_HIP_OK;
    if (same && count) {
        const size_t table = sizeof(uint64_t *) * (size_t)g->nranks;
        if (ctx->local_tmp.reserve(count * sizeof(uint64_t) + table) != hipSuccess) { ctx->err = "local all-reduce: out of device memory"; rc = SARPRO_HIP_ERR_OOM; }
        else {
            uint8_t *tmp = ctx->local_tmp.as<uint8_t>();
            const uint64_t **d_tab = reinterpret_cast<const uint64_t **>(tmp + count * sizeof(uint64_t));
            hipError_t e = hipMemcpyAsync(d_tab, g->bufs.data(), table, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); // (the table is host memory of the group: copied before anyone may change it)
            if (e == hipSuccess) e = launch_sum_rank_buffers(d_tab, g->nranks, reinterpret_cast<uint64_t *>(tmp), count, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { ctx->err = std::string("local all-reduce: ") + hipGetErrorString(e); rc = SARPRO_HIP_ERR_HIP; }
        }
    }
    if (!g->barrier()) { ctx->err = "local all-reduce: the group was aborted (another rank failed)"; return SARPRO_HIP_ERR_RCCL; } // every rank has read every buffer
    if (!stream_ok) { ctx->err = "local all-reduce: the stream failed"; return SARPRO_HIP_ERR_HIP; }
    if (!same) { ctx->err = "local all-reduce: a rank failed or the ranks disagree on the element count"; return SARPRO_HIP_ERR_INVALID_ARG; }
    if (rc == SARPRO_HIP_OK && count) {
        if (hipMemcpyAsync(d_buf, ctx->local_tmp.p, count * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { ctx->err = "local all-reduce: copy failed"; rc = SARPRO_HIP_ERR_HIP; }
    }
    return rc;
}

// A rank of an in-process group that fails between two collectives (the stripe entry points call this on every error return) releases
// its peers: their pending and later barriers return an error instead of waiting for ever.  The group stays aborted -- destroy it and
// create a new one.  (RCCL has no counterpart the library could use safely from one rank: a process whose rank failed must tear its
// communicator down, as with any NCCL program.)
void comm_abort_local_group(sarpro_hip_ctx *ctx) {
    if (ctx && ctx->local_group) ctx->local_group->abort();
}

// Record / replay (a measurement aid, bench.py `stripe_rank_model`).  COMM_RECORD = 1 on a rank of a real N-rank run keeps a copy of
// every all-reduce RESULT of its stripe call; COMM_REPLAY = 1 on the same context afterwards answers the all-reduces of an identical
// call from those copies (one device copy each, no communicator involved).  The context then executes, alone on its GPU, exactly the
// chain a rank of the N-rank run executed -- the scene's histograms, proofs, predictions and verdicts -- so that the per-rank cost of
// stripe mode can be timed where no second GPU exists.
void comm_replay_rewind(sarpro_hip_ctx *ctx) {
    ctx->comm_replay_pos = 0;
    if (ctx->attrs.on(A_COMM_RECORD)) { for (DevBuf *b : ctx->comm_saved) delete b; ctx->comm_saved.clear(); }
}
void comm_saved_release(sarpro_hip_ctx *ctx) { for (DevBuf *b : ctx->comm_saved) delete b; ctx->comm_saved.clear(); }

// all-reduce(sum, u64) enqueued on the context's stream, no host synchronisation (the stripe chain keeps
// running on the stream behind it)
static int comm_allreduce_sum_u64_async_raw(sarpro_hip_ctx *ctx, uint64_t *d_buf, size_t count);
int comm_allreduce_sum_u64_async(sarpro_hip_ctx *ctx, uint64_t *d_buf, size_t count) {
    const size_t bytes = count * sizeof(uint64_t);
    if (ctx->attrs.on(A_COMM_REPLAY)) {
        if (ctx->comm_replay_pos >= ctx->comm_saved.size() || ctx->comm_saved[ctx->comm_replay_pos]->cap != std::max<size_t>(bytes, 8)) {
            ctx->err = "COMM_REPLAY: this call's all-reduces differ from the recorded ones";
            return SARPRO_HIP_ERR_INVALID_ARG;
        }
        DevBuf *b = ctx->comm_saved[ctx->comm_replay_pos++];
        if (bytes && hipMemcpyAsync(d_buf, b->p, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { ctx->err = "COMM_REPLAY: copy failed"; return SARPRO_HIP_ERR_HIP; }
        return SARPRO_HIP_OK;
    }
    const int rc = comm_allreduce_sum_u64_async_raw(ctx, d_buf, count);
    if (rc == SARPRO_HIP_OK && ctx->attrs.on(A_COMM_RECORD)) {
        DevBuf *b = new DevBuf();
        if (b->reserve(std::max<size_t>(bytes, 8)) != hipSuccess || (bytes && hipMemcpyAsync(b->p, d_buf, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)) {
            delete b;
            ctx->err = "COMM_RECORD: out of device memory";
            return SARPRO_HIP_ERR_OOM;
        }
        ctx->comm_saved.push_back(b);
    }
    return rc;
}
static int comm_allreduce_sum_u64_async_raw(sarpro_hip_ctx *ctx, uint64_t *d_buf, size_t count) {
    if (ctx->local_group) return (d_buf || !count) ? local_allreduce_sum_u64(ctx, d_buf, count) : SARPRO_HIP_ERR_INVALID_ARG;
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
    if (ctx) ctx->local_group = nullptr; // (the group belongs to the caller)
    if (!ctx || !ctx->comm) return;
    Rccl r;
    std::string err;
    if (load_rccl(&r, &err)) r.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr;
}
