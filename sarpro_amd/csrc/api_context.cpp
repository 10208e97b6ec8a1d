// api_context.cpp -- the context of the C ABI (include/sarpro_hip.h): creation and teardown, attributes (the cross-check and tuning
// switches), per-kernel timing, the speculative chain's report.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <string>

#include "api_common.h"
#include "chain_kernels.h"
#include "context.h"
#include "internal.h"
#include "resize_kernels.h"

using namespace sarpro;

static thread_local std::string g_create_err;

// ---------------------------------------------------------------------------------------
// timing helpers (HIP events on the context's stream)
// ---------------------------------------------------------------------------------------
namespace sarpro {

TimingHold::TimingHold(sarpro_hip_ctx *c) : ctx(c) { timing_reset(c); ++c->timing_hold; }
TimingHold::~TimingHold() { --ctx->timing_hold; }

void timing_reset(sarpro_hip_ctx *ctx) {
    if (ctx->timing_hold > 0) return;
    if (ctx->async_pending && ctx->events_used < 4096) return; // calls enqueued without a synchronisation: their events are read (and dropped) together
    ctx->times.clear();
    ctx->host_times.clear();
    ctx->events_used = 0;
}

static hipEvent_t next_event(sarpro_hip_ctx *ctx) {
    if (ctx->events_used == ctx->event_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->event_pool.push_back(e);
    }
    return ctx->event_pool[ctx->events_used++];
}

KernelTimer::KernelTimer(sarpro_hip_ctx *c, const char *name, hipStream_t on) : ctx(c), stream(on ? on : c->stream) {
    if (!ctx->timing) return;
    if (!ctx->time_only.empty() && ctx->time_only != name) return;
    KernelTime t{name, next_event(ctx), next_event(ctx)};
    if (!t.start || !t.stop) return;
    (void)hipEventRecord(t.start, stream);
    ctx->times.push_back(t);
    active = true;
}
KernelTimer::~KernelTimer() {
    if (active) (void)hipEventRecord(ctx->times.back().stop, stream);
}

size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

HostTimer::HostTimer(sarpro_hip_ctx *c, const char *n) : ctx(c), name(n) {
    if (ctx->timing) t0 = std::chrono::steady_clock::now().time_since_epoch().count();
}
HostTimer::~HostTimer() {
    if (!ctx->timing) return;
    const long long t1 = std::chrono::steady_clock::now().time_since_epoch().count();
    const double ms = (double)(t1 - t0) * (double)std::chrono::steady_clock::period::num /
                      (double)std::chrono::steady_clock::period::den * 1e3;
    for (auto &h : ctx->host_times) // segments that repeat inside one call (reader / sink chunks) add up
        if (h.first == name) { h.second += (float)ms; return; }
    ctx->host_times.push_back({name, (float)ms});
}

} // namespace sarpro

// ---------------------------------------------------------------------------------------
// context attributes
// ---------------------------------------------------------------------------------------
namespace sarpro {
static const char *const kAttrNames[A_COUNT] = {
#define X(n) #n,
    SARPRO_ATTR_LIST(X)
#undef X
};
const char *attr_name(int a) { return a >= 0 && a < A_COUNT ? kAttrNames[a] : nullptr; }
int attr_index(const char *name) {
    if (!name) return -1;
    if (!strncmp(name, "SARPRO_HIP_", 11)) name += 11;
    for (int a = 0; a < A_COUNT; ++a)
        if (!strcmp(name, kAttrNames[a])) return a;
    return -1;
}
// The value of a switch as the environment (or a caller) spells it: a number; "" or any other word = 1 (the variable's presence
// used to be the switch); the two word-valued ones: SPEC_FORCE = "mispredict" (1), "nospec" (2), "lowmin" (4) in any combination, F32_ZONES = "tiny" (2).
static long long attr_parse(int a, const char *e) {
    if (a == A_SPEC_FORCE && !(e[0] >= '0' && e[0] <= '9'))
    {
        long long v = (strstr(e, "mispredict2") ? (long long)kSpecForceMispredict2 : 0) | (strstr(e, "nospec") ? (long long)kSpecForceNoSpec : 0) |
                      (strstr(e, "lowmin") ? (long long)kSpecForceMinMispredict : 0) | (strstr(e, "noretry") ? (long long)kSpecForceNoRetry : 0);
        for (const char *q = strstr(e, "mispredict"); q; q = strstr(q + 10, "mispredict"))
            if (q[10] != '2') v |= (long long)kSpecForceMispredict; // ("mispredict" on its own: one level off)
        return v;
    }
    if (a == A_F32_ZONES && !strcmp(e, "tiny")) return 2;
    char *end = nullptr;
    const long long v = strtoll(e, &end, 10);
    return (end && end != e && *end == 0) ? v : 1;
}
void attrs_from_environment(AttrSet *out) {
    for (int a = 0; a < A_COUNT; ++a) {
        const std::string var = std::string("SARPRO_HIP_") + kAttrNames[a];
        if (const char *e = getenv(var.c_str())) { out->set[a] = true; out->v[a] = attr_parse(a, e); }
    }
}
} // namespace sarpro

extern "C" int sarpro_hip_ctx_set_attr(sarpro_hip_ctx *ctx, const char *name, int64_t value) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const int a = sarpro::attr_index(name);
    if (a < 0) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "unknown context attribute");
    ctx->attrs.set[a] = true; ctx->attrs.v[a] = value;
    if (ctx->twin) { ctx->twin->attrs.set[a] = true; ctx->twin->attrs.v[a] = value; }
    for (sarpro_hip_ctx *l : ctx->lanes) { l->attrs.set[a] = true; l->attrs.v[a] = value; }
    return SARPRO_HIP_OK;
}
extern "C" int sarpro_hip_ctx_reset_attr(sarpro_hip_ctx *ctx, const char *name) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const int a = sarpro::attr_index(name);
    if (a < 0) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "unknown context attribute");
    ctx->attrs.set[a] = false; ctx->attrs.v[a] = 0;
    if (ctx->twin) { ctx->twin->attrs.set[a] = false; ctx->twin->attrs.v[a] = 0; }
    for (sarpro_hip_ctx *l : ctx->lanes) { l->attrs.set[a] = false; l->attrs.v[a] = 0; }
    return SARPRO_HIP_OK;
}
extern "C" int sarpro_hip_ctx_get_attr(const sarpro_hip_ctx *ctx, const char *name, int64_t *value, int *is_set) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const int a = sarpro::attr_index(name);
    if (a < 0) return SARPRO_HIP_ERR_INVALID_ARG;
    if (value) *value = ctx->attrs.v[a];
    if (is_set) *is_set = ctx->attrs.set[a] ? 1 : 0;
    return SARPRO_HIP_OK;
}
extern "C" const char *sarpro_hip_attr_name(int index) { return sarpro::attr_name(index); }

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
extern "C" const char *sarpro_hip_version(void) { return "sarpro-hip 0.1 (gfx950)"; }

extern "C" int sarpro_hip_ctx_create(int device, unsigned flags, sarpro_hip_ctx **out) {
    if (!out) return SARPRO_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return SARPRO_HIP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return SARPRO_HIP_ERR_INVALID_ARG; }
    if ((e = hipSetDevice(device)) != hipSuccess) { g_create_err = hipGetErrorString(e); return SARPRO_HIP_ERR_HIP; }
    sarpro_hip_ctx *ctx = new sarpro_hip_ctx();
    ctx->device = device;
    ctx->flags = flags;
    ctx->timing = (flags & SARPRO_HIP_CTX_TIMING) != 0;
    ctx->async_dev = (flags & SARPRO_HIP_CTX_ASYNC_DEV) != 0;
    sarpro::attrs_from_environment(&ctx->attrs); // the route switches' defaults: read here, never again
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_err = hipGetErrorString(e);
        delete ctx;
        return SARPRO_HIP_ERR_HIP;
    }
    (void)db_table_u16(); // build the constant dB table once, outside any timed region
    if (hipDeviceGetAttribute(&ctx->cu_count, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ctx->cu_count = 0;
    *out = ctx;
    return SARPRO_HIP_OK;
}

extern "C" void sarpro_hip_comm_destroy(sarpro_hip_ctx *ctx);

extern "C" void sarpro_hip_ctx_destroy(sarpro_hip_ctx *ctx) {
    if (!ctx) return;
    if (ctx->band_worker) { ctx->band_worker->stop(); delete ctx->band_worker; ctx->band_worker = nullptr; }
    if (ctx->twin) { sarpro_hip_ctx_destroy(ctx->twin); ctx->twin = nullptr; }
    for (sarpro_hip_ctx *l : ctx->lanes) sarpro_hip_ctx_destroy(l);
    ctx->lanes.clear();
    (void)hipSetDevice(ctx->device);
    for (hipEvent_t ev : ctx->pipe_events) (void)hipEventDestroy(ev);
    sarpro::comm_saved_release(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    sarpro_hip_comm_destroy(ctx);
    for (auto &kv : ctx->plans) {
        kv.second->release_all();
        delete kv.second;
    }
    // workspace buffers (DevBuf / PinnedBuf members) free themselves when the context is deleted, below
    for (hipEvent_t ev : ctx->event_pool) (void)hipEventDestroy(ev);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (hipEvent_t &e : ctx->ring_evt) if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char *sarpro_hip_last_error(const sarpro_hip_ctx *ctx) {
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}
extern "C" void *sarpro_hip_ctx_stream(sarpro_hip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int sarpro_hip_ctx_synchronize(sarpro_hip_ctx *ctx) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_ctx_time_only(sarpro_hip_ctx *ctx, const char *kernel_name) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    ctx->time_only = kernel_name ? kernel_name : "";
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_ctx_spec_report(sarpro_hip_ctx *ctx, sarpro_hip_spec_report *out) {
    if (!ctx || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    if (!ctx->spec_state.p) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "no speculative CLAHE chain has run on this context");
    sarpro::ChainSpecState st;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(&st, ctx->spec_state.p, sizeof(st), hipMemcpyDeviceToHost));
    out->spec_ok = st.spec_ok; out->verdict = st.verdict; out->floor_pred = st.floor_pred;
    out->n_lt[0] = st.n_lt[0]; out->n_lt[1] = st.n_lt[1]; out->target = st.target;
    out->est_lt[0] = st.est_lt[0]; out->est_lt[1] = st.est_lt[1];
    out->sample_valid[0] = st.sample_valid[0]; out->sample_valid[1] = st.sample_valid[1];
    out->pool_overflow = st.pool_overflow;
    out->n_below_min = st.n_below_min; out->min_pred[0] = st.min_pred[0]; out->min_pred[1] = st.min_pred[1];
    out->retried = st.retried; out->floor_first = st.floor_first;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_last_kernel_times(sarpro_hip_ctx *ctx, const char **names, float *ms, int max_entries) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    int n = 0;
    for (const KernelTime &t : ctx->times) {
        if (n >= max_entries) break;
        float v = 0.f;
        if (hipEventSynchronize(t.stop) != hipSuccess || hipEventElapsedTime(&v, t.start, t.stop) != hipSuccess) v = -1.f;
        if (names) names[n] = t.name;
        if (ms) ms[n] = v;
        ++n;
    }
    for (const auto &h : ctx->host_times) { // host segments (wall clock), names start with "host:"
        if (n >= max_entries) break;
        if (names) names[n] = h.first;
        if (ms) ms[n] = h.second;
        ++n;
    }
    for (const auto &h : ctx->lane_times) { // the kernels of the last resident batch, lane after lane (pipeline.cpp)
        if (n >= max_entries) break;
        if (names) names[n] = h.first;
        if (ms) ms[n] = h.second;
        ++n;
    }
    ctx->lane_times.clear();
    ctx->async_pending = false; // read: the next call starts a fresh list
    return n;
}

