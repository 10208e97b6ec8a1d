// api.cpp -- C ABI of libsarpro_hip.so: context, work planning and the pass orchestration of
// the u16 (integer-DN) flavour.  See include/sarpro_hip.h for the reference functions each
// entry point replaces.  No CPU fallback exists anywhere in this file: every raster result is
// produced by the kernels in kernels.hip.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <string>

#include "chain_kernels.h"
#include "context.h"
#include "internal.h"
#include "resize_kernels.h"

using namespace sarpro;

static thread_local std::string g_create_err;

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                      \
            return e__ == hipErrorOutOfMemory ? SARPRO_HIP_ERR_OOM : SARPRO_HIP_ERR_HIP;          \
        }                                                                                         \
    } while (0)

#define RETCHK(expr)                                   \
    do {                                               \
        int rc__ = (expr);                             \
        if (rc__ != SARPRO_HIP_OK) return rc__;        \
    } while (0)

static int fail(sarpro_hip_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}

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

// ---------------------------------------------------------------------------------------
// planning: split a stripe into work items
// ---------------------------------------------------------------------------------------
namespace sarpro {

// Column strips of the vector kernels start on a multiple of this many pixels (read when a plan is built): with 64, the 1-KiB
// row segment a wave reads (64 lanes x 8 u16) is 128-byte aligned and covers 8 memory lines instead of 9, the 512 bytes of
// levels it writes cover 4.5 instead of 5.  Cells and tiles start at multiples of tile_w / 2 (1250 px on the headline scene),
// so with the vector width alone (8) nearly every segment straddled a line at both ends: +13 % of HBM traffic on the apply
// pass by the PMC counters (profiles/r2_traffic.json).  The leading lanes of a cell's first strip are masked instead.
constexpr size_t kRgbItemRows = 256, kSampleItemRows = 1024, kU16ItemRows = 1024;
constexpr size_t kRgbTailRows = 1024, kRgbTailItemRows = 96; // fused CLAHE -> RGB pass: the stripe's last rows in small items (see get_plan)
constexpr size_t kRgbItemRowsLarge = 512, kRgbTailRowsLarge = 2500, kRgbTailItemRowsLarge = 128;
static size_t strip_align(const StripePlan &P, int vecw) { // (STRIP_ALIGN: planner tuning, read when the plan is built)
    if (vecw != 8 && vecw != 4) return (size_t)vecw;
    return std::max<size_t>(vecw, P.strip_align_px / vecw * vecw);
}

static void push_strips(std::vector<Rect> &out, const StripePlan &P, size_t lo, size_t hi, size_t c0, size_t c1,
                        const int ids[4], size_t chunk_rows, int vecw, int flags = 0) {
    if (lo >= hi || c0 >= c1) return;
    const size_t strip = 64 * (size_t)vecw, align = strip_align(P, vecw);
    for (size_t cs = c0 / align * align; cs < c1; cs += strip) {
        Rect r{};
        r.c0 = (int32_t)std::max(c0, cs);
        r.c1 = (int32_t)std::min(c1, cs + strip);
        r.cstart = (int32_t)cs;
        for (int k = 0; k < 4; ++k) r.id[k] = ids[k];
        r.pad[0] = flags;
        for (size_t rr = lo; rr < hi; rr += chunk_rows) {
            r.r0 = (int32_t)(rr - P.row0);
            r.r1 = (int32_t)(std::min(rr + chunk_rows, hi) - P.row0);
            out.push_back(r);
        }
    }
}

// Global rows [gr0, gr1) x columns [c0, c1) clipped to the local stripe -> work items.  With
// `sliver` given (vecw == 8) the aligned interior goes to `out`, the edge leftovers to `sliver`.
static void add_rects(std::vector<Rect> &out, std::vector<Rect> *sliver, const StripePlan &P, size_t gr0, size_t gr1,
                      size_t c0, size_t c1, const int ids[4], size_t chunk_rows, int vecw, int flags = 0) {
    const size_t lo = std::max(gr0, P.row0), hi = std::min(gr1, P.row0 + P.rows_local);
    if (lo >= hi || c0 >= c1) return;
    if (!sliver) { push_strips(out, P, lo, hi, c0, c1, ids, chunk_rows, vecw, flags); return; }
    const size_t a = (c0 + vecw - 1) / vecw * vecw, b = c1 / vecw * vecw; // aligned interior [a, b)
    if (a < b) {
        push_strips(out, P, lo, hi, a, b, ids, chunk_rows, vecw);
        push_strips(*sliver, P, lo, hi, c0, a, ids, chunk_rows * 4, 1);
        push_strips(*sliver, P, lo, hi, b, c1, ids, chunk_rows * 4, 1);
    } else {
        push_strips(*sliver, P, lo, hi, c0, c1, ids, chunk_rows * 4, 1);
    }
}

static int upload_vec(sarpro_hip_ctx *ctx, DevBuf &d, const void *src, size_t bytes) {
    if (!bytes) return SARPRO_HIP_OK;
    HIPCHK(ctx, d.reserve(bytes));
    HIPCHK(ctx, hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // src may be a temporary
    return SARPRO_HIP_OK;
}

// Pieces of a whole scene for persistent workgroups (piece_kernels.hip): strips of 1, 2, 4, 8 or 16 wave columns (256 px each)
// inside one interpolation cell, cut into row ranges so that each of the `grid` workgroups gets the same cost (rows x wave
// columns, the ragged last column included).  Cell-major order: a workgroup's pieces are neighbours.
static void build_pieces(StripePlan *P, int grid) {
    const ClaheGeometry &g = P->geom;
    struct Strip { PieceItem it; double cost_per_row; };
    std::vector<Strip> strips;
    double total = 0.0;
    // cells = ranges of constant (t0, t1), cut again where the weight changes sign -- the first half tile extrapolates (d < 0,
    // autoscale.rs:308-313) and takes the wider speculation margin, the half tile after it has the same tiles but d >= 0 --
    // and at the tile boundaries (a cell is offset by half a tile: the sample pass's strata are per tile)
    auto cuts = [](const std::vector<size_t> &starts, const std::vector<RowWeight> &w, size_t tile) {
        std::vector<size_t> out;
        for (size_t i = 0; i + 1 < starts.size(); ++i) {
            out.push_back(starts[i]);
            for (size_t k = starts[i] + 1; k < starts[i + 1]; ++k)
                if ((w[k - 1].d < 0.0) != (w[k].d < 0.0) || k % tile == 0) out.push_back(k);
        }
        out.push_back(starts.empty() ? 0 : starts.back());
        return out;
    };
    const std::vector<size_t> rcut = cuts(g.row_cell_start, g.row_w, g.tile_h), ccut = cuts(g.col_cell_start, g.col_w, g.tile_w);
    for (size_t ri = 0; ri + 1 < rcut.size(); ++ri) {
        const size_t r0 = rcut[ri], r1 = rcut[ri + 1];
        if (r0 >= r1) continue;
        const RowWeight &rw = g.row_w[r0];
        for (size_t ci = 0; ci + 1 < ccut.size(); ++ci) {
            const size_t c0 = ccut[ci], c1 = ccut[ci + 1];
            if (c0 >= c1) continue;
            const RowWeight &cw = g.col_w[c0];
            const bool neg = rw.d < 0.0 || cw.d < 0.0; // constant sign inside the cut cell
            const size_t palign = P->piece_align; // (pieces: 64-px alignment measured 5 % SLOWER on the histogram pass, unlike the strips of the apply pass)
            const size_t cstart = c0 / palign * palign;
            constexpr size_t kCh = (size_t)kPieceChunk; // px per wave column
            size_t nch = (c1 - cstart + kCh - 1) / kCh, off = 0;
            while (nch > 0) {
                int lg = kPieceWavesLog2;
                while ((size_t(1) << lg) > nch) --lg;
                const size_t gw = size_t(1) << lg;
                Strip st{};
                st.it.r0 = (int32_t)r0; st.it.r1 = (int32_t)r1;
                st.it.cstart = (int32_t)(cstart + off * kCh);
                st.it.c0 = (int32_t)std::max(c0, cstart + off * kCh);
                st.it.c1 = (int32_t)std::min(c1, cstart + (off + gw) * kCh);
                st.it.gx_log2 = lg;
                st.it.flags = neg ? 1 : 0;
                st.it.id[0] = rw.t0 * kTiles + cw.t0; st.it.id[1] = rw.t0 * kTiles + cw.t1;
                st.it.id[2] = rw.t1 * kTiles + cw.t0; st.it.id[3] = rw.t1 * kTiles + cw.t1;
                st.it.tile = (int32_t)(std::min<size_t>(r0 / g.tile_h, kTiles - 1) * kTiles + std::min<size_t>(c0 / g.tile_w, kTiles - 1));
                st.cost_per_row = (double)gw;
                total += st.cost_per_row * (double)(r1 - r0);
                strips.push_back(st);
                nch -= gw; off += gw;
            }
        }
    }
    P->piece_grid = grid;
    P->piece_items.clear();
    P->piece_first.assign((size_t)grid + 1, 0);
    const double share = total / (double)grid;
    double acc = 0.0;
    int k = 0;
    for (const Strip &st : strips) {
        const int gy = kPieceWaves >> st.it.gx_log2;
        int r = st.it.r0;
        while (r < st.it.r1) {
            int take = st.it.r1 - r;
            if (k < grid - 1) {
                const double room = share * (double)(k + 1) - acc;
                int rows = (int)(room / st.cost_per_row);
                rows = std::max(gy, (rows + gy - 1) / gy * gy); // whole steps of the 16 waves
                take = std::min(take, rows);
            }
            PieceItem it = st.it;
            it.r0 = r; it.r1 = r + take;
            P->piece_items.push_back(it);
            P->piece_first[(size_t)k + 1] = (int32_t)P->piece_items.size();
            acc += st.cost_per_row * (double)take;
            r += take;
            if (k < grid - 1 && acc >= share * (double)(k + 1) - 1e-9) ++k;
        }
    }
    for (int i = 1; i <= grid; ++i) P->piece_first[(size_t)i] = std::max(P->piece_first[(size_t)i], P->piece_first[(size_t)i - 1]);
}

// The exact u16 kernel's work: the cell-major item list cut into `nwg` contiguous shares of equal rows (a share boundary falls
// inside an item: the item is cut there, at a multiple of the 16 waves' step).  A workgroup walks its share top to bottom, strip
// after strip: it rebuilds its tables only where the cell changes (three or four times per launch).
static void build_u16_shares(const StripePlan &P, int nwg, std::vector<Rect> *items, std::vector<int32_t> *first) {
    items->clear();
    first->assign((size_t)nwg + 1, 0);
    double total = 0.0;
    for (const Rect &r : P.u16_rects) total += (double)(r.r1 - r.r0);
    const double share = total / (double)nwg;
    double acc = 0.0;
    int k = 0;
    for (const Rect &src : P.u16_rects) {
        int r = src.r0;
        while (r < src.r1) {
            int take = src.r1 - r;
            if (k < nwg - 1) {
                const double room = share * (double)(k + 1) - acc;
                const int rows = std::max(16, ((int)std::ceil(room) + 15) / 16 * 16);
                take = std::min(take, rows);
            }
            Rect it = src;
            it.r0 = r; it.r1 = r + take;
            items->push_back(it);
            (*first)[(size_t)k + 1] = (int32_t)items->size();
            acc += (double)take;
            r += take;
            if (k < nwg - 1 && acc >= share * (double)(k + 1) - 1e-9) ++k;
        }
    }
    for (int i = 1; i <= nwg; ++i) (*first)[(size_t)i] = std::max((*first)[(size_t)i], (*first)[(size_t)i - 1]);
}

int get_plan(sarpro_hip_ctx *ctx, size_t rows_total, size_t cols, size_t row0, size_t rows_local, int vecw,
             StripePlan **out) {
    auto key = std::make_tuple(rows_total, cols, row0, rows_local, vecw);
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) { *out = it->second; return SARPRO_HIP_OK; }
    if (ctx->plans.size() > 16) { // bounded cache; plans held by an open stripe handle stay
        for (auto jt = ctx->plans.begin(); jt != ctx->plans.end();) {
            if (jt->second->refs > 0) { ++jt; continue; }
            jt->second->release_all();
            delete jt->second;
            jt = ctx->plans.erase(jt);
        }
    }
    StripePlan *P = new StripePlan();
    P->rows_total = rows_total; P->cols = cols; P->row0 = row0; P->rows_local = rows_local; P->vecw = vecw;
    if (ctx->attrs.is_set(A_STRIP_ALIGN)) P->strip_align_px = (size_t)std::max<long long>(1, ctx->attrs.val(A_STRIP_ALIGN, 64));
    P->piece_align = (size_t)kPieceVec;
    if (ctx->attrs.is_set(A_PIECE_ALIGN)) P->piece_align = (size_t)std::max<long long>(kPieceVec, ctx->attrs.val(A_PIECE_ALIGN, kPieceVec) / kPieceVec * kPieceVec);
    build_clahe_geometry(rows_total, cols, &P->geom);
    const size_t strips_across = (cols + 64 * vecw - 1) / (64 * vecw) + kTiles;
    const size_t target_items = 4096;
    size_t chunk_rows = std::min<size_t>(128, std::max<size_t>(16, (rows_local * strips_across + target_items - 1) / target_items)); // <= 128 rows: with line-aligned strips 112..160 rows measured 3-4 % faster than 234 and than 96, 64 rows 12 % slower (per-item table staging), 512 rows 10 % slower (the resident workgroups drift apart and lose the sweep's DRAM locality)
    const AttrSet &at = ctx->attrs; // planner tuning (experiments): read when a plan is built, the plan is cached per shape
    if (at.is_set(A_CHUNK_ROWS)) chunk_rows = (size_t)std::max<long long>(8, at.val(A_CHUNK_ROWS, 0));
    const ClaheGeometry &g = P->geom;
    const bool split = false; // edge lanes are masked inside the vector kernels; no separate sliver items
    for (size_t ty = 0; ty < (size_t)kTiles; ++ty) {
        const size_t r0 = std::min(ty * g.tile_h, rows_total), r1 = std::min((ty + 1) * g.tile_h, rows_total);
        for (size_t tx = 0; tx < (size_t)kTiles; ++tx) {
            const size_t c0 = std::min(tx * g.tile_w, cols), c1 = std::min((tx + 1) * g.tile_w, cols);
            const int ids[4] = {(int)(ty * kTiles + tx), 0, 0, 0};
            add_rects(P->hist_rects_tiled, split ? &P->hist_sliver_tiled : nullptr, *P, r0, r1, c0, c1, ids, chunk_rows, vecw);
        }
    }
    {
        const int ids[4] = {0, 0, 0, 0};
        add_rects(P->hist_rects_flat, split ? &P->hist_sliver_flat : nullptr, *P, 0, rows_total, 0, cols, ids, chunk_rows, vecw);
    }
    // interpolation cells = ranges of constant (t0, t1), cut again where the weight changes sign: the first half tile
    // extrapolates (d < 0, autoscale.rs:308-313) and takes the speculative kernel's wider margin, the half tile after it has the
    // same tiles but d >= 0 and takes the interior margin (uncut, a third of the scene ran with the wide one)
    auto sign_cuts = [](const std::vector<size_t> &starts, const std::vector<RowWeight> &w) {
        std::vector<size_t> out;
        for (size_t i = 0; i + 1 < starts.size(); ++i) {
            out.push_back(starts[i]);
            for (size_t k = starts[i] + 1; k < starts[i + 1]; ++k)
                if ((w[k - 1].d < 0.0) != (w[k].d < 0.0)) out.push_back(k);
        }
        out.push_back(starts.empty() ? 0 : starts.back());
        return out;
    };
    const std::vector<size_t> rcells = sign_cuts(g.row_cell_start, g.row_w), ccells = sign_cuts(g.col_cell_start, g.col_w);
    for (size_t ri = 0; ri + 1 < rcells.size(); ++ri) {
        const size_t r0 = rcells[ri], r1 = rcells[ri + 1];
        if (r0 >= r1) continue;
        const RowWeight &rw = g.row_w[r0];
        for (size_t ci = 0; ci + 1 < ccells.size(); ++ci) {
            const size_t c0 = ccells[ci], c1 = ccells[ci + 1];
            if (c0 >= c1) continue;
            const RowWeight &cw = g.col_w[c0];
            const int ids[4] = {rw.t0 * kTiles + cw.t0, rw.t0 * kTiles + cw.t1, rw.t1 * kTiles + cw.t0,
                                rw.t1 * kTiles + cw.t1};
            // bit 0: the cell holds negative blend weights (dy < 0 or dx < 0) -- the speculative apply kernel widens its f32
            // error margin there
            const bool neg = rw.d < 0.0 || cw.d < 0.0;
            const int cell_flags = (neg ? 1 : 0) | (rw.d < 0.0 ? 2 : 0) | (cw.d < 0.0 ? 4 : 0); // bit 1 / 2: which weight is negative (the margin depends on it)
            add_rects(P->apply_rects, split ? &P->apply_sliver : nullptr, *P, r0, r1, c0, c1, ids, chunk_rows, vecw, cell_flags);
            if (vecw == 8) { // the conflict-free exact kernel with u16 levels out (kernels.hip 4a)
                size_t urows = kU16ItemRows;
                if (at.is_set(A_U16_ITEM_ROWS)) urows = (size_t)std::min<long long>(1024, std::max<long long>(16, at.val(A_U16_ITEM_ROWS, 0)));
                P->u16_item_rows = urows;
                add_rects(P->u16_rects, nullptr, *P, r0, r1, c0, c1, ids, urows, vecw, cell_flags);
            }
            if (vecw == 8) { // the fused CLAHE -> RGB pass (whole scenes and row stripes): 16 waves walk an item, so items are taller
                // the pass hands its items out by a counter, in sweep order: the last rows of the stripe are cut into small items, so that
                // the workgroups that finish their last large item early find something left to do (a tail of at most one small item).
                // Large scenes (six 512-row items per workgroup or more) take the taller items: fewer prologues.  Scene A at 400 MP,
                // configurations interleaved on one box (tools/time_rgb_items.py, profiles/r5/rgb_items.txt): static stride 0.632 ms;
                // counter, 256-row items, no tail 0.607; + 1024 tail rows in 96-row items 0.601; 512 / 2500 / 128 0.597; 640 / 2500 / 128 0.614.
                const bool large = ((rows_local + 511) / 512) * ((cols + 511) / 512) >= (size_t)6 * (size_t)std::max(ctx->cu_count, 1);
                size_t frows = large ? kRgbItemRowsLarge : kRgbItemRows;
                if (at.is_set(A_RGB_ITEM_ROWS)) frows = (size_t)std::max<long long>(16, at.val(A_RGB_ITEM_ROWS, 0));
                size_t tail_rows = std::min<size_t>(large ? kRgbTailRowsLarge : kRgbTailRows, rows_local / 8), tail_item = large ? kRgbTailItemRowsLarge : kRgbTailItemRows;
                if (at.is_set(A_RGB_TAIL_ROWS)) tail_rows = (size_t)std::max<long long>(0, at.val(A_RGB_TAIL_ROWS, 0));
                if (at.is_set(A_RGB_TAIL_ITEM_ROWS)) tail_item = (size_t)std::max<long long>(16, at.val(A_RGB_TAIL_ITEM_ROWS, 0));
                const size_t stripe_end = row0 + rows_local, tail_start = stripe_end > tail_rows ? stripe_end - tail_rows : 0;
                if (r0 < tail_start) add_rects(P->rgb_rects, nullptr, *P, r0, std::min(r1, tail_start), c0, c1, ids, frows, vecw, cell_flags);
                if (r1 > tail_start) add_rects(P->rgb_rects, nullptr, *P, std::max(r0, tail_start), r1, c0, c1, ids, std::min(frows, tail_item), vecw, cell_flags);
                size_t srows = kSampleItemRows;
                if (at.is_set(A_SAMPLE_ITEM_ROWS)) srows = (size_t)std::max<long long>(16, at.val(A_SAMPLE_ITEM_ROWS, 0));
                add_rects(P->sample_rects, nullptr, *P, r0, r1, c0, c1, ids, srows, vecw, cell_flags);
            }
        }
    }
    // launch order = sweep order: consecutive work items cover adjacent column strips of the same row
    // chunk, so the workgroups resident at any moment read neighbouring 1-KiB segments of the same
    // image rows (DRAM page locality) instead of strips megabytes apart
    auto sweep_order = [](std::vector<Rect> &v) {
        std::stable_sort(v.begin(), v.end(), [](const Rect &x, const Rect &y) {
            return x.r0 != y.r0 ? x.r0 < y.r0 : x.cstart < y.cstart;
        });
    };
    if (!at.on(A_NO_SWEEP_ORDER)) {
        sweep_order(P->hist_rects_tiled);
        sweep_order(P->hist_rects_flat);
        sweep_order(P->apply_rects);
        sweep_order(P->rgb_rects);
        sweep_order(P->sample_rects);
    }
    if (vecw == 8 && row0 == 0 && rows_local == rows_total && ctx->cu_count > 0) {
        long long pg = ctx->cu_count; // PIECE_GRID: planner tuning (how many persistent workgroups share the histogram sweep)
        if (at.is_set(A_PIECE_GRID)) pg = std::max<long long>(1, at.val(A_PIECE_GRID, pg));
        build_pieces(P, (int)std::min<long long>(pg, kPieceMaxGrid));
    }
    int rc = upload_vec(ctx, P->d_hist_rects_tiled, P->hist_rects_tiled.data(), P->hist_rects_tiled.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_piece_items, P->piece_items.data(), P->piece_items.size() * sizeof(PieceItem));
    if (!rc) rc = upload_vec(ctx, P->d_piece_first, P->piece_first.data(), P->piece_first.size() * sizeof(int32_t));
    if (!rc) rc = upload_vec(ctx, P->d_hist_rects_flat, P->hist_rects_flat.data(), P->hist_rects_flat.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_apply_rects, P->apply_rects.data(), P->apply_rects.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_rgb_rects, P->rgb_rects.data(), P->rgb_rects.size() * sizeof(Rect));
    if (!rc && !P->rgb_rects.empty()) { // the fused pass's saturation tables (host_logic.h); the column table padded so that every lane's 8-byte load is in range
        std::vector<uint8_t> cc, rb;
        P->sat_ok = clahe_saturated_levels(g, &cc, &rb);
        if (P->sat_ok) {
            cc.resize(round_up(cols, 64) + 64, 0);
            rc = upload_vec(ctx, P->d_sat_col, cc.data(), cc.size());
            if (!rc) rc = upload_vec(ctx, P->d_sat_row, rb.data(), rb.size());
        }
    }
    if (!rc) rc = upload_vec(ctx, P->d_sample_rects, P->sample_rects.data(), P->sample_rects.size() * sizeof(Rect));
    for (int nb = 1; nb <= kMaxBands && !rc && !P->u16_rects.empty() && ctx->cu_count > 0; ++nb) { // the exact u16 kernel's shares, per band count of a launch
        build_u16_shares(*P, std::max(1, ctx->cu_count / nb), &P->u16_items[nb - 1], &P->u16_first[nb - 1]);
        P->u16_nwg[nb - 1] = (int)P->u16_first[nb - 1].size() - 1;
        rc = upload_vec(ctx, P->d_u16_items[nb - 1], P->u16_items[nb - 1].data(), P->u16_items[nb - 1].size() * sizeof(Rect));
        if (!rc) rc = upload_vec(ctx, P->d_u16_first[nb - 1], P->u16_first[nb - 1].data(), P->u16_first[nb - 1].size() * sizeof(int32_t));
    }
    if (!rc) rc = upload_vec(ctx, P->d_hist_sliver_tiled, P->hist_sliver_tiled.data(), P->hist_sliver_tiled.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_hist_sliver_flat, P->hist_sliver_flat.data(), P->hist_sliver_flat.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_apply_sliver, P->apply_sliver.data(), P->apply_sliver.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_row_w, g.row_w.data(), g.row_w.size() * sizeof(RowWeight));
    if (!rc) rc = upload_vec(ctx, P->d_col_w, g.col_w.data(), g.col_w.size() * sizeof(RowWeight));
    if (rc) {
        P->release_all();
        delete P;
        return rc;
    }
    ctx->plans[key] = P;
    *out = P;
    return SARPRO_HIP_OK;
}

// ---------------------------------------------------------------------------------------
// The u16 pipeline as explicit phases (also the row-stripe protocol: each phase ends in a
// small integer reduction that a multi-rank driver all-reduces before the next phase).
// ---------------------------------------------------------------------------------------
struct U16Job {
    sarpro_hip_ctx *ctx = nullptr;
    int nbands = 1;
    const uint16_t *d_in[kMaxBands] = {nullptr, nullptr};
    size_t rows_total = 0, cols = 0, row0 = 0, rows_local = 0, in_pitch = 0;
    int strategy = 0, bit_depth = 0, mode = 0;
    bool synrgb = false; // dual-pol JPEG branch (save.rs:317-367): always U8, Tamed uses tamed_synrgb
    int tamed_force = 0; // single band tamed_synrgb entry point: 1 copol, 2 crosspol
    bool vec = false;
    bool reduce = false; // row stripe of a multi-rank scene: histograms are all-reduced over ctx->comm, on the stream
    bool hist_done = false; // phase 1 already ran (streaming ingest: chunk by chunk, under the upload)
    bool clear_after_sum = false; // untiled chain: k_sum_tile_hists is the last reader of the tile histogram and zeroes it
    size_t tile_hist_bytes = 0;   // footprint of this job's histogram pass in ctx->tile_hist[0]
    bool allow_async = false; // the entry point may return once the device chain is enqueued (SARPRO_HIP_CTX_ASYNC_DEV)
    bool tables_only = false; // percentile chain: stop at the DN -> final u8 tables (band_u8_table_dev)
    StripePlan *plan = nullptr;
    // host-side state between phases
    sarpro_hip_stats stats[kMaxBands];
    DnLut lut[kMaxBands];
    uint8_t resc[kMaxBands][256];
    bool resc_identity[kMaxBands] = {true, true};
    uint64_t level_hist_h[kMaxBands][256];
    int floor_with_cushion = -1;
    // level rasters (u8) when an intermediate is needed
    uint8_t *d_levels[kMaxBands] = {nullptr, nullptr};
    size_t lvl_pitch = 0;

    bool clahe() const { return strategy == SARPRO_STRATEGY_CLAHE; }
    int tamed_kind(int band) const {
        if (tamed_force) return tamed_force;
        if ((synrgb || (tables_only && nbands == 2)) && strategy == SARPRO_STRATEGY_TAMED) return band == 0 ? kTamedCopol : kTamedCrosspol;
        return kNotTamedSynrgb;
    }
    bool u8_out() const { return synrgb || tamed_force || bit_depth == SARPRO_BITDEPTH_U8; }
};

static bool ptr_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int job_init(U16Job &J) {
    sarpro_hip_ctx *ctx = J.ctx;
    if (J.strategy < 0 || J.strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (J.bit_depth != SARPRO_BITDEPTH_U8 && J.bit_depth != SARPRO_BITDEPTH_U16)
        return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if (J.mode < 0 || J.mode > SARPRO_SYNRGB_ENHANCED) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synrgb mode");
    if (J.in_pitch < J.cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pitch < cols");
    if (J.rows_total > 0x7FFFFFFFull || J.cols > 0x7FFFFFFFull) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "raster too large");
    if (J.row0 + J.rows_local > J.rows_total) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "stripe outside the scene");
    if (J.clahe() && !clahe_shape_ok(J.rows_total, J.cols))
        return fail(ctx, SARPRO_HIP_ERR_UNSUPPORTED_SHAPE,
                    "CLAHE tile arithmetic underflows for this shape (reference panics: autoscale.rs:250,254)");
    J.vec = J.in_pitch % 8 == 0;
    for (int b = 0; b < J.nbands; ++b) J.vec = J.vec && ptr_aligned16(J.d_in[b]);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return get_plan(ctx, J.rows_total, J.cols, J.row0, J.rows_local, J.vec ? 8 : 1, &J.plan);
}

// the tile histograms were zeroed again by their last reader (enqueued): the next histogram pass may skip its fill
static void mark_tile_hist_clean(U16Job &J) {
    if (!J.tile_hist_bytes) return; // the pass was not begun by this job object (cannot know its footprint)
    J.ctx->tile_hist_clean_ptr = J.ctx->tile_hist[0].p;
    J.ctx->tile_hist_clean_bytes = J.tile_hist_bytes;
}

// phase 1: local DN histograms -> ctx->ghist (u64 [nbands][65536]) on the device
static uint32_t *tile_hist_of(sarpro_hip_ctx *ctx, int band, int ntiles) { return ctx->tile_hist[0].as<uint32_t>() + (size_t)band * 65536 * (size_t)ntiles; }

// The histogram pass can be issued in pieces (streaming ingest: the work items whose rows have arrived):
// `begin` clears the tile histograms, [first, last) are indices into the plan's item list (sorted by row),
// `end` folds the tile histograms into the band histogram.  The default is the whole pass.
static int job_phase1(U16Job &J, bool begin = true, int first = 0, int last = -1, bool end = true) {
    sarpro_hip_ctx *ctx = J.ctx;
    if (J.hist_done) return SARPRO_HIP_OK;
    const bool tiled = J.clahe();
    const int ntiles = tiled ? kTiles * kTiles : 1;
    HIPCHK(ctx, ctx->ghist.reserve(sizeof(uint64_t) * 65536 * kMaxBands));
    DnHistArgs a{};
    // both bands' tile histograms in one allocation (band b at tile_hist_of(ctx, b, ntiles)): one fill instead of two
    const size_t band_bytes = sizeof(uint32_t) * 65536 * (size_t)ntiles;
    HIPCHK(ctx, ctx->tile_hist[0].reserve(band_bytes * kMaxBands));
    if (begin && ctx->pipe_wait_before_hist) { // resident batch, PIPE_ORDER = 2: this scene's sweep beside the previous scene's fused pass, not before it
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_wait_before_hist, 0));
        ctx->pipe_wait_before_hist = nullptr;
    }
    if (begin) {
        // the chain's last reader of the tile histograms zeroes what it read: a scene that follows one of the same or a
        // larger footprint on this context starts on clean bins (the fill of 32 MiB and its launch: ~12 us)
        const size_t need = band_bytes * (size_t)J.nbands;
        if (!(ctx->tile_hist_clean_ptr == ctx->tile_hist[0].p && ctx->tile_hist_clean_bytes >= need))
            HIPCHK(ctx, hipMemsetAsync(ctx->tile_hist[0].p, 0, need, ctx->stream));
        ctx->tile_hist_clean_bytes = 0; // dirty from here on
        J.tile_hist_bytes = need;
    }
    for (int b = 0; b < J.nbands; ++b) {
        a.in[b] = J.d_in[b];
        a.tile_hist[b] = tile_hist_of(ctx, b, ntiles);
    }
    a.pitch = J.in_pitch;
    const int nall = (int)(tiled ? J.plan->hist_rects_tiled.size() : J.plan->hist_rects_flat.size());
    if (last < 0 || last > nall) last = nall;
    a.rects = (tiled ? J.plan->d_hist_rects_tiled : J.plan->d_hist_rects_flat).as<Rect>() + first;
    a.lds_bins = 8192;
    const int nrects = last - first;
    if (J.vec && tiled && J.nbands == 2 && J.plan->piece_grid > 0 && first == 0 && last == nall && nall > 0 && !ctx->attrs.on(A_NO_PIECE_HIST)) {
        // whole tiled pass, both bands: persistent workgroups on balanced pieces (piece_kernels.hip k_dn_hist_pieces)
        DnHistPiecesArgs pa{};
        for (int b = 0; b < 2; ++b) { pa.in[b] = a.in[b]; pa.tile_hist[b] = a.tile_hist[b]; }
        pa.pitch = a.pitch; pa.items = J.plan->d_piece_items.as<PieceItem>(); pa.wg_first = J.plan->d_piece_first.as<int32_t>();
        pa.lds_bins = kPieceLdsBins;
        {
            KernelTimer t(ctx, "dn_hist_u16");
            HIPCHK(ctx, launch_dn_hist_pieces(pa, J.plan->piece_grid, ctx->stream));
        }
        if (ctx->pipe_record_after_hist) { HIPCHK(ctx, hipEventRecord(ctx->pipe_record_after_hist, ctx->stream)); ctx->pipe_record_after_hist = nullptr; }
    } else if (J.vec && !tiled && first == 0 && last == nall && nall > 0 && !ctx->attrs.on(A_NO_LINEAR_HIST)) {
        KernelTimer t(ctx, "dn_hist_u16"); // whole untiled pass in one go: the in-order sweep
        HIPCHK(ctx, launch_dn_hist_u16_linear(a, (uint32_t)J.rows_local, (uint32_t)J.cols, J.nbands, ctx->stream));
    } else if (nrects > 0) {
        KernelTimer t(ctx, "dn_hist_u16");
        if (J.vec) HIPCHK(ctx, launch_dn_hist_u16_interior(a, nrects, J.nbands, ctx->stream));
        else HIPCHK(ctx, launch_dn_hist_u16(a, nrects, J.nbands, false, ctx->stream));
    }
    if (!end) return SARPRO_HIP_OK;
    if (J.vec && !(tiled ? J.plan->hist_sliver_tiled : J.plan->hist_sliver_flat).empty()) { // unused unless the planner splits slivers
        const int ns = (int)(tiled ? J.plan->hist_sliver_tiled.size() : J.plan->hist_sliver_flat.size());
        a.rects = (tiled ? J.plan->d_hist_sliver_tiled : J.plan->d_hist_sliver_flat).as<Rect>();
        a.lds_bins = 2048;
        KernelTimer t(ctx, "dn_hist_u16_sliver");
        HIPCHK(ctx, launch_dn_hist_u16(a, ns, J.nbands, false, ctx->stream));
    }
    {
        SumTileHistArgs sa{};
        for (int b = 0; b < J.nbands; ++b) {
            sa.tile_hist[b] = tile_hist_of(ctx, b, ntiles);
            sa.out[b] = ctx->ghist.as<unsigned long long>() + (size_t)b * 65536;
        }
        sa.clear = J.clear_after_sum ? 1u : 0u;
        KernelTimer t(ctx, "sum_tile_hists");
        HIPCHK(ctx, launch_sum_tile_hists(sa, ntiles, J.nbands, ctx->stream));
        if (sa.clear) mark_tile_hist_clean(J);
    }
    return SARPRO_HIP_OK;
}

// after the (optional) all-reduce of ghist: stats, window, DN tables
static int job_after_phase1(U16Job &J) {
    sarpro_hip_ctx *ctx = J.ctx;
    const size_t bytes = sizeof(uint64_t) * 65536 * (size_t)J.nbands;
    HIPCHK(ctx, ctx->h_ghist.reserve(sizeof(uint64_t) * 65536 * kMaxBands));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_ghist.p, ctx->ghist.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, ctx->luts.reserve(2 * 131072));
    HIPCHK(ctx, ctx->h_upload.reserve(2 * 131072 + 2 * 64 * 256 * 8 + 66048 + 1024));
    // the bands are independent: band 1 is worked on by a helper thread while this thread does band 0
    auto band_work = [&J, ctx](int b) -> int {
        uint64_t *h = ctx->h_ghist.as<uint64_t>() + (size_t)b * 65536;
        { // the interior kernel does not count DN = 0: it is what is left of the scene
            uint64_t others = 0;
            for (uint32_t dn = 1; dn < 65536; ++dn) others += h[dn];
            h[0] = (uint64_t)J.rows_total * J.cols - others;
        }
        RETCHK(stats_from_dn_hist(h, &J.stats[b]));
        RETCHK(select_window(&J.stats[b], J.strategy, J.tamed_kind(b)));
        if (J.clahe()) build_clahe_bin_lut_u16(J.stats[b], &J.lut[b]);
        else build_level_lut_u16(J.stats[b], J.u8_out() ? SARPRO_BITDEPTH_U8 : SARPRO_BITDEPTH_U16, J.tamed_kind(b), &J.lut[b]);
        if (!J.clahe() && J.u8_out()) {
            // levels are a function of DN: their histogram, min and max follow from the DN histogram
            std::memset(J.level_hist_h[b], 0, sizeof(J.level_hist_h[b]));
            for (uint32_t dn = 0; dn < 65536; ++dn)
                if (h[dn]) J.level_hist_h[b][dn ? J.lut[b].full[dn] : 0] += h[dn];
        }
        return SARPRO_HIP_OK;
    };
    if (J.nbands == 2) {
        std::future<int> other = std::async(std::launch::async, band_work, 1);
        const int rc0 = band_work(0), rc1 = other.get();
        if (rc0) return rc0;
        if (rc1) return rc1;
    } else {
        RETCHK(band_work(0));
    }
    return SARPRO_HIP_OK;
}

// u8 rescale (autoscale.rs:348-364) from the level histogram; tamed_synrgb has none (:731-741)
static void job_rescale_from_level_hist(U16Job &J, int b) {
    if (J.tamed_kind(b) != kNotTamedSynrgb) {
        for (int i = 0; i < 256; ++i) J.resc[b][i] = (uint8_t)i;
        J.resc_identity[b] = true;
        return;
    }
    unsigned mn = 0, mx = 0;
    bool any = false;
    for (unsigned i = 0; i < 256; ++i)
        if (J.level_hist_h[b][i]) { if (!any) mn = i; mx = i; any = true; }
    u8_rescale_lut(mn, mx, J.resc[b]);
    J.resc_identity[b] = true;
    for (unsigned i = 0; i < 256; ++i)
        if (J.level_hist_h[b][i] && J.resc[b][i] != i) J.resc_identity[b] = false;
}

// phase 2 (CLAHE): per-tile bin histograms -> ctx->tile_bins (u64 [nbands][64][256])
static int job_phase2(U16Job &J) {
    sarpro_hip_ctx *ctx = J.ctx;
    if (!J.clahe()) return SARPRO_HIP_OK;
    HIPCHK(ctx, ctx->tile_bins.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands));
    TileBinHistArgs ta{};
    for (int b = 0; b < J.nbands; ++b) {
        uint8_t *stage = ctx->h_upload.as<uint8_t>() + (size_t)b * 65536;
        for (int i = 0; i < 65536; ++i) stage[i] = (uint8_t)J.lut[b].full[i];
        HIPCHK(ctx, hipMemcpyAsync(ctx->luts.as<uint8_t>() + (size_t)b * 131072, stage, 65536, hipMemcpyHostToDevice, ctx->stream));
        ta.tile_hist[b] = tile_hist_of(ctx, b, kTiles * kTiles);
        ta.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
        ta.out[b] = ctx->tile_bins.as<unsigned long long>() + (size_t)b * 64 * 256;
    }
    KernelTimer t(ctx, "tile_bin_hist");
    HIPCHK(ctx, launch_tile_bin_hist(ta, kTiles * kTiles, J.nbands, ctx->stream));
    return SARPRO_HIP_OK;
}

static int ensure_levels(U16Job &J) {
    sarpro_hip_ctx *ctx = J.ctx;
    J.lvl_pitch = round_up(J.cols, 64);
    for (int b = 0; b < J.nbands; ++b) {
        HIPCHK(ctx, ctx->levels[b].reserve(J.lvl_pitch * std::max<size_t>(J.rows_local, 1)));
        J.d_levels[b] = ctx->levels[b].as<uint8_t>();
    }
    return SARPRO_HIP_OK;
}

// phase 3: apply.  d_out[b] (+ out_pitch) receive the per-band raster when the caller wants it
// (single-band entry points: the final raster; dual-pol: optional u8 copies, may be null).
// Leaves the u8 level histogram in ctx->level_hist (CLAHE u8) for the reduction.
static int job_phase3(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch) {
    sarpro_hip_ctx *ctx = J.ctx;
    const bool u8o = J.u8_out();
    HIPCHK(ctx, ctx->level_hist.reserve(sizeof(uint64_t) * 256 * kMaxBands));
    if (J.clahe()) {
        // CDFs from the (reduced) tile histograms
        const size_t tb_bytes = sizeof(uint64_t) * 64 * 256 * (size_t)J.nbands;
        HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands));
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->tile_bins.p, tb_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, ctx->cdfs.reserve(sizeof(double) * 64 * 256 * kMaxBands));
        double *h_cdfs = reinterpret_cast<double *>(ctx->h_upload.as<uint8_t>() + 2 * 131072);
        for (int b = 0; b < J.nbands; ++b)
            RETCHK(clahe_cdfs(ctx->h_small.as<uint64_t>() + (size_t)b * 64 * 256, J.rows_total, J.cols, h_cdfs + (size_t)b * 64 * 256));
        HIPCHK(ctx, hipMemcpyAsync(ctx->cdfs.p, h_cdfs, sizeof(double) * 64 * 256 * (size_t)J.nbands, hipMemcpyHostToDevice, ctx->stream));

        ClaheApplyArgs a{};
        const bool direct = !J.synrgb && d_out[0] != nullptr; // single band: write the caller's raster
        if (!direct) RETCHK(ensure_levels(J));
        size_t win_max = 0;
        for (int b = 0; b < J.nbands; ++b) {
            a.in[b] = J.d_in[b];
            a.out[b] = direct ? d_out[b] : (void *)J.d_levels[b];
            a.cdfs[b] = ctx->cdfs.as<double>() + (size_t)b * 64 * 256;
            a.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
            a.win_lo[b] = J.lut[b].win_lo;
            a.win_hi[b] = J.lut[b].win_hi;
            win_max = std::max<size_t>(win_max, J.lut[b].win_hi - J.lut[b].win_lo + 1);
            a.level_hist[b] = u8o ? ctx->level_hist.as<unsigned long long>() + (size_t)b * 256 : nullptr;
        }
        a.in_pitch = J.in_pitch;
        a.out_pitch = direct ? out_pitch : J.lvl_pitch;
        a.rects = J.plan->d_apply_rects.as<Rect>();
        a.lut_in_lds = win_max <= kLutLdsMaxBytes;
        a.row_w = J.plan->d_row_w.as<RowWeight>();
        a.col_w = J.plan->d_col_w.as<RowWeight>();
        a.row_off = (int32_t)J.row0;
        a.max_val = u8o ? 255.0 : 65535.0;
        if (u8o) HIPCHK(ctx, hipMemsetAsync(ctx->level_hist.p, 0, sizeof(uint64_t) * 256 * kMaxBands, ctx->stream));
        const bool vec = J.vec && a.out_pitch % 8 == 0 && ptr_aligned16(a.out[0]) && (J.nbands < 2 || ptr_aligned16(a.out[1]));
        if (vec != J.vec) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "output raster must be 16-byte aligned with pitch % 8 == 0 when the input is");
        if (J.vec && u8o && clahe_apply_spec_ok(a, J.nbands) && !ctx->attrs.on(A_NO_SPEC)) {
            HIPCHK(ctx, ctx->spec_dump.reserve(kSpecDumpBytes));
            a.dump = ctx->spec_dump.as<uint8_t>();
            KernelTimer t(ctx, "clahe_apply_u8_spec");
            HIPCHK(ctx, launch_clahe_apply_u8_spec(a, (int)J.plan->apply_rects.size(), J.nbands, ctx->stream));
        } else {
            KernelTimer t(ctx, "clahe_apply_u16");
            HIPCHK(ctx, launch_clahe_apply_u16(a, (int)J.plan->apply_rects.size(), J.nbands, J.vec, !u8o, ctx->stream));
        }
        if (J.vec && !J.plan->apply_sliver.empty()) { // < 8-column leftovers at cell edges: scalar exact kernel
            a.rects = J.plan->d_apply_sliver.as<Rect>();
            KernelTimer t(ctx, "clahe_apply_sliver");
            HIPCHK(ctx, launch_clahe_apply_u16(a, (int)J.plan->apply_sliver.size(), J.nbands, false, !u8o, ctx->stream));
        }
        return SARPRO_HIP_OK;
    }
    // percentile strategies: the level histogram is known on the host already; publish it on the
    // device too so the stripe protocol reduces the same buffer in both modes
    if (u8o) {
        uint64_t *stage = reinterpret_cast<uint64_t *>(ctx->h_upload.as<uint8_t>() + 2 * 131072);
        for (int b = 0; b < J.nbands; ++b) std::memcpy(stage + (size_t)b * 256, J.level_hist_h[b], sizeof(uint64_t) * 256);
        HIPCHK(ctx, hipMemcpyAsync(ctx->level_hist.p, stage, sizeof(uint64_t) * 256 * (size_t)J.nbands, hipMemcpyHostToDevice, ctx->stream));
    }
    return SARPRO_HIP_OK;
}

// phase 4: finish.  Single band: final raster into d_out[0].  Dual-pol: RGB into d_rgb and
// (optionally) the per-band u8 rasters into d_out[b].
static int job_phase4(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px,
                      bool level_hist_reduced_on_device) {
    sarpro_hip_ctx *ctx = J.ctx;
    const bool u8o = J.u8_out();
    const uint32_t rows = (uint32_t)J.rows_local, cols = (uint32_t)J.cols;

    if (u8o && (J.clahe() || level_hist_reduced_on_device)) {
        HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands));
        uint64_t *h = ctx->h_small.as<uint64_t>() + 64 * 256 * kMaxBands;
        HIPCHK(ctx, hipMemcpyAsync(h, ctx->level_hist.p, sizeof(uint64_t) * 256 * (size_t)J.nbands, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int b = 0; b < J.nbands; ++b) {
            std::memcpy(J.level_hist_h[b], h + (size_t)b * 256, sizeof(uint64_t) * 256);
            if (J.clahe()) { // the speculative kernel does not count level 0: it is what is left of the scene
                uint64_t others = 0;
                for (int i = 1; i < 256; ++i) others += J.level_hist_h[b][i];
                J.level_hist_h[b][0] = (uint64_t)J.rows_total * J.cols - others;
            }
        }
    }
    if (u8o) for (int b = 0; b < J.nbands; ++b) job_rescale_from_level_hist(J, b);

    uint8_t *up = ctx->h_upload.as<uint8_t>();
    // dual-pol percentile strategies without per-band outputs: ONE fused pass DN,DN -> RGB (7 B/px), no
    // intermediate u8 rasters
    bool fused = false;
    if (!J.clahe() && J.synrgb && !d_out[0] && !d_out[1] && J.vec && J.in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 &&
        ptr_aligned16(d_rgb) && !ctx->attrs.on(A_NO_FUSED)) {
        LutComposeArgs probe{};
        probe.win_hi[0] = J.lut[0].win_hi; probe.win_hi[1] = J.lut[1].win_hi;
        fused = lut_compose_fits(probe);
    }
    if (!J.clahe()) {
        // table apply: final = resc[level[DN]] (u8) or level[DN] (u16)
        const bool need_levels = !fused && J.synrgb && (d_out[0] == nullptr || d_out[1] == nullptr);
        if (need_levels) RETCHK(ensure_levels(J));
        for (int b = 0; b < J.nbands; ++b) {
            LutApplyArgs a{};
            a.in = J.d_in[b];
            a.in_pitch = J.in_pitch;
            a.rows = rows; a.cols = cols;
            if (d_out[b]) { a.out = d_out[b]; a.out_pitch = out_pitch; }
            else { a.out = J.d_levels[b]; a.out_pitch = J.lvl_pitch; }
            a.win_lo = J.lut[b].win_lo; a.win_hi = J.lut[b].win_hi;
            const size_t esz = u8o ? 1 : 2;
            a.lut_in_lds = (size_t)(a.win_hi - a.win_lo + 1) * esz <= kLutLdsMaxBytes;
            uint8_t *stage = up + (size_t)b * 131072;
            if (u8o) for (int i = 0; i < 65536; ++i) stage[i] = J.resc[b][J.lut[b].full[i] & 0xFF];
            else std::memcpy(stage, J.lut[b].full.data(), 131072);
            if (u8o) stage[0] = J.resc[b][0];
            void *d_lut = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
            HIPCHK(ctx, hipMemcpyAsync(d_lut, stage, 65536 * esz, hipMemcpyHostToDevice, ctx->stream));
            a.lut = d_lut;
            if (fused) continue; // the tables are consumed by the fused pass below
            const bool vec = J.vec && a.out_pitch % 8 == 0 && ptr_aligned16(a.out);
            KernelTimer t(ctx, "lut_apply_u16");
            HIPCHK(ctx, launch_lut_apply_u16(a, vec, !u8o, ctx->stream));
        }
    }
    if (!J.synrgb) {
        if (J.clahe() && u8o && !J.resc_identity[0]) { // rare: CLAHE levels did not span 0..255
            HIPCHK(ctx, ctx->tables.reserve(66048));
            std::memcpy(up, J.resc[0], 256);
            HIPCHK(ctx, hipMemcpyAsync(ctx->tables.p, up, 256, hipMemcpyHostToDevice, ctx->stream));
            KernelTimer t(ctx, "remap_u8");
            HIPCHK(ctx, launch_remap_u8(reinterpret_cast<uint8_t *>(d_out[0]), out_pitch, rows, cols, ctx->tables.as<uint8_t>(), ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return SARPRO_HIP_OK;
    }

    // ---- dual-pol composition (synthetic_rgb.rs:182-197) ----
    // combined histogram of the FINAL u8 bands = level histograms pushed through each band's rescale
    uint64_t combined[256];
    std::memset(combined, 0, sizeof(combined));
    for (int b = 0; b < 2; ++b)
        for (int i = 0; i < 256; ++i) combined[J.resc[b][i]] += J.level_hist_h[b][i];
    std::vector<uint8_t> luts(66048), tables(66048);
    const bool suppressed = J.strategy == SARPRO_STRATEGY_TAMED || J.strategy == SARPRO_STRATEGY_CLAHE;
    if (suppressed) {
        J.floor_with_cushion = synrgb_floor_from_hist(combined, (uint64_t)J.rows_total * J.cols);
        synrgb_luts_suppressed(J.floor_with_cushion, luts.data());
    } else {
        J.floor_with_cushion = -1;
        synrgb_luts_default(luts.data());
    }
    uint8_t ident[256];
    for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
    // CLAHE: the compose kernel reads LEVELS, so the rescale is folded into the tables.
    // Percentile strategies: the table-apply pass already wrote final u8 values.
    const uint8_t *r1 = J.clahe() ? J.resc[0] : ident, *r2 = J.clahe() ? J.resc[1] : ident;
    fold_compose_tables(luts.data(), J.floor_with_cushion, r1, r2, tables.data());
    HIPCHK(ctx, ctx->tables.reserve(66048 + 512));
    uint8_t *tstage = up + 2 * 131072 + 2 * 64 * 256 * 8;
    std::memcpy(tstage, tables.data(), 66048);
    HIPCHK(ctx, hipMemcpyAsync(ctx->tables.p, tstage, 66048, hipMemcpyHostToDevice, ctx->stream));

    if (fused) {
        LutComposeArgs f{};
        for (int b = 0; b < 2; ++b) {
            f.in[b] = J.d_in[b];
            f.lut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
            f.win_hi[b] = J.lut[b].win_hi;
        }
        f.rgb = d_rgb; f.in_pitch = J.in_pitch; f.rgb_pitch_px = rgb_pitch_px; f.rows = rows; f.cols = cols;
        f.tables = ctx->tables.as<uint8_t>();
        {
            KernelTimer t(ctx, "lut_compose_u16");
            HIPCHK(ctx, launch_lut_compose_u16(f, ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return SARPRO_HIP_OK;
    }
    ComposeArgs c{};
    if (J.clahe()) { c.b1 = J.d_levels[0]; c.b2 = J.d_levels[1]; c.in_pitch = J.lvl_pitch; }
    else if (d_out[0] && d_out[1]) { c.b1 = (const uint8_t *)d_out[0]; c.b2 = (const uint8_t *)d_out[1]; c.in_pitch = out_pitch; }
    else {
        // percentile strategies wrote into d_out[b] when given, else into the level rasters; compose
        // needs one pitch for both bands, so mixed destinations are not offered by the entry points
        c.b1 = J.d_levels[0]; c.b2 = J.d_levels[1]; c.in_pitch = J.lvl_pitch;
    }
    c.rgb = d_rgb; c.rgb_pitch_px = rgb_pitch_px; c.rows = rows; c.cols = cols;
    c.tables = ctx->tables.as<uint8_t>();
    const int cvec = (c.in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 && ptr_aligned16(c.b1) && ptr_aligned16(c.b2) && ptr_aligned16(d_rgb)) ? 16 : 1;
    {
        KernelTimer t(ctx, "compose_u8");
        HIPCHK(ctx, launch_compose_u8(c, cvec, ctx->stream));
    }
    if (J.clahe()) { // optional per-band u8 rasters: levels pushed through the rescale
        for (int b = 0; b < 2; ++b) {
            if (!d_out[b]) continue;
            HIPCHK(ctx, hipMemcpy2DAsync(d_out[b], out_pitch, J.d_levels[b], J.lvl_pitch, cols, rows, hipMemcpyDeviceToDevice, ctx->stream));
            if (!J.resc_identity[b]) {
                uint8_t *m = tstage + 66048 + (size_t)b * 256;
                std::memcpy(m, J.resc[b], 256);
                HIPCHK(ctx, hipMemcpyAsync(ctx->tables.as<uint8_t>() + 66048 + (size_t)b * 256, m, 256, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, launch_remap_u8(reinterpret_cast<uint8_t *>(d_out[b]), out_pitch, rows, cols,
                                            ctx->tables.as<uint8_t>() + 66048 + (size_t)b * 256, ctx->stream));
            }
        }
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}


// ---------------------------------------------------------------------------------------
// Device-resident chain: CLAHE, u8 output, vector layout, whole scene on this GPU.  The host only
// enqueues: statistics, CLAHE bins, CDFs, rescale and compose tables are computed by small kernels
// (chain_kernels.hip), so there is ONE stream synchronisation per scene, at the end.
// ---------------------------------------------------------------------------------------
constexpr size_t kChainOffDb = 0, kChainOffSupp = 65536 * 8, kChainOffBlue = kChainOffSupp + 21504;
constexpr size_t kChainOffBlueDef = kChainOffBlue + 65536, kChainOffDefRg = kChainOffBlueDef + 65536;
constexpr size_t kChainOffGamma = kChainOffDefRg + 512, kChainOffBluePQ = kChainOffGamma + 3 * 256 * 8, kChainConstBytes = kChainOffBluePQ + 512 * 4;
constexpr size_t kTablesOffPQ = 66048 + 512, kTablesBytes = kTablesOffPQ + 2 * 256 * 4; // compose tables | per-band maps | Pv[256] f32 | Qv[256] f32 (the blue factors by LEVEL)
constexpr size_t kStateOffResc = 2 * sizeof(ChainBandState), kStateOffIdent = kStateOffResc + 512,
                 kStateOffFloor = kStateOffIdent + 16, kStateBytes = kStateOffFloor + 16;

extern "C" int sarpro_hip_ctx_chain_report(sarpro_hip_ctx *ctx, sarpro_hip_chain_report *out) {
    if (!ctx || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    out->floor_with_cushion = -1;
    if (!ctx->chain_state.p || !ctx->last_final_hist) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "no CLAHE chain has run on this context");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint8_t *state = ctx->chain_state.as<uint8_t>();
    HIPCHK(ctx, hipMemcpy(out->rescale, state + kStateOffResc, 512, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out->identity, state + kStateOffIdent, 2, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(&out->floor_with_cushion, state + kStateOffFloor, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out->level_hist, ctx->last_final_hist, sizeof(out->level_hist), hipMemcpyDeviceToHost));
    return SARPRO_HIP_OK;
}


static int chain_prepare(sarpro_hip_ctx *ctx) {
    if (ctx->chain_ready) return SARPRO_HIP_OK;
    HIPCHK(ctx, ctx->chain_consts.reserve(kChainConstBytes));
    HIPCHK(ctx, ctx->chain_state.reserve(kStateBytes));
    uint8_t *d = ctx->chain_consts.as<uint8_t>();
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffDb, db_table_u16(), 65536 * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffSupp, synrgb_supp_rg_tables(), 41 * 512, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffBlue, synrgb_blue_pair_supp(), 65536, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffBlueDef, synrgb_blue_pair_default(), 65536, hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint8_t> dflt(66048);
    synrgb_luts_default(dflt.data());
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffDefRg, dflt.data(), 512, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffGamma, gamma_level_thresholds_u8(), 3 * 256 * 8, hipMemcpyHostToDevice, ctx->stream));
    if (const float *pq = synrgb_blue_factors_supp()) { // the suppressed blue as a product of two factors (verified against the pair table for all pairs)
        HIPCHK(ctx, hipMemcpyAsync(d + kChainOffBluePQ, pq, 512 * 4, hipMemcpyHostToDevice, ctx->stream));
        ctx->blue_factors_ok = true;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->chain_ready = true;
    return SARPRO_HIP_OK;
}

static bool chain_eligible(const U16Job &J) {
    if (J.ctx->attrs.on(A_NO_CHAIN)) return false;
    // u16 output (no rescale, no composition) takes the same chain up to the blend, with the exact f64 kernel
    return J.clahe() && (J.u8_out() || !J.synrgb) && J.vec && !J.tamed_force && (J.reduce || (J.row0 == 0 && J.rows_local == J.rows_total));
}

static int chain_stats_scratch(sarpro_hip_ctx *ctx, ChainStatsArgs *sa) {
    const size_t part_bytes = (sizeof(ChainStatsPartial) * kChainStatsParts * kMaxBands + 255) & ~(size_t)255;
    HIPCHK(ctx, ctx->chain_scratch.reserve(part_bytes + sizeof(uint64_t) * 4096 * kMaxBands));
    sa->partials = ctx->chain_scratch.as<ChainStatsPartial>();
    sa->bins4096 = reinterpret_cast<unsigned long long *>(ctx->chain_scratch.as<uint8_t>() + part_bytes);
    return SARPRO_HIP_OK;
}

// row-stripe mode: merge a small integer buffer across ranks without leaving the stream
static int chain_reduce(U16Job &J, void *d_buf, size_t count_u64, const char *what) {
    if (!J.reduce) return SARPRO_HIP_OK;
    KernelTimer t(J.ctx, what);
    return comm_allreduce_sum_u64_async(J.ctx, reinterpret_cast<uint64_t *>(d_buf), count_u64);
}

// End of a device-resident chain: return at once on a stream-ordered context, else read the statistics back.
static int chain_tail(U16Job &J, sarpro_hip_stats *stats_out, ChainBandState *d_state) {
    sarpro_hip_ctx *ctx = J.ctx;
    if (ctx->async_dev && J.allow_async && !stats_out && !J.reduce) { // stream-ordered: nothing is read back, the LDS capacity keeps its value
        ctx->async_pending = ctx->timing;
        return SARPRO_HIP_OK;
    }
    ChainBandState *h_state = ctx->h_small.as<ChainBandState>();
    HIPCHK(ctx, hipMemcpyAsync(h_state, d_state, sizeof(ChainBandState) * (size_t)J.nbands, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // the only synchronisation of the chain
    uint32_t hi = 0;
    for (int b = 0; b < J.nbands; ++b) {
        J.stats[b] = h_state[b].stats;
        if (stats_out) stats_out[b] = J.stats[b];
        hi = std::max(hi, h_state[b].win_hi);
    }
    // size the LDS offset table of the NEXT scene from this scene's window (speed only: a window larger
    // than the capacity is gathered from global memory, with identical results)
    ctx->chain_lut_cap = std::min<uint32_t>(16384, std::max<uint32_t>(1024, (hi + 1 + 255) / 256 * 256));
    return SARPRO_HIP_OK;
}

// Scenes below this size keep the exact partial histogram: their chain is launch-bound, the gated kernels would cost more than
// the sampled histogram saves.  SARPRO_HIP_SAMPLED_HIST_MIN_PX overrides (the tests run the speculative chain on small rasters).
constexpr size_t kSampledHistMinPx = 32u << 20;
static uint32_t spec_force_flags(const sarpro_hip_ctx *ctx) { // SPEC_FORCE = mispredict (1) | nospec (2) | predicted lowest level + 1 (4): every rare branch of the speculative chain is testable
    return (uint32_t)ctx->attrs.val(A_SPEC_FORCE, 0) & (kSpecForceMispredict | kSpecForceNoSpec | kSpecForceMinMispredict | kSpecForceNoRetry | kSpecForceMispredict2);
}

static int chain_tail(U16Job &J, sarpro_hip_stats *stats_out, ChainBandState *d_state);

struct FusedTail {
    sarpro_hip_ctx *ctx; U16Job *J; ClaheRgbArgs fa; ClaheApplyArgs a; ChainSpecState *d_spec; ChainBandState *d_state;
    unsigned long long *exact_hist; StripePlan *plan; uint8_t *d_levels[2]; size_t lvl_pitch; uint32_t rows, cols;
    unsigned long long total_px; uint8_t *d_rgb; size_t rgb_pitch_px;
    ChainPredictArgs pa; // (the prediction's arguments: its second launch, behind an undercut lowest level, takes the same)
};
// The fused pass and what is gated on its verdict (job_run_fused_rgb's second half).  T.J is null when the call is deferred (never a row stripe).
static int fused_rgb_tail(const FusedTail &T) {
    sarpro_hip_ctx *ctx = T.ctx;
    uint8_t *consts = ctx->chain_consts.as<uint8_t>(), *state = ctx->chain_state.as<uint8_t>();
    ChainSpecState *d_spec = T.d_spec;
    ClaheApplyArgs a = T.a;
    const bool reduce = T.J && T.J->reduce;
    {
        // resident batch (pipeline.cpp): this scene's pass behind the previous scene's pass (another lane's stream), its own completion
        // published for the next one -- the passes own whole compute units (160 KiB of LDS each), two of them at once only split the chip
        if (ctx->pipe_wait_before_fused) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_wait_before_fused, 0));
        if (ctx->pipe_record_before_fused) { HIPCHK(ctx, hipEventRecord(ctx->pipe_record_before_fused, ctx->stream)); ctx->pipe_record_before_fused = nullptr; }
        int grid = std::max(ctx->cu_count, 1);
        if (ctx->attrs.is_set(A_RGB_GRID)) grid = (int)std::min<long long>(1024, std::max<long long>(1, ctx->attrs.val(A_RGB_GRID, grid)));
        {
            KernelTimer t(ctx, "clahe_rgb_fused");
            HIPCHK(ctx, launch_clahe_rgb_fused(T.fa, grid, ctx->stream));
        }
        if (ctx->pipe_record_after_fused) {
            HIPCHK(ctx, hipEventRecord(ctx->pipe_record_after_fused, ctx->stream));
            ctx->pipe_record_after_fused = nullptr; // recorded (the batch records it itself behind a chain that never got here)
        }
    }
    if (reduce) { // the verification counts of all stripes, then the verdict every rank shares
        static_assert(offsetof(ChainSpecState, n_below_min) == offsetof(ChainSpecState, n_lt) + 16, "the verification counts are one buffer");
        RETCHK(chain_reduce(*T.J, &d_spec->n_lt[0], 3, "allreduce_spec_counts"));
        HIPCHK(ctx, launch_spec_verdict(d_spec, T.d_state, ctx->stream));
    }
    {   // a refuted floor gets one second pass with the floor the first pass's counts point to (both launches return at once otherwise)
        ChainRepredictArgs ra{};
        ra.spec = d_spec; ra.resc_in = state + kStateOffResc; ra.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        ra.tables = ctx->tables.as<uint8_t>(); ra.supp_rg = consts + kChainOffSupp; ra.blue_pair_supp = consts + kChainOffBlue;
        ra.blue_pq = ctx->blue_factors_ok ? reinterpret_cast<const float *>(consts + kChainOffBluePQ) : nullptr;
        ra.blue_by_level = reinterpret_cast<float *>(ctx->tables.as<uint8_t>() + kTablesOffPQ);
        ra.stripes = reduce ? 1u : 0u;
        {
            KernelTimer t(ctx, "chain_repredict");
            HIPCHK(ctx, launch_chain_repredict(ra, ctx->stream));
            if (!reduce) { // an undercut lowest level whose true value the pass recorded: the prediction again, on that level (one device)
                ChainPredictArgs p2 = T.pa;
                p2.second = 1u;
                HIPCHK(ctx, launch_chain_predict(p2, ctx->stream));
            }
        }
        ClaheRgbArgs fr = T.fa;
        fr.retry = 1u;
        int grid = std::max(ctx->cu_count, 1);
        if (ctx->attrs.is_set(A_RGB_GRID)) grid = (int)std::min<long long>(1024, std::max<long long>(1, ctx->attrs.val(A_RGB_GRID, grid)));
        {
            KernelTimer t(ctx, "clahe_rgb_fused_retry");
            HIPCHK(ctx, launch_clahe_rgb_fused_retry(fr, grid, ctx->stream));
        }
        if (reduce) {
            RETCHK(chain_reduce(*T.J, &d_spec->n_lt[0], 3, "allreduce_spec_counts_retry"));
            HIPCHK(ctx, launch_spec_verdict(d_spec, T.d_state, ctx->stream, 1));
        }
    }
    {   // gated on the verdict: levels of every pixel with the full histogram -> exact tables -> composition
        KernelTimer t(ctx, "spec_fallback_apply");
        a.hist_mode = 0u; a.gate = d_spec;
        a.rects = T.plan->d_apply_rects.as<Rect>();
        for (int b = 0; b < 2; ++b) a.level_hist[b] = T.exact_hist + (size_t)b * 256;
        HIPCHK(ctx, launch_clahe_apply_u8_spec(a, (int)T.plan->apply_rects.size(), 2, ctx->stream));
    }
    if (reduce) RETCHK(chain_reduce(*T.J, T.exact_hist, (size_t)256 * kMaxBands, "allreduce_level_hist")); // (all zero when the fused RGB stood: the gated recount did not run)
    {
        ChainFinishArgs f{};
        f.level_hist = T.exact_hist; f.gate = d_spec;
        f.total_px = T.total_px; f.nbands = 2;
        f.resc_out = state + kStateOffResc; f.identity_out = state + kStateOffIdent;
        f.tables = ctx->tables.as<uint8_t>();
        f.supp_rg = consts + kChainOffSupp; f.blue_pair_supp = consts + kChainOffBlue;
        f.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        f.suppressed = 1;
        KernelTimer t(ctx, "chain_finish");
        HIPCHK(ctx, launch_chain_finish(f, ctx->stream));
    }
    {
        ComposeArgs c{};
        c.b1 = T.d_levels[0]; c.b2 = T.d_levels[1]; c.in_pitch = T.lvl_pitch;
        c.rgb = T.d_rgb; c.rgb_pitch_px = T.rgb_pitch_px; c.rows = T.rows; c.cols = T.cols;
        c.tables = ctx->tables.as<uint8_t>();
        c.spec = d_spec; c.speculative = 0;
        KernelTimer t(ctx, "spec_fallback_compose");
        HIPCHK(ctx, launch_compose_u8(c, 16, ctx->stream));
    }
    return SARPRO_HIP_OK;
}

// The fused CLAHE -> RGB route of job_run_chain, from the CDFs on (everything before it is shared with the other routes).
static int job_run_fused_rgb(U16Job &J, const ClaheRgbArgs &fa, uint8_t *d_rgb, size_t rgb_pitch_px, uint32_t sample_stride,
                             sarpro_hip_stats *stats_out) {
    sarpro_hip_ctx *ctx = J.ctx;
    uint8_t *consts = ctx->chain_consts.as<uint8_t>(), *state = ctx->chain_state.as<uint8_t>();
    ChainBandState *d_state = reinterpret_cast<ChainBandState *>(state);
    ChainSpecState *d_spec = ctx->spec_state.as<ChainSpecState>();
    const uint32_t rows = (uint32_t)J.rows_local, cols = (uint32_t)J.cols;
    RETCHK(ensure_levels(J)); // the fallback's level rasters (allocated once per shape; untouched when the fused RGB stands)
    HIPCHK(ctx, ctx->spec_dump.reserve(kSpecDumpBytes));
    unsigned long long *sample_hist = ctx->level_hist.as<unsigned long long>(), *exact_hist = sample_hist + 256 * kMaxBands * kSampleReplicas;
    ctx->last_final_hist = exact_hist;
    ClaheApplyArgs a{};
    for (int b = 0; b < 2; ++b) {
        a.in[b] = J.d_in[b];
        a.out[b] = J.d_levels[b];
        a.cdfs[b] = fa.cdfs[b];
        a.binlut[b] = fa.binlut[b];
        a.level_hist[b] = sample_hist + (size_t)b * 256;
    }
    a.in_pitch = J.in_pitch; a.out_pitch = J.lvl_pitch;
    a.row_w = fa.row_w; a.col_w = fa.col_w; a.row_off = fa.row_off;
    a.max_val = 255.0; a.dev_state = d_state; a.lut_cap = ctx->chain_lut_cap;
    a.dump = ctx->spec_dump.as<uint8_t>();
    a.sample_stride = sample_stride; a.sample_phase = sample_stride / 2; a.sample_valid = d_spec->sample_valid_rep;
    {   // the sampled rows of both bands through the blend: level histogram + valid counts, nothing stored
        a.hist_mode = 3u;
        a.rects = J.plan->d_sample_rects.as<Rect>();
        KernelTimer t(ctx, "clahe_sample");
        HIPCHK(ctx, launch_clahe_apply_u8_spec(a, (int)J.plan->sample_rects.size(), 2, ctx->stream));
    }
    // row stripes: the sample of the SCENE (every replica of the sampled histogram, the valid counts beside them)
    RETCHK(chain_reduce(J, sample_hist, (size_t)256 * kMaxBands * kSampleReplicas, "allreduce_sample_hist"));
    RETCHK(chain_reduce(J, d_spec->sample_valid_rep, (size_t)kSampleReplicas * 2, "allreduce_sample_valid"));
    ChainPredictArgs pa_first{};
    {
        ChainPredictArgs pa{};
        pa.sample_hist = sample_hist; pa.exact_hist = exact_hist; pa.spec = d_spec; pa.state = d_state;
        pa.total_px = (unsigned long long)J.rows_total * J.cols;
        pa.resc_out = state + kStateOffResc; pa.identity_out = state + kStateOffIdent;
        pa.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        pa.tables = ctx->tables.as<uint8_t>();
        pa.supp_rg = consts + kChainOffSupp; pa.blue_pair_supp = consts + kChainOffBlue;
        pa.force = spec_force_flags(ctx);
        pa.allow_rescaled = ctx->attrs.on(A_NO_SPEC_RESCALE) ? 0u : 1u; // the fused pass verifies a predicted lowest level
        pa.blue_pq = ctx->blue_factors_ok ? reinterpret_cast<const float *>(consts + kChainOffBluePQ) : nullptr;
        pa.blue_by_level = reinterpret_cast<float *>(ctx->tables.as<uint8_t>() + kTablesOffPQ);
        pa_first = pa;
        KernelTimer t(ctx, "chain_predict");
        HIPCHK(ctx, launch_chain_predict(pa, ctx->stream));
    }
    // everything from the fused pass on: at once, or -- resident batch with PIPE_ORDER = 3 -- when the batch driver says so (the next
    // scene's histogram sweep is enqueued on another lane FIRST, so that this pass can wait for it: an event must be recorded
    // before a stream can be made to wait for it)
    FusedTail T{};
    T.ctx = ctx; T.J = &J; T.fa = fa; T.a = a; T.d_spec = d_spec; T.d_state = d_state; T.exact_hist = exact_hist;
    T.plan = J.plan; T.d_levels[0] = J.d_levels[0]; T.d_levels[1] = J.d_levels[1]; T.lvl_pitch = J.lvl_pitch;
    T.rows = rows; T.cols = cols; T.total_px = (unsigned long long)J.rows_total * J.cols; T.d_rgb = d_rgb; T.rgb_pitch_px = rgb_pitch_px;
    T.pa = pa_first;
    if (ctx->pipe_defer && !J.reduce && ctx->async_dev && J.allow_async && !stats_out) {
        T.J = nullptr; // (the job object is the caller's: gone when the tail runs)
        ctx->pipe_deferred = [T]() { return fused_rgb_tail(T); };
        ctx->async_pending = ctx->timing; // what chain_tail does on a stream-ordered context
        return SARPRO_HIP_OK;
    }
    RETCHK(fused_rgb_tail(T));
    return chain_tail(J, stats_out, d_state);
}

static int job_run_chain(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px,
                         sarpro_hip_stats *stats_out) {
    sarpro_hip_ctx *ctx = J.ctx;
    RETCHK(chain_prepare(ctx));
    const uint32_t rows = (uint32_t)J.rows_local, cols = (uint32_t)J.cols;
    HIPCHK(ctx, ctx->luts.reserve(2 * 131072));
    HIPCHK(ctx, ctx->tile_bins.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands));
    HIPCHK(ctx, ctx->cdfs.reserve(sizeof(double) * 64 * 256 * kMaxBands));
    // [kSampleReplicas][2][256]: the apply / sampling pass's histogram (replica 0 alone unless it is sampled); then [2][256]: the gated recount
    HIPCHK(ctx, ctx->level_hist.reserve(sizeof(uint64_t) * 256 * kMaxBands * (kSampleReplicas + 1)));
    HIPCHK(ctx, ctx->tables.reserve(kTablesBytes));
    HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands));
    uint8_t *consts = ctx->chain_consts.as<uint8_t>(), *state = ctx->chain_state.as<uint8_t>();
    ChainBandState *d_state = reinterpret_cast<ChainBandState *>(state);
    // Dual-pol u8 scene on one device: the level histogram is counted on sampled rows only and the composition is speculative
    // (k_chain_predict); SARPRO_HIP_NO_SAMPLED_HIST=1 keeps the partial histogram of every row and the unconditional tail.
    const bool exact_only = ctx->attrs.on(A_NO_SPEC); // cross-check: every pixel through the exact f64 blend
    const size_t rgb_pitch_ok = rgb_pitch_px % 16 == 0 && ptr_aligned16(d_rgb);
    // (a row stripe takes the speculative route in its fused form only -- the sampled histogram, the valid counts and the pass's
    // verification counts are summed over the ranks, so every rank proves, predicts and decides the same; decided below from
    // what all ranks share: an empty stripe still joins every reduction)
    const bool whole = J.row0 == 0 && J.rows_local == J.rows_total;
    const bool fused_wanted = !d_out[0] && !d_out[1] && !ctx->attrs.on(A_NO_FUSED_RGB) && J.in_pitch % 8 == 0;
    bool sampled = J.synrgb && J.nbands == 2 && J.u8_out() && !exact_only && !ctx->attrs.on(A_FULL_LEVEL_HIST) &&
                   !ctx->attrs.on(A_NO_SAMPLED_HIST) && rgb_pitch_px % 16 == 0 && (J.reduce ? fused_wanted : (d_rgb && rgb_pitch_ok && whole));
    // every 17th row; every 33rd on scenes of 12000 rows and more (606 sampled rows at 20000).  The sample pass is 5 % of the fused pass's work
    // and, in a resident batch, runs beside another lane's histogram sweep: nine-scene cycle on three lanes 0.967 -> 0.949 ms per scene with
    // 33, 0.944 with 65 (all nine accepted either way); the estimate's error on these scenes is a bias of the row phase against the scene's
    // structure (1-2.5 % of a level's population at 9, 17, 33 and 65 alike: profiles/r3/spec_accuracy.txt), its random part grows with sqrt(stride)
    uint32_t sample_stride = J.rows_total >= 12000 ? 33 : 17;
    if (sampled) {
        size_t min_px = kSampledHistMinPx;
        if (ctx->attrs.is_set(A_SAMPLED_HIST_MIN_PX)) min_px = (size_t)std::max<long long>(0, ctx->attrs.val(A_SAMPLED_HIST_MIN_PX, 0));
        if (ctx->attrs.is_set(A_SAMPLE_STRIDE)) sample_stride = (uint32_t)std::max<long long>(5, ctx->attrs.val(A_SAMPLE_STRIDE, 0));
        if ((size_t)J.rows_total * J.cols < min_px) sampled = false;
    }
    ChainSpecState *d_spec = nullptr;
    ctx->spec_ran = sampled;
    if (sampled) {
        HIPCHK(ctx, ctx->spec_state.reserve(sizeof(ChainSpecState)));
        d_spec = ctx->spec_state.as<ChainSpecState>();
    }

    RETCHK(job_phase1(J)); // per-tile DN histograms -> ctx->ghist
    RETCHK(chain_reduce(J, ctx->ghist.p, 65536 * (size_t)J.nbands, "allreduce_dn_hist"));
    {
        ChainStatsArgs sa{};
        sa.ghist = ctx->ghist.as<unsigned long long>();
        sa.db = reinterpret_cast<const double *>(consts + kChainOffDb);
        sa.state = d_state;
        sa.binlut = ctx->luts.as<uint8_t>();
        sa.binlut_stride = 131072;
        sa.level_hist = ctx->level_hist.as<unsigned long long>(); // cleared here for the apply kernel (one fill kernel less)
        sa.sample_valid = d_spec ? d_spec->sample_valid_rep : nullptr;
        KernelTimer t(ctx, "chain_stats");
        RETCHK(chain_stats_scratch(ctx, &sa));
        HIPCHK(ctx, launch_chain_stats(sa, J.nbands, ctx->stream));
    }
    {
        TileBinHistArgs ta{};
        for (int b = 0; b < J.nbands; ++b) {
            ta.tile_hist[b] = tile_hist_of(ctx, b, kTiles * kTiles);
            ta.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
            ta.out[b] = ctx->tile_bins.as<unsigned long long>() + (size_t)b * 64 * 256;
        }
        ta.clear = 1u; // the last reader of the tile histograms
        if (!J.reduce && !ctx->attrs.on(A_SEPARATE_CDFS)) { // one device: the CDFs in the same launch
            for (int b = 0; b < J.nbands; ++b) ta.cdfs_out[b] = ctx->cdfs.as<double>() + (size_t)b * 64 * 256;
            ta.rows = (uint32_t)J.rows_total; ta.cols = cols;
        }
        KernelTimer t(ctx, "tile_bin_hist");
        HIPCHK(ctx, launch_tile_bin_hist(ta, kTiles * kTiles, J.nbands, ctx->stream));
        mark_tile_hist_clean(J);
    }
    RETCHK(chain_reduce(J, ctx->tile_bins.p, 64 * 256 * (size_t)J.nbands, "allreduce_tile_hists"));
    if (J.reduce || ctx->attrs.on(A_SEPARATE_CDFS)) {
        KernelTimer t(ctx, "chain_cdfs");
        HIPCHK(ctx, launch_chain_cdfs(ctx->tile_bins.as<unsigned long long>(), ctx->cdfs.as<double>(), (uint32_t)J.rows_total, cols,
                                      J.nbands, ctx->stream));
    }
    // Whole dual-pol u8 scene, RGB only: the fused pass (kernels.hip 6a) -- sample-only pass -> identity proof + predicted floor +
    // tables -> ONE sweep DN, DN -> RGB that verifies the floor; refuted (or unproven, or windows beyond the pass's LDS pool), the
    // gated apply -> finish -> compose kernels below produce the raster.  SARPRO_HIP_NO_FUSED_RGB=1: the apply + compose route.
    if (sampled && fused_wanted && (J.reduce || !J.plan->rgb_rects.empty())) {
        ClaheRgbArgs fa{};
        for (int b = 0; b < 2; ++b) {
            fa.in[b] = J.d_in[b];
            fa.cdfs[b] = ctx->cdfs.as<double>() + (size_t)b * 64 * 256;
            fa.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
        }
        fa.in_pitch = J.in_pitch; fa.rgb = d_rgb; fa.rgb_pitch_px = rgb_pitch_px;
        fa.rects = J.plan->d_rgb_rects.as<Rect>(); fa.nrects = (int)J.plan->rgb_rects.size();
        fa.row_w = J.plan->d_row_w.as<RowWeight>(); fa.col_w = J.plan->d_col_w.as<RowWeight>(); fa.row_off = (int32_t)J.row0;
        fa.dev_state = d_state; fa.spec = d_spec; fa.tables = ctx->tables.as<uint8_t>();
        fa.blue_by_level = ctx->blue_factors_ok ? reinterpret_cast<const float *>(ctx->tables.as<uint8_t>() + kTablesOffPQ) : nullptr;
        fa.sat_ok = J.plan->sat_ok ? 1u : 0u; fa.sat_col = J.plan->d_sat_col.as<uint8_t>(); fa.sat_row = J.plan->d_sat_row.as<uint8_t>() + J.row0; // (the table is indexed by the scene's row, the kernel by the stripe's)
        fa.sat_cols = (uint32_t)(round_up(J.cols, 64) + 64);
        // the sample-only pass costs ~0.025 ms + (apply pass) / stride: 0.056 ms at 17, 0.033 at 33; the wider stride's larger sigma (x 1.4: ~2.4 %
        // of scenes refuted instead of ~1.7 %, 1 ms each) costs 0.007 ms in expectation
        const uint32_t fused_stride = ctx->attrs.is_set(A_SAMPLE_STRIDE) ? sample_stride : 33u;
        fa.no_verdict = J.reduce ? 1u : 0u;
        if (clahe_rgb_fused_supported(fa)) return job_run_fused_rgb(J, fa, d_rgb, rgb_pitch_px, fused_stride, stats_out);
        if (J.reduce) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "row stripe: the rasters of every rank must be 16-byte aligned (pitches % 8 / % 16)");
    }
    // apply: levels into the internal rasters (dual-pol) or straight into the caller's raster (single band)
    const bool direct = !J.synrgb;
    if (!direct) RETCHK(ensure_levels(J));
    ClaheApplyArgs a{};
    for (int b = 0; b < J.nbands; ++b) {
        a.in[b] = J.d_in[b];
        a.out[b] = direct ? d_out[b] : (void *)J.d_levels[b];
        a.cdfs[b] = ctx->cdfs.as<double>() + (size_t)b * 64 * 256;
        a.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
        a.level_hist[b] = ctx->level_hist.as<unsigned long long>() + (size_t)b * 256;
    }
    a.in_pitch = J.in_pitch;
    a.out_pitch = direct ? out_pitch : J.lvl_pitch;
    a.rects = J.plan->d_apply_rects.as<Rect>();
    a.row_w = J.plan->d_row_w.as<RowWeight>();
    a.col_w = J.plan->d_col_w.as<RowWeight>();
    a.row_off = (int32_t)J.row0;
    const bool u16o = !J.u8_out();
    a.max_val = u16o ? 65535.0 : 255.0;
    a.dev_state = d_state;
    a.lut_cap = ctx->chain_lut_cap;
    if (a.out_pitch % 8 != 0 || !ptr_aligned16(a.out[0]) || (J.nbands > 1 && !ptr_aligned16(a.out[1])))
        return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "output raster must be 16-byte aligned with pitch % 8 == 0 when the input is");
    if (u16o) { // autoscale.rs:600-606 at max_val 65535: no u8 rescale, nothing downstream needs a level histogram
        for (int b = 0; b < J.nbands; ++b) a.level_hist[b] = nullptr;
        {
            KernelTimer t(ctx, "clahe_apply_u16");
            if (!ctx->attrs.on(A_NO_U16_CF) && J.nbands <= kMaxBands && J.plan->u16_nwg[J.nbands - 1] > 0 &&
                clahe_apply_u16_cf_supported(a, J.nbands, J.plan->u16_item_rows)) { // the conflict-free form: one persistent workgroup per share
                a.rects = J.plan->d_u16_items[J.nbands - 1].as<Rect>();
                HIPCHK(ctx, launch_clahe_apply_u16_cf(a, J.plan->d_u16_first[J.nbands - 1].as<int32_t>(), J.plan->u16_nwg[J.nbands - 1], J.nbands, ctx->stream));
            } else {
                HIPCHK(ctx, launch_clahe_apply_u16(a, (int)J.plan->apply_rects.size(), J.nbands, true, true, ctx->stream));
            }
        }
        ChainBandState *h_state = ctx->h_small.as<ChainBandState>();
        HIPCHK(ctx, hipMemcpyAsync(h_state, d_state, sizeof(ChainBandState) * (size_t)J.nbands, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        uint32_t hi = 0;
        for (int b = 0; b < J.nbands; ++b) {
            J.stats[b] = h_state[b].stats;
            if (stats_out) stats_out[b] = J.stats[b];
            hi = std::max(hi, h_state[b].win_hi);
        }
        ctx->chain_lut_cap = std::min<uint32_t>(16384, std::max<uint32_t>(1024, (hi + 1 + 255) / 256 * 256));
        return SARPRO_HIP_OK;
    }
    // (the level histogram was cleared by the statistics kernels)
    // whole scene on this device: levels >= 64 are only counted in bulk (chain_kernels.hip k_level_hist_guard); a row
    // stripe keeps the full histogram, which is what the ranks sum
    a.hist_mode = sampled ? 2u : (!J.reduce && !exact_only && !ctx->attrs.on(A_FULL_LEVEL_HIST)) ? 1u : 0u;
    if (sampled) {
        a.sample_stride = sample_stride;
        a.sample_phase = sample_stride / 2; // mid-phase: the row weights of the sampled rows average to those of all rows
        a.sample_valid = d_spec->sample_valid_rep;
    }
    if (exact_only) {
        KernelTimer t(ctx, "clahe_apply_u16");
        HIPCHK(ctx, launch_clahe_apply_u16(a, (int)J.plan->apply_rects.size(), J.nbands, true, false, ctx->stream));
    } else {
        HIPCHK(ctx, ctx->spec_dump.reserve(kSpecDumpBytes));
        a.dump = ctx->spec_dump.as<uint8_t>();
        KernelTimer t(ctx, "clahe_apply_u8_spec");
        HIPCHK(ctx, launch_clahe_apply_u8_spec(a, (int)J.plan->apply_rects.size(), J.nbands, ctx->stream));
    }
    if (a.hist_mode == 1) {
        KernelTimer t(ctx, "level_hist_guard");
        HIPCHK(ctx, ctx->hist_flags.reserve(sizeof(uint32_t) * kMaxBands));
        HIPCHK(ctx, launch_level_hist_guard(ctx->level_hist.as<unsigned long long>(), (unsigned long long)J.rows_total * J.cols, J.nbands,
                                            ctx->hist_flags.as<uint32_t>(), ctx->stream));
        LevelRecountArgs ra{};
        for (int b = 0; b < J.nbands; ++b) ra.levels[b] = reinterpret_cast<const uint8_t *>(a.out[b]);
        ra.pitch = a.out_pitch; ra.rows = (uint32_t)J.rows_local; ra.cols = cols;
        ra.level_hist = ctx->level_hist.as<unsigned long long>(); ra.flags = ctx->hist_flags.as<uint32_t>();
        HIPCHK(ctx, launch_level_hist_if_flagged(ra, J.nbands, ctx->stream));
    }
    RETCHK(chain_reduce(J, ctx->level_hist.p, 256 * kMaxBands, "allreduce_level_hist"));
    ComposeArgs c{};
    int cvec = 1;
    if (J.synrgb) {
        c.b1 = J.d_levels[0]; c.b2 = J.d_levels[1]; c.in_pitch = J.lvl_pitch;
        c.rgb = d_rgb; c.rgb_pitch_px = rgb_pitch_px; c.rows = rows; c.cols = cols;
        c.tables = ctx->tables.as<uint8_t>();
        cvec = (c.in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 && ptr_aligned16(c.b1) && ptr_aligned16(c.b2) && ptr_aligned16(d_rgb)) ? 16 : 1;
    }
    unsigned long long *final_hist = ctx->level_hist.as<unsigned long long>();
    if (sampled) { // identity proof + predicted floor + tables -> speculative composition (counts and verdict) -> gated exact recount
        if (cvec != 16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "sampled level histogram without the vector compose pass");
        ChainPredictArgs pa{};
        pa.sample_hist = ctx->level_hist.as<unsigned long long>();
        pa.exact_hist = final_hist = ctx->level_hist.as<unsigned long long>() + 256 * kMaxBands * kSampleReplicas;
        pa.spec = d_spec;
        pa.state = d_state;
        pa.total_px = (unsigned long long)J.rows_total * J.cols;
        pa.resc_out = state + kStateOffResc;
        pa.identity_out = state + kStateOffIdent;
        pa.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        pa.tables = ctx->tables.as<uint8_t>();
        pa.supp_rg = consts + kChainOffSupp;
        pa.blue_pair_supp = consts + kChainOffBlue;
        pa.force = spec_force_flags(ctx);
        {
            KernelTimer t(ctx, "chain_predict");
            HIPCHK(ctx, launch_chain_predict(pa, ctx->stream));
        }
        c.spec = d_spec;
        c.speculative = 1;
        {
            KernelTimer t(ctx, "compose_u8");
            HIPCHK(ctx, launch_compose_u8(c, 16, ctx->stream));
        }
        c.speculative = 0; // the composition below is the gated fallback
        KernelTimer t(ctx, "spec_fallback_recount");
        LevelRecountArgs ra{};
        for (int b = 0; b < J.nbands; ++b) ra.levels[b] = reinterpret_cast<const uint8_t *>(a.out[b]);
        ra.pitch = a.out_pitch; ra.rows = (uint32_t)J.rows_local; ra.cols = cols;
        ra.level_hist = final_hist; ra.gate = d_spec;
        HIPCHK(ctx, launch_level_hist_if_flagged(ra, J.nbands, ctx->stream));
    }
    {
        ChainFinishArgs fa{};
        fa.level_hist = final_hist;
        ctx->last_final_hist = final_hist;
        fa.gate = d_spec;
        fa.total_px = (unsigned long long)J.rows_total * J.cols;
        fa.nbands = J.nbands;
        fa.resc_out = state + kStateOffResc;
        fa.identity_out = state + kStateOffIdent;
        fa.tables = J.synrgb ? ctx->tables.as<uint8_t>() : nullptr;
        fa.supp_rg = consts + kChainOffSupp;
        fa.blue_pair_supp = consts + kChainOffBlue;
        fa.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        fa.suppressed = 1; // CLAHE always composes with the suppressed variant (synthetic_rgb.rs:188-194)
        KernelTimer t(ctx, "chain_finish");
        HIPCHK(ctx, launch_chain_finish(fa, ctx->stream));
    }
    if (J.synrgb) {
        {
            KernelTimer t(ctx, sampled ? "spec_fallback_compose" : "compose_u8");
            HIPCHK(ctx, launch_compose_u8(c, cvec, ctx->stream));
        }
        for (int b = 0; b < 2; ++b) // optional per-band u8 rasters: levels through the band's rescale
            if (d_out[b])
                HIPCHK(ctx, launch_chain_remap(J.d_levels[b], J.lvl_pitch, reinterpret_cast<uint8_t *>(d_out[b]), out_pitch, rows, cols,
                                               state + kStateOffResc + (size_t)b * 256, nullptr, ctx->stream));
    } else {
        // single band: the apply pass wrote levels into the caller's raster; rescale in place unless it is the identity
        HIPCHK(ctx, launch_chain_remap(reinterpret_cast<uint8_t *>(d_out[0]), out_pitch, reinterpret_cast<uint8_t *>(d_out[0]), out_pitch, rows,
                                       cols, state + kStateOffResc, state + kStateOffIdent, ctx->stream));
    }
    return chain_tail(J, stats_out, d_state);
}

// Device-resident chain for the percentile strategies, dual-pol, RGB only: histogram -> statistics + window +
// u8 level of every DN + level histogram (k_chain_stats, levels mode) -> rescale / floor / tables and DN -> final
// u8 tables (k_chain_finish) -> ONE fused pass DN,DN -> RGB (k_lut_compose_u16).  No host synchronisation in
// between; gamma != 1 is resolved against host-built thresholds, so no pow runs on the device.
// dual-pol -> RGB (fused pass) and / or per-band u8 rasters (table pass), all with device-built tables
static bool chain_levels_eligible(const U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, const uint8_t *d_rgb,
                                  size_t rgb_pitch_px) {
    if (J.ctx->attrs.on(A_NO_CHAIN)) return false;
    if (J.clahe() || !J.vec || !(J.reduce || (J.row0 == 0 && J.rows_local == J.rows_total))) return false;
    if (!J.u8_out() && J.synrgb) return false; // u16 levels: per-band rasters only
    if (J.synrgb && !(J.nbands == 2 && J.in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 && ptr_aligned16(d_rgb))) return false;
    bool any_out = false;
    for (int b = 0; b < J.nbands; ++b)
        if (d_out[b]) { any_out = true; if (out_pitch % 8 != 0 || !ptr_aligned16(d_out[b])) return false; }
    return J.synrgb || any_out || J.tables_only;
}

constexpr int kRerunOnHostRoute = 1; // u16 levels with gamma != 1 that the device could not certify (see k_chain_stats_c)

static int job_run_chain_levels(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px,
                                sarpro_hip_stats *stats_out) {
    sarpro_hip_ctx *ctx = J.ctx;
    const int nb = J.nbands;
    const bool u16o = !J.u8_out();
    RETCHK(chain_prepare(ctx));
    HIPCHK(ctx, ctx->luts.reserve(2 * 131072));
    HIPCHK(ctx, ctx->level_hist.reserve(sizeof(uint64_t) * 256 * kMaxBands));
    HIPCHK(ctx, ctx->tables.reserve(kTablesBytes));
    HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands));
    uint8_t *consts = ctx->chain_consts.as<uint8_t>(), *state = ctx->chain_state.as<uint8_t>();
    ChainBandState *d_state = reinterpret_cast<ChainBandState *>(state);
    const bool suppressed = J.strategy == SARPRO_STRATEGY_TAMED; // CLAHE is not handled here

    J.clear_after_sum = true; // untiled: k_sum_tile_hists is the only reader of the histogram
    RETCHK(job_phase1(J)); // DN histograms -> ctx->ghist
    RETCHK(chain_reduce(J, ctx->ghist.p, 65536 * (size_t)nb, "allreduce_dn_hist"));
    {
        ChainStatsArgs sa{};
        sa.ghist = ctx->ghist.as<unsigned long long>();
        sa.db = reinterpret_cast<const double *>(consts + kChainOffDb);
        sa.state = d_state;
        sa.binlut = ctx->luts.as<uint8_t>();
        sa.binlut_stride = 131072;
        sa.levels_mode = u16o ? 2 : 1;
        sa.lut16 = ctx->luts.as<uint16_t>();
        sa.strategy = J.strategy;
        for (int b = 0; b < nb; ++b) sa.tamed_kind[b] = J.tamed_kind(b);
        sa.total_px = (unsigned long long)J.rows_total * J.cols;
        sa.level_hist = ctx->level_hist.as<unsigned long long>();
        sa.gamma_thr = reinterpret_cast<const double *>(consts + kChainOffGamma);
        KernelTimer t(ctx, "chain_stats");
        RETCHK(chain_stats_scratch(ctx, &sa));
        HIPCHK(ctx, launch_chain_stats(sa, nb, ctx->stream));
    }
    if (!u16o) { // u16 levels have no rescale (autoscale.rs:689-703) and are never composed
        ChainFinishArgs fa{};
        fa.level_hist = ctx->level_hist.as<unsigned long long>();
        fa.total_px = (unsigned long long)J.rows_total * J.cols;
        fa.nbands = nb;
        fa.resc_out = state + kStateOffResc;
        fa.identity_out = state + kStateOffIdent;
        fa.tables = J.synrgb ? ctx->tables.as<uint8_t>() : nullptr; // no composition: only the DN -> final u8 tables
        fa.supp_rg = consts + kChainOffSupp;
        fa.blue_pair_supp = consts + kChainOffBlue;
        fa.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        fa.levels_mode = 1;
        for (int b = 0; b < nb; ++b) fa.no_rescale[b] = J.tamed_kind(b) != kNotTamedSynrgb;
        fa.suppressed = suppressed ? 1 : 0;
        fa.dn_tables = ctx->luts.as<uint8_t>();
        fa.dn_table_stride = 131072;
        fa.default_rg = consts + kChainOffDefRg;
        fa.blue_pair_default = consts + kChainOffBlueDef;
        KernelTimer t(ctx, "chain_finish");
        HIPCHK(ctx, launch_chain_finish(fa, ctx->stream));
    }
    for (int b = 0; b < nb; ++b) { // per-band u8 rasters: out = table[DN]
        if (!d_out[b]) continue;
        LutApplyArgs la{};
        la.in = J.d_in[b]; la.out = d_out[b]; la.in_pitch = J.in_pitch; la.out_pitch = out_pitch;
        la.rows = (uint32_t)J.rows_local; la.cols = (uint32_t)J.cols;
        la.lut = ctx->luts.as<uint8_t>() + (size_t)b * 131072; // u8 final values, or u16 levels (65536 x 2 bytes) at the same offset
        la.dev_state = d_state; la.band = b; la.lut_cap = ctx->chain_levels_cap;
        KernelTimer t(ctx, "lut_apply_u16");
        HIPCHK(ctx, launch_lut_apply_u16(la, true, u16o, ctx->stream));
    }
    if (J.synrgb) {
        LutComposeArgs f{};
        for (int b = 0; b < 2; ++b) { f.in[b] = J.d_in[b]; f.lut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072; }
        f.rgb = d_rgb; f.in_pitch = J.in_pitch; f.rgb_pitch_px = rgb_pitch_px;
        f.rows = (uint32_t)J.rows_local; f.cols = (uint32_t)J.cols;
        f.tables = ctx->tables.as<uint8_t>();
        f.dev_state = d_state;
        f.lut_cap = ctx->chain_levels_cap;
        KernelTimer t(ctx, "lut_compose_u16");
        HIPCHK(ctx, launch_lut_compose_u16(f, ctx->stream));
    }
    if (ctx->async_dev && J.allow_async && !stats_out && !u16o && !J.reduce) { // stream-ordered (u16 levels need their `uncertain` flag read back)
        ctx->async_pending = ctx->timing;
        return SARPRO_HIP_OK;
    }
    ChainBandState *h_state = ctx->h_small.as<ChainBandState>();
    HIPCHK(ctx, hipMemcpyAsync(h_state, d_state, sizeof(ChainBandState) * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // the only synchronisation of the chain
    uint32_t hi = 0;
    bool uncertain = false;
    for (int b = 0; b < nb; ++b) {
        J.stats[b] = h_state[b].stats;
        if (stats_out) stats_out[b] = J.stats[b];
        hi = std::max(hi, h_state[b].win_hi);
        uncertain = uncertain || (u16o && h_state[b].uncertain);
    }
    if (u16o && ctx->attrs.on(A_FORCE_UNCERTAIN)) uncertain = true; // test hook: exercise the rerun
    if (uncertain) return kRerunOnHostRoute;
    // LDS capacity (bytes per band) of the NEXT scene's DN tables (speed only)
    ctx->chain_levels_cap = std::min<uint32_t>(16384, std::max<uint32_t>(2048, (hi + 1 + 1023) / 1024 * 1024));
    return SARPRO_HIP_OK;
}

static int job_run_all(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px,
                       sarpro_hip_stats *stats_out) {
    timing_reset(J.ctx);
    J.ctx->spec_ran = false;
    RETCHK(job_init(J));
    if ((J.rows_local == 0 && !J.reduce) || J.cols == 0) { // a rank with an empty stripe still joins the reductions
        if (stats_out) std::memset(stats_out, 0, sizeof(*stats_out) * (size_t)J.nbands);
        return SARPRO_HIP_OK;
    }
    if (chain_eligible(J)) {
        HostTimer t(J.ctx, "host:chain(enqueue+final sync)");
        return job_run_chain(J, d_out, out_pitch, d_rgb, rgb_pitch_px, stats_out);
    }
    if (chain_levels_eligible(J, d_out, out_pitch, d_rgb, rgb_pitch_px)) {
        HostTimer t(J.ctx, "host:chain(enqueue+final sync)");
        const int rc = job_run_chain_levels(J, d_out, out_pitch, d_rgb, rgb_pitch_px, stats_out);
        if (rc != kRerunOnHostRoute) return rc;
    }
    { HostTimer t(J.ctx, "host:phase1_launch"); RETCHK(job_phase1(J)); }
    // (row-stripe mode without the device chain: the same phases with a synchronous all-reduce after each)
    if (J.reduce) RETCHK(sarpro_hip_comm_allreduce_sum_u64(J.ctx, J.ctx->ghist.as<uint64_t>(), 65536 * (size_t)J.nbands));
    { HostTimer t(J.ctx, "host:after_phase1(sync+stats+tables)"); RETCHK(job_after_phase1(J)); }
    { HostTimer t(J.ctx, "host:phase2_launch"); RETCHK(job_phase2(J)); }
    if (J.reduce && J.clahe()) RETCHK(sarpro_hip_comm_allreduce_sum_u64(J.ctx, J.ctx->tile_bins.as<uint64_t>(), 64 * 256 * (size_t)J.nbands));
    { HostTimer t(J.ctx, "host:phase3(sync+cdfs+launch)"); RETCHK(job_phase3(J, d_out, out_pitch)); }
    if (J.reduce && J.clahe() && J.u8_out()) RETCHK(sarpro_hip_comm_allreduce_sum_u64(J.ctx, J.ctx->level_hist.as<uint64_t>(), 256 * kMaxBands));
    { HostTimer t(J.ctx, "host:phase4(sync+tables+launch+sync)"); RETCHK(job_phase4(J, d_out, out_pitch, d_rgb, rgb_pitch_px, false)); }
    if (stats_out) for (int b = 0; b < J.nbands; ++b) stats_out[b] = J.stats[b];
    return SARPRO_HIP_OK;
}

} // namespace sarpro

// ---------------------------------------------------------------------------------------
// device-pointer entry points (u16)
// ---------------------------------------------------------------------------------------
extern "C" int sarpro_hip_autoscale_band_u16_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows, size_t cols,
                                                 size_t in_pitch, int strategy, int bit_depth, void *d_out,
                                                 size_t out_pitch, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if ((!d_in || !d_out) && rows * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (out_pitch < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "out_pitch < cols");
    U16Job J;
    J.ctx = ctx; J.nbands = 1; J.d_in[0] = d_in;
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = strategy; J.bit_depth = bit_depth;
    void *outs[kMaxBands] = {d_out, nullptr};
    return job_run_all(J, outs, out_pitch, nullptr, 0, stats_out);
}

namespace sarpro {
// nb = 1: one band (tamed: 0 / 1 copol / 2 crosspol, as band_u8_dev).  nb = 2: the two bands of a dual-pol product in ONE chain
// (one histogram pass, one statistics chain, one synchronisation; strategy Tamed: band 0 copol, band 1 crosspol; `tamed` unused).
int bands_u8_table_dev(sarpro_hip_ctx *ctx, const uint16_t *const d_in[], int nb, size_t rows, size_t cols, size_t in_pitch, int strategy, int tamed,
                       ResizeLutSrc *out) {
    for (int b = 0; b < nb; ++b) { out[b].lut = nullptr; out[b].dev_state = nullptr; out[b].band = b; out[b].lut_cap = 0; }
    U16Job J;
    J.ctx = ctx; J.nbands = nb;
    for (int b = 0; b < nb; ++b) J.d_in[b] = d_in[b];
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = (nb == 1 && tamed) ? SARPRO_STRATEGY_TAMED : strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.tamed_force = nb == 1 ? tamed : 0;
    J.tables_only = true;
    timing_reset(ctx);
    RETCHK(job_init(J));
    void *outs[kMaxBands] = {nullptr, nullptr};
    if (!rows || !cols || !chain_levels_eligible(J, outs, 0, nullptr, 0)) return SARPRO_HIP_OK;
    HostTimer t(ctx, "host:chain(enqueue+final sync)");
    const int rc = job_run_chain_levels(J, outs, 0, nullptr, 0, nullptr);
    if (rc == kRerunOnHostRoute) return SARPRO_HIP_OK;
    RETCHK(rc);
    for (int b = 0; b < nb; ++b) {
        out[b].lut = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
        out[b].dev_state = reinterpret_cast<const ChainBandState *>(ctx->chain_state.as<uint8_t>());
        out[b].lut_cap = (ctx->chain_levels_cap + 15u) & ~15u;
    }
    return SARPRO_HIP_OK;
}
int band_u8_table_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows, size_t cols, size_t in_pitch, int strategy, int tamed, ResizeLutSrc *out) {
    const uint16_t *const in[1] = {d_in};
    return bands_u8_table_dev(ctx, in, 1, rows, cols, in_pitch, strategy, tamed, out);
}

int band_u8_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows, size_t cols, size_t in_pitch, int strategy, int tamed,
                uint8_t *d_out, size_t out_pitch) {
    U16Job J;
    J.ctx = ctx; J.nbands = 1; J.d_in[0] = d_in;
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = tamed ? SARPRO_STRATEGY_TAMED : strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.tamed_force = tamed;
    void *outs[kMaxBands] = {d_out, nullptr};
    return job_run_all(J, outs, out_pitch, nullptr, 0, nullptr);
}
int band_u8_stripe_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t in_pitch,
                       int strategy, int tamed, uint8_t *d_out, size_t out_pitch) {
    U16Job J;
    J.ctx = ctx; J.nbands = 1; J.d_in[0] = d_in;
    J.rows_total = rows_total; J.row0 = row0; J.rows_local = rows_local; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = tamed ? SARPRO_STRATEGY_TAMED : strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.tamed_force = tamed;
    J.reduce = true;
    void *outs[kMaxBands] = {d_out, nullptr};
    return job_run_all(J, outs, out_pitch, nullptr, 0, nullptr);
}
} // namespace sarpro

extern "C" int sarpro_hip_dualpol_synrgb_u16_dev(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2,
                                                 size_t rows, size_t cols, size_t in_pitch, int strategy, int mode,
                                                 uint8_t *d_rgb, size_t rgb_pitch_px, uint8_t *d_u8_band1,
                                                 uint8_t *d_u8_band2, size_t u8_pitch, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if ((!d_band1 || !d_band2 || !d_rgb) && rows * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (rgb_pitch_px < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "rgb_pitch_px < cols");
    if ((d_u8_band1 == nullptr) != (d_u8_band2 == nullptr))
        return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pass both per-band u8 outputs or neither");
    if (d_u8_band1 && u8_pitch < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "u8_pitch < cols");
    U16Job J;
    J.ctx = ctx; J.nbands = 2; J.d_in[0] = d_band1; J.d_in[1] = d_band2;
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.mode = mode; J.synrgb = true;
    J.allow_async = true; // SARPRO_HIP_CTX_ASYNC_DEV applies to this entry point only
    void *outs[kMaxBands] = {d_u8_band1, d_u8_band2};
    return job_run_all(J, outs, u8_pitch, d_rgb, rgb_pitch_px, stats_out);
}

extern "C" int sarpro_hip_polop_f32_dev(sarpro_hip_ctx *ctx, int op, const float *d_a, const float *d_b, size_t n,
                                        float *d_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (op < SARPRO_OP_SUM || op > SARPRO_OP_LOGRATIO) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad polarization operation");
    if ((!d_a || !d_b || !d_out) && n) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null array");
    timing_reset(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        KernelTimer t(ctx, "polop_f32");
        HIPCHK(ctx, launch_polop_f32(op, d_a, d_b, n, d_out, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_synrgb_u8_dev(sarpro_hip_ctx *ctx, int mode, int strategy, const uint8_t *d_b1,
                                        const uint8_t *d_b2, size_t n, uint8_t *d_rgb) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (mode < 0 || mode > SARPRO_SYNRGB_ENHANCED) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synrgb mode");
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if ((!d_b1 || !d_b2 || !d_rgb) && n) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null array");
    timing_reset(ctx);
    if (n == 0) return SARPRO_HIP_OK;
    return sarpro::synrgb_flat_dev(ctx, mode, strategy, d_b1, d_b2, n, n, false, d_rgb);
}

// The composition of n pixels that are a PART of a product of n_total pixels (`reduce`: the other parts are other ranks' -- the
// suppressed variant's combined histogram is summed over the ranks before the floor is taken, synthetic_rgb.rs:92-113; every rank calls,
// also one with n = 0).  n == n_total, reduce false: the whole product (sarpro_hip_synrgb_u8_dev).
int sarpro::synrgb_flat_dev(sarpro_hip_ctx *ctx, int mode, int strategy, const uint8_t *d_b1, const uint8_t *d_b2, size_t n, size_t n_total,
                            bool reduce, uint8_t *d_rgb) {
    (void)mode; // synthetic_rgb.rs:72-79: the mode is ignored
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // view the flat arrays as rows of 4096 px (+ one short row) so the 2-D kernels apply
    const size_t W = 4096;
    const size_t full_rows = n / W, tail = n - full_rows * W;
    const bool suppressed = strategy == SARPRO_STRATEGY_TAMED || strategy == SARPRO_STRATEGY_CLAHE;
    std::vector<uint8_t> luts(66048), tables(66048);
    int fwc = -1;
    if (suppressed) {
        HIPCHK(ctx, ctx->level_hist.reserve(sizeof(uint64_t) * 256 * kMaxBands));
        HIPCHK(ctx, hipMemsetAsync(ctx->level_hist.p, 0, sizeof(uint64_t) * 256, ctx->stream));
        for (const uint8_t *p : {d_b1, d_b2}) {
            KernelTimer t(ctx, "hist256_u8");
            if (full_rows) HIPCHK(ctx, launch_hist256_u8(p, W, (uint32_t)full_rows, (uint32_t)W, ctx->level_hist.as<unsigned long long>(), ctx->stream));
            if (tail) HIPCHK(ctx, launch_hist256_u8(p + full_rows * W, W, 1, (uint32_t)tail, ctx->level_hist.as<unsigned long long>(), ctx->stream));
        }
        if (reduce) RETCHK(comm_allreduce_sum_u64_async(ctx, ctx->level_hist.as<uint64_t>(), 256));
        uint64_t h[256];
        HIPCHK(ctx, hipMemcpyAsync(h, ctx->level_hist.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        fwc = synrgb_floor_from_hist(h, n_total);
        synrgb_luts_suppressed(fwc, luts.data());
    } else {
        synrgb_luts_default(luts.data());
    }
    uint8_t ident[256];
    for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
    fold_compose_tables(luts.data(), fwc, ident, ident, tables.data());
    HIPCHK(ctx, ctx->tables.reserve(66048 + 512));
    HIPCHK(ctx, ctx->h_upload.reserve(2 * 131072 + 2 * 64 * 256 * 8 + 66048 + 1024));
    std::memcpy(ctx->h_upload.p, tables.data(), 66048); // pinned and the context's own: the copy needs no wait of its own
    HIPCHK(ctx, hipMemcpyAsync(ctx->tables.p, ctx->h_upload.p, 66048, hipMemcpyHostToDevice, ctx->stream));
    const bool al = ptr_aligned16(d_b1) && ptr_aligned16(d_b2) && ptr_aligned16(d_rgb);
    ComposeArgs c{};
    c.tables = ctx->tables.as<uint8_t>();
    c.in_pitch = W; c.rgb_pitch_px = W;
    if (full_rows) {
        c.b1 = d_b1; c.b2 = d_b2; c.rgb = d_rgb; c.rows = (uint32_t)full_rows; c.cols = (uint32_t)W;
        KernelTimer t(ctx, "compose_u8");
        HIPCHK(ctx, launch_compose_u8(c, al ? 16 : 1, ctx->stream));
    }
    if (tail) {
        c.b1 = d_b1 + full_rows * W; c.b2 = d_b2 + full_rows * W; c.rgb = d_rgb + full_rows * W * 3;
        c.rows = 1; c.cols = (uint32_t)tail;
        KernelTimer t(ctx, "compose_u8");
        HIPCHK(ctx, launch_compose_u8(c, al ? 16 : 1, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

// ---------------------------------------------------------------------------------------
// host-pointer entry points: stage through pitched device buffers, run the device path
// ---------------------------------------------------------------------------------------
namespace sarpro {

int stage_in_2d(sarpro_hip_ctx *ctx, DevBuf &buf, const void *host, size_t rows, size_t cols, size_t esz, size_t *pitch_elems) {
    const size_t pitch = round_up(std::max<size_t>(cols, 1), 64);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, buf.reserve(std::max<size_t>(rows, 1) * pitch * esz));
    if (rows && cols)
        HIPCHK(ctx, hipMemcpy2DAsync(buf.p, pitch * esz, host, cols * esz, cols * esz, rows, hipMemcpyHostToDevice, ctx->stream));
    *pitch_elems = pitch;
    return SARPRO_HIP_OK;
}

int fetch_out_2d(sarpro_hip_ctx *ctx, void *host, const void *dev, size_t pitch_bytes, size_t row_bytes, size_t rows) {
    if (rows && row_bytes) {
        HIPCHK(ctx, hipMemcpy2DAsync(host, row_bytes, dev, pitch_bytes, row_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SARPRO_HIP_OK;
}

} // namespace sarpro

static int host_band_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols, int strategy, int bit_depth,
                         int tamed_force, uint8_t *out_u8, uint16_t *out_u16, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const bool u8o = tamed_force || bit_depth == SARPRO_BITDEPTH_U8;
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if (rows * cols && (!in || (u8o ? (void *)out_u8 : (void *)out_u16) == nullptr)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    size_t pitch = 0;
    RETCHK(stage_in_2d(ctx, ctx->stage_in[0], in, rows, cols, 2, &pitch));
    const size_t osz = u8o ? 1 : 2;
    HIPCHK(ctx, ctx->stage_out[0].reserve(std::max<size_t>(rows, 1) * pitch * osz));
    U16Job J;
    J.ctx = ctx; J.nbands = 1; J.d_in[0] = ctx->stage_in[0].as<uint16_t>();
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = pitch;
    J.strategy = tamed_force ? SARPRO_STRATEGY_TAMED : strategy;
    J.bit_depth = u8o ? SARPRO_BITDEPTH_U8 : SARPRO_BITDEPTH_U16;
    J.tamed_force = tamed_force;
    if (!tamed_force && (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    void *outs[kMaxBands] = {ctx->stage_out[0].p, nullptr};
    RETCHK(job_run_all(J, outs, pitch, nullptr, 0, stats_out));
    return fetch_out_2d(ctx, u8o ? (void *)out_u8 : (void *)out_u16, ctx->stage_out[0].p, pitch * osz, cols * osz, rows);
}

// ---------------------------------------------------------------------------------------
// Streaming ingest / egress: the two u16 bands arrive through a row-chunk reader (a GDAL
// RasterIO loop, the strip-TIFF reader of tiff_io.cpp, ...) into a ring of pinned buffers; each chunk goes to
// the device with hipMemcpy2DAsync on a side stream while the reader fills the next one, and the DN-histogram
// work items whose rows have arrived run on the compute stream behind an event -- when the last chunk lands
// the first pass of the chain is already done.  The RGB leaves the same way, chunk by chunk, to a row sink.
// ---------------------------------------------------------------------------------------
static int stream_prepare(sarpro_hip_ctx *ctx, size_t ring_bytes) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (hipEvent_t &e : ctx->ring_evt)
        if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(ctx, ctx->h_ring.reserve(ring_bytes));
    return SARPRO_HIP_OK;
}

namespace sarpro {
int stream_upload_band(sarpro_hip_ctx *ctx, sarpro_hip_row_reader reader, void *user, int band, size_t rows, size_t cols,
                       uint16_t *d_dst, size_t pitch, size_t chunk_rows) {
    if (!reader) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null reader");
    if (rows * cols == 0) return SARPRO_HIP_OK;
    constexpr int kSlots = 3;
    if (!chunk_rows) chunk_rows = std::max<size_t>(16, (32u << 20) / (pitch * 2));
    chunk_rows = std::min(chunk_rows, rows);
    const size_t slot_bytes = chunk_rows * pitch * sizeof(uint16_t);
    RETCHK(stream_prepare(ctx, kSlots * slot_bytes));
    uint8_t *ring = ctx->h_ring.as<uint8_t>();
    size_t chunk = 0;
    for (size_t r0 = 0; r0 < rows; r0 += chunk_rows, ++chunk) {
        const int slot = (int)(chunk % kSlots);
        const size_t n = std::min(chunk_rows, rows - r0);
        if (chunk >= (size_t)kSlots) HIPCHK(ctx, hipEventSynchronize(ctx->ring_evt[slot]));
        uint16_t *h = reinterpret_cast<uint16_t *>(ring + slot * slot_bytes);
        {
            HostTimer t(ctx, "host:reader");
            if (int rc = reader(user, band, r0, n, h, pitch)) {
                (void)hipStreamSynchronize(ctx->copy_stream);
                ctx->err = "row reader failed (code " + std::to_string(rc) + ")";
                return SARPRO_HIP_ERR_IO;
            }
        }
        HIPCHK(ctx, hipMemcpyAsync(d_dst + r0 * pitch, h, n * pitch * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->ring_evt[slot], ctx->copy_stream));
    }
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_evt[(chunk - 1) % kSlots], 0)); // the compute stream sees the band
    return SARPRO_HIP_OK;
}
} // namespace sarpro

extern "C" int sarpro_hip_dualpol_synrgb_stream_u16(sarpro_hip_ctx *ctx, sarpro_hip_row_reader reader, void *reader_user, size_t rows,
                                                    size_t cols, int strategy, int mode, size_t chunk_rows, sarpro_hip_row_sink sink,
                                                    void *sink_user, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (!reader || !sink) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null reader / sink");
    timing_reset(ctx);
    if (rows * cols == 0) return SARPRO_HIP_OK;
    constexpr int kSlots = 3;
    const size_t pitch = round_up(cols, 64);
    if (!chunk_rows) chunk_rows = std::max<size_t>(16, (32u << 20) / (pitch * 2 * 2)); // ~32 MiB per slot (both bands)
    chunk_rows = std::min(chunk_rows, rows);
    const size_t in_slot = 2 * chunk_rows * pitch * sizeof(uint16_t), out_chunk_rows = std::max<size_t>(1, in_slot / (pitch * 3));
    const size_t slot_bytes = std::max(in_slot, out_chunk_rows * pitch * 3);
    RETCHK(stream_prepare(ctx, kSlots * slot_bytes));
    HIPCHK(ctx, ctx->stage_in[0].reserve(rows * pitch * 2));
    HIPCHK(ctx, ctx->stage_in[1].reserve(rows * pitch * 2));
    HIPCHK(ctx, ctx->stage_out[0].reserve(rows * pitch * 3));
    U16Job J;
    J.ctx = ctx; J.nbands = 2; J.d_in[0] = ctx->stage_in[0].as<uint16_t>(); J.d_in[1] = ctx->stage_in[1].as<uint16_t>();
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = pitch;
    J.strategy = strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.mode = mode; J.synrgb = true;
    RETCHK(job_init(J));

    // ---- ingest: reader -> pinned slot -> device, histogram items behind each chunk ----
    const std::vector<Rect> &items = J.clahe() ? J.plan->hist_rects_tiled : J.plan->hist_rects_flat;
    bool sorted = true;
    for (size_t i = 1; i < items.size() && sorted; ++i) sorted = items[i - 1].r0 <= items[i].r0;
    int issued = 0;
    bool begun = false;
    uint8_t *ring = ctx->h_ring.as<uint8_t>();
    size_t chunk = 0;
    for (size_t r0 = 0; r0 < rows; r0 += chunk_rows, ++chunk) {
        const int slot = (int)(chunk % kSlots);
        const size_t n = std::min(chunk_rows, rows - r0);
        if (chunk >= (size_t)kSlots) HIPCHK(ctx, hipEventSynchronize(ctx->ring_evt[slot])); // the slot's previous upload has left it
        uint16_t *h[2] = {reinterpret_cast<uint16_t *>(ring + slot * slot_bytes), reinterpret_cast<uint16_t *>(ring + slot * slot_bytes) + chunk_rows * pitch};
        {
            HostTimer t(ctx, "host:reader");
            for (int b = 0; b < 2; ++b)
                if (int rc = reader(reader_user, b, r0, n, h[b], pitch)) {
                    (void)hipStreamSynchronize(ctx->copy_stream);
                    (void)hipStreamSynchronize(ctx->stream);
                    ctx->err = "row reader failed (code " + std::to_string(rc) + ")";
                    return SARPRO_HIP_ERR_IO;
                }
        }
        for (int b = 0; b < 2; ++b)
            HIPCHK(ctx, hipMemcpyAsync(const_cast<uint16_t *>(J.d_in[b]) + r0 * pitch, h[b], n * pitch * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->ring_evt[slot], ctx->copy_stream));
        if (sorted) {
            int last = issued;
            while (last < (int)items.size() && (size_t)items[last].r1 <= r0 + n) ++last;
            if (last > issued || !begun) {
                HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_evt[slot], 0));
                RETCHK(job_phase1(J, !begun, issued, last, false));
                begun = true;
                issued = last;
            }
        }
    }
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_evt[(chunk - 1) % kSlots], 0));
    RETCHK(job_phase1(J, !begun, issued, -1, true)); // whatever is left, then the band histograms
    J.hist_done = true;

    // ---- the rest of the chain, device-resident ----
    void *outs[kMaxBands] = {nullptr, nullptr};
    uint8_t *d_rgb = ctx->stage_out[0].as<uint8_t>();
    {
        HostTimer t(ctx, "host:chain(enqueue+final sync)");
        if (chain_eligible(J)) RETCHK(job_run_chain(J, outs, 0, d_rgb, pitch, stats_out));
        else if (chain_levels_eligible(J, outs, 0, d_rgb, pitch)) RETCHK(job_run_chain_levels(J, outs, 0, d_rgb, pitch, stats_out));
        else return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "streaming ingest needs the device chain (SARPRO_HIP_NO_CHAIN is set)");
    }

    // ---- egress: device -> pinned slot -> sink, the copy of chunk k+1 under the sink of chunk k ----
    size_t ochunk = 0;
    const size_t nout = (rows + out_chunk_rows - 1) / out_chunk_rows;
    auto enqueue_out = [&](size_t k) -> int {
        const size_t r0 = k * out_chunk_rows, n = std::min(out_chunk_rows, rows - r0);
        HIPCHK(ctx, hipMemcpyAsync(ring + (k % kSlots) * slot_bytes, d_rgb + r0 * pitch * 3, n * pitch * 3, hipMemcpyDeviceToHost, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->ring_evt[k % kSlots], ctx->copy_stream));
        return SARPRO_HIP_OK;
    };
    for (size_t k = 0; k < std::min<size_t>(kSlots - 1, nout); ++k) RETCHK(enqueue_out(k));
    for (; ochunk < nout; ++ochunk) {
        if (ochunk + kSlots - 1 < nout) RETCHK(enqueue_out(ochunk + kSlots - 1));
        HIPCHK(ctx, hipEventSynchronize(ctx->ring_evt[ochunk % kSlots]));
        const size_t r0 = ochunk * out_chunk_rows, n = std::min(out_chunk_rows, rows - r0);
        HostTimer t(ctx, "host:sink");
        if (int rc = sink(sink_user, r0, n, ring + (ochunk % kSlots) * slot_bytes, pitch * 3)) {
            (void)hipStreamSynchronize(ctx->copy_stream);
            ctx->err = "row sink failed (code " + std::to_string(rc) + ")";
            return SARPRO_HIP_ERR_IO;
        }
    }
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_autoscale_band_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols,
                                             int strategy, int bit_depth, uint8_t *out_u8, uint16_t *out_u16,
                                             sarpro_hip_stats *stats_out) {
    return host_band_u16(ctx, in, rows, cols, strategy, bit_depth, 0, out_u8, out_u16, stats_out);
}

extern "C" int sarpro_hip_tamed_synrgb_u8_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols,
                                              int is_copol, uint8_t *out_u8) {
    return host_band_u16(ctx, in, rows, cols, SARPRO_STRATEGY_TAMED, SARPRO_BITDEPTH_U8,
                         is_copol ? kTamedCopol : kTamedCrosspol, out_u8, nullptr, nullptr);
}

extern "C" int sarpro_hip_dualpol_synrgb_u16(sarpro_hip_ctx *ctx, const uint16_t *band1, const uint16_t *band2,
                                             size_t rows, size_t cols, int strategy, int mode, uint8_t *rgb_out,
                                             uint8_t *u8_band1, uint8_t *u8_band2, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (rows * cols && (!band1 || !band2 || !rgb_out)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    size_t pitch = 0;
    RETCHK(stage_in_2d(ctx, ctx->stage_in[0], band1, rows, cols, 2, &pitch));
    RETCHK(stage_in_2d(ctx, ctx->stage_in[1], band2, rows, cols, 2, &pitch));
    const size_t r1 = std::max<size_t>(rows, 1);
    HIPCHK(ctx, ctx->stage_out[0].reserve(r1 * pitch * 3));
    const bool want_u8 = u8_band1 || u8_band2;
    if (want_u8) { HIPCHK(ctx, ctx->stage_out[1].reserve(r1 * pitch)); HIPCHK(ctx, ctx->stage_out[2].reserve(r1 * pitch)); }
    int rc = sarpro_hip_dualpol_synrgb_u16_dev(ctx, ctx->stage_in[0].as<uint16_t>(), ctx->stage_in[1].as<uint16_t>(), rows, cols,
                                               pitch, strategy, mode, ctx->stage_out[0].as<uint8_t>(), pitch,
                                               want_u8 ? ctx->stage_out[1].as<uint8_t>() : nullptr,
                                               want_u8 ? ctx->stage_out[2].as<uint8_t>() : nullptr, pitch, stats_out);
    if (rc) return rc;
    RETCHK(fetch_out_2d(ctx, rgb_out, ctx->stage_out[0].p, pitch * 3, cols * 3, rows));
    if (u8_band1) RETCHK(fetch_out_2d(ctx, u8_band1, ctx->stage_out[1].p, pitch, cols, rows));
    if (u8_band2) RETCHK(fetch_out_2d(ctx, u8_band2, ctx->stage_out[2].p, pitch, cols, rows));
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_polop_f32(sarpro_hip_ctx *ctx, int op, const float *a, const float *b, size_t n, float *out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (op < SARPRO_OP_SUM || op > SARPRO_OP_LOGRATIO) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad polarization operation");
    if ((!a || !b || !out) && n) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null array");
    if (!n) return SARPRO_HIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->stage_in[0].reserve(n * 4));
    HIPCHK(ctx, ctx->stage_in[1].reserve(n * 4));
    HIPCHK(ctx, ctx->stage_out[0].reserve(n * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_in[0].p, a, n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_in[1].p, b, n * 4, hipMemcpyHostToDevice, ctx->stream));
    RETCHK(sarpro_hip_polop_f32_dev(ctx, op, ctx->stage_in[0].as<float>(), ctx->stage_in[1].as<float>(), n, ctx->stage_out[0].as<float>()));
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->stage_out[0].p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_synrgb_u8(sarpro_hip_ctx *ctx, int mode, int strategy, const uint8_t *band1, const uint8_t *band2,
                                    size_t n, uint8_t *rgb_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if ((!band1 || !band2 || !rgb_out) && n) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null array");
    if (mode < 0 || mode > SARPRO_SYNRGB_ENHANCED) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synrgb mode");
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (!n) return SARPRO_HIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->stage_in[0].reserve(n));
    HIPCHK(ctx, ctx->stage_in[1].reserve(n));
    HIPCHK(ctx, ctx->stage_out[0].reserve(n * 3));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_in[0].p, band1, n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_in[1].p, band2, n, hipMemcpyHostToDevice, ctx->stream));
    RETCHK(sarpro_hip_synrgb_u8_dev(ctx, mode, strategy, ctx->stage_in[0].as<uint8_t>(), ctx->stage_in[1].as<uint8_t>(), n,
                                    ctx->stage_out[0].as<uint8_t>()));
    HIPCHK(ctx, hipMemcpyAsync(rgb_out, ctx->stage_out[0].p, n * 3, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

// ---------------------------------------------------------------------------------------
// synthetic scene
// ---------------------------------------------------------------------------------------
extern "C" int sarpro_hip_synth_scene_u16_dev_ex(sarpro_hip_ctx *ctx, uint64_t seed, int band, const uint16_t *q_tables_host,
                                                 size_t rows_total, size_t cols, size_t row0, size_t rows_local,
                                                 uint16_t *d_out, size_t pitch, uint32_t flags) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (band < 0 || band > 1 || !q_tables_host || (!d_out && rows_local * cols) || pitch < cols)
        return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synthetic scene arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t qbytes = sizeof(uint16_t) * 4 * 65536;
    HIPCHK(ctx, ctx->qtab.reserve(qbytes));
    HIPCHK(ctx, hipMemcpyAsync(ctx->qtab.p, q_tables_host + (size_t)band * 4 * 65536, qbytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, launch_synth_scene_u16(seed, band, ctx->qtab.as<uint16_t>(), rows_total, cols, row0, rows_local, d_out, pitch, flags, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}
extern "C" int sarpro_hip_synth_scene_u16_dev(sarpro_hip_ctx *ctx, uint64_t seed, int band, const uint16_t *q_tables_host,
                                              size_t rows_total, size_t cols, size_t row0, size_t rows_local,
                                              uint16_t *d_out, size_t pitch) {
    return sarpro_hip_synth_scene_u16_dev_ex(ctx, seed, band, q_tables_host, rows_total, cols, row0, rows_local, d_out, pitch, 0u);
}

// ---------------------------------------------------------------------------------------
// row-stripe protocol
// ---------------------------------------------------------------------------------------
struct sarpro_hip_stripe {
    U16Job job;
    int phase = 0;
};

extern "C" int sarpro_hip_stripe_begin_u16(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2,
                                           size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t in_pitch,
                                           int strategy, int mode, sarpro_hip_stripe **out) {
    if (!ctx || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    if ((!d_band1 || !d_band2) && rows_local * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    sarpro_hip_stripe *s = new sarpro_hip_stripe();
    U16Job &J = s->job;
    J.ctx = ctx; J.nbands = 2; J.d_in[0] = d_band1; J.d_in[1] = d_band2;
    J.rows_total = rows_total; J.cols = cols; J.row0 = row0; J.rows_local = rows_local; J.in_pitch = in_pitch;
    J.strategy = strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.mode = mode; J.synrgb = true;
    timing_reset(ctx);
    int rc = job_init(J);
    if (rc) { delete s; return rc; }
    J.plan->refs += 1; // the plan cache never evicts a plan an open stripe handle works on
    *out = s;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_phase1(sarpro_hip_stripe *s, uint64_t **d_buf, size_t *count) {
    if (!s || !d_buf || !count || s->phase != 0) return SARPRO_HIP_ERR_INVALID_ARG;
    RETCHK(job_phase1(s->job));
    HIPCHK(s->job.ctx, hipStreamSynchronize(s->job.ctx->stream)); // the buffer is complete when we return
    *d_buf = s->job.ctx->ghist.as<uint64_t>();
    *count = 65536 * 2;
    s->phase = 1;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_phase2(sarpro_hip_stripe *s, uint64_t **d_buf, size_t *count) {
    if (!s || !d_buf || !count || s->phase != 1) return SARPRO_HIP_ERR_INVALID_ARG;
    RETCHK(job_after_phase1(s->job));
    RETCHK(job_phase2(s->job));
    HIPCHK(s->job.ctx, hipStreamSynchronize(s->job.ctx->stream));
    if (s->job.clahe()) { *d_buf = s->job.ctx->tile_bins.as<uint64_t>(); *count = 64 * 256 * 2; }
    else { *d_buf = nullptr; *count = 0; }
    s->phase = 2;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_phase3(sarpro_hip_stripe *s, uint64_t **d_buf, size_t *count) {
    if (!s || !d_buf || !count || s->phase != 2) return SARPRO_HIP_ERR_INVALID_ARG;
    void *outs[kMaxBands] = {nullptr, nullptr};
    RETCHK(job_phase3(s->job, outs, 0));
    HIPCHK(s->job.ctx, hipStreamSynchronize(s->job.ctx->stream));
    if (s->job.clahe()) { *d_buf = s->job.ctx->level_hist.as<uint64_t>(); *count = 256 * 2; }
    else { *d_buf = nullptr; *count = 0; } // percentile strategies: level histogram follows from the reduced DN histogram
    s->phase = 3;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_phase4(sarpro_hip_stripe *s, uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out) {
    if (!s || s->phase != 3) return SARPRO_HIP_ERR_INVALID_ARG;
    if (!d_rgb && s->job.rows_local * s->job.cols) return fail(s->job.ctx, SARPRO_HIP_ERR_INVALID_ARG, "null rgb raster");
    if (rgb_pitch_px < s->job.cols) return fail(s->job.ctx, SARPRO_HIP_ERR_INVALID_ARG, "rgb_pitch_px < cols");
    void *outs[kMaxBands] = {nullptr, nullptr};
    if (s->job.rows_local * s->job.cols) RETCHK(job_phase4(s->job, outs, 0, d_rgb, rgb_pitch_px, false));
    if (stats_out) { stats_out[0] = s->job.stats[0]; stats_out[1] = s->job.stats[1]; }
    s->phase = 4;
    return SARPRO_HIP_OK;
}

extern "C" void sarpro_hip_stripe_end(sarpro_hip_stripe *s) {
    if (!s) return;
    if (s->job.plan && s->job.plan->refs > 0) s->job.plan->refs -= 1;
    delete s;
}

// One call per rank for one row stripe of a scene, reductions over the library's RCCL communicator
// (sarpro_hip_comm_init): the device-resident chains run unchanged with three (CLAHE) or one (percentile
// strategies) small all-reduces enqueued on the stream between their kernels -- no host synchronisation
// until the stripe's RGB is complete.
namespace sarpro {
// Which route a stripe takes (device-resident chain or host phases, fused pass or apply + compose) follows from the LAYOUT of its
// rasters -- alignment of the pointers, pitches -- and a route fixes the sequence of collectives the rank joins.  Ranks with
// different layouts would therefore join different sequences and wait for each other for ever.  The one-call stripe entry point
// makes the layout a property of the library, not of the caller: a stripe whose rasters are not in the aligned form (16-byte
// aligned pointers, in_pitch % 16 == 0, rgb_pitch_px % 16 == 0) is staged through library-owned rasters that are (one device copy
// in, one out: 11 B/px on that rank only), so every rank of a scene takes the same route whatever it was handed.
static int stripe_run_u16_impl(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2, size_t rows_total,
                               size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int mode,
                               uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out) {
    if (!ctx->comm && !ctx->local_group && !ctx->attrs.on(A_COMM_REPLAY)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "no communicator on this context (sarpro_hip_comm_init / _init_local)");
    comm_replay_rewind(ctx);
    if ((!d_band1 || !d_band2 || !d_rgb) && rows_local * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (rgb_pitch_px < cols || in_pitch < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pitch < cols");
    const bool empty = rows_local == 0 || cols == 0;
    const bool aligned = !empty && in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 && ptr_aligned16(d_band1) && ptr_aligned16(d_band2) && ptr_aligned16(d_rgb);
    const size_t lib_pitch = round_up(std::max<size_t>(cols, 1), 64);
    uint8_t *rgb_user = nullptr;
    size_t rgb_user_pitch = 0;
    if (empty) { // nothing of this rank is read or written; its pitches still decide the route: the library's
        in_pitch = rgb_pitch_px = lib_pitch;
    } else if (!aligned) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        const uint16_t *src[2] = {d_band1, d_band2};
        for (int b = 0; b < 2; ++b) {
            HIPCHK(ctx, ctx->stage_in[b].reserve(rows_local * lib_pitch * sizeof(uint16_t)));
            HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage_in[b].p, lib_pitch * sizeof(uint16_t), src[b], in_pitch * sizeof(uint16_t), cols * sizeof(uint16_t), rows_local,
                                         hipMemcpyDeviceToDevice, ctx->stream));
        }
        HIPCHK(ctx, ctx->stage_out[0].reserve(rows_local * lib_pitch * 3));
        d_band1 = ctx->stage_in[0].as<uint16_t>(); d_band2 = ctx->stage_in[1].as<uint16_t>();
        rgb_user = d_rgb; rgb_user_pitch = rgb_pitch_px;
        d_rgb = ctx->stage_out[0].as<uint8_t>();
        in_pitch = rgb_pitch_px = lib_pitch;
    }
    U16Job J;
    J.ctx = ctx; J.nbands = 2; J.d_in[0] = d_band1; J.d_in[1] = d_band2;
    J.rows_total = rows_total; J.cols = cols; J.row0 = row0; J.rows_local = rows_local; J.in_pitch = in_pitch;
    J.strategy = strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.mode = mode; J.synrgb = true; J.reduce = true;
    void *outs[kMaxBands] = {nullptr, nullptr};
    RETCHK(job_run_all(J, outs, 0, d_rgb, rgb_pitch_px, stats_out));
    if (rgb_user) {
        HIPCHK(ctx, hipMemcpy2DAsync(rgb_user, rgb_user_pitch * 3, d_rgb, rgb_pitch_px * 3, cols * 3, rows_local, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SARPRO_HIP_OK;
}
} // namespace sarpro

extern "C" int sarpro_hip_stripe_run_u16(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2, size_t rows_total,
                                         size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int mode,
                                         uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const int rc = sarpro::stripe_run_u16_impl(ctx, d_band1, d_band2, rows_total, cols, row0, rows_local, in_pitch, strategy, mode, d_rgb, rgb_pitch_px, stats_out);
    if (rc != SARPRO_HIP_OK) sarpro::comm_abort_local_group(ctx); // an in-process group: the peers of a rank that failed must not wait for it (comm.cpp)
    return rc;
}
